// exec_mask_probe.hip -- does gfx950 skip VALU passes whose lanes are all inactive?  Every SIMD holds 8 waves; every
// wave runs ITER x 8 independent instructions with only some lanes enabled (EXEC mask patterns below).
// build: hipcc --offload-arch=gfx950 -O3 tools/exec_mask_probe.hip -o gpurun_out/exec_mask_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096

enum { FULL = 0, LOW32 = 1, LOW16 = 2, EVEN = 3, HIGH32 = 4, Q1AND3 = 5 };
__device__ bool lane_on(int mode)
{
    const int l = threadIdx.x & 63;
    switch (mode) {
    case LOW32: return l < 32;
    case LOW16: return l < 16;
    case EVEN: return (l & 1) == 0;
    case HIGH32: return l >= 32;
    case Q1AND3: return ((l >> 4) & 1) == 0;
    default: return true;
    }
}

#define DEF(name, T, INIT, OP)                                                                       \
    __global__ void __launch_bounds__(256) name(float *out, float seed, int mode)                   \
    {                                                                                                \
        T a0 = INIT(seed), a1 = INIT(seed + 1), a2 = INIT(seed + 2), a3 = INIT(seed + 3);            \
        T a4 = INIT(seed + 4), a5 = INIT(seed + 5), a6 = INIT(seed + 6), a7 = INIT(seed + 7);        \
        T b = INIT(seed * 0.999f);                                                                   \
        if (lane_on(mode)) {                                                                         \
            for (int i = 0; i < ITER; ++i) {                                                         \
                asm volatile(OP : "+v"(a0) : "v"(b)); asm volatile(OP : "+v"(a1) : "v"(b));          \
                asm volatile(OP : "+v"(a2) : "v"(b)); asm volatile(OP : "+v"(a3) : "v"(b));          \
                asm volatile(OP : "+v"(a4) : "v"(b)); asm volatile(OP : "+v"(a5) : "v"(b));          \
                asm volatile(OP : "+v"(a6) : "v"(b)); asm volatile(OP : "+v"(a7) : "v"(b));          \
            }                                                                                        \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);       \
    }
#define F32(x) (x)
#define F64(x) ((double)(x))
DEF(k_add_f32, float, F32, "v_add_f32 %0, %0, %1")
DEF(k_fma_f32, float, F32, "v_fma_f32 %0, %0, %1, %1")
DEF(k_fma_f64, double, F64, "v_fma_f64 %0, %0, %1, %1")
DEF(k_mul_f64, double, F64, "v_mul_f64 %0, %0, %1")
DEF(k_sqrt_f32, float, F32, "v_sqrt_f32 %0, %0")

int main()
{
    int cus = 0, clk_khz = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int waves_per_simd = 8, blocks = cus * 4 * waves_per_simd / 4;     // 256-thread blocks = 4 waves
    float *out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *modes[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "even lanes", "lanes 32-63", "lanes 0-15 + 32-47"};
    struct { const char *name; void (*k)(float *, float, int); } tests[] = {
        {"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_sqrt_f32", k_sqrt_f32}};
    printf("%d CUs, clock %d MHz, %d waves per SIMD; cycles per wave-instruction per SIMD\n", cus, clk_khz / 1000, waves_per_simd);
    for (auto &t : tests) {
        for (int mode = 0; mode < 6; ++mode) {
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f, mode);   // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = 5.0 * waves_per_simd * (double)ITER * 8.0;
            printf("%-11s %-20s %.2f\n", t.name, modes[mode], ms * 1e-3 * clk_khz * 1e3 / instr_per_simd);
        }
    }
    return 0;
}
