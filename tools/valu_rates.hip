// valu_rates.hip -- measures per-instruction VALU issue cost on gfx950 for the instruction mix of the
// path tracer (f32, packed f32, f64, transcendental, select, LDS broadcast read).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
// Each test: every wave runs ITER iterations of 8 independent instructions; every SIMD of the chip holds
// `waves` waves.  Reports cycles per wave-instruction per SIMD = clk * time / (instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096
typedef float float2_ __attribute__((ext_vector_type(2)));

template <typename T> __device__ T splat(float s);
template <> __device__ float splat<float>(float s) { return s; }
template <> __device__ double splat<double>(float s) { return (double)s; }
template <> __device__ float2_ splat<float2_>(float s) { float2_ r = {s, s + 0.5f}; return r; }
__device__ float fold(float a) { return a; }
__device__ float fold(double a) { return (float)a; }
__device__ float fold(float2_ a) { return a.x + a.y; }

#define DEFX(name, T, OP, BCONSTRAINT, CLOB)                                                              \
    __global__ void __launch_bounds__(256) name(float *out, float seed)                            \
    {                                                                                              \
        T a0 = splat<T>(seed), a1 = splat<T>(seed + 1), a2 = splat<T>(seed + 2), a3 = splat<T>(seed + 3);  \
        T a4 = splat<T>(seed + 4), a5 = splat<T>(seed + 5), a6 = splat<T>(seed + 6), a7 = splat<T>(seed + 7); \
        T b = splat<T>(seed * 0.999f);                                                             \
        for (int i = 0; i < ITER; ++i) {                                                           \
            asm volatile(OP : "+v"(a0) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a1) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a2) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a3) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a4) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a5) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a6) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a7) : BCONSTRAINT(b) : CLOB);                                  \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = fold(a0) + fold(a1) + fold(a2) + fold(a3) + fold(a4) + fold(a5) + fold(a6) + fold(a7); \
    }

#define DEF(name, T, OP, B) DEFX(name, T, OP, B, "memory")
#define DEFC(name, T, OP, B) DEFX(name, T, OP, B, "vcc")
DEF(k_add_f32, float, "v_add_f32 %0, %0, %1", "v")
DEF(k_mul_f32_s, float, "v_mul_f32 %0, %1, %0", "s")
DEF(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %1", "v")
DEF(k_pk_mul_f32, float2_, "v_pk_mul_f32 %0, %0, %1", "v")
DEF(k_pk_add_f32, float2_, "v_pk_add_f32 %0, %0, %1", "v")
DEF(k_pk_fma_f32, float2_, "v_pk_fma_f32 %0, %0, %1, %1", "v")
DEF(k_mul_f64, double, "v_mul_f64 %0, %0, %1", "v")
DEF(k_add_f64, double, "v_add_f64 %0, %0, %1", "v")
DEF(k_fma_f64, double, "v_fma_f64 %0, %0, %1, %1", "v")
DEF(k_sqrt_f32, float, "v_sqrt_f32 %0, %0", "v")
DEF(k_rcp_f32, float, "v_rcp_f32 %0, %0", "v")
DEF(k_mov_b32, float, "v_mov_b32 %0, %1", "v")
DEF(k_cvt_f32_i32, float, "v_cvt_f32_i32 %0, %0", "v")
DEFC(k_cndmask, float, "v_cndmask_b32 %0, %0, %1, vcc", "v")
DEF(k_cndmask_nc, float, "v_cndmask_b32 %0, %0, %1, vcc", "v")
DEF(k_cndmask_s, float, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "v")
DEFC(k_cmp_f32, float, "v_cmp_lt_f32 vcc, %0, %1", "v")
DEF(k_cmp_f32_s, float, "v_cmp_lt_f32_e64 s[10:11], %0, %1", "v")
DEF(k_max_f32, float, "v_max_f32 %0, %0, %1", "v")
DEF(k_sub_f32, float, "v_sub_f32 %0, %1, %0", "v")
DEF(k_fmac_f32, float, "v_fmac_f32 %0, %1, %1", "v")
DEF(k_add_u32, float, "v_add_u32 %0, %0, %1", "v")
DEF(k_lshrrev, float, "v_lshrrev_b32 %0, 9, %0", "v")
DEF(k_add_f32_e64, float, "v_add_f32_e64 %0, %0, -%1", "v")
DEF(k_fma_f32_sgpr, float, "v_fma_f32 %0, %0, %1, %1", "s")
DEF(k_xor_b32, float, "v_xor_b32 %0, %0, %1", "v")
DEF(k_lshl_add, float, "v_lshl_add_u32 %0, %0, 3, %1", "v")
DEF(k_alignbit, float, "v_alignbit_b32 %0, %0, %0, 11", "v")

__global__ void __launch_bounds__(256) k_cvt_f64_f32(float *out, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    double d0, d1, d2, d3;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(a0));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(a1));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(a2));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d3) : "v"(a3));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a0) : "v"(d0));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a1) : "v"(d1));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a2) : "v"(d2));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a3) : "v"(d3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ void __launch_bounds__(256) k_lds_bcast(float *out, float seed)
{
    __shared__ float4 tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = float4{seed, seed, seed, seed};
    __syncthreads();
    float acc = 0;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 v = tab[(i + j) & 63];
            acc += v.x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K>
void run(K k, const char *name, int waves, float *d_out, double clk_hz)
{
    const int blocks = 256 * waves;   // 256 CUs x (one 256-thread block = 1 wave per SIMD)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves * ITER * 8.0;
    printf("%-24s waves/SIMD %d  %8.3f ms  %6.2f cycles/wave-instr/SIMD (at %.2f GHz nominal)\n",
           name, waves, ms, ms * 1e-3 * clk_hz / per_simd, clk_hz / 1e9);
}

int main()
{
    float *d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz\n", p.name, p.multiProcessorCount, clk / 1e6);
    for (int w : {4, 8}) {
        run(k_add_f32, "v_add_f32", w, d_out, clk);
        run(k_mul_f32_s, "v_mul_f32 (sgpr src)", w, d_out, clk);
        run(k_fma_f32, "v_fma_f32", w, d_out, clk);
        run(k_pk_mul_f32, "v_pk_mul_f32", w, d_out, clk);
        run(k_pk_add_f32, "v_pk_add_f32", w, d_out, clk);
        run(k_pk_fma_f32, "v_pk_fma_f32", w, d_out, clk);
        run(k_mul_f64, "v_mul_f64", w, d_out, clk);
        run(k_add_f64, "v_add_f64", w, d_out, clk);
        run(k_fma_f64, "v_fma_f64", w, d_out, clk);
        run(k_cvt_f64_f32, "v_cvt_f64_f32/f32_f64", w, d_out, clk);
        run(k_sqrt_f32, "v_sqrt_f32", w, d_out, clk);
        run(k_rcp_f32, "v_rcp_f32", w, d_out, clk);
        run(k_mov_b32, "v_mov_b32", w, d_out, clk);
        run(k_cvt_f32_i32, "v_cvt_f32_i32", w, d_out, clk);
        run(k_cndmask, "v_cndmask_b32 vcc (clob)", w, d_out, clk);
        run(k_cndmask_nc, "v_cndmask_b32 vcc", w, d_out, clk);
        run(k_cndmask_s, "v_cndmask_b32_e64 sgpr", w, d_out, clk);
        run(k_cmp_f32_s, "v_cmp_lt_f32_e64 sgpr", w, d_out, clk);
        run(k_max_f32, "v_max_f32", w, d_out, clk);
        run(k_sub_f32, "v_sub_f32", w, d_out, clk);
        run(k_fmac_f32, "v_fmac_f32", w, d_out, clk);
        run(k_add_u32, "v_add_u32", w, d_out, clk);
        run(k_lshrrev, "v_lshrrev_b32", w, d_out, clk);
        run(k_add_f32_e64, "v_add_f32_e64 (neg mod)", w, d_out, clk);
        run(k_fma_f32_sgpr, "v_fma_f32 (sgpr srcs)", w, d_out, clk);
        run(k_cmp_f32, "v_cmp_lt_f32", w, d_out, clk);
        run(k_xor_b32, "v_xor_b32", w, d_out, clk);
        run(k_lshl_add, "v_lshl_add_u32", w, d_out, clk);
        run(k_alignbit, "v_alignbit_b32", w, d_out, clk);
        run(k_lds_bcast, "ds_read_b128 bcast + add", w, d_out, clk);
    }
    return 0;
}
