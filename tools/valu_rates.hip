// valu_rates.hip -- per-opcode VALU issue cost on gfx950, measured in the SHADER-CLOCK domain.
//
//   build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o build/valu_rates     run on the GPU box:
//          build/valu_rates [waves_per_simd ...]  > profiles/rNN_valu_rates.json
//
// Method (what round 1's table lacked, VERDICT r01 "What's weak"):
//   * every SIMD of the chip holds W waves (one 256-thread workgroup = one wave per SIMD of a CU, W workgroups per CU);
//     each wave runs ITERS iterations of 8 INDEPENDENT instances of the instruction (asm volatile, separate registers);
//   * a run lasts >= 50 ms (ITERS is scaled per opcode from a short calibration run), is preceded by ramp launches
//     until two consecutive runs agree, and is repeated; the median is reported;
//   * cycles are read on the device: every wave stamps s_memtime (the shader clock: it follows DVFS) before and after
//     its loop and the host takes, per launch, the median elapsed cycles over all waves -- launch overhead, clock ramp
//     and the nominal-clock assumption drop out.  cost = elapsed_cycles / (W * ITERS * 8) SIMD-cycles per wave-instruction;
//   * the same loop with the 8 instructions replaced by nothing (s_nop-free, just the scalar loop control) is measured
//     and SUBTRACTED per iteration;
//   * the effective clock (elapsed shader cycles / elapsed wall time of the same launch) is reported beside each figure.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <typename T> __device__ T splat(float s);
template <> __device__ float splat<float>(float s) { return s; }
template <> __device__ double splat<double>(float s) { return (double)s; }
template <> __device__ float2_ splat<float2_>(float s) { float2_ r = {s, s + 0.5f}; return r; }
__device__ float fold(float a) { return a; }
__device__ float fold(double a) { return (float)a; }
__device__ float fold(float2_ a) { return a.x + a.y; }

struct Stamp { unsigned long long cycles; };

#define BODY8(OP, BCONSTRAINT, CLOB)                                                               \
            asm volatile(OP : "+v"(a0) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a1) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a2) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a3) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a4) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a5) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a6) : BCONSTRAINT(b) : CLOB);                                  \
            asm volatile(OP : "+v"(a7) : BCONSTRAINT(b) : CLOB);

#define BODY32(OP, BCONSTRAINT, CLOB) BODY8(OP, BCONSTRAINT, CLOB) BODY8(OP, BCONSTRAINT, CLOB) BODY8(OP, BCONSTRAINT, CLOB) BODY8(OP, BCONSTRAINT, CLOB)

#define DEFX(name, T, OP, BCONSTRAINT, CLOB)                                                              \
    __global__ void __launch_bounds__(256) name(float *out, Stamp *stamps, float seed, int iters)         \
    {                                                                                              \
        T a0 = splat<T>(seed), a1 = splat<T>(seed + 1), a2 = splat<T>(seed + 2), a3 = splat<T>(seed + 3);  \
        T a4 = splat<T>(seed + 4), a5 = splat<T>(seed + 5), a6 = splat<T>(seed + 6), a7 = splat<T>(seed + 7); \
        T b = splat<T>(seed * 0.999f);                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                \
        for (int i = 0; i < iters; ++i) {                                                          \
            BODY32(OP, BCONSTRAINT, CLOB) BODY32(OP, BCONSTRAINT, CLOB)                            \
            BODY32(OP, BCONSTRAINT, CLOB) BODY32(OP, BCONSTRAINT, CLOB)                            \
        }                                                                                          \
        asm volatile("s_nop 0" ::: "memory");                                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                \
        if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * 256 + threadIdx.x) >> 6].cycles = t1 - t0;   \
        out[blockIdx.x * 256 + threadIdx.x] = fold(a0) + fold(a1) + fold(a2) + fold(a3) + fold(a4) + fold(a5) + fold(a6) + fold(a7); \
    }

#define DEF(name, T, OP, B) DEFX(name, T, OP, B, "memory")
#define DEFC(name, T, OP, B) DEFX(name, T, OP, B, "vcc")
DEF(k_add_f32, float, "v_add_f32 %0, %0, %1", "v")
DEF(k_sub_f32, float, "v_sub_f32 %0, %1, %0", "v")
DEF(k_mul_f32, float, "v_mul_f32 %0, %0, %1", "v")
DEF(k_mul_f32_s, float, "v_mul_f32 %0, %1, %0", "s")
DEF(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %1", "v")
DEF(k_fmac_f32, float, "v_fmac_f32 %0, %1, %1", "v")
DEF(k_fma_f32_sgpr, float, "v_fma_f32 %0, %0, %1, %1", "s")
DEF(k_max_f32, float, "v_max_f32 %0, %0, %1", "v")
DEF(k_add_f32_e64, float, "v_add_f32_e64 %0, %0, -%1", "v")
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
DEF(k_cndmask_s, float, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "v")
DEFC(k_cmp_f32, float, "v_cmp_lt_f32 vcc, %0, %1", "v")
DEF(k_cmp_f32_s, float, "v_cmp_lt_f32_e64 s[10:11], %0, %1", "v")
DEF(k_add_u32, float, "v_add_u32 %0, %0, %1", "v")
DEF(k_lshrrev, float, "v_lshrrev_b32 %0, 9, %0", "v")
DEF(k_xor_b32, float, "v_xor_b32 %0, %0, %1", "v")
DEF(k_and_b32, float, "v_and_b32 %0, %0, %1", "v")
DEF(k_lshl_add, float, "v_lshl_add_u32 %0, %0, 3, %1", "v")
DEF(k_alignbit, float, "v_alignbit_b32 %0, %0, %0, 11", "v")
DEF(k_mul_lo_u32, float, "v_mul_lo_u32 %0, %0, %1", "v")
DEF(k_bfe_u32, float, "v_bfe_u32 %0, %0, 3, 9", "v")
DEF(k_nop, float, "s_nop 0", "v")

__global__ void __launch_bounds__(256) k_empty(float *out, Stamp *stamps, float seed, int iters)
{
    float a0 = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) asm volatile("" : "+v"(a0) :: "memory");
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * 256 + threadIdx.x) >> 6].cycles = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a0;
}

__global__ void __launch_bounds__(256) k_cvt_f64_f32(float *out, Stamp *stamps, float seed, int iters)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    double d0, d1, d2, d3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(a0));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(a1));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(a2));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d3) : "v"(a3));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a0) : "v"(d0));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a1) : "v"(d1));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a2) : "v"(d2));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a3) : "v"(d3));
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * 256 + threadIdx.x) >> 6].cycles = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ void __launch_bounds__(256) k_lds_bcast(float *out, Stamp *stamps, float seed, int iters)
{
    __shared__ float4 tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = float4{seed, seed, seed, seed};
    __syncthreads();
    float acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 v = tab[(i + j) & 63];
            acc += v.x;
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * 256 + threadIdx.x) >> 6].cycles = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

struct Run { double cycles_per_iter; double clock_ghz; double ms; };

template <typename K>
Run launch_once(K k, int waves, int iters, float *d_out, Stamp *d_stamps, std::vector<Stamp> &host, int cus)
{
    const int blocks = cus * waves;   // one 256-thread block = one wave on every SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_stamps, 1.0f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    const size_t n = (size_t)blocks * 4;
    hipMemcpy(host.data(), d_stamps, n * sizeof(Stamp), hipMemcpyDeviceToHost);
    // The waves of a SIMD do not progress evenly: VALU issue goes to the oldest wave first (two waves of simple
    // instructions saturate a SIMD), so waves finish one after the other.  All start within microseconds of each other;
    // the LAST finisher's elapsed cycles are the SIMD's busy time for the work of all its W waves.
    std::vector<unsigned long long> c(n);
    for (size_t i = 0; i < n; ++i) c[i] = host[i].cycles;
    const size_t hi = n - 1 - n / 100;                  // 99th percentile: robust against a straggling CU
    std::nth_element(c.begin(), c.begin() + hi, c.end());
    Run r;
    r.cycles_per_iter = (double)c[hi] / iters;
    r.clock_ghz = (double)c[hi] / (ms * 1e6);          // shader cycles per wall ns: the effective clock of this launch
    r.ms = ms;
    return r;
}

template <typename K>
Run measure(K k, int waves, float *d_out, Stamp *d_stamps, std::vector<Stamp> &host, int cus)
{
    // calibrate ITERS for >= 50 ms, ramp until two runs agree within 1 %, then take the median of 5
    int iters = 1 << 14;
    Run r = launch_once(k, waves, iters, d_out, d_stamps, host, cus);
    r = launch_once(k, waves, iters, d_out, d_stamps, host, cus);
    const double per_iter_ms = r.ms / iters;
    iters = (int)std::min(2.0e9, std::max(1.0 * (1 << 14), 55.0 / per_iter_ms));
    Run prev = launch_once(k, waves, iters, d_out, d_stamps, host, cus);
    for (int tries = 0; tries < 6; ++tries) {
        r = launch_once(k, waves, iters, d_out, d_stamps, host, cus);
        const bool settled = std::abs(r.cycles_per_iter - prev.cycles_per_iter) <= 0.01 * prev.cycles_per_iter;
        prev = r;
        if (settled) break;
    }
    std::vector<Run> runs;
    for (int i = 0; i < 5; ++i) runs.push_back(launch_once(k, waves, iters, d_out, d_stamps, host, cus));
    std::sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.cycles_per_iter < b.cycles_per_iter; });
    return runs[2];
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<int> wave_counts;
    for (int i = 1; i < argc; ++i) wave_counts.push_back(atoi(argv[i]));
    if (wave_counts.empty()) wave_counts = {1, 6, 8};
    float *d_out; hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float));
    Stamp *d_stamps; hipMalloc(&d_stamps, (size_t)cus * 8 * 4 * sizeof(Stamp));
    std::vector<Stamp> host((size_t)cus * 8 * 4);
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"nominal_clock_mhz\": %.0f,\n", p.gcnArchName, cus, p.clockRate / 1e3);
    printf(" \"method\": \"per-wave s_memtime stamps around ITERS x 128 independent instructions (8 registers round-robin), >= 50 ms per run, ramped, median of 5 runs; per run the LAST-finishing wave (p99) gives the SIMD busy cycles for the work of its W waves; cost = those cycles / (W x instructions per wave); the scalar loop control is subtracted for W = 1 only (it hides behind other waves otherwise)\",\n");
    printf(" \"results\": {\n");
    bool first_w = true;
    for (int w : wave_counts) {
        if (w < 1 || w > 8) continue;
        const Run empty = measure(k_empty, w, d_out, d_stamps, host, cus);
        printf("%s  \"waves_per_simd_%d\": {\"empty_loop_cycles_per_iteration_per_wave\": %.3f, \"ops\": {\n", first_w ? "" : ",\n", w, empty.cycles_per_iter);
        first_w = false;
        bool first = true;
        auto report = [&](const char *name, const Run &r, int per_iter) {
            // W waves share the SIMD: the SIMD issued W * per_iter instructions in cycles_per_iter (minus the loop control)
            // loop control (scalar) hides behind the other waves' VALU as soon as W > 1: subtract it for W == 1 only
            const double cost = (r.cycles_per_iter - (w == 1 ? empty.cycles_per_iter : 0.0)) / ((double)w * per_iter);
            const double raw = r.cycles_per_iter / ((double)w * per_iter);
            printf("%s   \"%s\": {\"cycles\": %.3f, \"cycles_before_subtracting_loop\": %.3f, \"run_ms\": %.1f, \"effective_clock_ghz\": %.3f}",
                   first ? "" : ",\n", name, cost, raw, r.ms, r.clock_ghz);
            first = false;
        };
#define RUN(k, name) report(name, measure(k, w, d_out, d_stamps, host, cus), 128)
#define RUN8(k, name) report(name, measure(k, w, d_out, d_stamps, host, cus), 8)
        RUN(k_add_f32, "v_add_f32"); RUN(k_sub_f32, "v_sub_f32"); RUN(k_mul_f32, "v_mul_f32"); RUN(k_mul_f32_s, "v_mul_f32 (sgpr src)");
        RUN(k_fma_f32, "v_fma_f32"); RUN(k_fmac_f32, "v_fmac_f32"); RUN(k_fma_f32_sgpr, "v_fma_f32 (sgpr srcs)");
        RUN(k_max_f32, "v_max_f32"); RUN(k_add_f32_e64, "v_add_f32_e64 (neg mod)");
        RUN(k_pk_mul_f32, "v_pk_mul_f32"); RUN(k_pk_add_f32, "v_pk_add_f32"); RUN(k_pk_fma_f32, "v_pk_fma_f32");
        RUN(k_mul_f64, "v_mul_f64"); RUN(k_add_f64, "v_add_f64"); RUN(k_fma_f64, "v_fma_f64");
        RUN8(k_cvt_f64_f32, "v_cvt_f64_f32 / v_cvt_f32_f64"); RUN(k_cvt_f32_i32, "v_cvt_f32_i32");
        RUN(k_sqrt_f32, "v_sqrt_f32"); RUN(k_rcp_f32, "v_rcp_f32");
        RUN(k_mov_b32, "v_mov_b32"); RUN(k_cndmask, "v_cndmask_b32 (vcc)"); RUN(k_cndmask_s, "v_cndmask_b32_e64 (sgpr mask)");
        RUN(k_cmp_f32, "v_cmp_lt_f32 (vcc)"); RUN(k_cmp_f32_s, "v_cmp_lt_f32_e64 (sgpr dst)");
        RUN(k_add_u32, "v_add_u32"); RUN(k_lshrrev, "v_lshrrev_b32"); RUN(k_xor_b32, "v_xor_b32"); RUN(k_and_b32, "v_and_b32");
        RUN(k_lshl_add, "v_lshl_add_u32"); RUN(k_alignbit, "v_alignbit_b32"); RUN(k_mul_lo_u32, "v_mul_lo_u32"); RUN(k_bfe_u32, "v_bfe_u32");
        RUN(k_nop, "s_nop 0"); RUN8(k_lds_bcast, "ds_read_b128 broadcast + v_add_f32");
        printf("\n  }}");
        fflush(stdout);
    }
    printf("\n }\n}\n");
    return 0;
}
