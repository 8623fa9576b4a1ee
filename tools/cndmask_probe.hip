// cndmask_probe.hip -- why did v_cndmask_b32 (VCC form) measure ~23 cycles in valu_rates?  Variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define BODY8(OP) \
    asm volatile(OP : "+v"(a0) : "v"(b)); asm volatile(OP : "+v"(a1) : "v"(b)); \
    asm volatile(OP : "+v"(a2) : "v"(b)); asm volatile(OP : "+v"(a3) : "v"(b)); \
    asm volatile(OP : "+v"(a4) : "v"(b)); asm volatile(OP : "+v"(a5) : "v"(b)); \
    asm volatile(OP : "+v"(a6) : "v"(b)); asm volatile(OP : "+v"(a7) : "v"(b));
#define K(name, PRE, OP) \
__global__ void __launch_bounds__(256) name(float *out, float seed) { \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5f; \
    asm volatile(PRE :: "v"(a0), "v"(b)); \
    for (int i = 0; i < ITER; ++i) { BODY8(OP) } \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }

K(k_vcc_uninit, "s_nop 0", "v_cndmask_b32 %0, %0, %1, vcc")
K(k_vcc_cmp, "v_cmp_lt_f32 vcc, %0, %1", "v_cndmask_b32 %0, %0, %1, vcc")
K(k_vcc_smov, "s_mov_b64 vcc, 0x5555", "v_cndmask_b32 %0, %0, %1, vcc")
K(k_vcc_allones, "s_mov_b64 vcc, -1", "v_cndmask_b32 %0, %0, %1, vcc")
K(k_vcc_zero, "s_mov_b64 vcc, 0", "v_cndmask_b32 %0, %0, %1, vcc")
K(k_sgpr, "s_mov_b64 s[10:11], 0x5555", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
K(k_vcc_e64, "s_mov_b64 vcc, 0x5555", "v_cndmask_b32_e64 %0, %0, %1, vcc")
K(k_distinct_dst, "s_mov_b64 vcc, 0x5555", "v_cndmask_b32 %0, %1, %1, vcc")
K(k_cmp_then_sel, "s_nop 0", "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc")
K(k_cmp_then_sel_s, "s_nop 0", "v_cmp_lt_f32_e64 s[10:11], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
K(k_add, "s_nop 0", "v_add_f32 %0, %0, %1")
K(k_cmp_3sel, "s_nop 0", "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc")
K(k_cmp_3sel_e64, "s_nop 0", "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %0, %1, %0, vcc\n v_cndmask_b32_e64 %0, %0, %1, vcc")
K(k_cmp_2sel, "s_nop 0", "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc")
K(k_sel_add_sel, "s_mov_b64 vcc, 0x5555", "v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc\n v_add_f32 %0, %0, %1")
K(k_sel_nop_sel, "s_mov_b64 vcc, 0x5555", "v_cndmask_b32 %0, %0, %1, vcc\n s_nop 0\n v_cndmask_b32 %0, %1, %0, vcc\n s_nop 0")

template <typename Kn> void run(Kn k, const char *name, float *d, double n_instr) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 1.0f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 1.0f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %6.2f cycles per op-group per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / (8.0 * ITER * 8.0) );
}
int main() {
    float *d; (void)hipMalloc(&d, 2048 * 256 * 4);
    run(k_add, "v_add_f32", d, 1);
    run(k_vcc_uninit, "cndmask vcc uninit", d, 1);
    run(k_vcc_cmp, "cndmask vcc after v_cmp", d, 1);
    run(k_vcc_smov, "cndmask vcc=0x5555", d, 1);
    run(k_vcc_allones, "cndmask vcc=-1", d, 1);
    run(k_vcc_zero, "cndmask vcc=0", d, 1);
    run(k_vcc_e64, "cndmask_e64 vcc=0x5555", d, 1);
    run(k_sgpr, "cndmask_e64 s[10:11]", d, 1);
    run(k_distinct_dst, "cndmask vcc src0=src1", d, 1);
    run(k_cmp_then_sel, "v_cmp vcc + cndmask vcc", d, 1);
    run(k_cmp_then_sel_s, "v_cmp sgpr + cndmask sgpr", d, 1);
    run(k_cmp_3sel, "cmp + 3 cndmask e32", d, 1);
    run(k_cmp_3sel_e64, "cmp + 3 cndmask e64", d, 1);
    run(k_cmp_2sel, "cmp + 2 cndmask e32", d, 1);
    run(k_sel_add_sel, "2x (cndmask e32 + add)", d, 1);
    run(k_sel_nop_sel, "2x (cndmask e32 + s_nop)", d, 1);
    return 0;
}
