// lds_dma_probe.hip -- what global_load_lds_dword does with its operands on gfx950, asked of the hardware:
//   hipcc --offload-arch=gfx950 -O2 -o build/lds_dma_probe tools/lds_dma_probe.hip && build/lds_dma_probe
// A wave loads word (4 lane + k) of a table through "global_load_lds_dword v[..], off offset:4k" with M0 = row k of an LDS array, for the
// lanes of an exec mask, waits with s_waitcnt vmcnt(0), and copies the array out.  Printed: where each word landed.  Findings (round 4):
// lane i's word lands at M0 + offset + 4 i (the offset field counts for the memory address AND the LDS address); lanes outside the exec mask
// write nothing; vmcnt covers the LDS write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t *table, uint32_t *out, unsigned long long mask, int compensate)
{
    __shared__ uint32_t rows[4][64];
    for (int k = 0; k < 4; ++k) rows[k][threadIdx.x] = 0xdead0000u + (uint32_t)k;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&rows[0][0];
    const uint32_t *mine = table + 4 * threadIdx.x;
    const uint32_t step = compensate ? 0xfcu : 0x100u;
    if ((mask >> threadIdx.x) & 1ull) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dword %0, off\n\ts_add_u32 m0, m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dword %0, off offset:4\n\ts_add_u32 m0, m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dword %0, off offset:8\n\ts_add_u32 m0, m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dword %0, off offset:12"
                     :: "v"(mine), "s"(base), "s"(step) : "memory", "m0", "scc");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 4; ++k) out[k * 64 + threadIdx.x] = rows[k][threadIdx.x];
}

int main()
{
    std::vector<uint32_t> table(256), out(256);
    for (int i = 0; i < 256; ++i) table[i] = (uint32_t)i;          // word k of lane i = 4 i + k
    uint32_t *d_table, *d_out;
    if (hipMalloc(&d_table, 1024) != hipSuccess || hipMalloc(&d_out, 1024) != hipSuccess) return 2;
    hipMemcpy(d_table, table.data(), 1024, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int compensate = 0; compensate < 2; ++compensate) {
        for (unsigned long long mask : {~0ull, 0x00000000000000f5ull}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_table, d_out, mask, compensate);
            if (hipDeviceSynchronize() != hipSuccess) return 3;
            hipMemcpy(out.data(), d_out, 1024, hipMemcpyDeviceToHost);
            int as_rows = 0, shifted = 0, untouched_ok = 0, other = 0;
            for (int k = 0; k < 4; ++k)
                for (int i = 0; i < 64; ++i) {
                    const uint32_t v = out[k * 64 + i];
                    const bool active = (mask >> i) & 1ull;
                    if (active && v == (uint32_t)(4 * i + k)) ++as_rows;
                    else if (!active && v == 0xdead0000u + (uint32_t)k) ++untouched_ok;
                    else if (v < 256u) ++shifted;
                    else ++other;
                }
            printf("{\"m0_step\": \"%s\", \"exec\": \"%016llx\", \"words_in_their_row_and_column\": %d, \"inactive_untouched\": %d, \"words_elsewhere\": %d, \"other\": %d}\n",
                   compensate ? "0xfc" : "0x100", mask, as_rows, untouched_ok, shifted, other);
            if (compensate && (shifted || other)) ++bad_total;
        }
    }
    return bad_total ? 1 : 0;
}
