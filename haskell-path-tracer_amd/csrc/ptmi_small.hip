// ptmi_small.hip -- the HBM-streaming kernels around the render kernels: genSeeds / createWith / initialOutput / reseed
// (src/Util.hs:122-135, 204-205), present (app/Main.hs:351, app/assets/fs.glsl:12), the group read-out's stitch, the cost order of
// the tiled kernels' dispatch, and the point queries behind the reference's unit-test surface (test/Scene/Intersection/Tests.hs).
#include "ptmi_device.h"

namespace ptmi {

namespace {

// ---------------------------------------------------------------------------------------
// genSeeds / createWith / initialOutput / reseed  (src/Util.hs:122-135, 204-205)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) seed_kernel(Planes p, int width, int rows_local, int stripe_rows,
                                                      int n_parts, int part, uint64_t seed0, int clear_color)
{
    const long long n_local = (long long)rows_local * width;
    const long long pixel = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= n_local) return;
    const int local_row = (int)(pixel / width);
    const int col = (int)(pixel - (long long)local_row * width);
    const uint64_t index = (uint64_t)global_row(local_row, stripe_rows, n_parts, part) * (uint64_t)width + (uint64_t)col;
    const Sfc32 s = sfc32_seed3(seed_word(seed0, index, 0), seed_word(seed0, index, 1), seed_word(seed0, index, 2));
    p.sa[pixel] = s.a; p.sb[pixel] = s.b; p.sc[pixel] = s.c; p.sctr[pixel] = s.counter;
    if (clear_color) { p.r[pixel] = 0.0f; p.g[pixel] = 0.0f; p.b[pixel] = 0.0f; }
}

__global__ void __launch_bounds__(kBlock) create_with_kernel(Planes p, const uint32_t *w0, const uint32_t *w1,
                                                             const uint32_t *w2, long long n)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Sfc32 s = sfc32_seed3(w0[i], w1[i], w2[i]);
    p.sa[i] = s.a; p.sb[i] = s.b; p.sc[i] = s.c; p.sctr[i] = s.counter;
}

// ---------------------------------------------------------------------------------------
// present: interleave + divide by the iteration count (app/Main.hs:351, app/assets/fs.glsl:12) and the
// framebuffer's float -> unorm8 conversion.  HBM-streaming: 12 B read and 12 (+4) B written per pixel.
// Four pixels per lane: three 16-byte plane loads, three 16-byte interleaved stores, one 16-byte RGBA store.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t unorm8(float c)
{
    const float x = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);    // NaN -> 0, as a GL clamp does
    return (uint32_t)(x * 255.0f + 0.5f);
}

__global__ void __launch_bounds__(kBlock) present_kernel(Planes p, long long n, float count, float *rgb, uint32_t *rgba, int aligned16)
{
    const long long q = ((long long)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (q >= n) return;
    const bool wide = aligned16 && q + 3 < n;               // caller-owned planes need not be 16-byte aligned
    float r[4], g[4], b[4];
    if (wide) {
        const float4 vr = *reinterpret_cast<const float4 *>(p.r + q);
        const float4 vg = *reinterpret_cast<const float4 *>(p.g + q);
        const float4 vb = *reinterpret_cast<const float4 *>(p.b + q);
        r[0] = vr.x; r[1] = vr.y; r[2] = vr.z; r[3] = vr.w;
        g[0] = vg.x; g[1] = vg.y; g[2] = vg.z; g[3] = vg.w;
        b[0] = vb.x; b[1] = vb.y; b[2] = vb.z; b[3] = vb.w;
    } else {
        for (int k = 0; k < 4; ++k) { const bool in = q + k < n; r[k] = in ? p.r[q + k] : 0.0f; g[k] = in ? p.g[q + k] : 0.0f; b[k] = in ? p.b[q + k] : 0.0f; }
    }
    for (int k = 0; k < 4; ++k) { r[k] = r[k] / count; g[k] = g[k] / count; b[k] = b[k] / count; }
    if (wide) {
        if (rgb) {
            float4 *o = reinterpret_cast<float4 *>(rgb + 3 * q);
            o[0] = float4{r[0], g[0], b[0], r[1]};
            o[1] = float4{g[1], b[1], r[2], g[2]};
            o[2] = float4{b[2], r[3], g[3], b[3]};
        }
        if (rgba) {
            uint4 v;
            v.x = unorm8(r[0]) | unorm8(g[0]) << 8 | unorm8(b[0]) << 16 | 0xff000000u;
            v.y = unorm8(r[1]) | unorm8(g[1]) << 8 | unorm8(b[1]) << 16 | 0xff000000u;
            v.z = unorm8(r[2]) | unorm8(g[2]) << 8 | unorm8(b[2]) << 16 | 0xff000000u;
            v.w = unorm8(r[3]) | unorm8(g[3]) << 8 | unorm8(b[3]) << 16 | 0xff000000u;
            *reinterpret_cast<uint4 *>(rgba + q) = v;
        }
    } else {
        for (int k = 0; k < 4 && q + k < n; ++k) {
            if (rgb) { rgb[3 * (q + k)] = r[k]; rgb[3 * (q + k) + 1] = g[k]; rgb[3 * (q + k) + 2] = b[k]; }
            if (rgba) rgba[q + k] = unorm8(r[k]) | unorm8(g[k]) << 8 | unorm8(b[k]) << 16 | 0xff000000u;
        }
    }
}

// ---------------------------------------------------------------------------------------
// stitch: a member's rows (local order) into the whole image's planes (group read-out).  16 bytes per lane when aligned.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) stitch_kernel(const float *src, int rows, int width, int stripe_rows, int n_parts, int part,
                                                        float *r, float *g, float *b)
{
    const long long n = (long long)rows * width;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int local_row = (int)(i / width);
    const int col = (int)(i - (long long)local_row * width);
    const long long o = (long long)global_row(local_row, stripe_rows, n_parts, part) * width + col;
    r[o] = src[i]; g[o] = src[n + i]; b[o] = src[2 * n + i];
}

// ---------------------------------------------------------------------------------------
// point queries: the reference's unit-test surface (test/Scene/Intersection/Tests.hs)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) eval_sphere_kernel(const float *sph, const float *rays, int n,
                                                             int32_t *is_just, float *t_out, float *normalp)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *s = sph + 10 * (size_t)i;
    float4 g[2]; g[0].x = s[0]; g[0].y = s[1]; g[0].z = s[2]; g[0].w = s[3] * s[3];
    g[1] = g[0];                                             // check_hit prefetches one element ahead
    const V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    const V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    const HitSel h = check_hit(g, 1, 0, o, d);
    is_just[i] = h.just ? 1 : 0;
    t_out[i] = h.just ? h.t : 0.0f;
    if (normalp) {
        V3 hp = mk(0, 0, 0), nr = mk(0, 0, 0);
        if (h.just) hit_record(g, 1, 0, o, d, h.t, hp, nr);
        float *q = normalp + 6 * (size_t)i;
        q[0] = hp.x; q[1] = hp.y; q[2] = hp.z; q[3] = nr.x; q[4] = nr.y; q[5] = nr.z;
    }
}

__global__ void __launch_bounds__(kBlock) eval_plane_kernel(const float *pl, const float *rays, int n,
                                                            int32_t *is_just, float *t_out, float *normalp)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *s = pl + 12 * (size_t)i;
    float4 g[3];
    g[0].x = s[0]; g[0].y = s[1]; g[0].z = s[2]; g[0].w = 0.0f;
    g[1].x = s[3]; g[1].y = s[4]; g[1].z = s[5]; g[1].w = 0.0f;
    g[2] = g[0];                                             // check_hit prefetches one element ahead
    const V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    const V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    const HitSel h = check_hit(g, 0, 1, o, d);
    is_just[i] = h.just ? 1 : 0;
    t_out[i] = h.just ? h.t : 0.0f;
    if (normalp) {
        V3 hp = mk(0, 0, 0), nr = mk(0, 0, 0);
        if (h.just) hit_record(g, 0, 0, o, d, h.t, hp, nr);
        float *q = normalp + 6 * (size_t)i;
        q[0] = hp.x; q[1] = hp.y; q[2] = hp.z; q[3] = nr.x; q[4] = nr.y; q[5] = nr.z;
    }
}

__global__ void __launch_bounds__(kBlock) eval_sincos_kernel(const float *x, int n, float *s, float *c)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincos(x[i], sn, cs);
    s[i] = sn; c[i] = cs;
}

// Quads by decreasing recorded cost, in 256 cost classes (order inside a class does not matter): one workgroup,
// LDS histogram, scan, scatter.  n is a few thousand to a few ten thousand.  Every cost is read ONCE and its class
// kept in `cls`, so the result is a permutation even if somebody were still adding to the costs.
// tail_start (optional): the first position of the order's TAIL -- the cheapest classes that together hold at most tail_permille
// thousandths of the recorded cost (the quads without any cost among them), rounded up to a multiple of 8 positions.
__global__ void __launch_bounds__(1024) quad_order_kernel(const unsigned int *cost, unsigned int *order, unsigned int *cls, unsigned int n,
                                                          unsigned int *tail_start, unsigned int tail_permille)
{
    __shared__ unsigned int hist[256], start[256], top;
    __shared__ unsigned long long class_cost[256];
    if (threadIdx.x < 256) { hist[threadIdx.x] = 0; class_cost[threadIdx.x] = 0ull; }
    if (threadIdx.x == 0) top = 1;
    __syncthreads();
    unsigned int mine = 0;
    for (unsigned int i = threadIdx.x; i < n; i += 1024) { const unsigned int c = cost[i]; cls[i] = c; mine = c > mine ? c : mine; }
    atomicMax(&top, mine);
    __syncthreads();
    const unsigned long long scale = top;
    for (unsigned int i = threadIdx.x; i < n; i += 1024) {                 // a thread revisits only its own elements
        const unsigned int c = cls[i];
        const unsigned int b = 255u - (unsigned int)(((unsigned long long)c * 255ull) / scale);   // 0 = most expensive
        cls[i] = b;
        atomicAdd(&hist[b], 1u);
        if (tail_start && c) atomicAdd(&class_cost[b], (unsigned long long)c);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int b = 0; b < 256; ++b) { start[b] = run; run += hist[b]; }
        if (tail_start) {
            unsigned long long total = 0, tail = 0;
            for (int b = 0; b < 256; ++b) total += class_cost[b];
            int cut = 256;                                     // classes cut .. 255 are the tail
            while (cut > 0 && (tail + class_cost[cut - 1]) * 1000ull <= total * (unsigned long long)tail_permille) { --cut; tail += class_cost[cut]; }
            unsigned int t = (cut < 256 && total) ? start[cut] : n;
            t = (t + 7u) & ~7u;
            *tail_start = t < n ? t : n;
        }
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n; i += 1024) order[atomicAdd(&start[cls[i]], 1u)] = i;
}

}  // namespace

// the default (variant 0 = auto) takes the cost order; the explicit variants, 13 and 17 included, keep the image order
bool uses_quad_order(const RenderArgs &a, int algorithm_inline, int variant)
{
    if (variant != 0 || !tiles_pay(a) || a.screen_x) return false;
    return algorithm_inline ? (a.bounce_limit > 0 && a.n_spp > 0) : true;
}


hipError_t launch_quad_order(const unsigned int *cost, unsigned int *order, unsigned int *cls, unsigned int n, unsigned int *tail_start,
                             unsigned int tail_permille, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    return launch(quad_order_kernel, dim3(1), dim3(1024), 0, stream, cost, order, cls, n, tail_start, tail_permille);
}


hipError_t launch_seed(Planes p, int width, int rows_local, int stripe_rows, int n_parts, int part,
                       uint64_t seed0, bool clear_color, hipStream_t stream)
{
    const long long n = (long long)rows_local * width;
    if (n <= 0) return hipSuccess;
    return launch(seed_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, p, width, rows_local,
                  stripe_rows, n_parts, part, seed0, clear_color ? 1 : 0);
}

hipError_t launch_create_with(Planes p, const uint32_t *w0, const uint32_t *w1, const uint32_t *w2,
                              int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    return launch(create_with_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, p, w0, w1, w2, (long long)n);
}

hipError_t launch_eval_sphere(const float *spheres10, const float *rays, int n,
                              int32_t *is_just, float *t, float *normalp, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    return launch(eval_sphere_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, spheres10, rays, n, is_just, t, normalp);
}

hipError_t launch_eval_plane(const float *planes12, const float *rays, int n,
                             int32_t *is_just, float *t, float *normalp, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    return launch(eval_plane_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, planes12, rays, n, is_just, t, normalp);
}

hipError_t launch_present(Planes p, long long n, int iterations, float *rgb, uint32_t *rgba, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const uintptr_t bits = (uintptr_t)p.r | (uintptr_t)p.g | (uintptr_t)p.b | (uintptr_t)rgb | (uintptr_t)rgba;
    return launch(present_kernel, dim3(blocks_for((n + 3) / 4)), dim3(kBlock), 0, stream, p, n, (float)iterations, rgb, rgba,
                  (bits & 15u) == 0 ? 1 : 0);
}

hipError_t launch_stitch(const float *src, int rows, int width, int stripe_rows, int n_parts, int part,
                         float *r, float *g, float *b, hipStream_t stream)
{
    const long long n = (long long)rows * width;
    if (n <= 0) return hipSuccess;
    return launch(stitch_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, src, rows, width, stripe_rows, n_parts, part, r, g, b);
}

hipError_t launch_eval_sincos(const float *x, int n, float *s, float *c, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    return launch(eval_sincos_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, x, n, s, c);
}


}  // namespace ptmi
