// ptmi_kernels.hip -- hand-written gfx950 kernels for the `render` hot path of
// robbert-vdh/haskell-path-tracer (src/Scene/Trace.hs:135-200).
//
// Shape of the main kernel (render Inline, Trace.hs:193-200 + 344-383):
//   * one lane per pixel; the seven state planes are read once and written once per launch,
//     coalesced (x fastest), whatever the sample count;
//   * the sample loop AND the bounce loop live in the kernel.  A lane whose path ends starts
//     its pixel's next sample at once ("regeneration"), so the 64 lanes of a wave stay busy
//     although paths end after different numbers of bounces -- the per-pixel order of RNG
//     draws and of floating-point additions is exactly that of n_spp successive `render` calls;
//   * the primitive list is staged into LDS once per workgroup and read as wave-wide
//     broadcasts (every lane walks the same primitive at the same time);
//   * no MFMA: the work is scalar-per-lane f32/f64 VALU with divergent control flow.
#include "ptmi_kernels.h"

namespace ptmi {

namespace {

constexpr int kBlock = 256;

struct HitSel { float t; int idx; bool just; };

// Correctly rounded binary32 square root (== IEEE sqrtf, which is what the reference's `sqrt`
// lowers to) without the compiler's always-on denormal scaling: v_sqrt_f32 is within 1 ulp, two
// exact FMA residuals pick the neighbour.  The residual test needs x == 0 or x >= 2^-96; if ANY
// lane of the wave holds a smaller positive x the whole wave takes the compiler's scaled sequence.
__device__ __forceinline__ float sqrt_rn(float x)
{
    const bool tiny = (f2u(x) - 1u) < (0x0f800000u - 1u);          // 0 < x < 2^-96
    if (__builtin_expect(__any(tiny), 0)) return __builtin_sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = u2f(f2u(s) - 1u), s_up = u2f(f2u(s) + 1u);
    const float e_dn = __builtin_fmaf(-s_dn, s, x);
    const float e_up = __builtin_fmaf(-s_up, s, x);
    float r = (e_dn <= 0.0f) ? s_dn : s;
    r = (e_up > 0.0f) ? s_up : r;
    return r;
}

// checkHit (Trace.hs:443-447): mapScene over spheres ++ planes (Util.hs:156-158), then
// expMinWith (Util.hs:171-178): left fold keeping the accumulated element iff keyA <= keyB.
// The reference builds every hit record and selects; selecting the index first and building
// one record afterwards gives the same value.
//
// Shape for the SIMD: the cheap part of every test (16 f32 operations for a sphere) runs for all
// lanes; the square root / division and the fold update run only when some lane of the wave can
// still be hit (wave-uniform branches on __any), which is the common case to skip once rays are
// incoherent and primitives are small.  best_key starts as NaN so that element 0 always replaces
// the accumulator (`NaN <= key` is false), which is expMinWith seeding the fold with its head.
template <typename ScenePtr>
__device__ __forceinline__ HitSel check_hit(ScenePtr S, int ns, int np, V3 o, V3 d)
{
    HitSel best; best.t = 0.0f; best.idx = 0; best.just = false;
    float best_key = __builtin_nanf("");
    float4 g = S[0];
    for (int i = 0; i < ns; ++i) {
        const float4 g_next = S[i + 1];                      // prefetch; S has a readable tail element
        // distanceTo @Sphere (Intersection.hs:39-48)
        const V3 l = mk(g.x, g.y, g.z) - o;
        const float tca = dot(l, d);
        const float d2 = dot(l, l) - (tca * tca);
        const float x = g.w - d2;                            // rad ** 2 - d2 (rad ** 2 squared at upload)
        // Nothing iff tca < 0 || d2 > rad**2 || t < 0;  d2 > r2 <=> r2 - d2 < 0 (exact: gradual underflow)
        const bool cand = !(tca < 0.0f) && !(x < 0.0f);
        if (__any(cand)) {
            const float t = tca - sqrt_rn(x);                // min t0 t1 == t0 (thc >= 0 or NaN)
            const bool just = cand && !(t < 0.0f);
            const float key = just ? t : kInfinite;          // maybe infinite fst
            if (!(best_key <= key)) { best_key = key; best.t = t; best.idx = i; best.just = just; }
        } else if (__any(!(best_key <= kInfinite))) {        // a Nothing still replaces a NaN / +inf key
            if (!(best_key <= kInfinite)) { best_key = kInfinite; best.t = 0.0f; best.idx = i; best.just = false; }
        }
        g = g_next;
    }
    for (int j = 0; j < np; ++j) {
        // distanceTo @Plane (Intersection.hs:57-62); g holds (px, py, pz, 0)
        const float4 gn = S[ns + 2 * j + 1];
        const float4 g_next = S[ns + 2 * j + 2];
        const V3 nor = mk(gn.x, gn.y, gn.z);
        const float denom = dot(d, nor);
        const bool cand = !(denom > 1e-6f);
        if (__any(cand)) {
            const float t = dot(mk(g.x, g.y, g.z) - o, nor) / denom;
            const bool just = cand && !(t < 0.0f);
            const float key = just ? t : kInfinite;
            if (!(best_key <= key)) { best_key = key; best.t = t; best.idx = ns + j; best.just = just; }
        } else if (__any(!(best_key <= kInfinite))) {
            if (!(best_key <= kInfinite)) { best_key = kInfinite; best.t = 0.0f; best.idx = ns + j; best.just = false; }
        }
        g = g_next;
    }
    return best;
}

// hit (Intersection.hs:29-32) + normal (:50 / :64) for the selected primitive
template <typename ScenePtr>
__device__ __forceinline__ void hit_record(ScenePtr S, int ns, int idx, V3 o, V3 d, float t,
                                           V3 &hit_pos, V3 &normal)
{
    hit_pos = o + scale_r(d, t);
    if (idx < ns) {
        const float4 g = S[idx];
        normal = normalize(hit_pos - mk(g.x, g.y, g.z));
    } else {
        const float4 gn = S[ns + 2 * (idx - ns) + 1];
        normal = mk(gn.x, gn.y, gn.z);
    }
}

// genVec (Util.hs:114-118) for the device: component = (random * 2.0) - 1.0 with
// random = (float(int32 w) * 2^-32 + 0.5) + 2^-33.  Doubling is exact and commutes with rounding
// here (no value leaves the normal range), so the doubled form below is the same binary32 value
// with one multiplication less: ((I * 2^-31 + 1.0) + 2^-32) - 1.0.
__device__ __forceinline__ float gen_component(Sfc32 &seed)
{
    const float i = (float)(int32_t)sfc32_next(seed);
    return ((i * 4.656612873077392578125e-10f + 1.0f) + 2.3283064365386962890625e-10f) - 1.0f;
}

// computeRay (Trace.hs:374-383) + calcNextRay (Trace.hs:394-435), both BRDF arms evaluated
// through selects so that Matte and Glossy lanes of one wave do not serialise.
template <typename ScenePtr>
__device__ __forceinline__ void shade(ScenePtr M, int idx, V3 hit_pos, V3 normal,
                                      V3 &o, V3 &d, V3 &throughput, V3 &result, Sfc32 &seed)
{
    const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
    const V3 color = mk(ma.x, ma.y, ma.z);
    const float illuminance = ma.w;
    const bool matte = f2u(mb.x) == 0u;
    const float p_over_pi = mb.z;          // p / pi            (Trace.hs:411), divided at upload
    const float half_k_glossy = mb.w;      // 0.5 * (1 - p)     (Trace.hs:424 and Util.hs:62-67), exact halving

    const V3 emittance = scale_r(color, illuminance);
    V3 rv;
    rv.x = gen_component(seed); rv.y = gen_component(seed); rv.z = gen_component(seed);
    // Matte:  rotate (anglesToQuaternion $ pi *^ rv) iNormal
    // Glossy: rotate (anglesToQuaternion $ (1 - p) *^ rv) reflection
    // anglesToQuaternion halves every angle; (k * rv) * 0.5 == (0.5 k) * rv bit for bit (power-of-two scaling).
    const float ia = dot(d, normal);
    const V3 reflection = d - scale_l(2.0f * ia, normal);
    const V3 axis = matte ? normal : reflection;
    const float hk = matte ? 0.5f * kPi : half_k_glossy;
    const V3 next = rotate(quaternion_from_half_angles(hk * rv.x, hk * rv.y, hk * rv.z), axis);
    const float nd = dot(next, axis);
    const float brdf = matte ? p_over_pi * nd : __builtin_fmaxf(0.0f, nd);
    constexpr float next_ray_prob = 1.0f / (kPi * 2.0f);

    o = hit_pos + scale_r(next, kEpsilon);
    d = next;
    const V3 tmod = scale_r(color, brdf * next_ray_prob);
    result = result + (emittance * throughput);
    throughput = throughput * tmod;
}

__device__ __forceinline__ int global_row(int local_row, int stripe_rows, int n_parts, int part)
{
    return ((local_row / stripe_rows) * n_parts + part) * stripe_rows + local_row % stripe_rows;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned int v)
{
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// ---------------------------------------------------------------------------------------
// render Inline.  LDS_SCENE: primitives staged in LDS (default) or read straight from
// global memory through scalar loads (ablation).  MODE selects the loop shape:
//   kCached      (default) primary hit evaluated once per pixel, two shade rounds per trace round
//   kRegenerate  lanes start their next sample as soon as a path ends, one shade per trace
//   kLockstep    all lanes of the wave run sample s together (what a per-sample launch would do)
// ---------------------------------------------------------------------------------------
enum { kCached = 0, kRegenerate = 1, kLockstep = 2 };

template <bool LDS_SCENE, int MODE>
__global__ void __launch_bounds__(kBlock) render_inline_kernel(const RenderArgs a)
{
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    const long long n_local = (long long)a.rows_local * a.width;
    const long long pixel = (long long)blockIdx.x * kBlock + threadIdx.x;
    unsigned int live = 0;
    if (pixel < n_local) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        if (a.screen_x) { px = a.screen_x[pixel]; py = a.screen_y[pixel]; }

        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);

        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        Sfc32 seed;
        seed.a = a.planes.sa[pixel]; seed.b = a.planes.sb[pixel];
        seed.c = a.planes.sc[pixel]; seed.counter = a.planes.sctr[pixel];

        const int limit = a.bounce_limit, n_spp = a.n_spp;

        if (limit <= 0) {
            // iterate 0: every sample returns (0, seed); new + old
            if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;
        } else if (MODE == kCached) {
            // primaryRays has no sub-pixel jitter (Trace.hs:244-262): every sample of a pixel shoots the
            // same primary ray, so its checkHit + hit are evaluated ONCE per pixel and every sample starts
            // from that record.  A sample then costs k shades and k-1 traces (k = its live bounces).
            // Loop shape: [shade][shade again for lanes whose sample just ended][trace].  A lane that
            // ends a sample in the first shade round starts the next one in the second, so all lanes
            // enter the trace round with a ray and the expensive round runs at full occupancy.
            const HitSel h0 = check_hit(S, ns, np, origin, primary);
            if (!h0.just) {
                if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;     // every sample: result 0, seed untouched
            } else {
                V3 p0, n0;
                hit_record(S, ns, h0.idx, origin, primary, h0.t, p0, n0);
                const int idx0 = h0.idx;
                int s = 0, it = 0, idx = idx0;
                V3 hit_pos = p0, normal = n0;                         // the hit waiting to be shaded
                V3 o = origin, d = primary;                           // the ray that produced it / the next ray
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                bool pending = n_spp > 0, has_ray = false;
                while (pending) {
                    for (int round = 0; round < 2; ++round) {
                        if (pending && !has_ray) {
                            shade(M, idx, hit_pos, normal, o, d, throughput, result, seed);
                            ++it; ++live;
                            // the next prepareRay would freeze the path (Trace.hs:364-365)
                            if (it >= limit || near_zero(throughput)) {
                                acc = result + acc;                   // \(new, seed') (old, _) -> (new + old, seed')
                                ++s; it = 0;
                                throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                                hit_pos = p0; normal = n0; idx = idx0; d = primary;
                                pending = s < n_spp;
                            } else {
                                pending = false; has_ray = true;
                            }
                        }
                    }
                    if (has_ray) {
                        const HitSel h = check_hit(S, ns, np, o, d);
                        has_ray = false;
                        if (h.just) {
                            hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                            idx = h.idx;
                            pending = true;
                        } else {
                            acc = result + acc;
                            ++s; it = 0;
                            throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                            hit_pos = p0; normal = n0; idx = idx0; d = primary;
                            pending = s < n_spp;
                        }
                    }
                }
            }
        } else if (MODE == kRegenerate) {
            int s = 0, it = 0;
            V3 o = origin, d = primary;
            V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
            while (s < n_spp) {
                const HitSel h = check_hit(S, ns, np, o, d);
                bool end = true;
                if (h.just) {
                    V3 hit_pos, normal;
                    hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                    shade(M, h.idx, hit_pos, normal, o, d, throughput, result, seed);
                    ++it; ++live;
                    // the next prepareRay would freeze the path (Trace.hs:364-365)
                    end = (it >= limit) || near_zero(throughput);
                }
                if (end) {
                    acc = result + acc;                      // \(new, seed') (old, _) -> (new + old, seed')
                    ++s; it = 0;
                    o = origin; d = primary;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                }
            }
        } else {
            for (int s = 0; s < n_spp; ++s) {
                V3 o = origin, d = primary;
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                for (int it = 0; it < limit; ++it) {
                    if (near_zero(throughput)) break;
                    const HitSel h = check_hit(S, ns, np, o, d);
                    if (!h.just) break;
                    V3 hit_pos, normal;
                    hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                    shade(M, h.idx, hit_pos, normal, o, d, throughput, result, seed);
                    ++live;
                }
                acc = result + acc;
            }
        }

        a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
        a.planes.sa[pixel] = seed.a; a.planes.sb[pixel] = seed.b;
        a.planes.sc[pixel] = seed.c; a.planes.sctr[pixel] = seed.counter;
    }

    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter, total);
    }
}

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

inline unsigned int blocks_for(long long n) { return (unsigned int)((n + kBlock - 1) / kBlock); }

}  // namespace

hipError_t launch_render_inline(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local)), block(kBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    switch (variant) {
    case 1:  hipLaunchKernelGGL((render_inline_kernel<false, kCached>), grid, block, 0, stream, a); break;
    case 2:  hipLaunchKernelGGL((render_inline_kernel<true, kLockstep>), grid, block, lds, stream, a); break;
    case 3:  hipLaunchKernelGGL((render_inline_kernel<true, kRegenerate>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((render_inline_kernel<true, kCached>), grid, block, lds, stream, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_render_streams(const RenderArgs &, int, hipStream_t)
{
    return hipErrorNotSupported;
}

hipError_t launch_seed(Planes p, int width, int rows_local, int stripe_rows, int n_parts, int part,
                       uint64_t seed0, bool clear_color, hipStream_t stream)
{
    const long long n = (long long)rows_local * width;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(seed_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, p, width, rows_local,
                       stripe_rows, n_parts, part, seed0, clear_color ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_create_with(Planes p, const uint32_t *w0, const uint32_t *w1, const uint32_t *w2,
                              int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(create_with_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, p, w0, w1, w2, (long long)n);
    return hipGetLastError();
}

hipError_t launch_eval_sphere(const float *spheres10, const float *rays, int n,
                              int32_t *is_just, float *t, float *normalp, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_sphere_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, spheres10, rays, n, is_just, t, normalp);
    return hipGetLastError();
}

hipError_t launch_eval_plane(const float *planes12, const float *rays, int n,
                             int32_t *is_just, float *t, float *normalp, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_plane_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, planes12, rays, n, is_just, t, normalp);
    return hipGetLastError();
}

hipError_t launch_eval_sincos(const float *x, int n, float *s, float *c, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_sincos_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, x, n, s, c);
    return hipGetLastError();
}

}  // namespace ptmi
