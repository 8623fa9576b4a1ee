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
//   * a shade whose outcome the next prepareRay is certain to freeze only adds its emittance and draws (surely_frozen_after);
//   * every render loop is written so that each LARGE block -- "start the pixel's next sample", "fetch the lineage's next
//     piece of work" -- is expanded once per trip: the sites that end a sample or a lineage only set a per-lane flag, and
//     one block at the top of the next trip acts on it.  (Three inlined copies of such a block cost 3-9 % per kernel.)
//   * no MFMA: the work is scalar-per-lane f32/f64 VALU with divergent control flow.
// Kernels in this file: render_inline_kernel, render_streams_kernel (Streams, one chain per pixel), render_streams_tree_kernel
// (Streams with ray splitting, one tree per pixel), the stream ("wavefront") form of Streams -- streams_primary_kernel,
// streams_pixels_kernel, streams_split_kernel, streams_level_kernel, streams_seeds_kernel, streams_advance_missed_kernel --,
// seed / create_with / present / stitch / quad_order / point-query kernels; with -DPTMI_ABLATIONS also the pooled and persistent
// forms of render Inline.  Compiled a second time with -DPTMI_CONTRACTED_BUILD (render Inline only, a * b + c fused: a
// measurement mode, see the end of the file).
#include "ptmi_kernels.h"

namespace ptmi {

namespace {

constexpr int kBlock = 256;      // small streaming kernels
constexpr size_t kMaxSceneLds = 3 * 1024;  // bytes of staged scene per one-wave workgroup before LDS would cap occupancy (~60 primitives)
constexpr int kRenderBlock = 64; // render kernels: one wave per workgroup, so a finished wave's slot is refilled at once (+1 % on C2)
#ifndef PTMI_FETCH_BATCH
#define PTMI_FETCH_BATCH 1
#endif
constexpr int kFetchBatch = PTMI_FETCH_BATCH;   // lanes that must be idle before the wave fetches pixels
constexpr int kChunk = 64;      // pixels a wave takes from the global counter per atomic (persistent kernel)

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

// v ^/ s for the three components of a sphere normal: three IEEE divisions by ONE denominator.  The compiler's division is
// div_scale (both operands), rcp, two refinements of the reciprocal, quotient, two residual corrections, div_fmas, div_fixup;
// when neither operand needs scaling and nothing is special -- the denominator within [2^-20, 2^20], the numerators at least
// 2^-100 in magnitude (so every quotient is a normal number, >= 2^-120) and, being components of the vector whose length the
// denominator is, not above it -- div_scale is the
// identity, div_fmas a plain fma and div_fixup passes the quotient through, so the same operations with the reciprocal and its
// refinements formed ONCE give the same three quotients bit for bit (18 instead of 33 instructions, one v_rcp_f32 instead of
// three).  If any lane of the wave falls outside (a zero component, a huge sphere, a NaN) the wave takes the compiler's form.
__device__ __forceinline__ V3 div3_by_length(V3 v, float s)
{
#ifdef PTMI_PLAIN_DIVISION
    return div_r(v, s);
#else
    const float amin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z));
    const bool plain = s >= 0x1p-20f && s <= 0x1p20f && amin >= 0x1p-100f;
    if (__builtin_expect(!__all(plain), 0)) return div_r(v, s);
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float r = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
    auto quotient = [&](float a) {
        float q = a * r;
        q = __builtin_fmaf(__builtin_fmaf(-s, q, a), r, q);
        return __builtin_fmaf(__builtin_fmaf(-s, q, a), r, q);
    };
    return mk(quotient(v.x), quotient(v.y), quotient(v.z));
#endif
}

// checkHit (Trace.hs:443-447): mapScene over spheres ++ planes (Util.hs:156-158), then
// expMinWith (Util.hs:171-178): left fold keeping the accumulated element iff keyA <= keyB.
// The reference builds every hit record and selects; selecting the index first and building
// one record afterwards gives the same value.
//
// check_hit_exact is the fold written out literally (every primitive, every lane, no shortcuts).
template <typename ScenePtr>
__device__ __noinline__ HitSel check_hit_exact(ScenePtr S, int ns, int np, V3 o, V3 d)
{
    HitSel best; best.t = 0.0f; best.idx = 0; best.just = false;
    float best_key = 0.0f;
    for (int i = 0; i < ns + np; ++i) {
        bool just; float t;
        if (i < ns) {                                        // distanceTo @Sphere (Intersection.hs:39-48)
            const float4 g = S[i];
            const V3 l = mk(g.x, g.y, g.z) - o;
            const float tca = dot(l, d);
            const float d2 = dot(l, l) - (tca * tca);
            const float thc = __builtin_sqrtf(g.w - d2);
            t = tca - thc;                                   // min t0 t1 == t0
            just = !(tca < 0.0f || d2 > g.w || t < 0.0f);
        } else {                                             // distanceTo @Plane (Intersection.hs:57-62)
            const float4 gp = S[ns + 2 * (i - ns)], gn = S[ns + 2 * (i - ns) + 1];
            const V3 nor = mk(gn.x, gn.y, gn.z);
            const float denom = dot(d, nor);
            t = dot(mk(gp.x, gp.y, gp.z) - o, nor) / denom;
            just = !(denom > 1e-6f || t < 0.0f);
        }
        const float key = just ? t : kInfinite;              // maybe infinite fst
        if (i == 0 || !(best_key <= key)) { best_key = key; best.t = t; best.idx = i; best.just = just; }
    }
    return best;
}

// check_hit is the same fold shaped for the SIMD:
//   * the cheap part of every test (16 f32 operations for a sphere) runs for all lanes; the square
//     root / division and the fold update run only when some lane of the wave can still be hit
//     (wave-uniform branch on __any) -- the common case to skip once rays are incoherent;
//   * best_key starts as NaN so that the first evaluated element always replaces the accumulator
//     (`NaN <= key` is false), which is expMinWith seeding the fold with its head;
//   * a skipped element is a Nothing (key FLT_MAX).  In the literal fold a Nothing replaces the
//     accumulator only when the accumulated key is NaN or +inf; leaving the accumulator alone instead
//     can change the outcome only if the FINAL accumulator is a Just whose key is not < FLT_MAX, so that
//     one case (a ray or primitive with non-finite numbers) is detected at the end and redone literally.
//   * spheres are walked two per trip with the two register sets swapping roles, so the prefetch of
//     the next primitive costs no moves.
template <typename ScenePtr>
__device__ __forceinline__ HitSel check_hit(ScenePtr S, int ns, int np, V3 o, V3 d, unsigned int *diag = nullptr)
{
    float best_key = __builtin_nanf("");
    int best_idx = 0;
    bool best_just = false;

    auto sphere = [&](const float4 g, int i) {
        // distanceTo @Sphere (Intersection.hs:39-48)
        const V3 l = mk(g.x, g.y, g.z) - o;
        const float tca = dot(l, d);
        const float d2 = dot(l, l) - (tca * tca);
        const float x = g.w - d2;                            // rad ** 2 - d2 (rad ** 2 squared at upload)
        // Nothing iff tca < 0 || d2 > rad**2 || t < 0;  d2 > r2 <=> r2 - d2 < 0 (exact: gradual underflow)
        const bool cand = !(tca < 0.0f) && !(x < 0.0f);
#ifdef PTMI_SPHERE_STATS
        {
            const unsigned long long cm = __ballot(cand), am = __ballot(1);
            if (diag && (threadIdx.x & 63) == (int)__builtin_ctzll(am)) {
                atomicAdd(diag + 16, 1u);                                    // sphere tests (per wave)
                if (cm) atomicAdd(diag + 17, 1u);                            // ... that took the square-root path
                atomicAdd(diag + 18, (unsigned int)__builtin_popcountll(cm));   // candidate lanes
                atomicAdd(diag + 19, (unsigned int)__builtin_popcountll(am));   // active lanes
            }
        }
#endif
        if (__any(cand)) {
            const float t = tca - sqrt_rn(x);                // min t0 t1 == t0 (thc >= 0 or NaN)
            const bool just = cand && !(t < 0.0f);
            const float key = just ? t : kInfinite;          // maybe infinite fst
            if (!(best_key <= key)) { best_key = key; best_idx = i; best_just = just; }
        }
    };

    float4 ga = S[0], gb;
    int i = 0;
    for (; i + 1 < ns; i += 2) {
        gb = S[i + 1];
        sphere(ga, i);
        ga = S[i + 2];                                       // S has readable elements past the geometry
        sphere(gb, i + 1);
    }
    if (i < ns) {
        gb = S[i + 1];
        sphere(ga, i);
        ga = gb;
    }
    for (int j = 0; j < np; ++j) {
        // distanceTo @Plane (Intersection.hs:57-62); ga holds (px, py, pz, 0)
        const float4 gn = S[ns + 2 * j + 1];
        const float4 g_next = S[ns + 2 * j + 2];
        const V3 nor = mk(gn.x, gn.y, gn.z);
        const float denom = dot(d, nor);
        const bool cand = !(denom > 1e-6f);
        if (__any(cand)) {
            const float t = dot(mk(ga.x, ga.y, ga.z) - o, nor) / denom;
            const bool just = cand && !(t < 0.0f);
            const float key = just ? t : kInfinite;
            if (!(best_key <= key)) { best_key = key; best_idx = ns + j; best_just = just; }
        }
        ga = g_next;
    }
    if (__builtin_expect(__any(best_just && !(best_key < kInfinite)), 0))
        return check_hit_exact(S, ns, np, o, d);
    HitSel best; best.t = best_key; best.idx = best_idx; best.just = best_just;
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
        // normalize (linear): v unchanged if |v|^2 is within 1e-6 of 0 or 1, else v / sqrt |v|^2
        const V3 v = hit_pos - mk(g.x, g.y, g.z);
        const float len2 = dot(v, v);
        normal = (near_zero(len2) || near_zero(1.0f - len2)) ? v : div3_by_length(v, sqrt_rn(len2));
    } else {
        const float4 gn = S[ns + 2 * (idx - ns) + 1];
        normal = mk(gn.x, gn.y, gn.z);
    }
}

// normal (Intersection.hs:50 / :64) of primitive idx at a hit position computed earlier: the second half of hit_record
template <typename ScenePtr>
__device__ __forceinline__ V3 normal_at(ScenePtr S, int ns, int idx, V3 hit_pos)
{
    if (idx < ns) {
        const float4 g = S[idx];
        const V3 v = hit_pos - mk(g.x, g.y, g.z);
        const float len2 = dot(v, v);
        return (near_zero(len2) || near_zero(1.0f - len2)) ? v : div3_by_length(v, sqrt_rn(len2));
    }
    const float4 gn = S[ns + 2 * (idx - ns) + 1];
    return mk(gn.x, gn.y, gn.z);
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

// calcNextRay's direction part (Trace.hs:394-429): the three draws, the rotated direction `next` and the BRDF
// factor `b`.  Both BRDF arms are evaluated through selects so that Matte and Glossy lanes of one wave do not
// serialise.  M points at the material records.
//   Matte:  rotate (anglesToQuaternion $ pi *^ rv) iNormal
//   Glossy: rotate (anglesToQuaternion $ (1 - p) *^ rv) reflection
// anglesToQuaternion halves every angle; (k * rv) * 0.5 == (0.5 k) * rv bit for bit (power-of-two scaling).
__device__ __forceinline__ void bounce_axis(float4 mb, V3 normal, V3 d, V3 &axis, float &hk)
{
    const bool matte = f2u(mb.x) == 0u;
    const float ia = dot(d, normal);
    const V3 reflection = d - scale_l(2.0f * ia, normal);
    axis = matte ? normal : reflection;
    hk = matte ? 0.5f * kPi : mb.w;       // mb.w = 0.5 * (1 - p)  (Trace.hs:424 and Util.hs:62-67), exact halving
}

__device__ __forceinline__ void next_about_axis(float4 mb, V3 axis, float hk, Sfc32 &seed, V3 &next, float &brdf)
{
    const bool matte = f2u(mb.x) == 0u;
    V3 rv;
    rv.x = gen_component(seed); rv.y = gen_component(seed); rv.z = gen_component(seed);
    next = rotate(quaternion_from_half_angles(hk * rv.x, hk * rv.y, hk * rv.z), axis);
    const float nd = dot(next, axis);
    brdf = matte ? mb.z * nd : __builtin_fmaxf(0.0f, nd);      // mb.z = p / pi (Trace.hs:411), divided at upload
}

template <typename ScenePtr>
__device__ __forceinline__ void next_direction(ScenePtr M, int idx, V3 normal, V3 d, Sfc32 &seed, V3 &next, float &brdf)
{
    const float4 mb = M[2 * idx + 1];
    V3 axis; float hk;
    bounce_axis(mb, normal, d, axis, hk);
    next_about_axis(mb, axis, hk, seed, next, brdf);
}

// The rest of calcNextRay (Trace.hs:431-435) and computeRay (Trace.hs:374-383) once `next` and `b` are known.
template <typename ScenePtr>
__device__ __forceinline__ void apply_bounce(ScenePtr M, int idx, V3 hit_pos, V3 next, float brdf,
                                             V3 &o, V3 &d, V3 &throughput, V3 &result)
{
    const float4 ma = M[2 * idx];
    const V3 color = mk(ma.x, ma.y, ma.z);
    const float illuminance = ma.w;
    const V3 emittance = scale_r(color, illuminance);
    constexpr float next_ray_prob = 1.0f / (kPi * 2.0f);
    o = hit_pos + scale_r(next, kEpsilon);
    d = next;
    const V3 tmod = scale_r(color, brdf * next_ray_prob);
    result = result + (emittance * throughput);
    throughput = throughput * tmod;
}

// computeRay (Trace.hs:374-383) + calcNextRay (Trace.hs:394-435)
template <typename ScenePtr>
__device__ __forceinline__ void shade(ScenePtr M, int idx, V3 hit_pos, V3 normal,
                                      V3 &o, V3 &d, V3 &throughput, V3 &result, Sfc32 &seed)
{
    V3 next; float brdf;
    next_direction(M, idx, normal, d, seed, next, brdf);
    apply_bounce(M, idx, hit_pos, next, brdf, o, d, throughput, result);
}

// The first shade of a sample that starts from the pixel's cached primary hit: result = 0, throughput = 1 and the
// incoming ray is the primary ray, so the rotation axis, the half angle scale and 0 + emittance * 1 are per-pixel
// constants (evaluated once, by the same operations), and throughput * tmod = 1 * tmod = tmod.
// ACCUMULATE (render Streams): `result` is the pixel's accumulator and first_term = emittance * 1 is added to it;
// otherwise (render Inline) first_term = 0 + emittance * 1 is the sample's result so far.
template <bool ACCUMULATE, typename ScenePtr>
__device__ __forceinline__ void shade_first(ScenePtr M, int idx, V3 hit_pos, V3 axis, float hk, V3 first_term,
                                            V3 &o, V3 &d, V3 &throughput, V3 &result, Sfc32 &seed)
{
    const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
    V3 next; float brdf;
    next_about_axis(mb, axis, hk, seed, next, brdf);
    constexpr float next_ray_prob = 1.0f / (kPi * 2.0f);
    o = hit_pos + scale_r(next, kEpsilon);
    d = next;
    result = ACCUMULATE ? result + first_term : first_term;
    throughput = scale_r(mk(ma.x, ma.y, ma.z), brdf * next_ray_prob);
}

// Is the shade that is about to happen CERTAIN to leave a throughput that the next prepareRay freezes (nearZero,
// Trace.hs:364-365) -- whatever the three random draws turn out to be?  If so its `next` ray and throughput are never
// looked at again: the only things the reference keeps from that iteration are result += emittance * throughput and
// the seed after genVec's three draws (Trace.hs:374-383), and the expensive half of the shade (three sin/cos pairs,
// the quaternion, the rotation) can be skipped without changing any output bit.
// Bound: next = rotate q axis with |q| = 1 up to rounding, so |next . axis| <= |axis|^2 (1 + 2e-5) <= a2 below; hence
// |brdf| <= bmax (Matte: |p/pi| a2, Glossy: max(0, nd) <= a2) by monotonicity of rounding, |tmod_c| <= |color_c| (bmax prob)
// and |throughput'_c| <= |throughput_c| (|color_c| g), formed in the SAME association as the real product so that it
// overflows exactly when the real one can; 1 % of slack covers the four roundings of the real dot product.  Every
// comparison is written so that a NaN or an infinity anywhere answers "not certain".
__device__ __forceinline__ bool surely_frozen_after(float4 ma, float4 mb, V3 axis, V3 throughput)
{
    const bool matte = f2u(mb.x) == 0u;
    const float a2 = dot(axis, axis) * 1.001f + 1e-30f;             // >= |next . axis|, also when the products are denormal
    const float bmax = matte ? __builtin_fabsf(mb.z) * a2 : a2;
    const float g = bmax * (1.0f / (kPi * 2.0f));
    const V3 v = throughput * scale_r(mk(ma.x, ma.y, ma.z), g);
    return dot(v, v) * 1.01f <= 1e-6f;
}

// What the reference keeps of an iteration whose throughput is frozen right afterwards: the contribution and the seed.
__device__ __forceinline__ void finish_frozen(float4 ma, V3 throughput, V3 &result, Sfc32 &seed)
{
    result = result + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
    (void)sfc32_next(seed); (void)sfc32_next(seed); (void)sfc32_next(seed);     // genVec's three draws
}

__device__ __forceinline__ int global_row(int local_row, int stripe_rows, int n_parts, int part)
{
    return ((local_row / stripe_rows) * n_parts + part) * stripe_rows + local_row % stripe_rows;
}

// How many lanes below this one are set in `mask`: v_mbcnt_lo/hi, no per-lane 64-bit mask to keep in registers.
__device__ __forceinline__ unsigned int rank_in(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned int v)
{
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// Which pixel a lane of a one-wave workgroup works on.  TILE_W == 0: 64 consecutive pixels of a row (plane order).
// TILE_W > 0: a TILE_W x (64 / TILE_W) tile of the image -- the 64 primary rays of a wave meet the same few
// primitives and their paths have similar lengths, which is worth 1-2 % (8 x 8 measured best); the
// seven plane accesses of a lane happen once per launch, so their shorter runs do not matter.
template <int TILE_W>
__device__ __forceinline__ bool lane_pixel(const RenderArgs &a, long long &pixel, unsigned int &quad, unsigned int wg = blockIdx.x)
{
    quad = 0;
    if (TILE_W > 0) {
        constexpr int tw = TILE_W > 0 ? TILE_W : 64, th = 64 / tw;
        const int tiles_x = (a.width + tw - 1) / tw;
        // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  A tile row segment is only tw * 4
        // bytes of a 128-byte line, so x-adjacent tiles must meet in ONE L2 or the line is fetched from HBM once per
        // tile (measured: 232 MB instead of 58 MB per launch): runs of 4 consecutive tiles ("quads") go to the same
        // XCD, as consecutive workgroups of that XCD.  The grid is padded to a multiple of 32 so that this is a bijection.
        // Which quad a dispatch position gets is the image order, or -- once a launch with the same camera has recorded
        // what every quad costs -- the most expensive first (quad_order), which shortens the end of the kernel where
        // the last waves run with the chip half empty.
        const unsigned int xcd = wg & 7u, k = wg >> 3;
        const unsigned int position = (k >> 2) * 8u + xcd;
        quad = a.quad_order ? a.quad_order[position] : position;
#if defined(PTMI_TILE_REVERSE)
        if (!a.quad_order) quad = (gridDim.x >> 2) - 1u - position;   // experiment: bottom of the image first
#endif
        const unsigned int tile = (quad << 2) + (k & 3u);
        const int tx = (int)(tile % (unsigned)tiles_x), ty = (int)(tile / (unsigned)tiles_x);
        const int x = tx * tw + (int)(threadIdx.x % tw), y = ty * th + (int)(threadIdx.x / tw);
        pixel = (long long)y * a.width + x;
        return x < a.width && y < a.rows_local;              // also false for the padding tiles (ty beyond the image)
    }
    pixel = (long long)wg * kRenderBlock + threadIdx.x;
    return pixel < (long long)a.rows_local * a.width;
}

// what the wave paid: the loop trips of its slowest lane, added to its quad's cost
__device__ __forceinline__ void record_cost(const RenderArgs &a, unsigned int quad, unsigned int trips)
{
    if (!a.quad_cost) return;
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(trips, off, 64); trips = other > trips ? other : trips; }
    if ((threadIdx.x & 63) == 0) atomicAdd(a.quad_cost + quad, trips + 1u);
}

__host__ inline unsigned int tile_grid(const RenderArgs &a, int tw)
{
    const int th = 64 / tw;
    const unsigned int tiles = (unsigned int)(((a.width + tw - 1) / tw) * ((a.rows_local + th - 1) / th));
    return (tiles + 31u) & ~31u;                             // see lane_pixel
}

// Sample chunks (see the comment in render_inline_kernel): which copy of the tile grid this workgroup is, which slice of
// the samples it renders, and the wait for the previous copy of the same tile.
template <int TILE_W>
__device__ __forceinline__ void enter_sample_chunk(const RenderArgs &a, unsigned int &wg, int &chunk, int &n_spp_chunk)
{
    wg = blockIdx.x + (a.first_position ? 4u * *a.first_position : 0u);     // (the tail of the stream form: RenderArgs.first_position)
    chunk = 0; n_spp_chunk = a.n_spp;
    if (TILE_W > 0 && a.spp_chunks > 1) {
        // The workgroup's place in the chain of copies is a TICKET, not blockIdx: HIP promises nothing about the order in which
        // workgroups are dispatched, and a consumer that waited for a producer not yet dispatched -- with every slot held by
        // waiting consumers -- would hang the launch.  The producer of a ticket's tile holds the ticket per_copy lower: it was
        // taken by a workgroup that is running or has finished.
        unsigned int t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(a.chunk_done + a.chunk_capacity, 1u);
        wg = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        const unsigned int per_copy = gridDim.x / (unsigned int)a.spp_chunks;
        chunk = (int)(wg / per_copy);
        wg -= (unsigned int)chunk * per_copy;
        const int per = (a.n_spp + a.spp_chunks - 1) / a.spp_chunks;
        n_spp_chunk = a.n_spp - chunk * per;
        n_spp_chunk = n_spp_chunk < 0 ? 0 : (n_spp_chunk > per ? per : n_spp_chunk);
        if (chunk > 0) {
            // one relaxed poll, then ONE acquire (polling with acquire loads invalidates the L1 every time round)
            while (__hip_atomic_load(a.chunk_done + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)chunk)
                __builtin_amdgcn_s_sleep(16);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
}

template <int TILE_W>
__device__ __forceinline__ void leave_sample_chunk(const RenderArgs &a, unsigned int wg, int chunk)
{
    if (TILE_W > 0 && a.spp_chunks > 1 && chunk + 1 < a.spp_chunks) {     // publish: the next copy of this tile may start
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the write-back has completed before the flag leaves (the compiler may drop its own wait)
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(a.chunk_done + wg, (unsigned int)(chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------
// render Inline.  LDS_SCENE: primitives staged in LDS (default) or read straight from
// global memory through scalar loads (ablation).  MODE selects the loop shape:
//   kCached      (default) primary hit evaluated once per pixel; loop [finish frozen shades + restart][shade][trace]
//   kCachedR1    round 1's default: primary hit cached, two shade rounds per trace round, every shade in full
//   kRegenerate  lanes start their next sample as soon as a path ends, one shade per trace
//   kLockstep    all lanes of the wave run sample s together (what a per-sample launch would do)
// ---------------------------------------------------------------------------------------
enum { kCached = 0, kRegenerate = 1, kLockstep = 2, kCachedR1 = 3 };   // kCachedR1: round 1's loop [shade][shade][trace], without the frozen-shade shortcut (ablation)

#ifndef PTMI_INLINE_WAVES
#define PTMI_INLINE_WAVES 7      // 72 VGPRs (three registers spilled around the loop, not in it) and a 16-word LDS column: C2 3.10 -> 3.04 ms
#endif
template <bool LDS_SCENE, int MODE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, MODE == kCached ? PTMI_INLINE_WAVES : (MODE == kCachedR1 ? 6 : 4)) render_inline_kernel(const RenderArgs a)
{
    // kCached: per-lane restart record (see below), 16 words; round 1's loop (kCachedR1) keeps the first shade's result there too
    __shared__ float pixel_const[MODE == kCached ? 10 : (MODE == kCachedR1 ? 19 : 1)][kRenderBlock];
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    // SAMPLE CHUNKS (a.spp_chunks > 1; tiled kernels only).  A pixel's samples are one serial chain, so a launch has as
    // many waves as the image has tiles, each as long as n_spp; with few tiles and many samples -- one of 8 parts of a
    // 4K image at 1024 spp: 16 200 waves for 6 144 slots -- the last round of waves runs on a partly empty chip and
    // costs 15 %.  The grid is therefore spp_chunks copies of the tile grid: copy c of a tile renders samples
    // [c S, (c+1) S) of its pixels, after copy c-1 has stored the planes and published done[tile] = c.  A workgroup's place
    // in that chain is a ticket it draws when it starts (enter_sample_chunk), so the producer of what it waits for has
    // started before it, whatever order the hardware dispatches workgroups in (in practice the producer finished a whole
    // round earlier: the wait falls through).  The planes travel through
    // memory between copies: release / acquire at agent scope (L2 write-back, L1 invalidate); copies of one tile run on
    // the same XCD (the grid of a copy is a multiple of 32).  Results do not depend on the chunking (sample-split invariance).
    unsigned int wg; int chunk, n_spp_chunk;
    enter_sample_chunk<TILE_W>(a, wg, chunk, n_spp_chunk);
    long long pixel;
    unsigned int quad, trips = 0;
    const bool valid = lane_pixel<TILE_W>(a, pixel, quad, wg);
    unsigned int live = 0;
    if (valid) {
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

        const int limit = a.bounce_limit, n_spp = n_spp_chunk;

        if (limit <= 0) {
            // iterate 0: every sample returns (0, seed); new + old
            if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;
        } else if (MODE == kCached) {
            // primaryRays has no sub-pixel jitter (Trace.hs:244-262): every sample of a pixel shoots the same primary ray, so
            // its checkHit + hit are evaluated ONCE per pixel and every sample starts from that record.
            // Loop shape: [finish frozen shades][restart][shade][trace].  A lane comes round with a hit to shade (`pending`) or
            // with its sample over (`over`: the trace missed, or the last shade left a throughput that the next prepareRay
            // freezes).  The shades whose outcome is CERTAIN to be frozen (the iteration limit, or surely_frozen_after) are
            // finished first -- emittance + three draws, no sin/cos, no rotation -- and those lanes are `over` too; then ONE
            // block restarts every `over` lane on its pixel's next sample, from the cached primary hit, with the rotation axis
            // and half-angle scale that every first shade of the pixel uses; then one full shade and one trace for all.  A
            // sample whose path ends by a certain freeze -- 64 % of them on C2 -- costs k-1 full shades and k-1 traces.
            const HitSel h0 = check_hit(S, ns, np, origin, primary);
            if (!h0.just) {
                if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;     // every sample: result 0, seed untouched
            } else {
                // What a sample restarts from lives in a lane-private LDS column (10 words), not in VGPRs: the position of the
                // primary hit, the axis and half-angle scale of its bounce, and the pixel's accumulator (touched once per sample).
                float *mine = &pixel_const[0][threadIdx.x];
                auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
                auto get = [&](int k) { return mine[k * kRenderBlock]; };
                V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
                hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
                const int idx0 = h0.idx;
                {
                    const float4 mb0 = M[2 * idx0 + 1];
                    V3 axis0; float hk0;
                    bounce_axis(mb0, normal, primary, axis0, hk0);
                    put(0, pos.x); put(1, pos.y); put(2, pos.z);
                    put(3, axis0.x); put(4, axis0.y); put(5, axis0.z); put(6, hk0);
                    put(7, acc.x); put(8, acc.y); put(9, acc.z);
                }
                int s = -1, it = 0, idx = idx0;                       // s: the sample being rendered (the first restart makes it 0)
                V3 d = primary;
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                bool pending = false, has_ray = false, over = n_spp > 0;
#ifdef PTMI_PHASE_STATS
                unsigned int st_iter = 0, st_a = 0, st_b = 0, st_c = 0, st_f = 0;   // this lane's participation per round
                unsigned long long cyc_a = 0, cyc_b = 0, cyc_c = 0;
#endif
                while (pending || over) {
                    ++trips;
#ifdef PTMI_PHASE_STATS
                    ++st_iter;
                    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
                    if (pending || over) ++st_a;
#endif
                    float4 mb = M[2 * idx + 1];
                    V3 axis = mk(0.0f, 0.0f, 0.0f); float hk = 0.0f;
                    if (pending) {
                        bounce_axis(mb, normal, d, axis, hk);
                        const float4 ma = M[2 * idx];
                        if (it + 1 >= limit || surely_frozen_after(ma, mb, axis, throughput)) {
                            finish_frozen(ma, throughput, result, seed);
                            ++live;
#ifdef PTMI_PHASE_STATS
                            ++st_f;
#endif
                            pending = false; over = true;
                        }
                    }
                    if (over) {                                        // next sample of this pixel
                        // \(new, seed') (old, _) -> (new + old, seed') -- once a sample has been rendered (the first time round
                        // the lane only starts sample 0)
                        if (s >= 0) { put(7, result.x + get(7)); put(8, result.y + get(8)); put(9, result.z + get(9)); }
                        ++s; it = 0;
                        throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                        pos = mk(get(0), get(1), get(2)); idx = idx0;
                        mb = M[2 * idx0 + 1];
                        axis = mk(get(3), get(4), get(5)); hk = get(6);
                        over = false; pending = s < n_spp;
                    }
                    if (pending) {
                        V3 next; float brdf;
                        next_about_axis(mb, axis, hk, seed, next, brdf);
                        apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, result);
                        ++it; ++live;
                        pending = false;
                        // the next prepareRay would freeze the path (Trace.hs:364-365)
                        if (it >= limit || near_zero(throughput)) over = true;
                        else has_ray = true;
                    }
#ifdef PTMI_PHASE_STATS
                    { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); cyc_a += t_now - t_prev; t_prev = t_now; }
                    if (has_ray) ++st_c;
#endif
                    if (has_ray) {
#ifdef PTMI_SPHERE_STATS
                        const HitSel h = check_hit(S, ns, np, pos, d, a.work_counter);
#else
                        const HitSel h = check_hit(S, ns, np, pos, d);
#endif
                        has_ray = false;
                        if (h.just) {
                            hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                            idx = h.idx;
                            pending = true;
                        } else {
                            over = true;
                        }
                    }
#ifdef PTMI_PHASE_STATS
                    cyc_c += __builtin_amdgcn_s_memtime() - t_prev;
#endif
                }
                acc = mk(get(7), get(8), get(9));
#ifdef PTMI_PHASE_STATS
                {
                    const unsigned long long m = __ballot(1);
                    unsigned int mx = st_iter;
                    for (int off = 32; off > 0; off >>= 1) { const unsigned int o2 = __shfl_xor(mx, off, 64); mx = o2 > mx ? o2 : mx; }
                    atomicAdd(a.work_counter + 1, st_iter); atomicAdd(a.work_counter + 2, st_a);
                    atomicAdd(a.work_counter + 3, st_b); atomicAdd(a.work_counter + 4, st_c); atomicAdd(a.work_counter + 6, st_f);
                    if ((threadIdx.x & 63) == (int)__builtin_ctzll(m)) {
                        atomicAdd(a.work_counter + 5, mx * 64u);
                        atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 8), cyc_a);
                        atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 10), cyc_b);
                        atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 12), cyc_c);
                    }
                }
#endif
            }
        } else if (MODE == kCachedR1) {
            constexpr bool kFinish = false;
            // primaryRays has no sub-pixel jitter (Trace.hs:244-262): every sample of a pixel shoots the
            // same primary ray, so its checkHit + hit are evaluated ONCE per pixel and every sample starts
            // from that record.  A sample then costs k shades and k-1 traces (k = its live bounces).
            // Loop shape (kCached): [shade][trace], where the shade round first FINISHES the shades whose outcome the next
            // prepareRay is certain to freeze (emittance + three draws, no sin/cos, no rotation) and restarts those lanes
            // on their next sample, so that they take part in the round's full shade with it.  A sample whose path ends
            // that way -- 64 % of them on C2 -- then costs k-1 full shades and k-1 traces, exactly one of each per trip.
            // Loop shape (kCachedR1, round 1): [shade][shade again for lanes whose sample just ended][trace], all in full.
            const HitSel h0 = check_hit(S, ns, np, origin, primary);
            if (!h0.just) {
                if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;     // every sample: result 0, seed untouched
            } else {
                // The per-pixel constants a sample restarts from (primary hit record + primary direction, 10
                // words) are read once per sample: they live in a lane-private LDS column instead of VGPRs,
                // which is what lets the kernel fit 72 VGPRs = 7 waves per SIMD (16 words: 28 waves fit the LDS of a CU).
                float *mine = &pixel_const[0][threadIdx.x];
                auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
                auto get = [&](int k) { return mine[k * kRenderBlock]; };
                V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
                hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
                put(0, pos.x); put(1, pos.y); put(2, pos.z);
                put(3, normal.x); put(4, normal.y); put(5, normal.z);
                put(6, primary.x); put(7, primary.y); put(8, primary.z);
                put(9, acc.x); put(10, acc.y); put(11, acc.z);        // the accumulator is touched once per sample: LDS too
                const int idx0 = h0.idx;
                {   // what every first shade of this pixel uses (shade_first)
                    const float4 ma0 = M[2 * idx0], mb0 = M[2 * idx0 + 1];
                    V3 axis; float hk;
                    bounce_axis(mb0, normal, primary, axis, hk);
                    const V3 first_result = mk(0.0f, 0.0f, 0.0f) + (scale_r(mk(ma0.x, ma0.y, ma0.z), ma0.w) * mk(1.0f, 1.0f, 1.0f));
                    put(12, axis.x); put(13, axis.y); put(14, axis.z); put(15, hk);
                    if (MODE == kCachedR1) { put(16, first_result.x); put(17, first_result.y); put(18, first_result.z); }
                }
                int s = 0, it = 0, idx = idx0;
                V3 d = primary;                                       // the ray that produced the hit / the next ray
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                bool pending = n_spp > 0, has_ray = false;
                auto restart = [&]() {                                // next sample of this pixel
                    // \(new, seed') (old, _) -> (new + old, seed')
                    put(9, result.x + get(9)); put(10, result.y + get(10)); put(11, result.z + get(11));
                    ++s; it = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                    pos = mk(get(0), get(1), get(2)); normal = mk(get(3), get(4), get(5));
                    d = mk(get(6), get(7), get(8)); idx = idx0;
                    pending = s < n_spp;
                };
#ifdef PTMI_PHASE_STATS
                unsigned int st_iter = 0, st_a = 0, st_b = 0, st_c = 0, st_f = 0;   // this lane's participation per round
#endif
#ifdef PTMI_PHASE_STATS
                unsigned long long cyc_a = 0, cyc_b = 0, cyc_c = 0;
#endif
                while (pending) {
                    ++trips;
#ifdef PTMI_PHASE_STATS
                    ++st_iter;
                    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
                    // round A: whatever hit is pending.  First the shades whose outcome is certain to be frozen by the next
                    // prepareRay (the iteration limit, or a throughput that cannot stay above nearZero: surely_frozen_after)
                    // -- 29 % of all shades on C2: they only add their emittance and advance the seed, and the lane starts its
                    // pixel's next sample AT ONCE, so that it takes part in this round's full shade with that sample.
                    if (pending && !has_ray) {
#ifdef PTMI_PHASE_STATS
                        ++st_a;
#endif
                        float4 mb = M[2 * idx + 1];
                        V3 axis; float hk;
                        bounce_axis(mb, normal, d, axis, hk);
                        if (kFinish) {
                            const float4 ma = M[2 * idx];
                            if (it + 1 >= limit || surely_frozen_after(ma, mb, axis, throughput)) {
                                finish_frozen(ma, throughput, result, seed);
                                ++it; ++live;
#ifdef PTMI_PHASE_STATS
                                ++st_f;
#endif
                                restart();                             // the cached primary hit: its axis and half-angle scale are cached too
                                mb = M[2 * idx0 + 1];
                                axis = mk(get(12), get(13), get(14)); hk = get(15);
                            }
                        }
                        if (pending) {
                            V3 next; float brdf;
                            next_about_axis(mb, axis, hk, seed, next, brdf);
                            apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, result);
                            ++it; ++live;
                            // the next prepareRay would freeze the path (Trace.hs:364-365)
                            if (it >= limit || near_zero(throughput)) restart();
                            else { pending = false; has_ray = true; }
                        }
                    }
#ifdef PTMI_PHASE_STATS
                    { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); cyc_a += t_now - t_prev; t_prev = t_now; }
#endif
                    // round B: only lanes that restarted in round A get here with a pending hit, and that hit is the
                    // cached primary hit with result 0 and throughput 1: the specialised first shade
                    // (With the frozen-shade shortcut above, 87 % of the restarts happen BEFORE round A's full shade; running a
                    // second round for the few that happen after it -- an unpredicted nearZero -- costs more than letting
                    // those lanes wait for the next trip: C2 3.11 ms without it, 3.17 when run for >= 16 lanes, 3.77 always.)
                    if (!kFinish && pending && !has_ray) {
#ifdef PTMI_PHASE_STATS
                        ++st_b;
#endif
                        shade_first<false>(M, idx0, pos, mk(get(12), get(13), get(14)), get(15), mk(get(16), get(17), get(18)),
                                    pos, d, throughput, result, seed);
                        ++it; ++live;
                        if (it >= limit || near_zero(throughput)) restart();
                        else { pending = false; has_ray = true; }
                    }
#ifdef PTMI_PHASE_STATS
                    { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); cyc_b += t_now - t_prev; t_prev = t_now; }
#endif
#ifdef PTMI_PHASE_STATS
                    if (has_ray) ++st_c;
#endif
                    if (has_ray) {
#ifdef PTMI_SPHERE_STATS
                        const HitSel h = check_hit(S, ns, np, pos, d, a.work_counter);
#else
                        const HitSel h = check_hit(S, ns, np, pos, d);
#endif
                        has_ray = false;
                        if (h.just) {
                            hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                            idx = h.idx;
                            pending = true;
                        } else {
                            restart();
                        }
                    }
#ifdef PTMI_PHASE_STATS
                    cyc_c += __builtin_amdgcn_s_memtime() - t_prev;
#endif
                }
                acc = mk(get(9), get(10), get(11));
#ifdef PTMI_PHASE_STATS
                // diagnostic build only: [1] lane-iterations, [2..4] lane participations in rounds A, B, C,
                // [5] max lane-iterations of the wave x lanes that had work (what the wave paid for)
                {
                    const unsigned long long m = __ballot(1);
                    unsigned int mx = st_iter;
                    for (int off = 32; off > 0; off >>= 1) { const unsigned int o2 = __shfl_xor(mx, off, 64); mx = o2 > mx ? o2 : mx; }
                    atomicAdd(a.work_counter + 1, st_iter); atomicAdd(a.work_counter + 2, st_a);
                    atomicAdd(a.work_counter + 3, st_b); atomicAdd(a.work_counter + 4, st_c); atomicAdd(a.work_counter + 6, st_f);
                    if ((threadIdx.x & 63) == (int)__builtin_ctzll(m)) {
                        atomicAdd(a.work_counter + 5, mx * 64u);
                        // wave cycles spent in rounds A, B, C (the waves of a SIMD interleave, so these are shares, not costs)
                        atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 8), cyc_a);
                        atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 10), cyc_b);
                        atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 12), cyc_c);
                    }
                }
#endif
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
    leave_sample_chunk<TILE_W>(a, wg, chunk);

    if (TILE_W > 0) record_cost(a, quad, trips);
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
}

#ifdef PTMI_ABLATIONS     // the two ablation kernels of render Inline: only in builds with -DPTMI_ABLATIONS
// ---------------------------------------------------------------------------------------
// render Inline, pooled second shade round.  Same loop [shade A][shade B][trace C] and the same arithmetic as
// kCached, but the B round -- lanes whose sample ended in A and whose next sample starts from the cached
// primary hit; only ~49 % of a wave's lanes -- is shared by the W waves of a workgroup: a restarting lane
// posts (seed, owner) into an LDS pool (ballot + prefix inside the wave, one LDS atomic per wave for the base),
// the pool's items are shaded densely by as many waves as it takes (the others skip the round), and the
// owner picks up (seed', next, b) and finishes computeRay itself.  An item's inputs besides the seed are the
// owner's restart record, which already lives in LDS.  A pixel's arithmetic does not depend on which lane
// executes it, so the planes stay bit-identical.
// Requires bounce_limit >= 1 and n_spp >= 1 (the launcher routes the degenerate cases elsewhere).
// ---------------------------------------------------------------------------------------
template <bool LDS_SCENE, int W>
__global__ void __launch_bounds__(64 * W, 6) render_inline_pooled_kernel(const RenderArgs a)
{
    constexpr int kThreads = 64 * W;
    __shared__ float pixel_const[13][kThreads];             // restart record: hit pos, normal, primary dir, acc, primitive index
    __shared__ unsigned int pool[8][kThreads];              // in: seed[4], owner | out: seed'[4], b, next[3]
    __shared__ unsigned int pool_count[2];                  // items | (waves with work << 16), alternating per trip
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) pool_count[tid] = 0;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = tid; i < total; i += kThreads) lds_scene[i] = a.scene.packed[i];
    }
    __syncthreads();
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    const long long n_local = (long long)a.rows_local * a.width;
    const long long pixel = (long long)blockIdx.x * kThreads + tid;
    const bool valid = pixel < n_local;
    const int limit = a.bounce_limit, n_spp = a.n_spp;
    unsigned int live = 0;

    float *mine = &pixel_const[0][tid];
    auto put = [&](int k, float v) { mine[k * kThreads] = v; };
    auto get = [&](int k) { return mine[k * kThreads]; };

    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    V3 pos = mk(0, 0, 0), normal = pos, d = pos;
    V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
    int s = 0, it = 0, idx = 0, idx0 = 0;
    bool pending = false, has_ray = false;

    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        if (a.screen_x) { px = a.screen_x[pixel]; py = a.screen_y[pixel]; }
        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);
        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        seed.a = a.planes.sa[pixel]; seed.b = a.planes.sb[pixel];
        seed.c = a.planes.sc[pixel]; seed.counter = a.planes.sctr[pixel];
        const HitSel h0 = check_hit(S, ns, np, origin, primary);
        if (!h0.just) {
            acc = mk(0.0f, 0.0f, 0.0f) + acc;                // every sample: result 0, seed untouched
        } else {
            hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
            put(0, pos.x); put(1, pos.y); put(2, pos.z);
            put(3, normal.x); put(4, normal.y); put(5, normal.z);
            put(6, primary.x); put(7, primary.y); put(8, primary.z);
            put(12, u2f((uint32_t)h0.idx));
            idx0 = idx = h0.idx;
            d = primary;
            pending = true;
        }
        put(9, acc.x); put(10, acc.y); put(11, acc.z);
    }
    auto restart = [&]() {                                    // next sample of this pixel
        // \(new, seed') (old, _) -> (new + old, seed')
        put(9, result.x + get(9)); put(10, result.y + get(10)); put(11, result.z + get(11));
        ++s; it = 0;
        throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
        pos = mk(get(0), get(1), get(2)); normal = mk(get(3), get(4), get(5));
        d = mk(get(6), get(7), get(8)); idx = idx0;
        pending = s < n_spp;
    };

#ifdef PTMI_POOL_STATS
    unsigned int st_trips = 0, st_alive = 0, st_batches = 0, st_items = 0;
    unsigned long long cyc[5] = {0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int k) { const unsigned long long t = __builtin_amdgcn_s_memtime(); cyc[k] += t - t_prev; t_prev = t; };
#else
    auto stamp = [](int) {};
#endif
    for (unsigned int trip = 0;; ++trip) {
        // ---- round A: in place
        if (pending && !has_ray) {
            shade(M, idx, pos, normal, pos, d, throughput, result, seed);
            ++it; ++live;
            // the next prepareRay would freeze the path (Trace.hs:364-365)
            if (it >= limit || near_zero(throughput)) restart();
            else { pending = false; has_ray = true; }
        }
        stamp(0);
        // ---- round B: restarted lanes post their item
        const bool need_b = pending && !has_ray;
        const unsigned long long mask = __ballot(need_b);
        const bool wave_alive = __any(pending || has_ray);
        unsigned int base = 0;
        if (lane == 0 && wave_alive)
            base = atomicAdd(&pool_count[trip & 1], (unsigned int)__builtin_popcountll(mask) | (1u << 16));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base) & 0xffffu;
        const int slot = (int)base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (need_b) {
            pool[0][slot] = seed.a; pool[1][slot] = seed.b; pool[2][slot] = seed.c; pool[3][slot] = seed.counter;
            pool[4][slot] = (unsigned int)tid;
        }
        __syncthreads();
        stamp(1);
        const unsigned int count = pool_count[trip & 1];
        if ((count >> 16) == 0) break;                       // no wave of the workgroup has work left
        if (tid == 0) pool_count[(trip & 1) ^ 1] = 0;        // last read before this barrier, next written after the one below
        const int n_items = (int)(count & 0xffffu);
        // the waves take 64-item batches in an order that rotates with the trip, so the extra work moves over the SIMDs
        for (int b0 = (int)((wave + trip) % W) * 64; b0 < n_items; b0 += kThreads) {
            const int item = b0 + lane;
#ifdef PTMI_POOL_STATS
            ++st_batches; st_items += (unsigned int)__builtin_popcountll(__ballot(item < n_items));
#endif
            if (item < n_items) {
                Sfc32 sd; sd.a = pool[0][item]; sd.b = pool[1][item]; sd.c = pool[2][item]; sd.counter = pool[3][item];
                const float *rec = &pixel_const[0][pool[4][item]];
                const V3 n_i = mk(rec[3 * kThreads], rec[4 * kThreads], rec[5 * kThreads]);
                const V3 d_i = mk(rec[6 * kThreads], rec[7 * kThreads], rec[8 * kThreads]);
                const int idx_i = (int)f2u(rec[12 * kThreads]);
                V3 next; float brdf;
                next_direction(M, idx_i, n_i, d_i, sd, next, brdf);
                pool[0][item] = sd.a; pool[1][item] = sd.b; pool[2][item] = sd.c; pool[3][item] = sd.counter;
                pool[4][item] = f2u(brdf);
                pool[5][item] = f2u(next.x); pool[6][item] = f2u(next.y); pool[7][item] = f2u(next.z);
            }
        }
        stamp(2);
        __syncthreads();
        if (need_b) {
            seed.a = pool[0][slot]; seed.b = pool[1][slot]; seed.c = pool[2][slot]; seed.counter = pool[3][slot];
            const float brdf = u2f(pool[4][slot]);
            const V3 next = mk(u2f(pool[5][slot]), u2f(pool[6][slot]), u2f(pool[7][slot]));
            apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, result);
            ++it; ++live;
            if (it >= limit || near_zero(throughput)) restart();
            else { pending = false; has_ray = true; }
        }
        stamp(3);
#ifdef PTMI_POOL_STATS
        ++st_trips; st_alive += wave_alive ? 1u : 0u;
#endif
        // ---- round C: trace
        if (has_ray) {
            const HitSel h = check_hit(S, ns, np, pos, d);
            has_ray = false;
            if (h.just) {
                hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                idx = h.idx;
                pending = true;
            } else {
                restart();
            }
        }
        stamp(4);
    }

    if (valid) {
        a.planes.r[pixel] = get(9); a.planes.g[pixel] = get(10); a.planes.b[pixel] = get(11);
        a.planes.sa[pixel] = seed.a; a.planes.sb[pixel] = seed.b;
        a.planes.sc[pixel] = seed.c; a.planes.sctr[pixel] = seed.counter;
    }
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if (lane == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
#ifdef PTMI_POOL_STATS
    // diagnostic build only: per wave [1] trips, [2] trips with own work, [3] B batches executed, [4] B items executed,
    // [8..17] cycles in: round A | post + barrier | B batches | barrier + pick-up | trace
    if (lane == 0) {
        atomicAdd(a.work_counter + 1, st_trips); atomicAdd(a.work_counter + 2, st_alive);
        atomicAdd(a.work_counter + 3, st_batches); atomicAdd(a.work_counter + 4, st_items);
        for (int k = 0; k < 5; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 8 + 2 * k), cyc[k]);
    }
#endif
}

// ---------------------------------------------------------------------------------------
// render Inline, persistent form (default).  Same per-pixel arithmetic as kCached above, but a lane
// that finishes its pixel (all n_spp samples) takes the next unprocessed pixel from a global counter
// instead of idling until the slowest of the wave's 64 pixels is done: the number of trace rounds a
// pixel needs is a sum over its samples and varies by +-25 % inside a wave at 64 spp (measured:
// 47.5 of 64 lanes active per VALU instruction with the static mapping).  Pixels are handed out with
// one atomic per wave: ballot of the lanes that want one, popcount prefix for the rank, the lowest
// wanting lane adds the count -- so lanes that ask together get consecutive pixels (at start: 64
// consecutive pixels per wave, fully coalesced plane reads).  A fetched pixel's primary ray joins the
// wave's next trace round; its hit is cached for the pixel's remaining samples.
// Requires bounce_limit >= 1 and n_spp >= 1 (the launcher routes the degenerate cases elsewhere).
// ---------------------------------------------------------------------------------------
template <bool LDS_SCENE>
__global__ void __launch_bounds__(kRenderBlock) render_inline_persistent_kernel(const RenderArgs a)
{
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    const unsigned long long n_local = (unsigned long long)a.rows_local * (unsigned long long)a.width;
    const int limit = a.bounce_limit, n_spp = a.n_spp;
    const V3 origin = a.cam.pos;
    const int lane = threadIdx.x & 63;
    unsigned int live = 0;

    unsigned long long pixel = 0;
    V3 acc = mk(0, 0, 0), primary = mk(0, 0, 0), p0 = mk(0, 0, 0), n0 = mk(0, 0, 0);
    V3 hit_pos = p0, normal = n0, o = origin, d = primary;
    V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    int s = 0, it = 0, idx = 0, idx0 = 0;
    bool pending = false, has_ray = false, is_primary = false, finished = false, exhausted = false;
    unsigned long long pool_next = 0, pool_end = 0;          // wave-uniform
    bool queue_empty = false;                                // wave-uniform

#ifdef PTMI_PHASE_STATS
    unsigned int st_iter = 0, st_a = 0, st_b = 0, st_c = 0;
#endif
    for (;;) {
#ifdef PTMI_PHASE_STATS
        ++st_iter;
#endif
        // ---- shade: twice, so that a lane whose sample ends in the first round starts the next in the second
        for (int round = 0; round < 2; ++round) {
            if (pending && !has_ray) {
#ifdef PTMI_PHASE_STATS
                if (round == 0) ++st_a; else ++st_b;
#endif
                shade(M, idx, hit_pos, normal, o, d, throughput, result, seed);
                ++it; ++live;
                // the next prepareRay would freeze the path (Trace.hs:364-365)
                if (it >= limit || near_zero(throughput)) {
                    acc = result + acc;                       // \(new, seed') (old, _) -> (new + old, seed')
                    ++s; it = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                    hit_pos = p0; normal = n0; idx = idx0; d = primary;
                    pending = s < n_spp;
                    finished = !pending;
                } else {
                    pending = false; has_ray = true;
                }
            }
        }
        // ---- retire finished pixels (one store site), then hand out new ones
        if (finished) {
            a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
            a.planes.sa[pixel] = seed.a; a.planes.sb[pixel] = seed.b;
            a.planes.sc[pixel] = seed.c; a.planes.sctr[pixel] = seed.counter;
            finished = false;
        }
        const bool want = !pending && !has_ray && !exhausted;
        const unsigned long long want_mask = __ballot(want);
        // The hand-out block (index arithmetic, seven plane loads, primary ray set-up) runs for the whole wave
        // whenever it runs, so lanes wait until kFetchBatch of them want a pixel -- or nothing else is in flight.
        if (want_mask && (__builtin_popcountll(want_mask) >= kFetchBatch || !__any(has_ray || pending))) {   // wave-uniform
            // The wave owns a pool [pool_next, pool_end) of consecutive pixels, refilled kChunk at a time with
            // ONE atomic on the global counter (a single counter word serves only ~90 requests/us on this chip:
            // one atomic per fetched pixel made the kernel 3x slower).  Wanting lanes take pool entries by rank.
            const unsigned int n_want = (unsigned int)__builtin_popcountll(want_mask);
            const unsigned int rank = (unsigned int)__builtin_popcountll(want_mask & ((1ull << lane) - 1ull));
            unsigned long long mine = ~0ull;
            unsigned int avail = (unsigned int)(pool_end - pool_next);
            unsigned int take = n_want < avail ? n_want : avail;
            if (want && rank < take) mine = pool_next + rank;
            pool_next += take;
            if (take < n_want && !queue_empty) {
                const int leader = (int)__builtin_ctzll(want_mask);
                unsigned int base = 0;
                if (lane == leader) base = atomicAdd(a.work_counter, (unsigned int)kChunk);
                base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
                if ((unsigned long long)base >= n_local) {
                    queue_empty = true;
                } else {
                    pool_next = base;
                    pool_end = (unsigned long long)base + kChunk < n_local ? (unsigned long long)base + kChunk : n_local;
                    avail = (unsigned int)(pool_end - pool_next);
                    const unsigned int more = n_want - take < avail ? n_want - take : avail;
                    if (want && rank >= take && rank - take < more) mine = pool_next + (rank - take);
                    pool_next += more;
                }
            }
            if (want) {
                if (mine != ~0ull) {
                    pixel = mine;
                    const int local_row = (int)(pixel / (unsigned long long)a.width);
                    const int col = (int)(pixel - (unsigned long long)local_row * (unsigned long long)a.width);
                    int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
                    if (a.screen_x) { px = a.screen_x[pixel]; py = a.screen_y[pixel]; }
                    primary = primary_direction(a.cam, px, py);
                    acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
                    seed.a = a.planes.sa[pixel]; seed.b = a.planes.sb[pixel];
                    seed.c = a.planes.sc[pixel]; seed.counter = a.planes.sctr[pixel];
                    o = origin; d = primary;
                    s = 0; it = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                    has_ray = true; is_primary = true;
                } else if (queue_empty) {
                    exhausted = true;
                }
            }
        }
        if (!__any(has_ray || pending)) break;               // nothing left in flight in this wave
#ifdef PTMI_PHASE_STATS
        if (has_ray) ++st_c;
#endif
        // ---- trace: every lane that has a ray (next bounce, or the primary ray of a fresh pixel)
        if (has_ray) {
            const HitSel h = check_hit(S, ns, np, o, d);
            has_ray = false;
            if (h.just) {
                hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                idx = h.idx;
                pending = true;
                if (is_primary) { p0 = hit_pos; n0 = normal; idx0 = idx; is_primary = false; }
            } else if (is_primary) {
                acc = mk(0.0f, 0.0f, 0.0f) + acc;            // every sample: result 0, seed untouched
                is_primary = false; finished = true;
            } else {
                acc = result + acc;
                ++s; it = 0;
                throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                hit_pos = p0; normal = n0; idx = idx0; d = primary;
                pending = s < n_spp;
                finished = !pending;
            }
        }
    }

    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if (lane == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
#ifdef PTMI_PHASE_STATS
    // diagnostic build only: [1] wave-iterations x 64, [2..4] lane participations in rounds A, B, C
    atomicAdd(a.work_counter + 1, st_iter); atomicAdd(a.work_counter + 2, st_a);
    atomicAdd(a.work_counter + 3, st_b); atomicAdd(a.work_counter + 4, st_c);
#endif
}

#endif  // PTMI_ABLATIONS

#ifndef PTMI_CONTRACTED_BUILD   // (the contracted-arithmetic object holds render Inline only: see the end of the file)
// ---------------------------------------------------------------------------------------
// render Streams (Trace.hs:141-191, 272-331).  The reference keeps one ray per pixel in a stream
// that `expand` compacts after every step (numNewRays is 0 or 1, Trace.hs:329-331) and scatters the
// colours back with `permute (+)`; because a pixel never owns more than one ray, the stream is the
// per-pixel chain below and the compaction becomes "a lane whose ray died starts its pixel's next
// sample" -- the wave stays dense without moving ray state through memory.  What differs from
// Inline, and is reproduced literally:
//   * every hit adds emittance * throughput straight into the accumulator, also in the step where the
//     throughput is already near zero (computeResult runs for every intersection, Trace.hs:290-293);
//   * the ray dies when nearZero throughput || miss (Trace.hs:329-331); there is NO bounce limit -- a
//     non-empty stream is never stopped by the iteration count (Trace.hs:166-170).  a.stream_step_cap only
//     guarantees that the kernel terminates (rays it cuts are counted, stream_counters[kScTruncated]);
//   * which seed the pixel carries out of `combine` (Trace.hs:179-184) is Accelerate-backend behaviour
//     (assumption A5).  Default: the pixel keeps its OLD seed while the sample runs;
//     a.seed_from_result: the seed of the ray that made the sample's LAST hit replaces it (kept in the lane's LDS
//     column, not in registers).  Either way updateSeed then advances the pixel's seed by one
//     draw (Trace.hs:151, :190-191).
// ---------------------------------------------------------------------------------------

#ifndef PTMI_STREAMS_WAVES
#define PTMI_STREAMS_WAVES 7     // 72 VGPRs (one pair spilled around the loop, not in it): C2 4.27 -> 4.17 ms
#endif
template <bool LDS_SCENE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, PTMI_STREAMS_WAVES) render_streams_kernel(const RenderArgs a)
{
    __shared__ float pixel_const[11][kRenderBlock];         // per-lane restart record (rows 0..6) and the last hit's seed (7..10)
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    // As the tail of the stream form the grid covers every dispatch position, and its workgroups start where the stream form's part
    // ends (a device word): those that would pass the last position have nothing to do.
    if (TILE_W > 0 && a.first_position && blockIdx.x + 4u * *a.first_position >= gridDim.x) return;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    unsigned int wg; int chunk, n_spp_chunk;
    enter_sample_chunk<TILE_W>(a, wg, chunk, n_spp_chunk);            // sample chunks, as in render_inline_kernel
    long long pixel;
    unsigned int quad, trips = 0;
    const bool valid = lane_pixel<TILE_W>(a, pixel, quad, wg);
    unsigned int live = 0, longest = 0, cut = 0;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        const int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);
        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        Sfc32 pixel_seed;
        pixel_seed.a = a.planes.sa[pixel]; pixel_seed.b = a.planes.sb[pixel];
        pixel_seed.c = a.planes.sc[pixel]; pixel_seed.counter = a.planes.sctr[pixel];
        const int n_spp = n_spp_chunk;

        const HitSel h0 = check_hit(S, ns, np, origin, primary);   // same primary ray for every sample
        if (!h0.just) {
            for (int s = 0; s < n_spp; ++s) (void)random_float(pixel_seed);     // updateSeed only
        } else {
            float *mine = &pixel_const[0][threadIdx.x];
            auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
            auto get = [&](int k) { return mine[k * kRenderBlock]; };
            V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
            hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
            const int idx0 = h0.idx;
            {   // what every sample of this pixel starts from: the primary hit and the axis / half-angle scale of its bounce
                const float4 mb0 = M[2 * idx0 + 1];
                V3 axis0; float hk0;
                bounce_axis(mb0, normal, primary, axis0, hk0);
                put(0, pos.x); put(1, pos.y); put(2, pos.z);
                put(3, axis0.x); put(4, axis0.y); put(5, axis0.z); put(6, hk0);
            }
            // PTMI_SEED_FROM_RESULT (combine new old): every hit leaves the seed its ray carried in rows 7..10 of the lane's
            // LDS column (four ds_writes per hit instead of four more registers); the sample's last one survives.
            auto note_hit_seed = [&](const Sfc32 &sd) {
                if (a.seed_from_result) { put(7, u2f(sd.a)); put(8, u2f(sd.b)); put(9, u2f(sd.c)); put(10, u2f(sd.counter)); }
            };
            int s = -1, idx = idx0;                               // s: the sample being rendered (the first pass through the block below makes it 0)
            unsigned int steps = 0;
            V3 d = primary;
            V3 throughput = mk(1.0f, 1.0f, 1.0f);
            Sfc32 seed = pixel_seed;
            bool pending = false, has_ray = false, over = n_spp > 0;
            // Loop shape [finish dead rays][next sample][shade][trace]: a lane comes round with a hit to shade (`pending`) or with
            // its sample over (`over`: the trace missed, or the cap cut the child).  ONE block per trip ends the samples that
            // are over and starts the pixel's next one from the cached primary hit, so that those lanes take part in this
            // trip's full shade.
            while (pending || over) {
                ++trips;
                float4 mb = M[2 * idx + 1];
                V3 axis = mk(0.0f, 0.0f, 0.0f); float hk = 0.0f;
                if (pending) {
                    // A ray whose throughput is already near zero dies at this hit (numNewRays, Trace.hs:329-331): the hit still
                    // adds its emittance (computeResult runs for every intersection) and nothing else of it survives -- no
                    // child, and the ray's seed is discarded -- so such lanes skip the three sin/cos pairs and the rotation.
                    if (near_zero(throughput)) {
                        const float4 ma = M[2 * idx];
                        acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
                        note_hit_seed(seed);
                        ++steps;
                        pending = false; over = true;
                    } else {
                        bounce_axis(mb, normal, d, axis, hk);
                    }
                }
                if (over) {
                    if (s >= 0) {                                  // a sample has been rendered
                        if (a.seed_from_result && steps > 0u) {    // combine new old: the seed the sample's last hit carried
                            pixel_seed.a = f2u(get(7)); pixel_seed.b = f2u(get(8)); pixel_seed.c = f2u(get(9)); pixel_seed.counter = f2u(get(10));
                        }
                        (void)random_float(pixel_seed);            // updateSeed
                        longest = steps > longest ? steps : longest;
                    }
                    seed = pixel_seed;
                    ++s; steps = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f);
                    pos = mk(get(0), get(1), get(2)); idx = idx0;
                    mb = M[2 * idx0 + 1];
                    axis = mk(get(3), get(4), get(5)); hk = get(6);
                    over = false; pending = s < n_spp;
                }
                if (pending) {                                     // alive (a fresh sample starts with throughput 1)
                    const bool capped = steps + 1u >= step_cap;
                    note_hit_seed(seed);
                    // results: colour += emittance * throughput for EVERY hit; then the new ray (shade, with the axis in hand)
                    V3 next; float brdf;
                    next_about_axis(mb, axis, hk, seed, next, brdf);
                    apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, acc);
                    ++steps; ++live;                               // the child exists even if the cap then cuts it
                    pending = false;
                    if (capped) { ++cut; over = true; }
                    else has_ray = true;
                }
                if (has_ray) {
                    const HitSel h = check_hit(S, ns, np, pos, d);
                    has_ray = false;
                    if (h.just) {
                        hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                        idx = h.idx;
                        pending = true;
                    } else {
                        over = true;
                    }
                }
            }
        }
        a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
        a.planes.sa[pixel] = pixel_seed.a; a.planes.sb[pixel] = pixel_seed.b;
        a.planes.sc[pixel] = pixel_seed.c; a.planes.sctr[pixel] = pixel_seed.counter;
    }
    leave_sample_chunk<TILE_W>(a, wg, chunk);
    if (TILE_W > 0) {
        if (a.first_position) {                               // the stream form's unit: the hits the tile's pixels shaded (record_item_cost)
            const unsigned long long hits = wave_sum(live);
            if (a.quad_cost && (threadIdx.x & 63) == 0) atomicAdd(a.quad_cost + quad, (unsigned int)hits);
        } else {
            record_cost(a, quad, trips);
        }
    }
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
    if (a.stream_iterations) {
        for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(longest, off, 64); longest = other > longest ? other : longest; }
        if ((threadIdx.x & 63) == 0 && longest) atomicMax(a.stream_iterations + (size_t)(blockIdx.x & (kStatShards - 1)) * (2 * kStatStride), longest);
    }
    if (__any(cut != 0u)) {                                           // rare: only when the safety cap bites
        const unsigned long long total = wave_sum(cut);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.stream_counters + kScTruncated, total);
    }
}

// ---------------------------------------------------------------------------------------
// render Streams as a stream ("wavefront" path): used when a ray can split, i.e. when the scene holds the
// build-defined GLASS material (numNewRays = 2; the reference only announces such materials, Trace.hs:109-118,
// :306-307, :327-328 -- there are NO reference semantics for GLASS; the definition is glass_children below,
// mirrored by the oracle).  One launch per traceStep (Trace.hs:272-294):
//   map checkHit over the stream; computeResult for every hit -> `permute (+)` = float atomics into the
//   colour planes; `expand` = wave-level compaction: ballot of the lanes that emit a child, popcount prefix
//   for the slot, ONE atomic per wave on the next stream's length.  A lane never waits for another.
// Without GLASS every pixel has at most one ray per step, so every colour word sees one adder per launch and
// the result equals the per-pixel chain kernel bit for bit (tested); with GLASS several rays of one pixel may
// add in the same launch and the order of those additions is not defined (neither is it in Accelerate's
// permute), so GLASS scenes are compared with a tolerance.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void queue_store(const RayQueue &q, unsigned int i, V3 o, V3 d, V3 t, uint32_t pixel, Sfc32 s, uint32_t depth)
{
    float4 *r = q.record(i);
    r[0] = float4{o.x, o.y, o.z, d.x};
    r[1] = float4{d.y, d.z, t.x, t.y};
    r[2] = float4{t.z, u2f(pixel), u2f(s.a), u2f(s.b)};
    r[3] = float4{u2f(s.c), u2f(s.counter), u2f(depth), 0.0f};
}

// The two constants of a GLASS material that glass_children needs: eta = 1 / ior and Schlick's r0 = ((1 - ior) / (1 + ior))^2 -- two
// divisions per glass hit.  A kernel that keeps the scene in LDS computes them once per workgroup, with the same operations, into the
// two words of the material record that GLASS leaves unused (mb.z, mb.w of the LDS copy); one that reads the scene through scalar
// loads computes them at the hit.
struct GlassConstants { float eta, r0; };
__device__ __forceinline__ GlassConstants glass_constants(float ior)
{
    GlassConstants g;
    g.eta = 1.0f / ior;
    const float q = (1.0f - ior) / (1.0f + ior);
    g.r0 = q * q;
    return g;
}
__device__ __forceinline__ void stage_glass_constants(float4 *lds, const SceneView &scene)
{
    float4 *M = lds + scene.geom_f4();
    const int n = scene.n_spheres + scene.n_planes;
    for (int i = threadIdx.x; i < n; i += kRenderBlock) {
        float4 mb = M[2 * i + 1];
        if (f2u(mb.x) == 2u) { const GlassConstants g = glass_constants(mb.y); mb.z = g.eta; mb.w = g.r0; M[2 * i + 1] = mb; }
    }
}
template <bool LDS_SCENE>
__device__ __forceinline__ GlassConstants glass_constants_of(float4 mb)
{
    if (LDS_SCENE) { GlassConstants g; g.eta = mb.z; g.r0 = mb.w; return g; }
    return glass_constants(mb.y);
}

// GLASS ior (extension): reflection child + refraction child; see the oracle's glass_children for the spec.
__device__ __forceinline__ void glass_children(V3 color, GlassConstants gc, V3 p, V3 n, V3 d, V3 throughput, Sfc32 seed,
                                               V3 o_out[2], V3 d_out[2], V3 t_out[2], Sfc32 s_out[2])
{
    (void)gen_component(seed); (void)gen_component(seed); (void)gen_component(seed);   // genVec is drawn before the match
    const float dn = dot(d, n);
    const float cosi = -dn;
    const float eta = gc.eta;
    const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
    const V3 reflection = d - scale_l(2.0f * dn, n);
    const float r0 = gc.r0;
    const float mm = 1.0f - cosi;
    float R = r0 + (1.0f - r0) * (((mm * mm) * (mm * mm)) * mm);
    V3 refraction;
    if (k < 0.0f) { R = 1.0f; refraction = reflection; }
    else refraction = scale_l(eta, d) + scale_l(eta * cosi - __builtin_sqrtf(k), n);
    o_out[0] = p + scale_r(reflection, kEpsilon); d_out[0] = reflection;
    t_out[0] = throughput * scale_r(color, R);
    s_out[0] = seed;
    o_out[1] = p + scale_r(refraction, kEpsilon); d_out[1] = refraction;
    t_out[1] = throughput * scale_r(color, 1.0f - R);
    (void)random_float(seed);
    s_out[1] = seed;
}

// ---------------------------------------------------------------------------------------
// render Streams for scenes whose rays SPLIT (the build-defined GLASS extension), per-pixel form: the tree walk.
// A lane owns a pixel and walks each sample's ray TREE depth first: at a GLASS hit the reflection child continues in
// the lane and the refraction child waits on a lane-private stack (scratch memory); when a lineage ends the lane pops
// the most recent waiting child, and only when the stack is empty does it go on with the sample's next start hit or
// the pixel's next sample.  Compared with the stream form below: no ray ever travels through HBM queues, a colour word
// has ONE adder (no atomics, and the order of a pixel's additions is defined: depth first, reflection before
// refraction -- oracle: ora_render_streams_tree, bit-exact), and the waves are dispatched by recorded cost, exactly as in
// render_streams_kernel.  The set of rays traced is the stream algorithm's (same children, same seeds, same step
// indices); a child that finds kTreeStackDepth children waiting in its lane is dropped and counted.
//
// THE START RECORD.  Every sample of a pixel shoots the same primary ray (Trace.hs:244-262), and a GLASS hit involves no
// random draw that changes a direction (glass_children only ADVANCES the seed): if the primary hit is glass, its two
// children are the same two rays in every sample too.  So what is evaluated once per pixel and kept in a lane-private
// LDS column is not only the primary hit but, for a glass primary hit, the first hit of EACH child -- the hits a sample
// starts from (0, 1 or 2 of them; a child that misses contributes nothing).  A sample then adds the glass hit's
// emittance (the same value every time), counts its two children, and works through its start hits in tree order, each
// with the seed its ray would carry: the sample's seed advanced by 3 (reflection) or 4 (refraction) raw draws.  Two
// traces and one glass evaluation per sample disappear for such pixels; everything downstream -- including start hits
// that are glass themselves -- takes the general path.  (With a step cap below 3 the children could be cut: no caching.)
// ---------------------------------------------------------------------------------------
#ifndef PTMI_TREE_WAVES
#define PTMI_TREE_WAVES 6        // 80 VGPRs (three values spilled around the shade) and a start record of 2 x 10 words: 8.39 -> 7.96 ms on the glass scene
#endif
template <bool LDS_SCENE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, PTMI_TREE_WAVES) render_streams_tree_kernel(const RenderArgs a)
{
    // two start hits per lane: position (3), incoming direction (3), throughput (3), primitive | steps << 16 | draws << 24
    // (The start hits' normals are recomputed at every sample start -- normal_at, the second half of hit_record.  Kept in LDS they save
    // 2.4 % at 5 waves per SIMD (8.39 -> 8.19 ms), but six waves' columns then no longer fit a CU, and the sixth wave is worth 5 %.)
#ifndef PTMI_TREE_NORMAL_LDS
#define PTMI_TREE_NORMAL_LDS 0
#endif
    constexpr int kEntry = PTMI_TREE_NORMAL_LDS ? 13 : 10;   // words per start hit
    __shared__ uint32_t start_rec[2 * kEntry][kRenderBlock];

    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
        stage_glass_constants(lds_scene, a.scene);
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    unsigned int wg; int chunk, n_spp_chunk;
    enter_sample_chunk<TILE_W>(a, wg, chunk, n_spp_chunk);            // sample chunks, as in render_inline_kernel
    long long pixel;
    unsigned int quad, trips = 0;
    const bool valid = lane_pixel<TILE_W>(a, pixel, quad, wg);
    unsigned int live = 0, longest = 0, cut = 0, dropped = 0;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        const int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);
        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        Sfc32 pixel_seed;
        pixel_seed.a = a.planes.sa[pixel]; pixel_seed.b = a.planes.sb[pixel];
        pixel_seed.c = a.planes.sc[pixel]; pixel_seed.counter = a.planes.sctr[pixel];
        const int n_spp = n_spp_chunk;

        const HitSel h0 = check_hit(S, ns, np, origin, primary);   // same primary ray for every sample
        if (!h0.just) {
            for (int s = 0; s < n_spp; ++s) (void)random_float(pixel_seed);     // updateSeed only
        } else {
            uint32_t *rec = &start_rec[0][threadIdx.x];
            auto put_entry = [&](int e, V3 p, V3 nrm, V3 dir, V3 t, int prim, unsigned int steps_done, unsigned int draws) {
                uint32_t *q = rec + (size_t)e * kEntry * kRenderBlock;
                q[0] = f2u(p.x); q[kRenderBlock] = f2u(p.y); q[2 * kRenderBlock] = f2u(p.z);
                q[3 * kRenderBlock] = f2u(dir.x); q[4 * kRenderBlock] = f2u(dir.y); q[5 * kRenderBlock] = f2u(dir.z);
                q[6 * kRenderBlock] = f2u(t.x); q[7 * kRenderBlock] = f2u(t.y); q[8 * kRenderBlock] = f2u(t.z);
                q[9 * kRenderBlock] = (uint32_t)prim | (steps_done << 16) | (draws << 24);
                if (PTMI_TREE_NORMAL_LDS) { q[10 * kRenderBlock] = f2u(nrm.x); q[11 * kRenderBlock] = f2u(nrm.y); q[12 * kRenderBlock] = f2u(nrm.z); }
            };
            V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
            hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
            int n_entries = 0;
            bool first_is_reflection = false;
            V3 emit0 = mk(0.0f, 0.0f, 0.0f);
            const float4 ma0 = M[2 * h0.idx], mb0 = M[2 * h0.idx + 1];
#ifdef PTMI_TREE_NO_PREFIX
            const bool prefix = false;
#else
            const bool prefix = f2u(mb0.x) == 2u && step_cap >= 3u;   // a glass primary hit whose children cannot be cut
#endif
            if (prefix) {
                emit0 = scale_r(mk(ma0.x, ma0.y, ma0.z), ma0.w) * mk(1.0f, 1.0f, 1.0f);     // computeResult of the primary hit
                V3 ko[2], kd[2], kt[2]; Sfc32 ks[2];
                glass_children(mk(ma0.x, ma0.y, ma0.z), glass_constants_of<LDS_SCENE>(mb0), pos, normal, primary, mk(1.0f, 1.0f, 1.0f), pixel_seed, ko, kd, kt, ks);
                for (int k = 0; k < 2; ++k) {
                    const V3 ro = k == 0 ? ko[0] : ko[1], rd = k == 0 ? kd[0] : kd[1], rt = k == 0 ? kt[0] : kt[1];
                    const HitSel h = check_hit(S, ns, np, ro, rd);
                    if (h.just) {
                        V3 hp, hn;
                        hit_record(S, ns, h.idx, ro, rd, h.t, hp, hn);
                        if (n_entries == 0) first_is_reflection = k == 0;
                        put_entry(n_entries++, hp, hn, rd, rt, h.idx, 1u, (unsigned int)k);
                    }
                }
                // THE LEAD SEED.  A sample's reflection child carries the sample's seed advanced by 3 raw draws, its refraction
                // child by 4, and updateSeed moves the pixel's seed on by one: so from here to the end of the pixel `pixel_seed`
                // holds the pixel's seed advanced by 3 -- the reflection's seed as it stands, the refraction's one step further --
                // and is stepped back three times before it is stored (sfc32_prev, the exact inverse).  Seven SFC32 steps per
                // sample become at most two.
                (void)sfc32_next(pixel_seed); (void)sfc32_next(pixel_seed); (void)sfc32_next(pixel_seed);
            } else {
                put_entry(0, pos, normal, primary, mk(1.0f, 1.0f, 1.0f), h0.idx, 0u, 0u);
                n_entries = 1;
            }
            // children waiting for this lane: origin, direction, throughput, seed, step index (RayState, Trace.hs:45); scratch
            // memory.  (Before the start record existed, every sample of a glass pixel pushed a child and the first entry
            // lived in LDS: 23 GB -> 1.4 GB of scratch writes per launch.  With the primary split cached, pushes are rare,
            // and an LDS entry beside the start record would cost a wave of occupancy: 10.2 ms instead of 9.1.)
            // The lane's waiting children, a stack.  Its first kTreeFastLevels entries are 64-byte records in a global-memory block
            // laid out [tile][level][lane] -- a push is four 16-byte stores to ONE line, a pop four loads -- and only deeper
            // entries live in scratch memory, where a push is fourteen lane-strided dwords, each a partial line: with the
            // whole stack in scratch the kernel wrote 4 GB per 1080p / 64-spp launch (70 times the planes).
            uint32_t stack_w[kTreeStackDepth - kTreeFastLevels][14];
            float4 *const fast = a.tree_stack + ((size_t)wg * kTreeFastLevels * kRenderBlock + threadIdx.x) * 4;   // (never NULL: the call fails without the block)
            int sp = 0, entry_i = 0;
            int s = 0, idx = h0.idx;
            unsigned int steps = 0, deepest = 0;                 // deepest: traceSteps of the sample's longest lineage
            V3 d = primary;
            V3 throughput = mk(1.0f, 1.0f, 1.0f);
            Sfc32 seed = pixel_seed;
            bool pending = false, has_ray = false;
            auto begin_sample = [&]() {                           // what every sample of this pixel has already behind it
                if (prefix) { acc = acc + emit0; live += 2u; }
                entry_i = 0; deepest = prefix ? 2u : 1u;
            };
            // the next hit this sample starts from -- or, when it has none left, the end of the sample and the next one
            auto next_start = [&]() {
                for (;;) {
                    if (entry_i < n_entries) {
                        const uint32_t *q = rec + (size_t)entry_i * kEntry * kRenderBlock;
                        pos = mk(u2f(q[0]), u2f(q[kRenderBlock]), u2f(q[2 * kRenderBlock]));
                        d = mk(u2f(q[3 * kRenderBlock]), u2f(q[4 * kRenderBlock]), u2f(q[5 * kRenderBlock]));
                        throughput = mk(u2f(q[6 * kRenderBlock]), u2f(q[7 * kRenderBlock]), u2f(q[8 * kRenderBlock]));
                        idx = (int)(q[9 * kRenderBlock] & 0xffffu);
                        const uint32_t meta = q[9 * kRenderBlock] >> 16;
                        ++entry_i;
                        steps = meta & 0xffu;
                        normal = PTMI_TREE_NORMAL_LDS ? mk(u2f(q[10 * kRenderBlock]), u2f(q[11 * kRenderBlock]), u2f(q[12 * kRenderBlock])) : normal_at(S, ns, idx, pos);
                        seed = pixel_seed;                        // (with a prefix: the lead seed -- the draws the ray's ancestors made)
                        if (meta >> 8) (void)sfc32_next(seed);    // the refraction child: one more
                        pending = true; has_ray = false;
                        return;
                    }
                    (void)random_float(pixel_seed);               // updateSeed: the sample's tree is done
                    ++s; longest = deepest > longest ? deepest : longest;
                    if (s >= n_spp) { pending = false; has_ray = false; return; }
                    begin_sample();
                }
            };
            auto lineage_ended = [&]() {
                if (sp > 0) {                                     // the most recent waiting child
                    --sp;
                    uint32_t e[14];
                    if (sp < kTreeFastLevels) {
                        const float4 *r = fast + (size_t)sp * kRenderBlock * 4;
                        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                        e[0] = f2u(r0.x); e[1] = f2u(r0.y); e[2] = f2u(r0.z); e[3] = f2u(r0.w); e[4] = f2u(r1.x); e[5] = f2u(r1.y); e[6] = f2u(r1.z);
                        e[7] = f2u(r1.w); e[8] = f2u(r2.x); e[9] = f2u(r2.y); e[10] = f2u(r2.z); e[11] = f2u(r2.w); e[12] = f2u(r3.x); e[13] = f2u(r3.y);
                    } else {
                        const int q0 = sp - kTreeFastLevels;
                        for (int q = 0; q < 14; ++q) e[q] = stack_w[q0 < kTreeStackDepth - kTreeFastLevels ? q0 : 0][q];
                    }
                    pos = mk(u2f(e[0]), u2f(e[1]), u2f(e[2]));
                    d = mk(u2f(e[3]), u2f(e[4]), u2f(e[5]));
                    throughput = mk(u2f(e[6]), u2f(e[7]), u2f(e[8]));
                    seed.a = e[9]; seed.b = e[10]; seed.c = e[11]; seed.counter = e[12];
                    steps = e[13];
                    pending = false; has_ray = true;
                } else {
                    next_start();
                }
            };
            if (n_spp > 0) { begin_sample(); next_start(); }
#ifdef PTMI_TREE_STATS
            unsigned int st_dead = 0, st_shade = 0, st_trace = 0;      // this lane's participation per round (diagnostic build)
#endif
            bool ended = false;                                   // the lane's lineage is over: its next piece of work is fetched at the top of the trip
            while (pending || has_ray || ended) {
                ++trips;
#ifdef PTMI_TREE_STATS
                if (pending && !has_ray && near_zero(throughput)) ++st_dead;
#endif
                // shade round.  A ray whose throughput is already near zero dies at this hit (numNewRays): the hit adds its
                // emittance and nothing else of it survives, so such lanes skip the expensive half and go on with their most
                // recent waiting child, their sample's next start hit or the pixel's next sample -- in the latter cases they
                // take part in this round's full shade.  A lineage that ended in the previous trip (miss, cap) fetches its next
                // piece of work here too: lineage_ended is a large block -- next start hit, its normal, its seed -- and is expanded
                // once.
                if (pending && !has_ray && near_zero(throughput)) {
                    const float4 ma = M[2 * idx];
                    acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);   // computeResult (Trace.hs:318-323)
                    ++steps;
                    pending = false; ended = true;
                }
                if (ended) { lineage_ended(); ended = false; }     // the one expansion of that block (it is large)
                // (The start hit that the block above may just have loaded can itself belong to a dead ray -- a reflection of
                // weight ~0: it must not be shaded; it waits for the next trip's dead-ray block.  A test inside next_start
                // instead cost 12 %.)
                if (pending && !has_ray && !near_zero(throughput)) {   // alive
#ifdef PTMI_TREE_STATS
                    ++st_shade;
#endif
                    const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
                    const bool capped = steps + 1u >= step_cap;
                    if (f2u(mb.x) == 2u) {                        // GLASS: two children (extension; spec = the oracle's glass_children)
                        acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
                        ++steps;
                        V3 ko[2], kd[2], kt[2]; Sfc32 ks[2];
                        glass_children(mk(ma.x, ma.y, ma.z), glass_constants_of<LDS_SCENE>(mb), pos, normal, d, throughput, seed, ko, kd, kt, ks);
                        live += 2u;
                        if (capped) { cut += 2u; pending = false; ended = true; }
                        else {
                            // while the cached reflection's subtree is walked, the cached refraction "waits": one slot less
                            if (sp < kTreeStackDepth - ((prefix && entry_i == 1 && first_is_reflection) ? 1 : 0)) {
                                if (sp < kTreeFastLevels) {
                                    float4 *r = fast + (size_t)sp * kRenderBlock * 4;
                                    r[0] = float4{ko[1].x, ko[1].y, ko[1].z, kd[1].x};
                                    r[1] = float4{kd[1].y, kd[1].z, kt[1].x, kt[1].y};
                                    r[2] = float4{kt[1].z, u2f(ks[1].a), u2f(ks[1].b), u2f(ks[1].c)};
                                    r[3] = float4{u2f(ks[1].counter), u2f(steps), 0.0f, 0.0f};
                                } else {
                                    const uint32_t e[14] = {f2u(ko[1].x), f2u(ko[1].y), f2u(ko[1].z), f2u(kd[1].x), f2u(kd[1].y), f2u(kd[1].z),
                                                            f2u(kt[1].x), f2u(kt[1].y), f2u(kt[1].z), ks[1].a, ks[1].b, ks[1].c, ks[1].counter, steps};
                                    const int q0 = sp - kTreeFastLevels;
                                    for (int q = 0; q < 14; ++q) stack_w[q0 < kTreeStackDepth - kTreeFastLevels ? q0 : 0][q] = e[q];
                                }
                                ++sp;
                            } else {
                                ++dropped;
                            }
                            pos = ko[0]; d = kd[0]; throughput = kt[0]; seed = ks[0];
                            pending = false; has_ray = true;
                        }
                    } else {
                        // results: colour += emittance * throughput for EVERY hit; then the new ray
                        shade(M, idx, pos, normal, pos, d, throughput, acc, seed);
                        ++steps; ++live;
                        pending = false;
                        if (capped) { ++cut; ended = true; }
                        else has_ray = true;
                    }
                }
                if (has_ray) {
#ifdef PTMI_TREE_STATS
                    ++st_trace;
#endif
                    deepest = steps + 1u > deepest ? steps + 1u : deepest;
                    const HitSel h = check_hit(S, ns, np, pos, d);
                    has_ray = false;
                    if (h.just) {
                        hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                        idx = h.idx;
                        pending = true;
                    } else {
                        ended = true;
                    }
                }
            }
#ifdef PTMI_TREE_STATS
            // [1] lane-trips needed, [2] dead-ray finishes, [3] shades, [4] traces (lane participations)
            atomicAdd(a.work_counter + 1, trips); atomicAdd(a.work_counter + 2, st_dead);
            atomicAdd(a.work_counter + 3, st_shade); atomicAdd(a.work_counter + 4, st_trace);
#endif
            if (prefix) { sfc32_prev(pixel_seed); sfc32_prev(pixel_seed); sfc32_prev(pixel_seed); }     // the lead seed back to the pixel's
        }
#ifdef PTMI_TREE_STATS_MAP
        acc.x = (float)trips;                                         // diagnostic build: the red plane becomes the per-pixel cost map
#endif
        a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
        a.planes.sa[pixel] = pixel_seed.a; a.planes.sb[pixel] = pixel_seed.b;
        a.planes.sc[pixel] = pixel_seed.c; a.planes.sctr[pixel] = pixel_seed.counter;
    }
    leave_sample_chunk<TILE_W>(a, wg, chunk);
#ifdef PTMI_TREE_STATS
    {   // [5] lane-trips the wave paid for: its longest lane x 64
        unsigned int mx = trips;
        for (int off = 32; off > 0; off >>= 1) { const unsigned int o2 = __shfl_xor(mx, off, 64); mx = o2 > mx ? o2 : mx; }
        if ((threadIdx.x & 63) == 0) atomicAdd(a.work_counter + 5, mx * 64u);
    }
#endif
    if (TILE_W > 0) record_cost(a, quad, trips);
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
    if (a.stream_iterations) {
        for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(longest, off, 64); longest = other > longest ? other : longest; }
        if ((threadIdx.x & 63) == 0 && longest) atomicMax(a.stream_iterations + (size_t)(blockIdx.x & (kStatShards - 1)) * (2 * kStatStride), longest);
    }
    if (__any((cut | dropped) != 0u)) {                               // rare
        const unsigned long long n_cut = wave_sum(cut), n_dropped = wave_sum(dropped);
        if ((threadIdx.x & 63) == 0) {
            if (n_cut) atomicAdd(a.stream_counters + kScTruncated, n_cut);
            if (n_dropped) atomicAdd(a.stream_counters + kScDropped, n_dropped);
        }
    }
}

// ---------------------------------------------------------------------------------------
// render Streams as a stream ("wavefront" form), the kernels.
//
//   streams_primary_kernel   every sample of a pixel shoots the same primary ray (Trace.hs:244-262), so its checkHit + hit are
//                            evaluated ONCE per render call; the hits the samples start from go into the start-hit list, in
//                            regions of one 64-pixel tile each, compacted inside the wave by ballot + popcount prefix (HitList);
//                            pixels whose primary ray misses never enter a stream.
//   streams_pixels_kernel    scenes whose rays never split (and PTMI_OPT_STREAM_BATCH = 0): one launch, persistent waves take
//                            the regions as chunks (the first gridDim statically, later ones by ticket), a lane takes a start
//                            hit and renders ALL samples of its pixel from it, colour and seed in registers -- read once,
//                            written once, no atomics, additions in sample order: bit-identical to the per-pixel kernel and
//                            the oracle under both seed rules.  `expand` (Trace.hs:284-289) with numNewRays in {0, 1} is "the
//                            child is the lane's next ray"; a lane whose pixel is done REFILLS from the wave's chunk (ballot of
//                            the idle lanes + popcount prefix), so the rounds stay dense although pixels differ in cost.
//   streams_split_kernel     scenes with a ray-splitting material (the build-defined GLASS), or samples cut into unordered
//                            items (PTMI_OPT_STREAM_BATCH): the same persistent shape; an item is (start hit, a range of the
//                            pixel's samples).  At a GLASS hit the reflection stays in the lane and the refraction goes into the
//                            wave's CHILD RING in LDS -- `expand` as wave-level compaction: ballot of the emitting lanes,
//                            popcount prefix for the slot -- from which lanes that have no ray take their next one (ballot +
//                            prefix again) before they start their item's next sample.  Nearly every child is traced by the
//                            wave that emitted it, in the same launch; only when the ring is full does a child travel through
//                            the overflow stream in HBM (blocks of slots reserved per wave, one atomic per 256 children).
//                            `permute (+)` (Trace.hs:179-184): a lane's own lineages add into its LDS accumulator, flushed with
//                            one float atomic per colour word per item; rays taken from the ring add with float atomics (exact
//                            zeros skipped).  The order of a pixel's additions is undefined, as in Accelerate's permute.
//   streams_level_kernel     the overflow levels: one launch per level of what is left in the HBM stream (usually nothing).
//   streams_seeds_kernel     updateSeed (Trace.hs:190-191) for every pixel and sample of the call, with the seed each item
//                            starts from recorded on the way (split kernel only).
// Every ray carries its step index (the `awhile` iteration it belongs to), so the safety cap cuts the same rays as in
// the other forms; cut rays, dropped children (overflow stream full) and emitted children are counted.
// ---------------------------------------------------------------------------------------
constexpr unsigned int kHole = 0xffffffffu;                  // pixel word of an unused output slot
constexpr unsigned int kFirstBlock = 64, kNextBlock = 256;   // output slots a wave owns at start / reserves per atomic
#ifndef PTMI_LEVEL_WAVES
#define PTMI_LEVEL_WAVES 6
#endif
#ifndef PTMI_REFILL_BATCH
#define PTMI_REFILL_BATCH 8
#endif
constexpr unsigned int kRefillBatch = PTMI_REFILL_BATCH;     // overflow levels: idle lanes a wave waits for before it runs the refill block

// 16 bytes past the L1 (global_load_dwordx4 ... nt): for records this wave wrote itself a few trips ago
__device__ __forceinline__ float4 load_past_l1(const float4 *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return float4{v.x, v.y, v.z, v.w};
}
// Element `byte_off / 4` of a plane through a 32-bit byte offset (the stream form holds < 2^30 pixels): the access becomes
// scalar base + 32-bit vector offset instead of a 64-bit address pair per plane.
template <typename T> __device__ __forceinline__ T &plane_at(T *base, uint32_t byte_off)
{
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}
// agent-scope relaxed accesses (global_load / global_store ... sc1): the load passes the CU's L1 by, the store is written through
__device__ __forceinline__ float load_agent(const float *p) { return u2f(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ uint32_t load_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(float *p, float v) { __hip_atomic_store(reinterpret_cast<uint32_t *>(p), f2u(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The start-hit list.  One workgroup = one quad of four x-adjacent 8x8 tiles (TILES) or 256 consecutive pixels, one wave
// = one region.  Dispatch position p works on quad quad_order[p] (most expensive first, once costs are known), and its
// regions are 4 p .. 4 p + 3: the list is in dispatch order, which is the order the item kernels hand the chunks out.
// A glass primary hit is replaced by the first hits of its two children when that changes nothing observable: the
// step cap cannot cut the children (>= 3) and the glass hit itself emits nothing (its emittance would have to be added
// once per sample).  advance_missed: a pixel without start hits gets its updateSeeds here (streams_pixels_kernel does
// the others' itself).  `counters`: the stream form's counter block (kLvSplitPixels, kLvDeepest).
// which pixel a lane of the primary kernel (and of the kernel that advances the missed pixels' seeds) looks at
template <bool TILES>
__device__ __forceinline__ bool primary_pixel(const RenderArgs &a, unsigned int &quad, unsigned int &region, long long &pixel)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int position = blockIdx.x;
    quad = (TILES && a.quad_order) ? a.quad_order[position] : position;
    region = position * 4u + (unsigned int)wave;
    if (TILES) {
        const unsigned int tile = quad * 4u + (unsigned int)wave;
        const int tiles_x = (a.width + 7) / 8;
        const int tx = (int)(tile % (unsigned)tiles_x), ty = (int)(tile / (unsigned)tiles_x);
        const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
        pixel = (long long)y * a.width + x;
        return x < a.width && y < a.rows_local;
    }
    pixel = (long long)region * 64 + lane;
    return pixel < (long long)a.rows_local * a.width;
}

template <bool TILES>
__global__ void __launch_bounds__(256) streams_primary_kernel(const RenderArgs a, const HitList out, unsigned int *counters)
{
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    const float4 *S = a.scene.packed;                        // one evaluation per pixel and call: scalar loads will do
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    unsigned int quad, region; long long pixel;
    const bool valid = primary_pixel<TILES>(a, quad, region, pixel);
    int n_rec = 0;                                            // records this pixel contributes: 0, 1 or 2
    V3 pos[2], nor[2], dir[2], thr[2];
    int prim[2] = {0, 0};
    uint32_t meta[2] = {0u, 0u};
    bool split = false;
    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        const V3 primary = primary_direction(a.cam, col, global_row(local_row, a.stripe_rows, a.n_parts, a.part));
        const HitSel h = check_hit(S, ns, np, a.cam.pos, primary);
        if (h.just) {
            V3 p0, n0;
            hit_record(S, ns, h.idx, a.cam.pos, primary, h.t, p0, n0);
            const float4 ma = M[2 * h.idx], mb = M[2 * h.idx + 1];
            const V3 emit = scale_r(mk(ma.x, ma.y, ma.z), ma.w) * mk(1.0f, 1.0f, 1.0f);
            split = out.region_slots > 64u && f2u(mb.x) == 2u && a.stream_step_cap >= 3 && emit.x == 0.0f && emit.y == 0.0f && emit.z == 0.0f;
            if (split) {
                V3 ko[2], kd[2], kt[2]; Sfc32 ks[2]; Sfc32 dummy; dummy.a = dummy.b = dummy.c = dummy.counter = 0;
                glass_children(mk(ma.x, ma.y, ma.z), glass_constants(mb.y), p0, n0, primary, mk(1.0f, 1.0f, 1.0f), dummy, ko, kd, kt, ks);
                for (int k = 0; k < 2; ++k) {
                    const V3 ro = k == 0 ? ko[0] : ko[1], rd = k == 0 ? kd[0] : kd[1], rt = k == 0 ? kt[0] : kt[1];
                    const HitSel hc = check_hit(S, ns, np, ro, rd);
                    if (hc.just) {
                        V3 hp, hn;
                        hit_record(S, ns, hc.idx, ro, rd, hc.t, hp, hn);
                        const int e = n_rec++;
                        if (e == 0) { pos[0] = hp; nor[0] = hn; dir[0] = rd; thr[0] = rt; prim[0] = hc.idx; meta[0] = 1u | ((3u + (unsigned int)k) << 8); }
                        else        { pos[1] = hp; nor[1] = hn; dir[1] = rd; thr[1] = rt; prim[1] = hc.idx; meta[1] = 1u | ((3u + (unsigned int)k) << 8); }
                    }
                }
            } else if (out.region_slots == 64u) {
                // streams_pixels_kernel's record: what every first shade of the pixel uses -- the axis and the half-angle scale
                // of its bounce (the same operations on the same inputs, once per pixel) -- in the normal's and direction's place
                V3 axis0; float hk0;
                bounce_axis(mb, n0, primary, axis0, hk0);
                pos[0] = p0; nor[0] = axis0; dir[0] = mk(hk0, 0.0f, 0.0f); thr[0] = mk(1.0f, 1.0f, 1.0f); prim[0] = h.idx; meta[0] = 0u;
                n_rec = 1;
            } else {
                pos[0] = p0; nor[0] = n0; dir[0] = primary; thr[0] = mk(1.0f, 1.0f, 1.0f); prim[0] = h.idx; meta[0] = 0u;
                n_rec = 1;
            }
        }
    }
    const unsigned long long m0 = __ballot(n_rec > 0), m1 = __ballot(n_rec > 1), ms = __ballot(split);
    const unsigned long long missed = __ballot(valid && n_rec == 0);      // updateSeed is all a sample does to these pixels
    const unsigned int c0 = (unsigned int)__builtin_popcountll(m0), c1 = (unsigned int)__builtin_popcountll(m1);
    if (lane == 0) {
        out.counts[region] = c0 + c1;
        out.missed[region] = missed;
        if (ms) {
            atomicAdd(counters + kLvSplitPixels * kCounterStride, (unsigned int)__builtin_popcountll(ms));
            if (2u > __hip_atomic_load(counters + kLvDeepest * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(counters + kLvDeepest * kCounterStride, 2u);       // the children's traceStep
        }
    }
    const unsigned int slot0 = region * out.region_slots;
    for (int e = 0; e < 2; ++e) {
        if (n_rec > e) {
            const unsigned int i = slot0 + (e == 0 ? rank_in(m0) : c0 + rank_in(m1));
            const V3 p = e == 0 ? pos[0] : pos[1], n = e == 0 ? nor[0] : nor[1], dd = e == 0 ? dir[0] : dir[1], tt = e == 0 ? thr[0] : thr[1];
            float4 *r = out.record(i);
            r[0] = float4{p.x, p.y, p.z, n.x};
            r[1] = float4{n.y, n.z, dd.x, dd.y};
            r[2] = float4{dd.z, tt.x, tt.y, tt.z};
            r[3] = float4{u2f((uint32_t)(e == 0 ? prim[0] : prim[1])), u2f((uint32_t)pixel), u2f(e == 0 ? meta[0] : meta[1]), u2f(quad)};
        }
    }
}

// updateSeed (Trace.hs:190-191) for the pixels WITHOUT start hits (their primary ray misses, or both children of their glass
// primary hit do): `draws` draws each.  The pixels with start hits are streams_pixels_kernel's.  Same pixel mapping as the
// primary kernel, whose missed[] masks (one per region) say which lanes have work.
template <bool TILES>
__global__ void __launch_bounds__(256) streams_advance_missed_kernel(const RenderArgs a, const HitList hits, int draws, const unsigned int *tail_start)
{
    if (tail_start && blockIdx.x >= *tail_start) return;      // (the per-pixel tail renders those positions whole)
    unsigned int quad, region; long long pixel;
    const bool valid = primary_pixel<TILES>(a, quad, region, pixel);
    const unsigned long long missed = hits.missed[region];
    if (!valid || !((missed >> (threadIdx.x & 63)) & 1ull)) return;
    Sfc32 sd; sd.a = a.planes.sa[pixel]; sd.b = a.planes.sb[pixel]; sd.c = a.planes.sc[pixel]; sd.counter = a.planes.sctr[pixel];
    for (int j = 0; j < draws; ++j) (void)random_float(sd);
    a.planes.sa[pixel] = sd.a; a.planes.sb[pixel] = sd.b; a.planes.sc[pixel] = sd.c; a.planes.sctr[pixel] = sd.counter;
}

// The chunk cursor of the item kernels.  A chunk is 64 slots of a region of the start-hit list; the regions come in groups
// of four, one group per dispatch POSITION (the four tiles of a quad, most expensive quad first).  The positions are dealt to
// eight queues, position p to queue p mod 8 -- one queue per XCD -- so that the tiles of a quad, whose pixels share cache lines
// of the planes, are worked on behind ONE L2 (dealt to any XCD, every line of the planes was fetched four times).  A queue is a
// ticket counter: ticket j stands for chunk (j mod n) of pass (j div n), n = the queue's chunks, passes outermost and
// positions in dispatch order.  A wave takes tickets -- one returning atomic each; an item is tens to thousands of loop trips
// -- from the queue of the XCD it runs on (HW_REG_XCC_ID; which wave works on which chunk changes no result) and, when that one
// is exhausted, from the other XCDs' queues.  Eight counters instead of one: a single word serves ~90 atomics per microsecond
// and thousands of waves start together.  Regions without records (tiles whose primary rays all miss) are skipped.
struct ChunkCursor {
    unsigned int taken, len, first, pass;    // of the chunk in hand: records handed out, records, first slot, pass
    unsigned int region;                     // ... its region
    unsigned int home, tries;                // the wave's XCD; queues found exhausted (8: nothing is left)
    unsigned int n_positions;                // dispatch positions the kernel works on (the first ones of the order)
    bool ready;                              // (ordered passes) the chunk's previous pass has been published and acquired
};
__device__ __forceinline__ unsigned int xcc_id()
{
    return (unsigned int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;     // HW_REG_XCC_ID, bits [3:0]
}
__device__ __forceinline__ bool chunks_left(const ChunkCursor &c) { return c.tries < 8u; }
__device__ __forceinline__ void next_chunk(ChunkCursor &c, const ItemArgs &it)
{
    const unsigned int per = it.hits.region_slots >> 6;      // chunks per region: 1 or 2
    c.taken = 0; c.len = 0; c.ready = false;
    while (c.tries < 8u) {
        const unsigned int q = (c.home + c.tries) & 7u;
        // positions in queue q: p = 8 s + q < n_positions
        const unsigned int n_pos = c.n_positions > q ? (c.n_positions - q - 1u) / 8u + 1u : 0u;
        const unsigned int n = n_pos * 4u * per;
        unsigned int j = 0;
        if ((threadIdx.x & 63) == 0) j = atomicAdd(it.chunk_cursor + (size_t)q * kCounterStride, 1u);
        j = (unsigned int)__builtin_amdgcn_readfirstlane((int)j);
        if (n == 0u || j / n >= (unsigned int)it.passes) { ++c.tries; continue; }
        c.pass = j / n;
        const unsigned int k = j - c.pass * n, s_pos = k / (4u * per), r = k - s_pos * (4u * per);
        c.region = (s_pos * 8u + q) * 4u + r / per;
        const unsigned int half = r % per;
        const unsigned int have = it.hits.counts[c.region];
        c.first = c.region * it.hits.region_slots + half * 64u;
        c.len = have > half * 64u ? (have - half * 64u < 64u ? have - half * 64u : 64u) : 0u;
        if (c.len) return;
    }
}

// what an item cost, for the dispatch order of later launches with the same key: the loop trips the lane spent on it (or the
// hits it shaded, one per trip)
__device__ __forceinline__ void record_item_cost(const RenderArgs &a, unsigned int quad, unsigned int trips)
{
    if (a.quad_cost) atomicAdd(a.quad_cost + quad, trips);
}

// ---------------------------------------------------------------------------------------
// streams_pixels_kernel: the stream form for scenes whose rays never split.  The loop is render_streams_kernel's
// [finish dead rays][next sample][shade][trace] with a [refill] block in front: a lane whose pixel is done stores its
// seven words and becomes idle; idle lanes take the next start hits of the wave's chunk.
// ---------------------------------------------------------------------------------------
#ifndef PTMI_PIXELS_WAVES
#define PTMI_PIXELS_WAVES 7
#endif
constexpr int kMinPassSamples = 1;                           // the fewest samples an ordered pass may hold (a lane publishes an item before it takes the next)
template <bool LDS_SCENE, bool PASSES>
__global__ void __launch_bounds__(kRenderBlock, PTMI_PIXELS_WAVES) streams_pixels_kernel(const RenderArgs a, const ItemArgs it)
{
    // the lane's item: 0-2 position of the start hit, 3-5 axis and 6 half-angle scale of its bounce, 7 primitive, 8 quad,
    // 9 the lane's count of shaded hits when the item began, 10 the samples the item renders, 11 its region
    __shared__ float item_const[12][kRenderBlock];
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    // ORDERED PASSES (it.passes > 1).  A pixel's samples are a serial chain, and with few items per lane the end of the launch is as
    // long as the last items.  The samples are therefore cut into passes: an item renders one pass's samples of its pixel, and the
    // pixel's seven words travel through the planes to whichever lane -- of any wave, on any XCD -- takes its next pass.  What orders
    // them: an item of pass p is handed out only when region_done[its region] says that every item of the region's pass p - 1 has
    // been PUBLISHED.  The L2s of the eight XCDs are not coherent with each other and a CU's L1 is never refreshed, so the seven
    // words are stored WRITE-THROUGH (sc1: agent-scope relaxed atomic stores), the storing wave waits for its stores (vmcnt(0))
    // before its lanes add to the region's counter (agent-scope atomics), and the taking lanes read counter and words with sc1
    // loads, which pass the L1 by and are served coherently (MI355X_MICROARCH.md, "Valid forms": every store of the handed-off bytes
    // sc1 and drained before the counter moves, every load of them sc1; the loads come after the poll of the same wave).  No fence:
    // an agent-scope release is a write-back of the XCD's whole L2, and the L2 serves them one after the other -- 7 168 waves
    // releasing every 16 trips (112 write-backs per microsecond on the chip) DOUBLED the launch (1080p / 64 spp as four passes:
    // 4.5 -> 9.4 ms), and releasing in batches of up to 256 trips still cost more than short passes gained.
    // Nothing here depends on which XCD or CU a wave runs on.
    // (PASSES is a template parameter: carried as run-time branches the blocks below cost the one-pass kernel 3.8 % -- 4.57 -> 4.75 ms on
    // S16 -- in scalar registers spilled and instructions per trip)
    const int passes = PASSES ? it.passes : 1;
    float *mine = &item_const[0][threadIdx.x];
    auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
    auto get = [&](int k) { return mine[k * kRenderBlock]; };
    // PTMI_SEED_FROM_RESULT (combine new old): the seed the ray of the sample's last hit carried.  (Four registers, copied at every
    // hit; four LDS words written at every hit and four selects per sample cost 3 % more, stepping the lane's seed back over the
    // hit's draws at the sample's end -- sfc32_prev -- 2.5 %.)
    Sfc32 hit_seed; hit_seed.a = hit_seed.b = hit_seed.c = hit_seed.counter = 0;

    ChunkCursor cur; cur.home = xcc_id(); cur.tries = 0;
    // (the positions from *tail_start on -- the cheapest quads -- are the per-pixel kernel's, whose waves fill the slots this launch's
    // waves leave as they end: ptmi_api.cpp)
    cur.n_positions = it.n_positions;
    if (it.tail_start) { const unsigned int t = *it.tail_start; cur.n_positions = t < it.n_positions ? t : it.n_positions; }
    next_chunk(cur, it);
#ifdef PTMI_TAIL_STATS
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned long long lane_trips = 0, wave_trips = 0;        // lanes with an item, summed over the trips / trips
    unsigned long long ph_refill = 0, ph_over = 0, ph_refills = 0, ph_ends = 0, ph_taken = 0, ph_t = 0;   // (-DPTMI_TAIL_PHASES) cycles in the refill block / in the sample-end block of trips that end an item
#endif

    bool busy = false, pending = false, has_ray = false, over = false, unpublished = false;
    V3 acc = mk(0.0f, 0.0f, 0.0f), pos = acc, normal = acc, d = acc, throughput = acc;
    Sfc32 pixel_seed; pixel_seed.a = pixel_seed.b = pixel_seed.c = pixel_seed.counter = 0;
    Sfc32 seed = pixel_seed;
    int idx = 0, s = 0;
    uint32_t pixel4 = 0;                                      // byte offset of the lane's pixel in a plane
    unsigned int steps = 0, longest = 0, live = 0;
    for (;;) {
        // ---- publish (ordered passes): the items whose words this wave stored in its last trip (a trip ago: the wait is free)
        if (PASSES && __any(unpublished)) {                   // wave-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through stores have completed before the counters move
            if (unpublished) { atomicAdd(it.region_done + f2u(get(11)), 1u); unpublished = false; }
        }
        // ---- refill: idle lanes take the next start hits of the wave's chunk (at once: an item is a pixel's whole sample chain)
        const unsigned long long idle = __ballot(!busy);
        bool open = idle && chunks_left(cur);
        if (PASSES && open && cur.pass > 0u && !cur.ready) {           // wave-uniform: has the region's previous pass been published?
            unsigned int done = 0;
            if (lane == 0) done = __hip_atomic_load(it.region_done + cur.region, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done = (unsigned int)__builtin_amdgcn_readfirstlane((int)done);
            if (done >= cur.pass * cur.len) cur.ready = true;
            else { open = false; if (!__any(busy)) __builtin_amdgcn_s_sleep(8); }
        }
#ifdef PTMI_TAIL_STATS
        if (open) { ph_t = __builtin_readcyclecounter(); ++ph_refills; }
#endif
        if (open) {                                           // wave-uniform
            const unsigned int want = (unsigned int)__builtin_popcountll(idle), avail = cur.len - cur.taken;
            const unsigned int take = want < avail ? want : avail;
            const unsigned int rank = rank_in(idle);
            if (!busy && rank < take) {
                const float4 *r = it.hits.record(cur.first + cur.taken + rank);
                const float4 r0 = r[0], r1 = r[1], r3 = r[3];   // [position, axis x] [axis yz, half-angle scale, -] ... [primitive, pixel, -, quad]
                pixel4 = f2u(r3.y) << 2;
                put(0, r0.x); put(1, r0.y); put(2, r0.z);
                put(3, r0.w); put(4, r1.x); put(5, r1.y); put(6, r1.z);
                put(7, r3.x); put(8, r3.w); put(9, u2f(live));
                // the samples of this pass: n_spp over the passes, the first (n_spp mod passes) passes one more
                put(10, u2f((uint32_t)(a.n_spp / passes + ((int)cur.pass < a.n_spp % passes ? 1 : 0)))); put(11, u2f(cur.region));
                if (PASSES) {                                  // another wave's stores of a moment ago: sc1 loads
                    acc = mk(load_agent(&plane_at(a.planes.r, pixel4)), load_agent(&plane_at(a.planes.g, pixel4)), load_agent(&plane_at(a.planes.b, pixel4)));
                    pixel_seed.a = load_agent(&plane_at(a.planes.sa, pixel4)); pixel_seed.b = load_agent(&plane_at(a.planes.sb, pixel4));
                    pixel_seed.c = load_agent(&plane_at(a.planes.sc, pixel4)); pixel_seed.counter = load_agent(&plane_at(a.planes.sctr, pixel4));
                } else {
                    acc = mk(plane_at(a.planes.r, pixel4), plane_at(a.planes.g, pixel4), plane_at(a.planes.b, pixel4));
                    pixel_seed.a = plane_at(a.planes.sa, pixel4); pixel_seed.b = plane_at(a.planes.sb, pixel4);
                    pixel_seed.c = plane_at(a.planes.sc, pixel4); pixel_seed.counter = plane_at(a.planes.sctr, pixel4);
                }
                s = -1; busy = true; over = true; pending = false; has_ray = false;
            }
            cur.taken += take;
            if (cur.taken >= cur.len) next_chunk(cur, it);
#ifdef PTMI_TAIL_STATS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ph_taken += take; ph_refill += __builtin_readcyclecounter() - ph_t;
#endif
        }
        if (!__any(busy) && !chunks_left(cur)) break;      // (no lane busy, chunks left: nothing below has a lane to run for; the next trip refills)
#ifdef PTMI_TAIL_STATS
        lane_trips += (unsigned long long)__builtin_popcountll(__ballot(busy)); ++wave_trips;
#endif
        float4 mb = M[2 * idx + 1];
        V3 axis = mk(0.0f, 0.0f, 0.0f); float hk = 0.0f;
        if (pending) {
            // A ray whose throughput is already near zero dies at this hit (numNewRays, Trace.hs:329-331): the hit still
            // adds its emittance (computeResult runs for every intersection) and nothing else of it survives.
            if (near_zero(throughput)) {
                const float4 ma = M[2 * idx];
                acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
                hit_seed = seed;
                ++steps;
                pending = false; over = true;
            } else {
                bounce_axis(mb, normal, d, axis, hk);
            }
        }
#ifdef PTMI_TAIL_STATS
        const bool ph_ending = __any(over && s + 1 >= (int)f2u(get(10)));
        if (ph_ending) { ph_t = __builtin_readcyclecounter(); ++ph_ends; }
#endif
        if (over) {
            if (s >= 0) {                                      // a sample has been rendered
                if (a.seed_from_result && steps > 0u) {        // combine new old: the seed the sample's last hit carried
                    pixel_seed = hit_seed;
                }
                (void)random_float(pixel_seed);                // updateSeed
            }
            ++s; steps = 0;
            over = false;
            if (s < (int)f2u(get(10))) {                       // the pixel's next sample of this pass, from its cached start hit
                longest = longest > 1u ? longest : 1u;         // the primary ray's traceStep
                seed = pixel_seed;
                throughput = mk(1.0f, 1.0f, 1.0f);
                pos = mk(get(0), get(1), get(2)); idx = (int)f2u(get(7));
                mb = M[2 * idx + 1];
                axis = mk(get(3), get(4), get(5)); hk = get(6);
                pending = true;
            } else {                                           // the pixel is done: its seven words, once
                if (PASSES) {                                  // write-through: the pixel's next pass may run behind another L2
                    store_agent(&plane_at(a.planes.r, pixel4), acc.x); store_agent(&plane_at(a.planes.g, pixel4), acc.y); store_agent(&plane_at(a.planes.b, pixel4), acc.z);
                    store_agent(&plane_at(a.planes.sa, pixel4), pixel_seed.a); store_agent(&plane_at(a.planes.sb, pixel4), pixel_seed.b);
                    store_agent(&plane_at(a.planes.sc, pixel4), pixel_seed.c); store_agent(&plane_at(a.planes.sctr, pixel4), pixel_seed.counter);
                    unpublished = true;                        // published at the top of the next trip
                } else {
                    plane_at(a.planes.r, pixel4) = acc.x; plane_at(a.planes.g, pixel4) = acc.y; plane_at(a.planes.b, pixel4) = acc.z;
                    plane_at(a.planes.sa, pixel4) = pixel_seed.a; plane_at(a.planes.sb, pixel4) = pixel_seed.b;
                    plane_at(a.planes.sc, pixel4) = pixel_seed.c; plane_at(a.planes.sctr, pixel4) = pixel_seed.counter;
                }
                record_item_cost(a, f2u(get(8)), live - f2u(get(9)));       // its shaded hits stand for the loop trips it took
                busy = false;
            }
        }
#ifdef PTMI_TAIL_STATS
        if (ph_ending) ph_over += __builtin_readcyclecounter() - ph_t;
#endif
        if (pending) {                                         // alive (a fresh sample starts with throughput 1)
            const bool capped = steps + 1u >= step_cap;
            hit_seed = seed;
            // results: colour += emittance * throughput for EVERY hit; then the new ray (shade, with the axis in hand)
            V3 next; float brdf;
            next_about_axis(mb, axis, hk, seed, next, brdf);
            apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, acc);
            ++steps; ++live;                                   // the child exists even if the cap then cuts it
            pending = false;
            if (capped) {                                      // rare: only when the safety cap bites
                over = true;
                const unsigned long long cm = __ballot(1);
                if (lane == (int)__builtin_ctzll(cm)) atomicAdd(a.stream_counters + kScTruncated, (unsigned long long)__builtin_popcountll(cm));
            } else has_ray = true;
        }
        if (has_ray) {
            longest = steps + 1u > longest ? steps + 1u : longest;     // the traceStep this ray belongs to
            const HitSel h = check_hit(S, ns, np, pos, d);
            has_ray = false;
            if (h.just) {
                hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                idx = h.idx;
                pending = true;
            } else {
                over = true;
            }
        }
    }
    if (PASSES && __any(unpublished)) {                      // (nobody waits for the last pass; a wave that ends earlier owes its items)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (unpublished) atomicAdd(it.region_done + f2u(get(11)), 1u);
    }
#ifdef PTMI_TAIL_STATS
    if (lane == 0) {     // diagnostic build: [8] first start, [10] last end, [12] sum of ends, [14] waves, [16] lanes-with-item x trips, [18] trips (all u64, s_memtime ticks)
        unsigned long long *wc = reinterpret_cast<unsigned long long *>(a.work_counter + 8);
        const unsigned long long t_end = __builtin_readcyclecounter();
        atomicMax(wc + 0, ~t_start); atomicMax(wc + 1, t_end); atomicAdd(wc + 2, t_end - t_start); atomicAdd(wc + 3, 1ull);
        atomicAdd(wc + 4, lane_trips); atomicAdd(wc + 5, wave_trips);
        atomicMax(wc + 6, t_end - t_start);
#ifdef PTMI_TAIL_PHASES
        atomicAdd(wc + 8, ph_refill); atomicAdd(wc + 9, ph_over); atomicAdd(wc + 10, ph_refills); atomicAdd(wc + 11, ph_ends); atomicAdd(wc + 12, ph_taken);
#else
        const unsigned long long bin = (t_end - t_start) >> 19;               // [24, 64): waves by duration, bins of 2^19 cycles
        atomicAdd(a.work_counter + 24 + (bin < 39ull ? (unsigned int)bin : 39u), 1u);
#endif
    }
#endif
    // statistics: the per-pixel kernels' sharded counters
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(longest, off, 64); longest = other > longest ? other : longest; }
    const unsigned long long live_total = wave_sum(live);
    if (lane == 0) {
        if (a.live_counter && live_total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, live_total);
        if (a.stream_iterations && longest) atomicMax(a.stream_iterations + (size_t)(blockIdx.x & (kStatShards - 1)) * (2 * kStatStride), longest);
    }
}

// ---------------------------------------------------------------------------------------
// streams_split_kernel: items of (start hit, sample range), a child ring per wave.  Loop shape
// [finish dead hits][refill items][next ray: ring, else the wave's spill queue, else the item's next sample][shade][expand][trace].
// Where the children of a wave wait, in this order: the RING in LDS (kRing records: nearly all of them, for one or two
// trips); when the ring is full, the wave's own SPILL QUEUE in HBM (kSpill 64-byte records that only this wave writes and
// reads: same-wave program order, no cross-wave visibility question); when that is full too, the overflow stream that
// a later launch reads (streams_level_kernel) -- which a render call practically never needs.
// ---------------------------------------------------------------------------------------
#ifndef PTMI_SPLIT_WAVES
#define PTMI_SPLIT_WAVES 6
#endif
#ifndef PTMI_RING
#define PTMI_RING 16
#endif
constexpr unsigned int kRing = PTMI_RING;                     // records of a wave's child ring (a power of two, <= 64)
#ifndef PTMI_SPILL
#define PTMI_SPILL 4096
#endif
// records of a wave's spill queue in HBM (a power of two; 256 KB per wave, 1.6 GB for the 6 144 waves of a launch).  A wave that
// works through the inside of a glass sphere emits up to 64 children per trip and places 30: with 256 records 0.03 % of the glass
// scene's children went on to the overflow stream -- four more launches and read-backs per call, 8.76 ms; 1 024: 47 rays, 8.61;
// 4 096: none, 8.49.
constexpr unsigned int kSpill = PTMI_SPILL;
#ifndef PTMI_ITEM_BATCH
#define PTMI_ITEM_BATCH 1
#endif
constexpr unsigned int kItemBatch = PTMI_ITEM_BATCH;          // lanes without an item a wave waits for before it runs the refill block (its loads stall the whole wave)
template <bool LDS_SCENE, bool TILES>
__global__ void __launch_bounds__(kRenderBlock, PTMI_SPLIT_WAVES) streams_split_kernel(const RenderArgs a, const ItemArgs it)
{
    // the lane's own item: 0-2 position of the start hit, 3-5 normal, 6-8 incoming direction, 9-11 throughput,
    // 12 primitive | meta << 16, 13 pixel, 14-17 the seed the ray of its next sample carries
    __shared__ float item_rec[18][kRenderBlock];
    __shared__ uint32_t ring[15][kRing];                      // children waiting for a lane: RayQueue's record, word by word (15 words)
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
        stage_glass_constants(lds_scene, a.scene);
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    const unsigned int w = blockIdx.x;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    float *mine = &item_rec[0][threadIdx.x];
    auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
    auto get = [&](int k) { return mine[k * kRenderBlock]; };

    ChunkCursor cur; cur.home = xcc_id(); cur.tries = 0; cur.n_positions = it.n_positions;
    next_chunk(cur, it);
    unsigned int ring_head = 0, ring_n = 0;                   // wave-uniform
    unsigned int spill_head = 0, spill_n = 0;                 // wave-uniform: the wave's spill queue, records [w kSpill, (w + 1) kSpill) of it.spill
    unsigned int blk = w * kFirstBlock, blk_end = blk + kFirstBlock;   // wave-uniform: the overflow block being filled

    bool busy = false, foreign = false, has_ray = false, pending = false;   // own item in hand / the ray came from the ring / a ray to trace / a hit to shade
    V3 o = mk(0, 0, 0), d = o, throughput = o, normal = o;    // o: the ray's origin, or the position of the pending hit
    V3 own_acc = o;                                           // what the lane's own lineages have added for its item's pixel
    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    uint32_t pixel = 0, depth = 0;                            // depth: step index of the lane's current ray
    int idx = 0, samples_left = 0;
    unsigned int deepest = 0, item_trips = 0;                 // item_trips: loop trips since the lane took its item (its cost, for later launches' dispatch order)
    unsigned int live_w = 0, cut_w = 0, dropped_w = 0, stored_w = 0, spilled_w = 0;   // wave-uniform statistics

    // computeResult + permute (+) (Trace.hs:179-184, :318-323); adding an exact zero changes nothing
    auto add_colour = [&](V3 c) __attribute__((always_inline)) {
        if (foreign) {
            if (c.x != 0.0f) atomicAdd(&plane_at(a.planes.r, pixel << 2), c.x);
            if (c.y != 0.0f) atomicAdd(&plane_at(a.planes.g, pixel << 2), c.y);
            if (c.z != 0.0f) atomicAdd(&plane_at(a.planes.b, pixel << 2), c.z);
        } else {
            own_acc = own_acc + c;
        }
    };

#ifdef PTMI_SPLIT_STATS
    unsigned int st_trips = 0, st_dead = 0, st_free = 0, st_ring = 0, st_start = 0, st_end = 0, st_refill = 0, st_shade = 0, st_glass = 0, st_trace = 0, st_busy = 0;
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    for (;;) {
#ifdef PTMI_SPLIT_STATS
        ++st_trips;
        st_dead += (unsigned int)__builtin_popcountll(__ballot(pending && near_zero(throughput)));
        st_busy += (unsigned int)__builtin_popcountll(__ballot(busy));
#endif
        // ---- a hit whose ray arrived with near-zero throughput (numNewRays = 0, Trace.hs:329-331) adds its emittance and nothing
        // else of it survives: the lineage ends here
        if (pending && near_zero(throughput)) {
            const float4 ma = M[2 * idx];
            add_colour(scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
            pending = false;
        }
        // ---- refill: lanes without an item take the next ones of the wave's chunk (whatever ray they are tracing meanwhile)
        const unsigned long long empty = __ballot(!busy);
        if ((unsigned int)__builtin_popcountll(empty) >= kItemBatch && chunks_left(cur)) {     // wave-uniform
#ifdef PTMI_SPLIT_STATS
            ++st_refill;
#endif
            const unsigned int want = (unsigned int)__builtin_popcountll(empty), avail = cur.len - cur.taken;
            const unsigned int take = want < avail ? want : avail;
            const unsigned int rank = rank_in(empty);
            if (!busy && rank < take) {
                const float4 *r = it.hits.record(cur.first + cur.taken + rank);
                const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                const uint32_t px = f2u(r3.y);
                // initialState (Trace.hs:158-162) one step on: the cached start hit; the item's first sample starts from the
                // pixel's seed advanced by (pass * samples_per_pass) draws, which is what that many updateSeeds leave
                // ... and, for a child of a cached glass primary hit, by the 3 or 4 raw draws its ancestors made: the item keeps the
                // seed its next sample's RAY carries, which updateSeed moves on by one like the pixel's own
                const uint4 snap = it.seed_snapshots[(size_t)cur.pass * it.n_px + px];
                Sfc32 s0; s0.a = snap.x; s0.b = snap.y; s0.c = snap.z; s0.counter = snap.w;
                for (uint32_t q = 0; q < (f2u(r3.z) >> 8); ++q) (void)sfc32_next(s0);
                put(0, r0.x); put(1, r0.y); put(2, r0.z); put(3, r0.w); put(4, r1.x); put(5, r1.y);
                put(6, r1.z); put(7, r1.w); put(8, r2.x); put(9, r2.y); put(10, r2.z); put(11, r2.w);
                put(12, u2f(f2u(r3.x) | (f2u(r3.z) << 16))); put(13, r3.y);
                put(14, u2f(s0.a)); put(15, u2f(s0.b)); put(16, u2f(s0.c)); put(17, u2f(s0.counter));
                item_trips = 0;
                const int first_sample = (int)cur.pass * it.samples_per_pass;
                samples_left = a.n_spp - first_sample < it.samples_per_pass ? a.n_spp - first_sample : it.samples_per_pass;
                busy = true;
            }
            cur.taken += take;
            if (cur.taken >= cur.len) next_chunk(cur, it);
        }
        // ---- the next ray of every lane that holds neither a ray nor a hit: a child from the wave's ring first ...
        const bool free_lane = !pending && !has_ray;
        const unsigned long long free_m = __ballot(free_lane);
        bool took = false;
#ifdef PTMI_SPLIT_STATS
        st_free += (unsigned int)__builtin_popcountll(free_m);
        st_ring += ring_n < (unsigned int)__builtin_popcountll(free_m) ? ring_n : (unsigned int)__builtin_popcountll(free_m);
        st_start += (unsigned int)__builtin_popcountll(__ballot(free_lane && busy && samples_left > 0));
        st_end += (unsigned int)__builtin_popcountll(__ballot(free_lane && busy && samples_left <= 0));
#endif
        if (ring_n && free_m) {                               // wave-uniform
            const unsigned int want = (unsigned int)__builtin_popcountll(free_m);
            const unsigned int take = want < ring_n ? want : ring_n;
            const unsigned int rank = rank_in(free_m);
            if (free_lane && rank < take) {
                const unsigned int slot = (ring_head + rank) & (kRing - 1u);
                o = mk(u2f(ring[0][slot]), u2f(ring[1][slot]), u2f(ring[2][slot]));
                d = mk(u2f(ring[3][slot]), u2f(ring[4][slot]), u2f(ring[5][slot]));
                throughput = mk(u2f(ring[6][slot]), u2f(ring[7][slot]), u2f(ring[8][slot]));
                pixel = ring[9][slot];
                seed.a = ring[10][slot]; seed.b = ring[11][slot]; seed.c = ring[12][slot]; seed.counter = ring[13][slot];
                depth = ring[14][slot];
                has_ray = true; foreign = true; took = true;
            }
            ring_head = (ring_head + take) & (kRing - 1u); ring_n -= take;
        } else if (spill_n && free_m) {                       // ... or, the ring being empty, from the wave's spill queue (rare)
            const unsigned int want = (unsigned int)__builtin_popcountll(free_m);
            const unsigned int take = want < spill_n ? want : spill_n;
            const unsigned int rank = rank_in(free_m);
            // the records were written by this wave, at least a trip ago: once its stores have been acknowledged (they have: the
            // wait is free) they are in the L2, and loads that bypass the L1 see them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (free_lane && rank < take) {
                const float4 *r = it.spill.record(w * kSpill + ((spill_head + rank) & (kSpill - 1u)));
                const float4 r0 = load_past_l1(r), r1 = load_past_l1(r + 1), r2 = load_past_l1(r + 2), r3 = load_past_l1(r + 3);
                o = mk(r0.x, r0.y, r0.z);
                d = mk(r0.w, r1.x, r1.y);
                throughput = mk(r1.z, r1.w, r2.x);
                pixel = f2u(r2.y);
                seed.a = f2u(r2.z); seed.b = f2u(r2.w); seed.c = f2u(r3.x); seed.counter = f2u(r3.y);
                depth = f2u(r3.z);
                has_ray = true; foreign = true; took = true;
            }
            spill_head = (spill_head + take) & (kSpill - 1u); spill_n -= take;
        }
        // ---- ... else the next sample of its own item (its start hit and that sample's seed are in the lane's LDS column); an
        // item without samples left is over: its colour goes to the planes, one atomic per word
        if (free_lane && !took && busy) {
            if (samples_left > 0) {
                --samples_left;
                o = mk(get(0), get(1), get(2));
                normal = mk(get(3), get(4), get(5));
                d = mk(get(6), get(7), get(8));
                throughput = mk(get(9), get(10), get(11));
                const uint32_t pm = f2u(get(12));
                idx = (int)(pm & 0xffffu);
                pixel = f2u(get(13));
                Sfc32 ss; ss.a = f2u(get(14)); ss.b = f2u(get(15)); ss.c = f2u(get(16)); ss.counter = f2u(get(17));
                seed = ss;                                     // (already past the draws its ray's ancestors made)
                (void)sfc32_next(ss);                          // updateSeed (Trace.hs:190-191): the next sample starts one draw further
                put(14, u2f(ss.a)); put(15, u2f(ss.b)); put(16, u2f(ss.c)); put(17, u2f(ss.counter));
                depth = (pm >> 16) & 0xffu;
                deepest = deepest > 1u ? deepest : 1u;        // the primary ray's traceStep
                pending = true; foreign = false;              // (a start hit of a dead ray -- a reflection of weight ~0 -- waits for the next trip's first block)
            } else {
                const uint32_t px = f2u(get(13));
                if (own_acc.x != 0.0f) atomicAdd(&plane_at(a.planes.r, px << 2), own_acc.x);
                if (own_acc.y != 0.0f) atomicAdd(&plane_at(a.planes.g, px << 2), own_acc.y);
                if (own_acc.z != 0.0f) atomicAdd(&plane_at(a.planes.b, px << 2), own_acc.z);
                own_acc = mk(0.0f, 0.0f, 0.0f);
                if (TILES && a.quad_cost) {
                    const unsigned int y = px / (unsigned int)a.width, x = px - y * (unsigned int)a.width;
                    const unsigned int tile = (y >> 3) * (unsigned int)((a.width + 7) / 8) + (x >> 3);
                    record_item_cost(a, tile >> 2, item_trips);
                }
                busy = false;
            }
        }
        // no lane holds a ray or a hit: every lane was free, so ring and spill queue are empty (64 free lanes would have taken
        // from them) and no lane holds an item with samples left.  (With chunks left nothing below has a lane to run for and the
        // next trip refills: a `continue` here would be a second back edge, and cost the loop its register allocation.)
        if (!__any(has_ray || pending) && !chunks_left(cur) && !__any(busy) && ring_n == 0 && spill_n == 0) break;
        item_trips += busy ? 1u : 0u;

        // ---- shade round, for the hits of rays that are alive (a dead one just fetched waits for the next trip's first block)
        bool emits = false;
        V3 ko = o, kd = o, kt = o; Sfc32 ks = seed;
        const bool alive = pending && !near_zero(throughput);
        live_w += (unsigned int)__builtin_popcountll(__ballot(alive));        // one child per shaded hit ...
#ifdef PTMI_SPLIT_STATS
        st_shade += (unsigned int)__builtin_popcountll(__ballot(alive));
        st_glass += (unsigned int)__builtin_popcountll(__ballot(alive && f2u(M[2 * idx + 1].x) == 2u));
#endif
        if (alive) {
            const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
            V3 contribution;
            if (f2u(mb.x) == 2u) {                            // GLASS (extension): reflection stays, refraction is emitted
                contribution = scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput;
                V3 co[2], cd[2], ct[2]; Sfc32 cs[2];
                glass_children(mk(ma.x, ma.y, ma.z), glass_constants_of<LDS_SCENE>(mb), o, normal, d, throughput, seed, co, cd, ct, cs);
                o = co[0]; d = cd[0]; throughput = ct[0]; seed = cs[0];
                ko = co[1]; kd = cd[1]; kt = ct[1]; ks = cs[1];
                emits = true;
            } else {
                contribution = mk(0.0f, 0.0f, 0.0f);
                shade(M, idx, o, normal, o, d, throughput, contribution, seed);   // contribution = 0 + emittance * throughput
            }
            add_colour(contribution);
            ++depth; pending = false; has_ray = true;          // the child: next traceStep, same lane
        }
        // ---- expand: compaction of the emitted children into the wave's ring; what the ring cannot hold goes to the wave's spill
        // queue, what that cannot hold to the overflow stream
        const unsigned long long kids = it.may_emit ? __ballot(emits) : 0ull;
        if (kids) {                                           // wave-uniform
            const unsigned int cnt = (unsigned int)__builtin_popcountll(kids), rank = rank_in(kids);
            live_w += cnt;                                    // ... and a second one per GLASS hit
            const unsigned int room_ring = kRing - ring_n;
            const unsigned int to_ring = cnt < room_ring ? cnt : room_ring;
            if (emits && rank < to_ring) {
                const unsigned int slot = (ring_head + ring_n + rank) & (kRing - 1u);
                ring[0][slot] = f2u(ko.x); ring[1][slot] = f2u(ko.y); ring[2][slot] = f2u(ko.z);
                ring[3][slot] = f2u(kd.x); ring[4][slot] = f2u(kd.y); ring[5][slot] = f2u(kd.z);
                ring[6][slot] = f2u(kt.x); ring[7][slot] = f2u(kt.y); ring[8][slot] = f2u(kt.z);
                ring[9][slot] = pixel;
                ring[10][slot] = ks.a; ring[11][slot] = ks.b; ring[12][slot] = ks.c; ring[13][slot] = ks.counter;
                ring[14][slot] = depth;
            }
            ring_n += to_ring;
            if (cnt > to_ring) {                              // the ring is full (rare)
                const unsigned int rest = cnt - to_ring;
                const unsigned int room_spill = kSpill - spill_n;
                const unsigned int to_spill = rest < room_spill ? rest : room_spill;
                if (emits && rank >= to_ring && rank - to_ring < to_spill)
                    queue_store(it.spill, w * kSpill + ((spill_head + spill_n + (rank - to_ring)) & (kSpill - 1u)), ko, kd, kt, pixel, ks, depth);
                spill_n += to_spill; spilled_w += to_spill;
                if (rest > to_spill) {                        // the wave's spill queue is full too: the overflow stream (a later launch reads it)
                    const unsigned int cnt2 = rest - to_spill, first2 = to_ring + to_spill;
                    const bool spills = emits && rank >= first2;
                    const unsigned int rank2 = rank - first2;  // (meaningful where `spills`)
                    const unsigned int room = blk_end - blk;
                    unsigned int slot = blk + rank2;
                    if (cnt2 > room) {                        // the block is full: one atomic reserves the next for the whole wave
                        unsigned int fresh = 0;
                        if (lane == 0) fresh = it.out_base + atomicAdd(it.out_count, kNextBlock);
                        fresh = (unsigned int)__builtin_amdgcn_readfirstlane((int)fresh);
                        if (rank2 >= room) slot = fresh + (rank2 - room);
                        blk = fresh + (cnt2 - room); blk_end = fresh + kNextBlock;
                    } else {
                        blk += cnt2;
                    }
                    const unsigned int lost = (unsigned int)__builtin_popcountll(__ballot(spills && slot >= it.out.capacity));
                    stored_w += cnt2 - lost; dropped_w += lost;
                    if (spills && slot < it.out.capacity) queue_store(it.out, slot, ko, kd, kt, pixel, ks, depth);   // depth: the child's step index
                }
            }
        }
        // ---- trace round: one traceStep (Trace.hs:272-294) for every lane that holds a ray
        cut_w += (unsigned int)__builtin_popcountll(__ballot(has_ray && depth >= step_cap));
#ifdef PTMI_SPLIT_STATS
        st_trace += (unsigned int)__builtin_popcountll(__ballot(has_ray));
#endif
        if (has_ray) {
            if (depth >= step_cap) {                          // the safety cap (the reference has none): the ray exists, but is never traced
                has_ray = false;
            } else {
                deepest = depth + 1u > deepest ? depth + 1u : deepest;
                const HitSel h = check_hit(S, ns, np, o, d);
                has_ray = false;
                if (h.just) {
                    hit_record(S, ns, h.idx, o, d, h.t, o, normal);
                    idx = h.idx;
                    pending = true;
                }
            }
        }
    }
#ifdef PTMI_SPLIT_STATS
    if (lane == 0 && a.work_counter) {      // diagnostic build: per-round lane participation, summed over the waves; wave durations
        unsigned int *wc = a.work_counter;
        atomicAdd(wc + 1, st_trips); atomicAdd(wc + 2, st_dead); atomicAdd(wc + 3, st_free); atomicAdd(wc + 4, st_ring);
        atomicAdd(wc + 5, st_start); atomicAdd(wc + 6, st_end); atomicAdd(wc + 7, st_refill); atomicAdd(wc + 8, st_shade);
        atomicAdd(wc + 9, st_glass); atomicAdd(wc + 10, st_trace); atomicAdd(wc + 11, st_busy);
        const unsigned long long dur = __builtin_readcyclecounter() - t_start;
        atomicAdd(reinterpret_cast<unsigned long long *>(wc + 12), dur);
        atomicMax(reinterpret_cast<unsigned long long *>(wc + 14), dur);
        atomicAdd(wc + 16, 1u);
        const unsigned int x = cur.home;                              // per XCD: [24+x] waves, [32+x] sum of durations >> 12, [40+x] longest >> 12, [48+x] trips
        atomicAdd(wc + 24 + x, 1u); atomicAdd(wc + 32 + x, (unsigned int)(dur >> 12)); atomicMax(wc + 40 + x, (unsigned int)(dur >> 12));
        atomicAdd(wc + 48 + x, st_trips);
    }
#endif
    // what is left of this wave's last overflow block: holes
    if (it.may_emit) {
        const unsigned int end = blk_end < it.out.capacity ? blk_end : it.out.capacity;
        for (unsigned int i = blk + (unsigned int)lane; i < end; i += 64u) *it.out.pixel_word(i) = kHole;
    }
    // statistics: one set of atomics per wave, on counters sharded by workgroup
    unsigned int deep = deepest;
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(deep, off, 64); deep = other > deep ? other : deep; }
    if (lane == 0) {
        unsigned int *st = it.stats;
        if (live_w) atomicAdd(st + (kLvLive + (w & (unsigned int)(kLvLiveShards - 1))) * kCounterStride, live_w);
        if (stored_w) atomicAdd(it.emitted + (size_t)(w & (unsigned int)(kLvEmitShards - 1)) * kCounterStride, stored_w);
        // a maximum: most waves find it already there (a plain load first; the atomic only when it would raise the word)
        if (deep > __hip_atomic_load(st + kLvDeepest * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st + kLvDeepest * kCounterStride, deep);
        if (cut_w) atomicAdd(st + kLvCut * kCounterStride, cut_w);
        if (dropped_w) atomicAdd(st + kLvDropped * kCounterStride, dropped_w);
        if (spilled_w + stored_w) atomicAdd(st + kLvSpilled * kCounterStride, spilled_w + stored_w);
    }
}

// ---------------------------------------------------------------------------------------
// streams_level_kernel: one overflow level.  Input: the stream the previous level (or the split kernel) wrote -- ray
// states and holes; persistent waves take its 64-record chunks in a static stride; a lane follows its ray's lineage (at a
// GLASS hit the reflection stays, the refraction goes to the output stream); lanes whose lineage has ended refill from
// the wave's chunk in batches.  Colours through float atomics.
// ---------------------------------------------------------------------------------------
template <bool LDS_SCENE>
__global__ void __launch_bounds__(kRenderBlock, PTMI_LEVEL_WAVES) streams_level_kernel(const RenderArgs a, const LevelArgs lv)
{
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
        stage_glass_constants(lds_scene, a.scene);
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned int G = gridDim.x, w = blockIdx.x;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;

    unsigned int n_in = *lv.in_count + lv.in_base;            // the producer's cursor (real items and holes)
    n_in = n_in < lv.in.capacity ? n_in : lv.in.capacity;
    const unsigned int n_chunks = (n_in + 63u) / 64u;
    unsigned int chunk = w, taken = 0;                        // wave-uniform cursor: chunk index, items of it already handed out
    unsigned int blk = w * kFirstBlock, blk_end = blk + kFirstBlock;   // wave-uniform: the output block being filled
    unsigned int chunk_len = 0, chunk_first = 0;
    auto open = [&]() __attribute__((always_inline)) {
        if (chunk >= n_chunks) { chunk_len = 0; return; }
        chunk_first = chunk * 64u;
        chunk_len = n_in - chunk_first < 64u ? n_in - chunk_first : 64u;
    };
    open();

    bool has_ray = false, pending = false;                    // a ray to trace / a hit to shade
    V3 o = mk(0, 0, 0), d = o, throughput = o, normal = o;
    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    uint32_t pixel = 0, depth = 0;
    int idx = 0;
    unsigned int deepest = 0;
    unsigned int live_w = 0, cut_w = 0, dropped_w = 0, stored_w = 0;

    auto add_colour = [&](V3 c) __attribute__((always_inline)) {
        if (c.x != 0.0f) atomicAdd(a.planes.r + pixel, c.x);
        if (c.y != 0.0f) atomicAdd(a.planes.g + pixel, c.y);
        if (c.z != 0.0f) atomicAdd(a.planes.b + pixel, c.z);
    };

    for (;;) {
        if (pending && near_zero(throughput)) {
            const float4 ma = M[2 * idx];
            add_colour(scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
            pending = false;
        }
        const unsigned long long idle = __ballot(!has_ray && !pending);
        if (chunk < n_chunks && ((unsigned int)__builtin_popcountll(idle) >= kRefillBatch || (idle && !~idle))) {   // wave-uniform
            const unsigned int want = (unsigned int)__builtin_popcountll(idle), avail = chunk_len - taken;
            const unsigned int take = want < avail ? want : avail;
            const unsigned int rank = (unsigned int)__builtin_popcountll(idle & below);
            if (!has_ray && !pending && rank < take) {
                const float4 *r = lv.in.record(chunk_first + taken + rank);
                const float4 r2 = r[2];
                pixel = f2u(r2.y);
                if (pixel != kHole) {
                    const float4 r0 = r[0], r1 = r[1], r3 = r[3];
                    o = mk(r0.x, r0.y, r0.z);
                    d = mk(r0.w, r1.x, r1.y);
                    throughput = mk(r1.z, r1.w, r2.x);
                    seed.a = f2u(r2.z); seed.b = f2u(r2.w); seed.c = f2u(r3.x); seed.counter = f2u(r3.y);
                    depth = f2u(r3.z); has_ray = true;
                }
            }
            taken += take;
            if (taken >= chunk_len) { chunk += G; taken = 0; open(); }
        }
        if (!__any(has_ray || pending) && chunk >= n_chunks) break;     // (a chunk of holes: nothing below runs, the next trip looks at the next one)
        bool emits = false;
        V3 ko = o, kd = o, kt = o; Sfc32 ks = seed;
        const bool alive = pending && !near_zero(throughput);
        live_w += (unsigned int)__builtin_popcountll(__ballot(alive));
        if (alive) {
            const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
            V3 contribution;
            if (f2u(mb.x) == 2u) {
                contribution = scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput;
                V3 co[2], cd[2], ct[2]; Sfc32 cs[2];
                glass_children(mk(ma.x, ma.y, ma.z), glass_constants_of<LDS_SCENE>(mb), o, normal, d, throughput, seed, co, cd, ct, cs);
                o = co[0]; d = cd[0]; throughput = ct[0]; seed = cs[0];
                ko = co[1]; kd = cd[1]; kt = ct[1]; ks = cs[1];
                emits = true;
            } else {
                contribution = mk(0.0f, 0.0f, 0.0f);
                shade(M, idx, o, normal, o, d, throughput, contribution, seed);
            }
            add_colour(contribution);
            ++depth; pending = false; has_ray = true;
        }
        const unsigned long long kids = lv.may_emit ? __ballot(emits) : 0ull;
        if (kids) {
            const unsigned int cnt = (unsigned int)__builtin_popcountll(kids), rank = (unsigned int)__builtin_popcountll(kids & below);
            live_w += cnt;
            const unsigned int room = blk_end - blk;
            unsigned int slot = blk + rank;
            if (cnt > room) {
                unsigned int fresh = 0;
                if (lane == 0) fresh = lv.out_base + atomicAdd(lv.out_count, kNextBlock);
                fresh = (unsigned int)__builtin_amdgcn_readfirstlane((int)fresh);
                if (rank >= room) slot = fresh + (rank - room);
                blk = fresh + (cnt - room); blk_end = fresh + kNextBlock;
            } else {
                blk += cnt;
            }
            const unsigned int lost = (unsigned int)__builtin_popcountll(__ballot(emits && slot >= lv.out.capacity));
            stored_w += cnt - lost; dropped_w += lost;
            if (emits && slot < lv.out.capacity) queue_store(lv.out, slot, ko, kd, kt, pixel, ks, depth);
        }
        cut_w += (unsigned int)__builtin_popcountll(__ballot(has_ray && depth >= step_cap));
        if (has_ray) {
            if (depth >= step_cap) {
                has_ray = false;
            } else {
                deepest = depth + 1u > deepest ? depth + 1u : deepest;
                const HitSel h = check_hit(S, ns, np, o, d);
                has_ray = false;
                if (h.just) {
                    hit_record(S, ns, h.idx, o, d, h.t, o, normal);
                    idx = h.idx;
                    pending = true;
                }
            }
        }
    }
    if (lv.may_emit) {
        const unsigned int end = blk_end < lv.out.capacity ? blk_end : lv.out.capacity;
        for (unsigned int i = blk + (unsigned int)lane; i < end; i += 64u) *lv.out.pixel_word(i) = kHole;
    }
    unsigned int deep = deepest;
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(deep, off, 64); deep = other > deep ? other : deep; }
    if (lane == 0) {
        unsigned int *st = lv.stats;
        if (live_w) atomicAdd(st + (kLvLive + (w & (unsigned int)(kLvLiveShards - 1))) * kCounterStride, live_w);
        if (stored_w) atomicAdd(lv.emitted + (size_t)(w & (unsigned int)(kLvEmitShards - 1)) * kCounterStride, stored_w);
        if (deep > __hip_atomic_load(st + kLvDeepest * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st + kLvDeepest * kCounterStride, deep);
        if (cut_w) atomicAdd(st + kLvCut * kCounterStride, cut_w);
        if (dropped_w) atomicAdd(st + kLvDropped * kCounterStride, dropped_w);
    }
}

// updateSeed (Trace.hs:190-191) for every sample of the call: `draws` draws per pixel, and on the way the seed each of the
// `passes` items of the pixel starts from (snapshots[pass][pixel]: the pixel's seed after pass * samples_per_pass draws).
__global__ void __launch_bounds__(kBlock) streams_seeds_kernel(Planes p, uint4 *snapshots, long long n, int passes, int samples_per_pass, int draws)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Sfc32 s; s.a = p.sa[i]; s.b = p.sb[i]; s.c = p.sc[i]; s.counter = p.sctr[i];
    int done = 0;
    for (int k = 0; k < passes; ++k) {
        snapshots[(size_t)k * (size_t)n + (size_t)i] = uint4{s.a, s.b, s.c, s.counter};
        const int upto = (k + 1) * samples_per_pass < draws ? (k + 1) * samples_per_pass : draws;
        for (; done < upto; ++done) (void)random_float(s);
    }
    for (; done < draws; ++done) (void)random_float(s);
    p.sa[i] = s.a; p.sb[i] = s.b; p.sc[i] = s.c; p.sctr[i] = s.counter;
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

#endif  // PTMI_CONTRACTED_BUILD

inline unsigned int blocks_for(long long n, int block = kBlock) { return (unsigned int)((n + block - 1) / block); }

}  // namespace

// 8x8 tiles leave lanes idle on the right and bottom edges; rows of 64 leave them idle at the end only
static bool tiles_pay_dims(int width, int rows_local) { return width >= 64 && rows_local >= 16; }
bool tiles_pay(const RenderArgs &a) { return tiles_pay_dims(a.width, a.rows_local); }

unsigned int quad_positions(int width, int rows_local)
{
    if (!tiles_pay_dims(width, rows_local)) return 0;
    const unsigned int tiles = (unsigned int)(((width + 7) / 8) * ((rows_local + 7) / 8));
    return ((tiles + 31u) & ~31u) / 4u;
}

// the default (variant 0 = auto) takes the cost order; the explicit variants, 13 and 17 included, keep the image order
bool uses_quad_order(const RenderArgs &a, int algorithm_inline, int variant)
{
    if (variant != 0 || !tiles_pay(a) || a.screen_x) return false;
    return algorithm_inline ? (a.bounce_limit > 0 && a.n_spp > 0) : true;
}

#ifndef PTMI_CONTRACTED_BUILD
namespace {
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

hipError_t launch_quad_order(const unsigned int *cost, unsigned int *order, unsigned int *cls, unsigned int n, unsigned int *tail_start,
                             unsigned int tail_permille, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(quad_order_kernel, dim3(1), dim3(1024), 0, stream, cost, order, cls, n, tail_start, tail_permille);
    return hipGetLastError();
}

#endif  // PTMI_CONTRACTED_BUILD

// Sample chunks (render_inline_kernel): only when the launch has few rounds of waves and every copy keeps >= 64 samples
// (every copy re-evaluates the primary hit and moves the planes once more).  Sets b.spp_chunks (>= 1) and clears the flags.
static hipError_t choose_sample_chunks(RenderArgs &b, unsigned int per_copy, int waves_per_simd, hipStream_t stream)
{
    const int wanted = b.spp_chunks;                           // 0 = automatic, 1 = off, k = forced
    b.spp_chunks = 1;
    if (!b.chunk_done || b.chunk_capacity < per_copy || wanted == 1 || b.screen_x) return hipSuccess;
    const int cus = b.cus > 0 ? b.cus : 256;                   // of the context's device (ptmi_create)
    const unsigned long long slots = (unsigned long long)cus * 4ull * (unsigned long long)waves_per_simd;
    int k = wanted > 1 ? wanted : (int)((16ull * slots + per_copy - 1) / per_copy);    // aim at >= 16 rounds of waves
    if (wanted <= 0 && k > b.n_spp / 64) k = b.n_spp / 64;
    if (k > b.n_spp) k = b.n_spp;
    if (k > 64) k = 64;
    if (k < 2) return hipSuccess;
    b.spp_chunks = k;
    if (hipError_t e = hipMemsetAsync(b.chunk_done, 0, (size_t)per_copy * sizeof(unsigned int), stream)) return e;
    return hipMemsetAsync(b.chunk_done + b.chunk_capacity, 0, sizeof(unsigned int), stream);      // the ticket counter
}

// Which render Inline kernel a variant is (ptmi_set_variant):
//   0 auto | 4 cached, a wave = 64 consecutive pixels of a row (LDS scene) | 5 the same with the scene through scalar loads
//   13 = 4 with 8x8 pixel tiles per wave | 17 = 5 with 8x8 tiles -- these are what auto chooses from.
// Only in builds with -DPTMI_ABLATIONS (ptmi_set_variant refuses them otherwise):
//   1 / 6 persistent hand-out (LDS / scalar-load scene) | 2 lock step | 3 regenerate | 7 / 8 capped occupancy
//   10-12 pooled second shade round | 14-16 other tile shapes | 18 round 1's loop
hipError_t launch_render_inline(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const bool big_scene = lds > kMaxSceneLds;               // every route reads such a scene through scalar loads, not LDS
    const bool degenerate = a.bounce_limit <= 0 || a.n_spp <= 0;   // the cached kernel handles both (iterate 0; no sample at all)
    if (variant == 0 || degenerate) {
        // static mapping wins at every size measured; a scene so big that staging it per wave would cost more occupancy than scalar loads cost speed is
        // read through scalar loads; 8x8 tiles once the image is big enough for whole tiles to dominate
        const bool tiles = tiles_pay(a);
        variant = !big_scene ? (tiles ? 13 : 4) : (tiles ? 17 : 5);
    }
    if (big_scene) {                                         // the LDS forms would not fit or would cap occupancy
        if (variant == 1) variant = 6;
        else if (variant == 4 || variant == 7 || variant == 8) variant = 5;
        else if (variant >= 13 && variant <= 16) variant = 17;
    }
    if (variant == 17 || variant == 13) {
        RenderArgs b = a;
        const unsigned int per_copy = tile_grid(a, 8);
        if (hipError_t e = choose_sample_chunks(b, per_copy, PTMI_INLINE_WAVES, stream)) return e;
        const dim3 cgrid(per_copy * (unsigned int)b.spp_chunks);
        if (variant == 17) hipLaunchKernelGGL((render_inline_kernel<false, kCached, 8>), cgrid, block, 0, stream, b);
        else               hipLaunchKernelGGL((render_inline_kernel<true, kCached, 8>), cgrid, block, lds, stream, b);
        return hipGetLastError();
    }
    if (variant == 5) { hipLaunchKernelGGL((render_inline_kernel<false, kCached>), grid, block, 0, stream, a); return hipGetLastError(); }
    if (variant == 4) { hipLaunchKernelGGL((render_inline_kernel<true, kCached>), grid, block, lds, stream, a); return hipGetLastError(); }
#ifdef PTMI_ABLATIONS
    if (variant == 1 || variant == 6) {
        // persistent grid; more workgroups than fit would only start late and find the queue empty: cap at 8 per CU
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        const int max_blocks = (cus > 0 ? cus : 256) * 8 * (256 / kRenderBlock);
        const unsigned int blocks = grid.x < (unsigned int)max_blocks ? grid.x : (unsigned int)max_blocks;
        e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
        if (variant == 1) hipLaunchKernelGGL((render_inline_persistent_kernel<true>), dim3(blocks), block, lds, stream, a);
        else              hipLaunchKernelGGL((render_inline_persistent_kernel<false>), dim3(blocks), block, 0, stream, a);
        return hipGetLastError();
    }
    if (variant == 18) {                                      // round 1's loop (no frozen-shade shortcut), 8x8 tiles, LDS scene
        hipLaunchKernelGGL((render_inline_kernel<true, kCachedR1, 8>), dim3(tile_grid(a, 8)), block, lds, stream, a);
        return hipGetLastError();
    }
    if (variant >= 14 && variant <= 16) {                     // other pixel tiles per wave: 16x4 / 4x16 / 32x2 (8x8 is handled above)
        const int tw = variant == 14 ? 16 : variant == 15 ? 4 : 32;
        const dim3 tgrid(tile_grid(a, tw));
        if (tw == 16)      hipLaunchKernelGGL((render_inline_kernel<true, kCached, 16>), tgrid, block, lds, stream, a);
        else if (tw == 4)  hipLaunchKernelGGL((render_inline_kernel<true, kCached, 4>), tgrid, block, lds, stream, a);
        else               hipLaunchKernelGGL((render_inline_kernel<true, kCached, 32>), tgrid, block, lds, stream, a);
        return hipGetLastError();
    }
    if (variant >= 10 && variant <= 12) {                    // pooled second shade round, W = 2 / 4 / 8 waves per workgroup
        const int w = variant == 10 ? 2 : variant == 11 ? 4 : 8;
        const dim3 pgrid(blocks_for(n_local, 64 * w)), pblock(64 * w);
        if (big_scene) {                                       // a scene too big to stage per workgroup: scalar loads
            if (w == 2)      hipLaunchKernelGGL((render_inline_pooled_kernel<false, 2>), pgrid, pblock, 0, stream, a);
            else if (w == 4) hipLaunchKernelGGL((render_inline_pooled_kernel<false, 4>), pgrid, pblock, 0, stream, a);
            else             hipLaunchKernelGGL((render_inline_pooled_kernel<false, 8>), pgrid, pblock, 0, stream, a);
        } else {
            if (w == 2)      hipLaunchKernelGGL((render_inline_pooled_kernel<true, 2>), pgrid, pblock, lds, stream, a);
            else if (w == 4) hipLaunchKernelGGL((render_inline_pooled_kernel<true, 4>), pgrid, pblock, lds, stream, a);
            else             hipLaunchKernelGGL((render_inline_pooled_kernel<true, 8>), pgrid, pblock, lds, stream, a);
        }
        return hipGetLastError();
    }
    switch (variant) {
    case 2:  if (big_scene) hipLaunchKernelGGL((render_inline_kernel<false, kLockstep>), grid, block, 0, stream, a);
             else           hipLaunchKernelGGL((render_inline_kernel<true, kLockstep>), grid, block, lds, stream, a);
             return hipGetLastError();
    case 3:  if (big_scene) hipLaunchKernelGGL((render_inline_kernel<false, kRegenerate>), grid, block, 0, stream, a);
             else           hipLaunchKernelGGL((render_inline_kernel<true, kRegenerate>), grid, block, lds, stream, a);
             return hipGetLastError();
    case 7:  hipLaunchKernelGGL((render_inline_kernel<true, kCached>), grid, block, 33 * 1024, stream, a); return hipGetLastError();   // 4 waves/SIMD
    case 8:  hipLaunchKernelGGL((render_inline_kernel<true, kCached>), grid, block, 41 * 1024, stream, a); return hipGetLastError();   // 3 waves/SIMD
    default: break;
    }
#endif
    return hipErrorInvalidValue;                             // ptmi_set_variant admits only what the build holds
}

#ifndef PTMI_CONTRACTED_BUILD
bool variant_available(int variant)
{
    if (variant == 0 || variant == 4 || variant == 5 || variant == 9 || variant == 13 || variant == 17) return true;
#ifdef PTMI_ABLATIONS
    return variant >= 0 && variant <= 18;
#else
    return false;
#endif
}

hipError_t launch_render_streams(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    hipError_t e = hipMemsetAsync(a.stream_iterations, 0, (size_t)kStatShards * 2 * kStatStride * sizeof(unsigned int), stream);   // every shard: the figure is per launch
    if (e != hipSuccess) return e;
    const bool scalar_scene = variant == 5 || variant == 6 || variant == 17 || lds > kMaxSceneLds;
    const bool tiles = variant == 4 || variant == 5 ? false : tiles_pay(a);      // 4 / 5 keep the row mapping (ablation)
    if (tiles) {
        RenderArgs b = a;
        const unsigned int per_copy = tile_grid(a, 8);
        if (hipError_t ce = choose_sample_chunks(b, per_copy, PTMI_STREAMS_WAVES, stream)) return ce;
        const dim3 tgrid(per_copy * (unsigned int)b.spp_chunks);
        if (scalar_scene) hipLaunchKernelGGL((render_streams_kernel<false, 8>), tgrid, block, 0, stream, b);
        else              hipLaunchKernelGGL((render_streams_kernel<true, 8>), tgrid, block, lds, stream, b);
    } else {
        if (scalar_scene) hipLaunchKernelGGL((render_streams_kernel<false>), grid, block, 0, stream, a);
        else              hipLaunchKernelGGL((render_streams_kernel<true>), grid, block, lds, stream, a);
    }
    return hipGetLastError();
}

// The per-pixel chain kernel as the TAIL of a stream-form launch: a grid over every dispatch position whose workgroups start at
// *first_position (RenderArgs.first_position); no sample chunks; stream_iterations is the stream form's to clear.
hipError_t launch_render_streams_tail(const RenderArgs &a, const unsigned int *first_position, hipStream_t stream)
{
    if (!tiles_pay(a) || !first_position) return hipSuccess;
    RenderArgs b = a;
    b.spp_chunks = 1; b.first_position = first_position;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 grid(tile_grid(a, 8)), block(kRenderBlock);
    if (lds > kMaxSceneLds) hipLaunchKernelGGL((render_streams_kernel<false, 8>), grid, block, 0, stream, b);
    else                    hipLaunchKernelGGL((render_streams_kernel<true, 8>), grid, block, lds, stream, b);
    return hipGetLastError();
}

// workgroups (per copy of the grid) of the tree walk = records' worth of RenderArgs.tree_stack: x kTreeFastLevels x 64 lanes x 64 B
unsigned int tree_workgroups(int width, int rows_local)
{
    if (tiles_pay_dims(width, rows_local)) return quad_positions(width, rows_local) * 4u;
    return (unsigned int)(((long long)width * rows_local + kRenderBlock - 1) / kRenderBlock);
}

hipError_t launch_render_streams_tree(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    hipError_t e = hipMemsetAsync(a.stream_iterations, 0, (size_t)kStatShards * 2 * kStatStride * sizeof(unsigned int), stream);   // every shard: the figure is per launch
    if (e != hipSuccess) return e;
    const bool scalar_scene = variant == 5 || variant == 17 || lds > kMaxSceneLds;
    const bool tiles = variant == 4 || variant == 5 ? false : tiles_pay(a);
    if (tiles) {
        RenderArgs b = a;
        const unsigned int per_copy = tile_grid(a, 8);
        if (hipError_t ce = choose_sample_chunks(b, per_copy, PTMI_TREE_WAVES, stream)) return ce;
        const dim3 tgrid(per_copy * (unsigned int)b.spp_chunks);
        if (scalar_scene) hipLaunchKernelGGL((render_streams_tree_kernel<false, 8>), tgrid, block, 0, stream, b);
        else              hipLaunchKernelGGL((render_streams_tree_kernel<true, 8>), tgrid, block, lds, stream, b);
    } else {
        if (scalar_scene) hipLaunchKernelGGL((render_streams_tree_kernel<false>), grid, block, 0, stream, a);
        else              hipLaunchKernelGGL((render_streams_tree_kernel<true>), grid, block, lds, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_streams_level(const RenderArgs &a, const LevelArgs &lv, unsigned int grid, hipStream_t stream)
{
    if (grid == 0) return hipSuccess;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 g(grid), b(kRenderBlock);
    if (lds > kMaxSceneLds) hipLaunchKernelGGL((streams_level_kernel<false>), g, b, 0, stream, a, lv);      // a scene too big for LDS at this occupancy: scalar loads
    else                    hipLaunchKernelGGL((streams_level_kernel<true>), g, b, lds, stream, a, lv);
    return hipGetLastError();
}

hipError_t launch_streams_pixels(const RenderArgs &a, const ItemArgs &it, unsigned int grid, hipStream_t stream)
{
    if (grid == 0) return hipSuccess;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 g(grid), b(kRenderBlock);
    if (it.passes > 1) {
        if (lds > kMaxSceneLds) hipLaunchKernelGGL((streams_pixels_kernel<false, true>), g, b, 0, stream, a, it);
        else                    hipLaunchKernelGGL((streams_pixels_kernel<true, true>), g, b, lds, stream, a, it);
    } else {
        if (lds > kMaxSceneLds) hipLaunchKernelGGL((streams_pixels_kernel<false, false>), g, b, 0, stream, a, it);
        else                    hipLaunchKernelGGL((streams_pixels_kernel<true, false>), g, b, lds, stream, a, it);
    }
    return hipGetLastError();
}

hipError_t launch_streams_split(const RenderArgs &a, const ItemArgs &it, unsigned int grid, hipStream_t stream)
{
    if (grid == 0) return hipSuccess;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 g(grid), b(kRenderBlock);
    const bool tiles = tiles_pay(a);
    if (lds > kMaxSceneLds) {
        if (tiles) hipLaunchKernelGGL((streams_split_kernel<false, true>), g, b, 0, stream, a, it);
        else       hipLaunchKernelGGL((streams_split_kernel<false, false>), g, b, 0, stream, a, it);
    } else {
        if (tiles) hipLaunchKernelGGL((streams_split_kernel<true, true>), g, b, lds, stream, a, it);
        else       hipLaunchKernelGGL((streams_split_kernel<true, false>), g, b, lds, stream, a, it);
    }
    return hipGetLastError();
}

int streams_pixels_waves() { return PTMI_PIXELS_WAVES; }
int streams_min_pass_samples() { return kMinPassSamples; }
int streams_split_waves() { return PTMI_SPLIT_WAVES; }
unsigned int streams_spill_records() { return kSpill; }
unsigned int streams_first_block() { return kFirstBlock; }

// regions of the start-hit list: one per 8x8 tile (padded as the tiled render kernels pad their grids), or one per 64
// consecutive pixels for images too small for tiles, a multiple of four either way (a workgroup of the primary kernel makes four)
unsigned int streams_regions(int width, int rows_local)
{
    if (tiles_pay_dims(width, rows_local)) return quad_positions(width, rows_local) * 4u;
    const unsigned long long n = (unsigned long long)width * (unsigned long long)rows_local;
    return (unsigned int)((((n + 63ull) / 64ull) + 3ull) & ~3ull);
}

hipError_t launch_streams_primary(const RenderArgs &a, HitList hits, unsigned int *counters, hipStream_t stream)
{
    if (hits.n_regions == 0) return hipSuccess;
    const dim3 g(hits.n_regions / 4u), b(256);
    if (tiles_pay(a)) hipLaunchKernelGGL((streams_primary_kernel<true>), g, b, 0, stream, a, hits, counters);
    else              hipLaunchKernelGGL((streams_primary_kernel<false>), g, b, 0, stream, a, hits, counters);
    return hipGetLastError();
}

hipError_t launch_streams_advance_missed(const RenderArgs &a, HitList hits, int draws, const unsigned int *tail_start, hipStream_t stream)
{
    if (hits.n_regions == 0 || draws <= 0) return hipSuccess;
    const dim3 g(hits.n_regions / 4u), b(256);
    if (tiles_pay(a)) hipLaunchKernelGGL((streams_advance_missed_kernel<true>), g, b, 0, stream, a, hits, draws, tail_start);
    else              hipLaunchKernelGGL((streams_advance_missed_kernel<false>), g, b, 0, stream, a, hits, draws, tail_start);
    return hipGetLastError();
}

hipError_t launch_streams_seeds(Planes p, uint4 *snapshots, long long n, int passes, int samples_per_pass, int draws, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(streams_seeds_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, p, snapshots, n, passes, samples_per_pass, draws);
    return hipGetLastError();
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

hipError_t launch_present(Planes p, long long n, int iterations, float *rgb, uint32_t *rgba, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const uintptr_t bits = (uintptr_t)p.r | (uintptr_t)p.g | (uintptr_t)p.b | (uintptr_t)rgb | (uintptr_t)rgba;
    hipLaunchKernelGGL(present_kernel, dim3(blocks_for((n + 3) / 4)), dim3(kBlock), 0, stream, p, n, (float)iterations, rgb, rgba,
                       (bits & 15u) == 0 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_stitch(const float *src, int rows, int width, int stripe_rows, int n_parts, int part,
                         float *r, float *g, float *b, hipStream_t stream)
{
    const long long n = (long long)rows * width;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(stitch_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, src, rows, width, stripe_rows, n_parts, part, r, g, b);
    return hipGetLastError();
}

hipError_t launch_eval_sincos(const float *x, int n, float *s, float *c, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_sincos_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, x, n, s, c);
    return hipGetLastError();
}

#endif  // PTMI_CONTRACTED_BUILD

}  // namespace ptmi

#ifdef PTMI_CONTRACTED_BUILD
// THE CONTRACTED-ARITHMETIC OBJECT.  This file is compiled a second time with -ffp-contract=fast and -Dptmi=ptmi_contracted
// (every name above then lives in namespace ptmi_contracted, render Inline only): the same kernel with a * b + c contracted
// into fused multiply-adds wherever the source writes it -- dot products, cross products, the rotation, the quaternion.  It is
// NOT the reference's arithmetic as this repository reads it (every operation rounded on its own); it
// exists to MEASURE how much of the kernel's time that reading costs (PTMI_OPT_ARITHMETIC, never the default, never the
// headline).  One C entry, because the two objects' RenderArgs are distinct types of identical layout.
extern "C" int ptmi_contracted_launch_inline(const void *args, int variant, void *stream)
{
    return (int)ptmi::launch_render_inline(*static_cast<const ptmi::RenderArgs *>(args), variant, static_cast<hipStream_t>(stream));
}
#endif
