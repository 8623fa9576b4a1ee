// ptmi_device.h -- what every render kernel of libptmi shares: the device side of checkHit / hit / calcNextRay
// (src/Scene/Trace.hs:344-456, src/Scene/Intersection.hs:16-64, src/Util.hs:114-118, :156-178), the pixel -> wave mapping,
// the sample chunks, the GLASS extension's children.  Included by one translation unit per kernel family:
//   ptmi_inline.hip          render Inline (Trace.hs:193-200)             [+ ptmi_inline_ablations.hip with -DPTMI_ABLATIONS]
//   ptmi_streams_chain.hip   render Streams, one chain per pixel (Trace.hs:141-191)
//   ptmi_streams_tree.hip    render Streams with ray splitting, one tree per pixel
//   ptmi_stream_primary.hip / ptmi_stream_pixels.hip / ptmi_stream_split.hip   the stream ("wavefront") form (ptmi_stream_form.h)
//   ptmi_small.hip           seeds, createWith, present, stitch, dispatch order, point queries
// Everything here is in an unnamed namespace: each unit gets its own copy, nothing is linked across units (no -fgpu-rdc).
// Diagnostic builds (-DPTMI_*_STATS) live in ptmi_diag.h; the kernels only call its probes, which are empty otherwise.
#pragma once

#include "ptmi_kernels.h"
#include "ptmi_diag.h"

namespace ptmi {

namespace {

constexpr int kBlock = 256;      // small streaming kernels
constexpr size_t kMaxSceneLds = 3 * 1024;  // bytes of staged scene per one-wave workgroup before LDS would cap occupancy (~60 primitives)
constexpr int kRenderBlock = 64; // render kernels: one wave per workgroup, so a finished wave's slot is refilled at once (+1 % on C2)

struct HitSel { float t; int idx; bool just; };
#ifdef PTMI_NO_STAGED_WALK_OTHERS          // (A/B builds: the Streams kernels read the staged scene as S[i], as they did until round 6)
constexpr bool kStagedWalk = false;
#else
constexpr bool kStagedWalk = true;
#endif

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
//   * STAGED = true (the caller has staged the scene in LDS and S points there): the geometry is read off ONE vector register
//     that holds its LDS address, four spheres per trip, the offsets as immediates.  The address of a wave-uniform LDS read is a
//     scalar, but ds_read wants it in a VGPR: read as S[i] every ds_read_b128 had a `v_mov_b32 vN, sK` in front of it -- one
//     VALU issue slot in 22 of the sphere test, 56 M of them per C2 launch (round 6, tools/isa_other.py: C2 2.93 -> 2.86 ms).
template <bool STAGED = false, typename ScenePtr>
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
        diag::sphere_test(diag, cand);
        if (__any(cand)) {
            const float t = tca - sqrt_rn(x);                // min t0 t1 == t0 (thc >= 0 or NaN)
            const bool just = cand && !(t < 0.0f);
            const float key = just ? t : kInfinite;          // maybe infinite fst
            // (as a masked block -- s_and_saveexec, two v_mov, s_or exec -- not as three selects: the selects cost C2 + 1.6 %, round 6)
            if (!(best_key <= key)) { best_key = key; best_idx = i; best_just = just; }
        }
    };

    float4 ga, gb;
    int i = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const float4 *Staged;
    Staged P = nullptr;
    if constexpr (STAGED) {
        unsigned int lds_at = (unsigned int)(uintptr_t)&S[0];    // the low half of a generic LDS address is the LDS offset
        asm volatile("" : "+v"(lds_at));                          // ... kept in a VGPR: the loads below take it with immediate offsets
        P = (Staged)(uintptr_t)lds_at;
        ga = P[0];
        for (; i + 3 < ns; i += 4) {
            gb = P[1]; sphere(ga, i);
            ga = P[2]; sphere(gb, i + 1);
            gb = P[3]; sphere(ga, i + 2);
            ga = P[4]; sphere(gb, i + 3);                         // S has readable elements past the geometry
            P += 4;
        }
        for (; i + 1 < ns; i += 2) {
            gb = P[1]; sphere(ga, i);
            ga = P[2]; sphere(gb, i + 1);
            P += 2;
        }
        if (i < ns) {
            gb = P[1];
            sphere(ga, i);
            ga = gb;
            P += 1;
        }
    } else
#endif
    {
        ga = S[0];
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
    }
    for (int j = 0; j < np; ++j) {
        // distanceTo @Plane (Intersection.hs:57-62); ga holds (px, py, pz, 0)
        float4 gn, g_next;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (STAGED) { gn = P[2 * j + 1]; g_next = P[2 * j + 2]; } else
#endif
        { gn = S[ns + 2 * j + 1]; g_next = S[ns + 2 * j + 2]; }
        const V3 nor = mk(gn.x, gn.y, gn.z);
        const float denom = dot(d, nor);
        const bool cand = !(denom > 1e-6f);
        diag::plane_test(diag, cand);
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

// The REFRACTION child of a GLASS hit alone, for kernels that let the reflection child travel through the ordinary shade (the
// stream form's split kernel): with ia = dot(d, n) and reflection = d - (2 ia) n in hand -- what the ordinary shade's bounce_axis
// computes by the same operations -- and the seed already past genVec's three draws, this is glass_children's k, Schlick weight R,
// refraction direction and second child (origin, direction, throughput throughput * (color ^* (1 - R)), seed one draw further),
// operation for operation; the first child is (p + reflection ^* epsilon, reflection, throughput * (color ^* R), seed), which is
// what the ordinary path makes of `next = reflection` and the factor R.  sqrt_rn == IEEE sqrtf.
__device__ __forceinline__ float glass_refraction_child(V3 color, GlassConstants gc, V3 p, V3 n, V3 d, float ia, V3 reflection, V3 throughput, Sfc32 seed,
                                                        V3 &o1, V3 &d1, V3 &t1, Sfc32 &s1)
{
    const float cosi = -ia;
    const float eta = gc.eta;
    const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
    const float r0 = gc.r0;
    const float mm = 1.0f - cosi;
    float R = r0 + (1.0f - r0) * (((mm * mm) * (mm * mm)) * mm);
    V3 refraction;
    if (k < 0.0f) { R = 1.0f; refraction = reflection; }
    else refraction = scale_l(eta, d) + scale_l(eta * cosi - sqrt_rn(k), n);
    o1 = p + scale_r(refraction, kEpsilon); d1 = refraction;
    t1 = throughput * scale_r(color, 1.0f - R);
    (void)random_float(seed);
    s1 = seed;
    return R;
}

inline unsigned int blocks_for(long long n, int block = kBlock) { return (unsigned int)((n + block - 1) / block); }

// Sample chunks (render_inline_kernel): only when the launch has few rounds of waves and every copy keeps >= 64 samples
// (every copy re-evaluates the primary hit and moves the planes once more).  Sets b.spp_chunks (>= 1) and clears the flags.
// `rounds`: how many rounds of waves the launch should have at least.  16 for the kernels whose copies are cheap to start (Inline and the Streams
// chain: one part of 8 of a 4K image at 1024 spp is flat between 5 and 8 copies, 21.9 ms, against 22.6 with 2 and 22.4 with 16); 10 for the tree walk,
// every copy of which evaluates its pixels' start records -- up to three traces and a glass split -- again (C5 per part: 29.2-29.3 ms with 4 copies,
// 29.6-29.8 with 7, 30.7 with one; 1080p / 256 spp: 28.5 with one or two, 28.9 with 4).
inline hipError_t choose_sample_chunks(RenderArgs &b, unsigned int per_copy, int waves_per_simd, hipStream_t stream, unsigned long long rounds = 16)
{
    const int wanted = b.spp_chunks;                           // 0 = automatic, 1 = off, k = forced
    b.spp_chunks = 1;
    if (!b.chunk_done || b.chunk_capacity < per_copy || wanted == 1 || b.screen_x) return hipSuccess;
    const int cus = b.cus > 0 ? b.cus : 256;                   // of the context's device (ptmi_create)
    const unsigned long long slots = (unsigned long long)cus * 4ull * (unsigned long long)waves_per_simd;
    int k = wanted > 1 ? wanted : (int)((rounds * slots + per_copy - 1) / per_copy);
    if (wanted <= 0 && k > b.n_spp / 64) k = b.n_spp / 64;
    if (k > b.n_spp) k = b.n_spp;
    if (k > 64) k = 64;
    if (k < 2) return hipSuccess;
    b.spp_chunks = k;
    if (hipError_t e = hipMemsetAsync(b.chunk_done, 0, (size_t)per_copy * sizeof(unsigned int), stream)) return e;
    return hipMemsetAsync(b.chunk_done + b.chunk_capacity, 0, sizeof(unsigned int), stream);      // the ticket counter
}

}  // namespace

}  // namespace ptmi
