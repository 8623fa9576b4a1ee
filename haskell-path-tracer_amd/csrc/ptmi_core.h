// ptmi_core.h -- scalar building blocks of the path-tracing hot path, written for gfx950.
//
// Everything here is __host__ __device__ so that the per-launch camera uniforms (host)
// and the per-pixel work (device) use one definition of the arithmetic.  The contract:
// IEEE binary32, each operation rounded separately (the build passes -ffp-contract=off),
// correctly rounded sqrt / division (hipcc default), sin/cos evaluated with glibc's
// published sinf/cosf algorithm in binary64 (CDNA4 runs f64 VALU at half the f32 rate,
// which makes libm-exact trigonometry affordable on the device).
//
// Reference citations are relative to robbert-vdh/haskell-path-tracer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PTMI_HD __host__ __device__ __forceinline__

namespace ptmi {

struct V3 { float x, y, z; };
struct Quat { float w; V3 v; };
struct Sfc32 { uint32_t a, b, c, counter; };

constexpr float kPi       = 3.14159274101257324219f;   // Accelerate `pi :: Exp Float`
constexpr float kInfinite = 3.40282346638528859812e+38f; // Trace.hs:450-451 encodeFloat 16777215 104
constexpr float kEpsilon  = 0.002f;                      // Trace.hs:455-456

PTMI_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
PTMI_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- linear (V3, Quaternion) -- L0 semantics restated ------------------------------
PTMI_HD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PTMI_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PTMI_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PTMI_HD V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PTMI_HD V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
PTMI_HD V3 scale_r(V3 v, float a) { return mk(v.x * a, v.y * a, v.z * a); }   // v ^* a
PTMI_HD V3 scale_l(float a, V3 v) { return mk(a * v.x, a * v.y, a * v.z); }   // a *^ v
PTMI_HD V3 div_r(V3 v, float a) { return mk(v.x / a, v.y / a, v.z / a); }     // v ^/ a
PTMI_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PTMI_HD V3 cross(V3 a, V3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PTMI_HD bool near_zero(float a) { return __builtin_fabsf(a) <= 1e-6f; }       // Epsilon Float
PTMI_HD bool near_zero(V3 v) { return near_zero(dot(v, v)); }                 // Epsilon (V3 a)
PTMI_HD V3 normalize(V3 v)
{
    float l = dot(v, v);
    if (near_zero(l) || near_zero(1.0f - l)) return v;
    return div_r(v, __builtin_sqrtf(l));
}
PTMI_HD Quat qmul(Quat a, Quat b)
{
    Quat r;
    r.w = a.w * b.w - dot(a.v, b.v);
    r.v = (cross(a.v, b.v) + scale_l(a.w, b.v)) + scale_l(b.w, a.v);
    return r;
}
// rotate q v = vector part of  q * Quaternion 0 v * conjugate q   (left-associated)
PTMI_HD V3 rotate(Quat q, V3 v)
{
    Quat qv; qv.w = 0.0f; qv.v = v;
    Quat qc; qc.w = q.w; qc.v = neg(q.v);
    return qmul(qmul(q, qv), qc).v;
}

// ---- sinf / cosf: glibc 2.35 (ARM optimized-routines) algorithm, both at once --------
// sincosf.h reduce_large(): argument reduction for |y| >= 120 with 4/pi bit windows.
struct Reduced { double x; int n; };
__host__ __device__ __noinline__ inline Reduced sincos_reduce_large(uint32_t xi)
{
    constexpr uint32_t inv_pio4[24] = {
        0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
        0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
        0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
        0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041 };
    const int base = (xi >> 26) & 15;
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = (uint32_t)(xi * inv_pio4[base]);
    uint64_t res1 = (uint64_t)xi * inv_pio4[base + 4];
    uint64_t res2 = (uint64_t)xi * inv_pio4[base + 8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    Reduced r;
    r.x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
    r.n = (int)n;
    return r;
}

// One evaluation of s_sinf.c + s_cosf.c on the same argument.  The two libm routines share
// the reduction (n, x) and differ only in which of the two polynomials they return; for
// |y| < 0.75 glibc skips the reduction, which is the n = 0 case of the general path
// (x - 0*hpi == x exactly), so one code path serves both.
//
// FUSED = false is the literal restatement (every binary64 operation rounded on its own, like
// glibc's non-FMA build).  FUSED = true contracts the a*b+c of the two POLYNOMIALS into binary64
// FMAs (15 instead of 22 f64 operations; the argument reduction stays unfused); it is used by the kernels only because tools/verify_sincos.hip has compared the
// two on the device for ALL 2^32 binary32 arguments and found them bit-identical after the final
// rounding to binary32 (profiles/r01_verify_sincos.txt) -- glibc's own FMA build (`__sinf_fma`,
// the variant x86-64 hosts with FMA dispatch to) is the same kind of contraction.
// An FMA with two constant operands needs one of them in a vector register pair.  Kept live across the render loops those
// pairs were the registers that spilled; formed where it is used -- two moves of literals, the same issue cycles as the copy
// of a pair that the accumulating FMA needs anyway -- it occupies nothing.  (The empty asm keeps the halves from being hoisted.)
PTMI_HD double local_constant(double k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo = (uint32_t)__builtin_bit_cast(uint64_t, k), hi = (uint32_t)(__builtin_bit_cast(uint64_t, k) >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
#else
    return k;
#endif
}

template <bool FUSED>
PTMI_HD void sincos_t(float y, float &sn, float &cs)
{
    const uint32_t bits = f2u(y);
    const uint32_t top = (bits >> 20) & 0x7ff;          // abstop12
    double x = (double)y;
    int n, m;
    if (top < 0x42f) {                                   // |y| < 120: reduce_fast
        double r = x * 0x1.45F306DC9C883p+23;
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * 0x1.921FB54442D18p0;        // never contracted: a fused reduction changes 34 results
        m = n;
    } else if (top < 0x7f8) {
        const Reduced red = sincos_reduce_large(bits);
        x = red.x; n = red.n;
        m = n + (int)(bits >> 31);
    } else {                                             // inf / NaN
        sn = cs = (y - y) / (y - y);
        return;
    }
    const double x2 = x * x;
    // x * sign[m & 3], sign = {1,-1,-1,1}: an exact sign flip -- in the FUSED form on the bit pattern (bit 1 of m + 1 moved onto the
    // sign bit: a compare and a select less per evaluation; covered, like the contraction, by tools/verify_sincos.hip's comparison of
    // the two forms for all 2^32 arguments)
    const double xs = FUSED ? __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, x) ^ ((uint64_t)(((uint32_t)(m + 1) & 2u) << 30) << 32))
                            : (((m + 1) & 2) ? -x : x);
    float S, Cp;
    if (FUSED) {
        const double x3 = xs * x2;
        const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, local_constant(0x1.1107605230bc4p-7));
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, -0x1.555545995a603p-3, xs);
        S = (float)__builtin_fma(x7, s1, s);
        const double x4 = x2 * x2;
        const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, local_constant(-0x1.6c087e89a359dp-10));
        const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 0x1p0);
        const double x6 = x4 * x2;
        const double c = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
        Cp = (float)__builtin_fma(x6, c2, c);
    } else {
        // sine-type polynomial (identical coefficients in both table rows)
        const double x3 = xs * x2;
        const double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
        const double x7 = x3 * x2;
        const double s = xs + x3 * -0x1.555545995a603p-3;
        S = (float)(s + x7 * s1);
        // cosine-type polynomial; table row 1 (m & 2) negates every coefficient = negates the result
        const double x4 = x2 * x2;
        const double c2 = -0x1.6c087e89a359dp-10 + x2 * 0x1.99343027bf8c3p-16;
        const double c1 = 0x1p0 + x2 * -0x1.ffffffd0c621cp-2;
        const double x6 = x4 * x2;
        const double c = c1 + x4 * 0x1.55553e1068f19p-5;
        Cp = (float)(c + x6 * c2);
    }
    // table row 1 (m & 2) negates the cosine polynomial; n & 1 swaps which polynomial is the sine.  Done on the bit
    // patterns (xor / and) instead of selects: back-to-back v_cndmask pairs are slow on gfx950.
    const uint32_t cb = f2u(Cp) ^ (((uint32_t)m & 2u) << 30), sb = f2u(S);
    const uint32_t swap = (sb ^ cb) & (0u - ((uint32_t)n & 1u));
    sn = u2f(sb ^ swap);
    cs = u2f(cb ^ swap);
#if defined(__HIP_DEVICE_COMPILE__)
    // |y| < 2^-12 (of a random angle: one evaluation in ~10 000): in the FUSED form behind a wave-uniform branch instead of two selects in
    // every evaluation (round 6: C2 - 1.4 %)
    if (FUSED) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(top < 0x398) != 0, 0)) { if (top < 0x398) { sn = y; cs = 1.0f; } }
        return;
    }
#endif
    if (top < 0x398) { sn = y; cs = 1.0f; }              // |y| < 2^-12
}

#ifndef PTMI_SINCOS_FUSED
#define PTMI_SINCOS_FUSED 0
#endif
PTMI_HD void sincos(float y, float &sn, float &cs) { sincos_t<PTMI_SINCOS_FUSED != 0>(y, sn, cs); }

// anglesToQuaternion (src/Util.hs:55-67) from the three HALF angles (yaw*0.5 etc. already formed)
PTMI_HD Quat quaternion_from_half_angles(float half_roll, float half_pitch, float half_yaw)
{
    float sr, cr, sp, cp, sy, cy;
    sincos(half_roll, sr, cr);
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the three f64 evaluations from being interleaved: their temporaries are the kernel's VGPR peak
    __builtin_amdgcn_sched_barrier(0);
#endif
    sincos(half_pitch, sp, cp);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
    sincos(half_yaw, sy, cy);
    Quat q;
    q.w   = cy * cp * cr + sy * sp * sr;
    q.v.x = cy * cp * sr - sy * sp * cr;
    q.v.y = sy * cp * sr + cy * sp * cr;
    q.v.z = sy * cp * cr - cy * sp * sr;
    return q;
}

// src/Util.hs:55-67
PTMI_HD Quat angles_to_quaternion(V3 angles)
{
    return quaternion_from_half_angles(angles.x * 0.5f, angles.y * 0.5f, angles.z * 0.5f);
}

// ---- SFC32 (sfc-random-accelerate; PractRand sfc32) ---------------------------------
PTMI_HD uint32_t sfc32_next(Sfc32 &s)
{
    const uint32_t tmp = s.a + s.b + s.counter;
    s.counter += 1u;
    s.a = s.b ^ (s.b >> 9);
    s.b = s.c + (s.c << 3);
    s.c = ((s.c << 21) | (s.c >> 11)) + tmp;
    return tmp;
}
// The state sfc32_next came from: the step is a bijection of the 128-bit state (b -> b ^ (b >> 9) and c -> 9 c are invertible,
// the rest is additions).  Used where a kernel needs the seed a ray carried BEFORE its hit's three draws and kept only the
// one after them.
PTMI_HD void sfc32_prev(Sfc32 &s)
{
    const uint32_t c = s.b * 0x38e38e39u;                    // 9^-1 mod 2^32
    const uint32_t b = s.a ^ (s.a >> 9) ^ (s.a >> 18) ^ (s.a >> 27);
    const uint32_t counter = s.counter - 1u;
    const uint32_t tmp = s.c - ((c << 21) | (c >> 11));
    s.a = tmp - b - counter; s.b = b; s.c = c; s.counter = counter;
}
// random @Float -> (0, 1]   (mwc-random wordToFloat)
PTMI_HD float random_float(Sfc32 &s)
{
    const int32_t i = (int32_t)sfc32_next(s);
    return ((float)i * 2.3283064365386963e-10f + 0.5f) + 1.1641532182693481e-10f;
}
// src/Util.hs:114-118
PTMI_HD V3 gen_vec(Sfc32 &s)
{
    V3 r;
    r.x = (random_float(s) * 2.0f) - 1.0f;
    r.y = (random_float(s) * 2.0f) - 1.0f;
    r.z = (random_float(s) * 2.0f) - 1.0f;
    return r;
}
// createWith: 3-word seeding, counter = 1, 15 outputs discarded
PTMI_HD Sfc32 sfc32_seed3(uint32_t a, uint32_t b, uint32_t c)
{
    Sfc32 s; s.a = a; s.b = b; s.c = c; s.counter = 1u;
    for (int i = 0; i < 15; ++i) (void)sfc32_next(s);
    return s;
}
// deterministic stand-in for the OS-entropy words of genSeeds (src/Util.hs:122-127)
PTMI_HD uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
PTMI_HD uint32_t seed_word(uint64_t seed0, uint64_t index, uint32_t k)
{
    const uint64_t n = index * 3u + k;
    uint32_t h = fmix32((uint32_t)n ^ (uint32_t)seed0);
    return fmix32(h ^ (uint32_t)(n >> 32) ^ (uint32_t)(seed0 >> 32) ^ 0x9e3779b9u);
}

// ---- primaryRays (src/Scene/Trace.hs:205-262) ----------------------------------------
struct PrimaryUniforms { V3 pos, center, right, top; float size_x, size_y; };

// per pixel: Trace.hs:244-262, screenSize = (W, -H) (src/Util.hs:198-200)
PTMI_HD V3 primary_direction(const PrimaryUniforms &u, int64_t px, int64_t py)
{
    const float raster_x = (float)px, raster_y = (float)py;
    const float screen_x = raster_x / u.size_x * 2.0f + (-1.0f);
    const float screen_y = raster_y / u.size_y * 2.0f + 1.0f;
    const V3 virtual_point = (u.center + scale_r(u.right, screen_x)) + scale_r(u.top, screen_y);
    return normalize(virtual_point - u.pos);
}

}  // namespace ptmi
