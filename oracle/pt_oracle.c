/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, see pt_oracle.h for the parity
 * status: intersection PINNED by the reference's 8 properties, the rest of the
 * path PARITY UNPINNED).
 *
 * Every function is a literal restatement: same operations, same order, same
 * associativity as the Haskell source it cites (paths relative to the reference
 * root).  Where the arithmetic lives in an un-vendored dependency the published
 * definition is restated and named:
 *   linear 1.21 / linear-accelerate 0.7  (dot, cross, normalize, nearZero, rotate,
 *                                         ^*, *^, ^/)         tracer.cabal:46-47
 *   sfc-random-accelerate @16fe36ec      (SFC32, random @Float) cabal.project:61-65
 *   glibc 2.35 libm                      (sinf, cosf as lowered by accelerate-llvm-native)
 *
 * Build: see Makefile (-O2 -ffp-contract=off -fno-fast-math are REQUIRED).
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ========================================================================== */
/* sinf / cosf : glibc 2.35 sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h,
 * sincosf_data.c} (from ARM optimized-routines), non-TOINT_INTRINSICS (x86_64)
 * variant, restated.  accelerate-llvm-native lowers Accelerate's sin/cos to the
 * llvm.sin/cos.f32 intrinsics, which become libm calls on x86_64.  Evaluated in
 * binary64, result rounded once to binary32.                                   */
/* ========================================================================== */
typedef struct {
    double sign[4];
    double hpi_inv;                 /* 2/pi * 2^24 */
    double hpi;                     /* pi/2        */
    double c0, c1, c2, c3, c4;      /* cosine polynomial */
    double s1, s2, s3;              /* sine polynomial   */
} ora_sincos_t;

static const ora_sincos_t ora_sincos_table[2] = {
    { { 1.0, -1.0, -1.0, 1.0 },
      0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
      0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
      -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 },
    { { 1.0, -1.0, -1.0, 1.0 },
      0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
      -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
      -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 }
};

/* 4/pi as overlapping 32-bit windows, 8 bits apart (table for huge arguments). */
static const uint32_t ora_inv_pio4[24] = {
    0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
    0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
    0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
    0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041
};

static inline uint32_t ora_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t ora_abstop12(float x) { return (ora_asuint(x) >> 20) & 0x7ff; }

static inline float ora_sinf_poly(double x, double x2, const ora_sincos_t *p, int n)
{
    double x3, x4, x6, x7, s, c, c1, c2, s1;
    if ((n & 1) == 0) {
        x3 = x * x2;
        s1 = p->s2 + x2 * p->s3;
        x7 = x3 * x2;
        s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    } else {
        x4 = x2 * x2;
        c2 = p->c3 + x2 * p->c4;
        c1 = p->c0 + x2 * p->c1;
        x6 = x4 * x2;
        c = c1 + x4 * p->c2;
        return (float)(c + x6 * c2);
    }
}

static inline double ora_reduce_fast(double x, const ora_sincos_t *p, int *np)
{
    double r = x * p->hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * p->hpi;
}

static inline double ora_reduce_large(uint32_t xi, int *np)
{
    const uint32_t *arr = &ora_inv_pio4[(xi >> 26) & 15];
    int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;

    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;

    res0 = xi * arr[0];                    /* 32-bit product, as upstream */
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;

    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

#ifdef ORA_SINCOS_ALT
/* SENSITIVITY STUDY ONLY (tools/sensitivity.py): a different, equally legitimate single-precision sin/cos
 * (Cephes-style reduction and polynomials, ~1 ulp) standing in for "some other libm".  Never part of the
 * oracle proper. */
static void alt_sincosf(float x, float *sn, float *cs)
{
    float ax = fabsf(x);
    int j = (int)(ax * 1.27323954473516f);            /* 4/pi */
    j = (j + 1) & ~1;
    float y = (float)j;
    float z = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float zz = z * z;
    float s = ((-1.9515295891e-4f * zz + 8.3321608736e-3f) * zz - 1.6666654611e-1f) * zz * z + z;
    float c = ((2.443315711809948e-5f * zz - 1.388731625493765e-3f) * zz + 4.166664568298827e-2f) * zz * zz - 0.5f * zz + 1.0f;
    int q = (j >> 1) & 3;
    float rs = (q & 1) ? c : s, rc = (q & 1) ? s : c;
    if (q & 2) rs = -rs;
    if (((q + 1) >> 1) & 1) rc = -rc;
    if (x < 0) rs = -rs;
    *sn = rs; *cs = rc;
}
float ora_sinf(float y) { float s, c; alt_sincosf(y, &s, &c); return s; }
float ora_cosf(float y) { float s, c; alt_sincosf(y, &s, &c); return c; }
#define ora_sinf ora_sinf_glibc_unused
#define ora_cosf ora_cosf_glibc_unused
#endif

float ora_sinf(float y)
{
    double x = y, s;
    int n;
    const ora_sincos_t *p = &ora_sincos_table[0];

    if (ora_abstop12(y) < ora_abstop12(0x1.921FB6p-1f)) {
        s = x * x;
        if (ora_abstop12(y) < ora_abstop12(0x1p-12f))
            return y;
        return ora_sinf_poly(x, s, p, 0);
    } else if (ora_abstop12(y) < ora_abstop12(120.0f)) {
        x = ora_reduce_fast(x, p, &n);
        s = p->sign[n & 3];
        if (n & 2) p = &ora_sincos_table[1];
        return ora_sinf_poly(x * s, x * x, p, n);
    } else if (ora_abstop12(y) < ora_abstop12(INFINITY)) {
        uint32_t xi = ora_asuint(y);
        int sign = xi >> 31;
        x = ora_reduce_large(xi, &n);
        s = p->sign[(n + sign) & 3];
        if ((n + sign) & 2) p = &ora_sincos_table[1];
        return ora_sinf_poly(x * s, x * x, p, n);
    }
    return (y - y) / (y - y);              /* inf, NaN -> NaN */
}

float ora_cosf(float y)
{
    double x = y, s;
    int n;
    const ora_sincos_t *p = &ora_sincos_table[0];

    if (ora_abstop12(y) < ora_abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (ora_abstop12(y) < ora_abstop12(0x1p-12f))
            return 1.0f;
        return ora_sinf_poly(x, x2, p, 1);
    } else if (ora_abstop12(y) < ora_abstop12(120.0f)) {
        x = ora_reduce_fast(x, p, &n);
        s = p->sign[n & 3];
        if (n & 2) p = &ora_sincos_table[1];
        return ora_sinf_poly(x * s, x * x, p, n ^ 1);
    } else if (ora_abstop12(y) < ora_abstop12(INFINITY)) {
        uint32_t xi = ora_asuint(y);
        int sign = xi >> 31;
        x = ora_reduce_large(xi, &n);
        s = p->sign[(n + sign) & 3];
        if ((n + sign) & 2) p = &ora_sincos_table[1];
        return ora_sinf_poly(x * s, x * x, p, n ^ 1);
    }
    return (y - y) / (y - y);
}

#ifdef ORA_SINCOS_ALT
#undef ora_sinf
#undef ora_cosf
#endif

/* ========================================================================== */
/* linear / linear-accelerate vector algebra (L0, restated)                    */
/* ========================================================================== */
static const float ORA_PI = 3.14159274101257324219f;   /* Accelerate `pi :: Exp Float` */

static inline ora_v3 v3(float x, float y, float z) { ora_v3 r = { x, y, z }; return r; }
static inline ora_v3 v3_add(ora_v3 a, ora_v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ora_v3 v3_sub(ora_v3 a, ora_v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ora_v3 v3_mul(ora_v3 a, ora_v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ora_v3 v3_neg(ora_v3 a) { return v3(-a.x, -a.y, -a.z); }
/* v ^* a = fmap (*a) v ; a *^ v = fmap (a*) v ; v ^/ a = fmap (/a) v */
static inline ora_v3 v3_scale_r(ora_v3 v, float a) { return v3(v.x * a, v.y * a, v.z * a); }
static inline ora_v3 v3_scale_l(float a, ora_v3 v) { return v3(a * v.x, a * v.y, a * v.z); }
static inline ora_v3 v3_div(ora_v3 v, float a) { return v3(v.x / a, v.y / a, v.z / a); }
/* dot (V3 a b c) (V3 d e f) = a*d + b*e + c*f   (infixl 6 +) */
static inline float v3_dot(ora_v3 a, ora_v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* cross (V3 a b c) (V3 d e f) = V3 (b*f-c*e) (c*d-a*f) (a*e-b*d) */
static inline ora_v3 v3_cross(ora_v3 a, ora_v3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* Epsilon Float: nearZero a = abs a <= 1e-6 ; Epsilon (V3 a): nearZero = nearZero . quadrance */
static inline int near_zero_f(float a) { return fabsf(a) <= 1e-6f; }
int ora_near_zero_v3(ora_v3 v) { return near_zero_f(v3_dot(v, v)); }

/* normalize v = if nearZero l || nearZero (1-l) then v else fmap (/sqrt l) v where l = quadrance v */
ora_v3 ora_normalize(ora_v3 v)
{
    float l = v3_dot(v, v);
    if (near_zero_f(l) || near_zero_f(1.0f - l)) return v;
    return v3_div(v, sqrtf(l));
}

/* Quaternion s1 v1 * Quaternion s2 v2 =
 *   Quaternion (s1*s2 - (v1 `dot` v2)) ((v1 `cross` v2) + s1*^v2 + s2*^v1)      */
static inline ora_quat quat_mul(ora_quat a, ora_quat b)
{
    ora_quat r;
    r.w = a.w * b.w - v3_dot(a.v, b.v);
    r.v = v3_add(v3_add(v3_cross(a.v, b.v), v3_scale_l(a.w, b.v)), v3_scale_l(b.w, a.v));
    return r;
}

/* rotate q v = ijk where Quaternion _ ijk = q * Quaternion 0 v * conjugate q   (infixl 7 *) */
ora_v3 ora_rotate(ora_quat q, ora_v3 v)
{
    ora_quat qv = { 0.0f, v };
    ora_quat qc = { q.w, v3_neg(q.v) };
    return quat_mul(quat_mul(q, qv), qc).v;
}

/* src/Util.hs:55-67 */
ora_quat ora_angles_to_quaternion(ora_v3 angles)
{
    float roll = angles.x, pitch = angles.y, yaw = angles.z;
    float cy = ora_cosf(yaw * 0.5f),   sy = ora_sinf(yaw * 0.5f);
    float cp = ora_cosf(pitch * 0.5f), sp = ora_sinf(pitch * 0.5f);
    float cr = ora_cosf(roll * 0.5f),  sr = ora_sinf(roll * 0.5f);
    ora_quat q;
    q.w   = cy * cp * cr + sy * sp * sr;
    q.v.x = cy * cp * sr - sy * sp * cr;
    q.v.y = sy * cp * sr + cy * sp * cr;
    q.v.z = sy * cp * cr - cy * sp * sr;
    return q;
}

/* ========================================================================== */
/* SFC32 (sfc-random-accelerate, L0): PractRand sfc32, PARITY UNPINNED         */
/* ========================================================================== */
uint32_t ora_sfc32_next(ora_sfc32 *s)
{
    uint32_t tmp = s->a + s->b + s->counter;
    s->counter += 1u;
    s->a = s->b ^ (s->b >> 9);
    s->b = s->c + (s->c << 3);
    s->c = ((s->c << 21) | (s->c >> 11)) + tmp;
    return tmp;
}

/* random @Float: mwc-random's wordToFloat, result in (0,1] */
float ora_random_float(ora_sfc32 *s)
{
    int32_t i = (int32_t)ora_sfc32_next(s);
    return ((float)i * 2.3283064365386963e-10f + 0.5f) + 1.1641532182693481e-10f;
}

/* createWith: PractRand 3-word seeding -- counter = 1, discard 15 outputs */
ora_sfc32 ora_sfc32_seed3(uint32_t a, uint32_t b, uint32_t c)
{
    ora_sfc32 s = { a, b, c, 1u };
    for (int i = 0; i < 15; ++i) (void)ora_sfc32_next(&s);
    return s;
}

/* Deterministic stand-in for Rng.uniform on OS entropy (Util.hs:122-127): word k of
 * pixel `index` = murmur3 fmix32((index*3 + k) folded to 32 bits, xor seed0 halves). */
static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
void ora_seed_words(uint64_t seed0, uint64_t index, uint32_t out[3])
{
    uint32_t lo = (uint32_t)seed0, hi = (uint32_t)(seed0 >> 32);
    for (uint32_t k = 0; k < 3; ++k) {
        uint64_t n = index * 3u + k;
        uint32_t h = fmix32((uint32_t)n ^ lo);
        h = fmix32(h ^ (uint32_t)(n >> 32) ^ hi ^ 0x9e3779b9u);
        out[k] = h;
    }
}

void ora_gen_seeds(uint64_t seed0, int64_t first_index, int64_t n,
                   uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr)
{
    for (int64_t i = 0; i < n; ++i) {
        uint32_t w[3];
        ora_seed_words(seed0, (uint64_t)(first_index + i), w);
        ora_sfc32 s = ora_sfc32_seed3(w[0], w[1], w[2]);
        sa[i] = s.a; sb[i] = s.b; sc[i] = s.c; sctr[i] = s.counter;
    }
}

/* src/Util.hs:114-118: V3_ <$> rng <*> rng <*> rng, rng = (\x -> (x * 2.0) - 1.0) <$> random */
ora_v3 ora_gen_vec(ora_sfc32 *s)
{
    ora_v3 r;
    r.x = (ora_random_float(s) * 2.0f) - 1.0f;
    r.y = (ora_random_float(s) * 2.0f) - 1.0f;
    r.z = (ora_random_float(s) * 2.0f) - 1.0f;
    return r;
}

/* ========================================================================== */
/* src/Scene/Intersection.hs                                                   */
/* ========================================================================== */
static inline ora_v3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

/* Intersection.hs:39-48 */
ora_maybe_float ora_distance_to_sphere(ora_ray r, const ora_sphere *s)
{
    ora_v3 pos = ld3(s->position);
    float rad = s->radius;
    ora_v3 l = v3_sub(pos, r.origin);
    float tca = v3_dot(l, r.direction);
    float d2 = v3_dot(l, l) - (tca * tca);
    float rad2 = rad * rad;                         /* rad ** 2 */
    float thc = sqrtf(rad2 - d2);
    float t0 = tca - thc;
    float t1 = tca + thc;
    float t = fminf(t0, t1);
    ora_maybe_float m;
    if (tca < 0.0f || d2 > rad2 || t < 0.0f) { m.is_just = 0; m.value = 0.0f; }
    else { m.is_just = 1; m.value = t; }
    return m;
}

/* Intersection.hs:57-62 */
ora_maybe_float ora_distance_to_plane(ora_ray r, const ora_plane *p)
{
    ora_v3 pos = ld3(p->position), nor = ld3(p->direction);
    float denom = v3_dot(r.direction, nor);
    float dist = v3_dot(v3_sub(pos, r.origin), nor) / denom;
    ora_maybe_float m;
    if (denom > 1e-6f || dist < 0.0f) { m.is_just = 0; m.value = 0.0f; }
    else { m.is_just = 1; m.value = dist; }
    return m;
}

/* Intersection.hs:29-32 with normal = Intersection.hs:50 */
ora_maybe_hit ora_hit_sphere(ora_ray r, float t, const ora_sphere *s)
{
    ora_maybe_hit h;
    ora_v3 hit_position = v3_add(r.origin, v3_scale_r(r.direction, t));
    h.is_just = 1;
    h.normal_p.origin = hit_position;
    h.normal_p.direction = ora_normalize(v3_sub(hit_position, ld3(s->position)));
    h.material.color = ld3(s->color);
    h.material.illuminance = s->illuminance;
    h.material.brdf_tag = s->brdf_tag;
    h.material.brdf_param = s->brdf_param;
    return h;
}

/* Intersection.hs:29-32 with normal = Intersection.hs:64 */
ora_maybe_hit ora_hit_plane(ora_ray r, float t, const ora_plane *p)
{
    ora_maybe_hit h;
    ora_v3 hit_position = v3_add(r.origin, v3_scale_r(r.direction, t));
    h.is_just = 1;
    h.normal_p.origin = hit_position;
    h.normal_p.direction = ld3(p->direction);
    h.material.color = ld3(p->color);
    h.material.illuminance = p->illuminance;
    h.material.brdf_tag = p->brdf_tag;
    h.material.brdf_param = p->brdf_param;
    return h;
}

/* ========================================================================== */
/* src/Scene/Trace.hs                                                          */
/* ========================================================================== */
/* Trace.hs:450-451  infinite = encodeFloat 16777215 104 = FLT_MAX */
static const float ORA_INFINITE = 3.40282346638528859812e+38f;
/* Trace.hs:455-456 */
static const float ORA_EPSILON = 0.002f;

/* Trace.hs:443-447 with Util.hs:156-158 (mapScene: spheres ++ planes) and
 * Util.hs:171-178 (expMinWith: foldl', `cond (valA <= valB) a b`).            */
ora_maybe_hit ora_check_hit(const ora_scene *scene, ora_ray r)
{
    ora_maybe_hit acc; float acc_key = 0.0f; int first = 1;
    memset(&acc, 0, sizeof acc);
    int n = scene->n_spheres + scene->n_planes;
    for (int i = 0; i < n; ++i) {
        ora_maybe_float d;
        ora_maybe_hit h;
        if (i < scene->n_spheres) {
            d = ora_distance_to_sphere(r, &scene->spheres[i]);
            if (d.is_just) h = ora_hit_sphere(r, d.value, &scene->spheres[i]);
        } else {
            const ora_plane *p = &scene->planes[i - scene->n_spheres];
            d = ora_distance_to_plane(r, p);
            if (d.is_just) h = ora_hit_plane(r, d.value, p);
        }
        if (!d.is_just) memset(&h, 0, sizeof h);     /* Nothing */
        float key = d.is_just ? d.value : ORA_INFINITE;   /* maybe infinite fst */
        if (first) { acc = h; acc_key = key; first = 0; }
        else if (!(acc_key <= key)) { acc = h; acc_key = key; }   /* cond (valA <= valB) a b */
    }
    return acc;                                         /* fmap snd */
}

/* Trace.hs:394-435 */
void ora_calc_next_ray(const ora_material *m, ora_ray normal_p, ora_ray ray, ora_sfc32 seed,
                       ora_ray *next_ray, ora_v3 *throughput_mod, ora_sfc32 *seed_out)
{
    ora_v3 i_point = normal_p.origin, i_normal = normal_p.direction;
    ora_v3 rotation_vector = ora_gen_vec(&seed);
    ora_v3 next; float brdf;
    if (m->brdf_tag == ORA_MATTE) {
        float p = m->brdf_param;
        next = ora_rotate(ora_angles_to_quaternion(v3_scale_l(ORA_PI, rotation_vector)), i_normal);
        brdf = p / ORA_PI * v3_dot(next, i_normal);
    } else {
        float p = m->brdf_param;
        float intersection_angle = v3_dot(ray.direction, i_normal);
        ora_v3 reflection = v3_sub(ray.direction, v3_scale_l(2.0f * intersection_angle, i_normal));
        next = ora_rotate(ora_angles_to_quaternion(v3_scale_l(1.0f - p, rotation_vector)), reflection);
        brdf = fmaxf(0.0f, v3_dot(next, reflection));
    }
    next_ray->origin = v3_add(i_point, v3_scale_r(next, ORA_EPSILON));
    next_ray->direction = next;
    float next_ray_prob = 1.0f / (ORA_PI * 2.0f);
    *throughput_mod = v3_scale_r(m->color, brdf * next_ray_prob);
    *seed_out = seed;
}

/* Trace.hs:344-383 */
void ora_trace_inline(int limit, const ora_scene *scene, ora_ray primary, ora_sfc32 seed,
                      ora_v3 *color_out, ora_sfc32 *seed_out, int *live_bounces)
{
    ora_ray ray = primary;
    ora_v3 result = v3(0.0f, 0.0f, 0.0f);
    ora_v3 throughput = v3(1.0f, 1.0f, 1.0f);
    int live = 0;
    for (int it = 0; it < limit; ++it) {                 /* iterate limit prepareRay */
        ora_maybe_hit next_hit;
        if (ora_near_zero_v3(throughput) || !(next_hit = ora_check_hit(scene, ray)).is_just) {
            throughput = v3(0.0f, 0.0f, 0.0f);
        } else {                                          /* computeRay */
            ora_v3 emittance = v3_scale_r(next_hit.material.color, next_hit.material.illuminance);
            ora_ray next_ray; ora_v3 tmod; ora_sfc32 seed2;
            ora_calc_next_ray(&next_hit.material, next_hit.normal_p, ray, seed, &next_ray, &tmod, &seed2);
            result = v3_add(result, v3_mul(emittance, throughput));
            throughput = v3_mul(throughput, tmod);
            ray = next_ray; seed = seed2;
            ++live;
        }
    }
    *color_out = result; *seed_out = seed;
    if (live_bounces) *live_bounces = live;
}

/* Trace.hs:205-242 (per-launch values) */
ora_primary_uniforms ora_primary_setup(const ora_camera *cam, int width, int height)
{
    ora_primary_uniforms u;
    float c_fov = (float)cam->fov;
    float screen_angle = (c_fov * ORA_PI / 180.0f) / 2.0f;
    float screen_distance = 1.0f / tanf(screen_angle);
    float screen_half_width = tanf(screen_angle) * screen_distance;
    ora_v3 c_pos = ld3(cam->position);
    ora_v3 c_dir = ora_rotate(ora_angles_to_quaternion(ld3(cam->rotation)), v3(0.0f, 0.0f, -1.0f));
    float screen_aspect = (float)width / (float)height;             /* Util.hs:192-193 */
    ora_v3 center = v3_add(c_pos, v3_scale_r(c_dir, screen_distance));
    ora_v3 center_offset = v3_sub(center, c_pos);
    ora_v3 right = v3_div(ora_normalize(v3_cross(center_offset, v3(0.0f, 1.0f, 0.0f))), screen_half_width);
    ora_v3 top = v3_div(v3_cross(c_dir, right), screen_aspect);
    u.pos = c_pos; u.center = center; u.right = right; u.top = top; u.inv_w_dummy = 0.0f;
    return u;
}

/* Trace.hs:244-262 with screenSize = (W, -H) (Util.hs:198-200) */
ora_ray ora_primary_ray(const ora_primary_uniforms *u, int64_t x, int64_t y, int width, int height)
{
    float raster_x = (float)x, raster_y = (float)y;
    float size_x = (float)width, size_y = (float)(-height);
    float screen_x = raster_x / size_x * 2.0f + (-1.0f);
    float screen_y = raster_y / size_y * 2.0f + 1.0f;
    ora_v3 virtual_point = v3_add(v3_add(u->center, v3_scale_r(u->right, screen_x)),
                                  v3_scale_r(u->top, screen_y));
    ora_ray r;
    r.origin = u->pos;
    r.direction = ora_normalize(v3_sub(virtual_point, u->pos));
    return r;
}

/* Which image row a held row is: rows == NULL means the whole image (row i is image row i). */
static inline int image_row(const ora_opts *o, int local_row) { return (o && o->rows) ? o->rows[local_row] : local_row; }
static inline int held_rows(const ora_opts *o, int height) { return (o && o->rows) ? o->n_rows : height; }

/* Trace.hs:193-200, applied n_spp times, on the rows the planes hold (ora_opts.rows; default: all) */
int64_t ora_render_inline_ex(const ora_scene *scene, const ora_camera *cam,
                             int width, int height, int bounce_limit, int n_spp,
                             const int64_t *screen_x, const int64_t *screen_y,
                             float *r, float *g, float *b,
                             uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr,
                             int n_threads, const ora_opts *opts)
{
    ora_primary_uniforms u = ora_primary_setup(cam, width, height);
    int64_t live_total = 0;
    const int n_rows = held_rows(opts, height);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+:live_total) num_threads(n_threads > 1 ? n_threads : 1)
#endif
    for (int row = 0; row < n_rows; ++row) {
        for (int col = 0; col < width; ++col) {
            int64_t i = (int64_t)row * width + col;
            int64_t px = screen_x ? screen_x[i] : col;
            int64_t py = screen_y ? screen_y[i] : image_row(opts, row);
            ora_ray primary = ora_primary_ray(&u, px, py, width, height);
            ora_sfc32 seed = { sa[i], sb[i], sc[i], sctr[i] };
            ora_v3 acc = v3(r[i], g[i], b[i]);
            for (int s = 0; s < n_spp; ++s) {
                ora_v3 nw; ora_sfc32 seed2; int live;
                ora_trace_inline(bounce_limit, scene, primary, seed, &nw, &seed2, &live);
                acc = v3_add(nw, acc);                    /* new + old */
                seed = seed2;
                live_total += live;
            }
            r[i] = acc.x; g[i] = acc.y; b[i] = acc.z;
            sa[i] = seed.a; sb[i] = seed.b; sc[i] = seed.c; sctr[i] = seed.counter;
        }
    }
    (void)n_threads;
    return live_total;
}

int64_t ora_render_inline(const ora_scene *scene, const ora_camera *cam,
                          int width, int height, int bounce_limit, int n_spp,
                          const int64_t *screen_x, const int64_t *screen_y,
                          float *r, float *g, float *b,
                          uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr,
                          int n_threads)
{
    return ora_render_inline_ex(scene, cam, width, height, bounce_limit, n_spp, screen_x, screen_y,
                                r, g, b, sa, sb, sc, sctr, n_threads, NULL);
}

/* Trace.hs:141-191 + 272-331.  Each pixel owns exactly one ray per step
 * (numNewRays is 0 or 1, Trace.hs:329-331), so the stream algorithm is restated
 * per pixel; array-level steps (expand / permute) only move data.
 *   step: hit = checkHit ray                                     (:275-279)
 *         results gets (pixel, emittance*throughput, seed) iff hit  (:290-293, :318-323)
 *         combine: acc = acc + colour; WHICH seed survives depends on the order in which
 *                  Accelerate's permute hands (new value, old value) to the combination
 *                  function -- see ORA_SEED_* in pt_oracle.h                (:179-184)
 *         new ray iff not (nearZero throughput || isNothing hit)    (:284-289, :329-331)
 *   notFinished never stops a non-empty stream (:166-170) -> `max_iterations` is
 *   only a safety cap here (reference: none); updateSeed advances the pixel's
 *   seed by one draw (:151, :190-191).                                          */
int64_t ora_render_streams_ex(const ora_scene *scene, const ora_camera *cam,
                              int width, int height, int max_iterations, int n_spp,
                              float *r, float *g, float *b,
                              uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr,
                              const ora_opts *opts, int64_t *truncated)
{
    ora_primary_uniforms u = ora_primary_setup(cam, width, height);
    int64_t live_total = 0, n_truncated = 0;
    const int n_rows = held_rows(opts, height);
    const int from_result = opts && opts->streams_seed_rule == ORA_SEED_FROM_RESULT;
    const int n_threads = opts && opts->n_threads > 1 ? opts->n_threads : 1;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+:live_total, n_truncated) num_threads(n_threads)
#endif
    for (int row = 0; row < n_rows; ++row) {
        for (int col = 0; col < width; ++col) {
            int64_t i = (int64_t)row * width + col;
            ora_ray primary = ora_primary_ray(&u, col, image_row(opts, row), width, height);
            ora_sfc32 pixel_seed = { sa[i], sb[i], sc[i], sctr[i] };
            ora_v3 acc = v3(r[i], g[i], b[i]);
            for (int s = 0; s < n_spp; ++s) {
                ora_ray ray = primary;
                ora_v3 throughput = v3(1.0f, 1.0f, 1.0f);
                ora_sfc32 seed = pixel_seed;
                int it;
                for (it = 0; it < max_iterations; ++it) {
                    ora_maybe_hit hit = ora_check_hit(scene, ray);
                    if (hit.is_just) {
                        ora_v3 emittance = v3_scale_r(hit.material.color, hit.material.illuminance);
                        acc = v3_add(acc, v3_mul(emittance, throughput));
                        if (from_result) pixel_seed = seed;   /* combine new old: the RayResult's seed (:321) replaces the accumulator's */
                    }
                    if (ora_near_zero_v3(throughput) || !hit.is_just) break;
                    ora_ray next_ray; ora_v3 tmod; ora_sfc32 seed2;
                    ora_calc_next_ray(&hit.material, hit.normal_p, ray, seed, &next_ray, &tmod, &seed2);
                    throughput = v3_mul(throughput, tmod);
                    ray = next_ray; seed = seed2;
                    ++live_total;
                }
                if (it == max_iterations) ++n_truncated;      /* a ray was still alive when the safety cap stopped it */
                (void)ora_random_float(&pixel_seed);      /* updateSeed */
            }
            r[i] = acc.x; g[i] = acc.y; b[i] = acc.z;
            sa[i] = pixel_seed.a; sb[i] = pixel_seed.b; sc[i] = pixel_seed.c; sctr[i] = pixel_seed.counter;
        }
    }
    if (truncated) *truncated = n_truncated;
    return live_total;
}

int64_t ora_render_streams(const ora_scene *scene, const ora_camera *cam,
                           int width, int height, int max_iterations, int n_spp,
                           float *r, float *g, float *b,
                           uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr)
{
    return ora_render_streams_ex(scene, cam, width, height, max_iterations, n_spp, r, g, b, sa, sb, sc, sctr, NULL, NULL);
}

/* ========================================================================== */
/* Streams as a stream, with the build-defined GLASS extension                 */
/* ========================================================================== */
typedef struct { ora_ray ray; int64_t pixel; ora_v3 throughput; ora_sfc32 seed; } ora_ray_state;   /* Trace.hs:46 */

/* GLASS (extension, no reference semantics; the reference only says rays "will expand into zero, one, or
 * two new rays", Trace.hs:109-113).  Defined here, mirrored by the device:
 *   (rv, seed') = genVec seed                       -- drawn before the match on the BRDF (Trace.hs:402)
 *   dn = d . n ; cosi = -dn ; eta = 1 / ior ; k = 1 - (eta*eta) * (1 - cosi*cosi)
 *   reflection = d - (2*dn) *^ n                     -- the Glossy formula (Trace.hs:421)
 *   r0 = ((1-ior)/(1+ior))^2 ; m = 1 - cosi ; R = r0 + (1-r0) * (((m*m)*(m*m))*m)   -- Schlick
 *   k < 0 (total internal reflection):  R = 1, refraction = reflection
 *   else refraction = eta *^ d + (eta*cosi - sqrt k) *^ n
 *   child 0 = (iPoint + reflection ^* epsilon, reflection), throughput * (color ^* R),     seed'
 *   child 1 = (iPoint + refraction ^* epsilon, refraction), throughput * (color ^* (1-R)), seed' advanced one draw
 * Back faces are culled by distanceTo, so a refracted ray never meets the far side of its sphere.       */
static void glass_children(const ora_material *m, ora_ray normal_p, ora_ray ray, ora_v3 throughput, ora_sfc32 seed,
                           ora_ray_state out[2])
{
    ora_v3 n = normal_p.direction, d = ray.direction, p = normal_p.origin;
    (void)ora_gen_vec(&seed);
    float ior = m->brdf_param;
    float dn = v3_dot(d, n);
    float cosi = -dn;
    float eta = 1.0f / ior;
    float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
    ora_v3 reflection = v3_sub(d, v3_scale_l(2.0f * dn, n));
    float r0 = (1.0f - ior) / (1.0f + ior); r0 = r0 * r0;
    float mm = 1.0f - cosi;
    float R = r0 + (1.0f - r0) * (((mm * mm) * (mm * mm)) * mm);
    ora_v3 refraction;
    if (k < 0.0f) { R = 1.0f; refraction = reflection; }
    else refraction = v3_add(v3_scale_l(eta, d), v3_scale_l(eta * cosi - sqrtf(k), n));
    out[0].ray.origin = v3_add(p, v3_scale_r(reflection, ORA_EPSILON)); out[0].ray.direction = reflection;
    out[0].throughput = v3_mul(throughput, v3_scale_r(m->color, R));
    out[0].seed = seed;
    out[1].ray.origin = v3_add(p, v3_scale_r(refraction, ORA_EPSILON)); out[1].ray.direction = refraction;
    out[1].throughput = v3_mul(throughput, v3_scale_r(m->color, 1.0f - R));
    (void)ora_random_float(&seed);
    out[1].seed = seed;
}

int64_t ora_render_streams_wavefront_ex(const ora_scene *scene, const ora_camera *cam,
                                        int width, int height, int hard_cap, int n_spp, int capacity_factor,
                                        float *r, float *g, float *b,
                                        uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr,
                                        int64_t *dropped, int *steps_last_sample,
                                        const ora_opts *opts, int64_t *truncated)
{
    const int n_rows = held_rows(opts, height);
    const int from_result = opts && opts->streams_seed_rule == ORA_SEED_FROM_RESULT;
    const int64_t n_px = (int64_t)width * n_rows, cap = n_px * (capacity_factor > 0 ? capacity_factor : 1);
    ora_primary_uniforms u = ora_primary_setup(cam, width, height);
    ora_ray_state *cur = malloc(sizeof *cur * (size_t)cap), *nxt = malloc(sizeof *nxt * (size_t)cap);
    int64_t live_total = 0, n_dropped = 0, n_truncated = 0;
    int steps = 0;
    for (int s = 0; s < n_spp; ++s) {
        int64_t n_cur = n_px;                                   /* initialState (Trace.hs:158-162) */
        for (int64_t i = 0; i < n_px; ++i) {
            cur[i].ray = ora_primary_ray(&u, i % width, image_row(opts, (int)(i / width)), width, height);
            cur[i].pixel = i; cur[i].throughput = v3(1.0f, 1.0f, 1.0f);
            cur[i].seed.a = sa[i]; cur[i].seed.b = sb[i]; cur[i].seed.c = sc[i]; cur[i].seed.counter = sctr[i];
        }
        for (steps = 0; n_cur > 0 && steps < hard_cap; ++steps) {          /* awhile notFinished (:142-150, :166-170) */
            int64_t n_nxt = 0;
            for (int64_t i = 0; i < n_cur; ++i) {                         /* traceStep (:272-294) */
                const ora_ray_state *rs = &cur[i];
                ora_maybe_hit hit = ora_check_hit(scene, rs->ray);
                if (!hit.is_just) continue;
                ora_v3 emittance = v3_scale_r(hit.material.color, hit.material.illuminance);
                ora_v3 c = v3_mul(emittance, rs->throughput);                /* computeResult (:318-323) */
                r[rs->pixel] = r[rs->pixel] + c.x; g[rs->pixel] = g[rs->pixel] + c.y; b[rs->pixel] = b[rs->pixel] + c.z;   /* permute (+) */
                if (from_result) {                                            /* combine new old: the result's seed survives (:181) */
                    sa[rs->pixel] = rs->seed.a; sb[rs->pixel] = rs->seed.b; sc[rs->pixel] = rs->seed.c; sctr[rs->pixel] = rs->seed.counter;
                }
                if (ora_near_zero_v3(rs->throughput)) continue;              /* numNewRays (:329-331) */
                ora_ray_state kids[2]; int n_kids;
                if (hit.material.brdf_tag == ORA_GLASS) {
                    glass_children(&hit.material, hit.normal_p, rs->ray, rs->throughput, rs->seed, kids);
                    n_kids = 2;
                } else {
                    ora_v3 tmod;
                    ora_calc_next_ray(&hit.material, hit.normal_p, rs->ray, rs->seed, &kids[0].ray, &tmod, &kids[0].seed);
                    kids[0].throughput = v3_mul(rs->throughput, tmod);
                    n_kids = 1;
                }
                for (int kk = 0; kk < n_kids; ++kk) {                        /* expand: children in element order */
                    if (n_nxt < cap) { kids[kk].pixel = rs->pixel; nxt[n_nxt++] = kids[kk]; ++live_total; }
                    else ++n_dropped;
                }
            }
            ora_ray_state *t = cur; cur = nxt; nxt = t; n_cur = n_nxt;
        }
        n_truncated += n_cur;                                                /* rays still in the stream when the safety cap stopped awhile */
        for (int64_t i = 0; i < n_px; ++i) {                                 /* map updateSeed (:151, :190-191) */
            ora_sfc32 sd = { sa[i], sb[i], sc[i], sctr[i] };
            (void)ora_random_float(&sd);
            sa[i] = sd.a; sb[i] = sd.b; sc[i] = sd.c; sctr[i] = sd.counter;
        }
    }
    free(cur); free(nxt);
    if (dropped) *dropped = n_dropped;
    if (steps_last_sample) *steps_last_sample = steps;
    if (truncated) *truncated = n_truncated;
    return live_total;
}

int64_t ora_render_streams_wavefront(const ora_scene *scene, const ora_camera *cam,
                                     int width, int height, int hard_cap, int n_spp, int capacity_factor,
                                     float *r, float *g, float *b,
                                     uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr,
                                     int64_t *dropped, int *steps_last_sample)
{
    return ora_render_streams_wavefront_ex(scene, cam, width, height, hard_cap, n_spp, capacity_factor,
                                           r, g, b, sa, sb, sc, sctr, dropped, steps_last_sample, NULL, NULL);
}

/* Streams with ray splitting, the same rays as ora_render_streams_wavefront_ex but visited per pixel and DEPTH FIRST:
 * at a GLASS hit the first child (reflection) is followed at once, the second (refraction) waits on a stack and the
 * most recent waiting child is resumed when a lineage ends.  Every ray carries its step index (the awhile iteration it
 * belongs to in the stream form), so hard_cap cuts the same rays.  Only the ORDER in which a pixel's contributions are
 * added differs from the stream form -- Accelerate's permute leaves that order undefined -- which is why the two forms
 * agree to rounding only.  This is the order the device's tree-walk kernel uses (bit-exact against this function).
 * stack_depth bounds the waiting children of one pixel (the device holds kTreeStackDepth = 16); a child that finds
 * the stack full is dropped and counted. */
int64_t ora_render_streams_tree(const ora_scene *scene, const ora_camera *cam,
                                int width, int height, int hard_cap, int n_spp, int stack_depth,
                                float *r, float *g, float *b,
                                uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr,
                                int64_t *dropped, int *longest_lineage,
                                const ora_opts *opts, int64_t *truncated)
{
    typedef struct { ora_ray_state rs; int steps; } waiting;
    const int n_rows = held_rows(opts, height);
    ora_primary_uniforms u = ora_primary_setup(cam, width, height);
    int64_t live_total = 0, n_dropped = 0, n_truncated = 0;
    int longest = 0;
    const int n_threads = opts && opts->n_threads > 1 ? opts->n_threads : 1;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+:live_total, n_dropped, n_truncated) reduction(max:longest) num_threads(n_threads)
#endif
    for (int row = 0; row < n_rows; ++row) {
        waiting *stack = malloc(sizeof *stack * (size_t)(stack_depth > 0 ? stack_depth : 1));     /* (per row: the rows run on several threads) */
        for (int col = 0; col < width; ++col) {
            int64_t i = (int64_t)row * width + col;
            ora_ray primary = ora_primary_ray(&u, col, image_row(opts, row), width, height);
            ora_sfc32 pixel_seed = { sa[i], sb[i], sc[i], sctr[i] };
            ora_v3 acc = v3(r[i], g[i], b[i]);
            for (int s = 0; s < n_spp; ++s) {
                int sp = 0, steps = 0, have = 1;
                ora_ray_state cur;
                cur.ray = primary; cur.pixel = i; cur.throughput = v3(1.0f, 1.0f, 1.0f); cur.seed = pixel_seed;
                while (have) {
                    int ended = 1;                                        /* does this lineage end here? */
                    if (steps >= hard_cap) { ++n_truncated; }             /* awhile had stopped: the ray is never traced */
                    else {
                        ora_maybe_hit hit = ora_check_hit(scene, cur.ray);
                        ++steps;
                        if (steps > longest) longest = steps;
                        if (hit.is_just) {
                            ora_v3 emittance = v3_scale_r(hit.material.color, hit.material.illuminance);
                            acc = v3_add(acc, v3_mul(emittance, cur.throughput));   /* computeResult + permute (+) */
                            if (!ora_near_zero_v3(cur.throughput)) {                  /* numNewRays */
                                if (hit.material.brdf_tag == ORA_GLASS) {
                                    ora_ray_state kids[2];
                                    glass_children(&hit.material, hit.normal_p, cur.ray, cur.throughput, cur.seed, kids);
                                    live_total += 2;
                                    if (steps >= hard_cap) { n_truncated += 2; }
                                    else {
                                        if (sp < stack_depth) { stack[sp].rs = kids[1]; stack[sp].steps = steps; ++sp; }
                                        else ++n_dropped;
                                        cur.ray = kids[0].ray; cur.throughput = kids[0].throughput; cur.seed = kids[0].seed;
                                        ended = 0;
                                    }
                                } else {
                                    ora_ray next_ray; ora_v3 tmod; ora_sfc32 seed2;
                                    ora_calc_next_ray(&hit.material, hit.normal_p, cur.ray, cur.seed, &next_ray, &tmod, &seed2);
                                    cur.throughput = v3_mul(cur.throughput, tmod);
                                    cur.ray = next_ray; cur.seed = seed2;
                                    ++live_total;
                                    if (steps >= hard_cap) ++n_truncated; else ended = 0;
                                }
                            }
                        }
                    }
                    if (ended) {
                        if (sp > 0) { --sp; cur = stack[sp].rs; steps = stack[sp].steps; }
                        else have = 0;
                    }
                }
                (void)ora_random_float(&pixel_seed);                      /* updateSeed */
            }
            r[i] = acc.x; g[i] = acc.y; b[i] = acc.z;
            sa[i] = pixel_seed.a; sb[i] = pixel_seed.b; sc[i] = pixel_seed.c; sctr[i] = pixel_seed.counter;
        }
        free(stack);
    }
    if (dropped) *dropped = n_dropped;
    if (longest_lineage) *longest_lineage = longest;
    if (truncated) *truncated = n_truncated;
    return live_total;
}

int ora_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
