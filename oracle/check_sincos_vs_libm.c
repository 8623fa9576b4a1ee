/*
 * check_sincos_vs_libm -- pins ora_sinf / ora_cosf (the restated glibc algorithm the
 * oracle and the HIP kernel both use) against the libm of the machine it runs on.
 * TEST INFRASTRUCTURE.  usage: check_sincos_vs_libm [lo_bits hi_bits [stride]]
 * Walks every binary32 bit pattern in [lo_bits, hi_bits] (both signs) with the given
 * stride and prints the number of bitwise mismatches.  Exit status 0 iff none.
 * Default: every float with |x| < 4 (the hot path's range is |x| <= pi/2) -- 2.16e9
 * patterns -- plus a strided sweep of the rest up to +inf.
 */
#include "pt_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static long sweep(uint32_t lo, uint32_t hi, uint32_t stride, long *count)
{
    long bad = 0, n = 0;
#pragma omp parallel for reduction(+:bad,n) schedule(static)
    for (int64_t u = lo; u <= (int64_t)hi; u += stride) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            float x = from_bits((uint32_t)u | ((uint32_t)sgn << 31));
            float s0 = sinf(x), s1 = ora_sinf(x), c0 = cosf(x), c1 = ora_cosf(x);
            int sbad = bits(s0) != bits(s1) && !(isnan(s0) && isnan(s1));
            int cbad = bits(c0) != bits(c1) && !(isnan(c0) && isnan(c1));
            if (sbad || cbad) {
                ++bad;
                if (bad < 5)
                    fprintf(stderr, "mismatch x=%a sin %a/%a cos %a/%a\n", x, s0, s1, c0, c1);
            }
            ++n;
        }
    }
    *count += n;
    return bad;
}

int main(int argc, char **argv)
{
    long n = 0, bad = 0;
    if (argc >= 3) {
        uint32_t lo = (uint32_t)strtoul(argv[1], 0, 0), hi = (uint32_t)strtoul(argv[2], 0, 0);
        uint32_t stride = argc >= 4 ? (uint32_t)strtoul(argv[3], 0, 0) : 1;
        bad = sweep(lo, hi, stride, &n);
    } else {
        bad += sweep(0x00000000u, 0x407fffffu, 1, &n);      /* |x| < 4, exhaustive */
        bad += sweep(0x40800000u, 0x7f800000u, 61, &n);     /* the rest, strided   */
    }
    printf("{\"checked\": %ld, \"mismatches\": %ld}\n", n, bad);
    return bad != 0;
}
