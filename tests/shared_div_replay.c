/* Test-only CPU replay of SharedDiv::operator() (plaac_amd/csrc/kernels_windows_exact.hip.inc): the exact fma sequence
 *     q0 = RN(a * y);  r = fma(-d, q0, a);  q = fma(r, y, q0)        with y = RN(1 / d)
 * against the IEEE quotient a / d, for every denominator the window kernels tabulate (first level 1..41; second level
 * 41 + (820 - ml(ml+1)/2) + (820 - mr(mr+1)/2), ml, mr = 0..40) and a numerator set that stresses the final rounding:
 * random values over the tracks' ranges, quotients within a few ulp(a) of a rounding midpoint, exact quotients, the
 * smallest / largest magnitudes a window sum can take. Build: gcc -O2 -ffp-contract=off (fma() is libm's correctly
 * rounded fused multiply-add whether or not the CPU has the instruction). Returns the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd64(void) { /* splitmix64 */
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static int same(double a, double b) { return memcmp(&a, &b, 8) == 0; }

static int check(double a, double d, double y, double *first_a, double *first_d) {
    volatile double q0 = a * y;
    const double r = fma(-d, q0, a);
    const double q = fma(r, y, q0);
    volatile double want = a / d;
    if (same(q, want)) return 0;
    if (first_a && *first_d == 0.0) {
        *first_a = a;
        *first_d = d;
    }
    return 1;
}

/* per_den: numerators of each kind per denominator. out[0] = cases run, out[1] = denominators. */
long shared_div_replay(long per_den, double out[4]) {
    long bad = 0, cases = 0, ndens = 0;
    double fa = 0.0, fd = 0.0;
    int dens[41 + 41 * 41], nd = 0;
    for (int c = 1; c <= 41; ++c) dens[nd++] = c;
    for (int ml = 0; ml <= 40; ++ml)
        for (int mr = 0; mr <= 40; ++mr) dens[nd++] = 41 + (820 - ml * (ml + 1) / 2) + (820 - mr * (mr + 1) / 2);
    for (int k = 0; k < nd; ++k) {
        const double d = (double)dens[k];
        volatile double y = 1.0 / d; /* the reciprocal the context creation asserts on the device table */
        ++ndens;
        for (long i = 0; i < per_den; ++i) {
            /* (1) random numerators: sign, exponent 2^-30 .. 2^14, random mantissa */
            {
                const uint64_t u = rnd64();
                const int e = (int)(rnd64() % 45) - 30;
                double a = ldexp(1.0 + (double)(u >> 12) * 0x1p-52, e);
                if (u & 1) a = -a;
                bad += check(a, d, y, &fa, &fd);
                ++cases;
            }
            /* (2) hard cases: a = d * (m + 1/2) ulp-units, perturbed by -2..2 ulp(a): a/d next to a rounding midpoint */
            {
                const uint64_t m = (1ull << 52) | (rnd64() >> 12);
                const int e = (int)(rnd64() % 40) - 25;
                /* exact product d * (2m + 1) in long double (64-bit mantissa: d < 2^11, 2m+1 < 2^54 -> may round) */
                const long double prod = (long double)d * (long double)(2 * m + 1);
                double a = ldexp((double)prod, e - 53);
                for (int s = -2; s <= 2; ++s) {
                    double as = a;
                    for (int t = 0; t < (s < 0 ? -s : s); ++t) as = nextafter(as, s < 0 ? -INFINITY : INFINITY);
                    bad += check(as, d, y, &fa, &fd);
                    bad += check(-as, d, y, &fa, &fd);
                    cases += 2;
                }
            }
            /* (3) exact quotients and their neighbours */
            {
                const uint64_t m = (1ull << 40) | (rnd64() >> 24); /* 41-bit mantissa: d * m is exact */
                double a = ldexp(d * (double)m, (int)(rnd64() % 30) - 50);
                bad += check(a, d, y, &fa, &fd);
                bad += check(nextafter(a, INFINITY), d, y, &fa, &fd);
                bad += check(nextafter(a, -INFINITY), d, y, &fa, &fd);
                cases += 3;
            }
        }
        /* (4) fixed edge numerators (-0.0 is not one: every window sum starts from +0.0, plaac.java:2606, and
         * x + y is -0.0 only when both are; the sequence would return +0.0 for it) */
        const double edge[] = {0.0, 1.0, -1.0, 0x1p-60, 0x1.fffffffffffffp-40, 0x1.fffffffffffffp13, 41.0, 1681.0,
                               4.5 * 41, -4.5 * 41, 0.5, 0x1.0000000000001p0, 0x1.fffffffffffffp0};
        for (unsigned j = 0; j < sizeof edge / sizeof edge[0]; ++j) {
            bad += check(edge[j], d, y, &fa, &fd);
            ++cases;
        }
    }
    out[0] = (double)cases;
    out[1] = (double)ndens;
    out[2] = fa;
    out[3] = fd;
    return bad;
}
