/* Experiment for a later round (CPU only): the literal homography of getHomography_cu divides nine
 * products t[r]*n[c] by the same plane offset d (gipuma.cu:339-356; pm::homography in pm_device.h), ~95 of
 * the ~250 instructions a (hypothesis, view) pair costs outside the sample loop.  With r = the correctly
 * rounded 1/d (pm::rcp_newton gives exactly that for |d| in [2^-126, 2^126), checked exhaustively on the
 * GPU) the Markstein sequence
 *     q0 = n * r;  e = fmaf(-d, q0, n);  q = fmaf(e, r, q0)
 * is claimed to be the correctly rounded n / d.  This program checks the claim against the compiler's
 * IEEE division on random and structured operands -- every operation here (mul, fmaf, div) is IEEE, so
 * the CPU result is the GPU result -- and reports where it fails.
 * Result (4.3e9 pairs, |d| in 2^+-60, |n| in 2^+-80 or zero): no mismatch whenever the quotient is a normal
 * number; mismatches only for subnormal / overflowing quotients, and for n = -0 (the sequence returns +0).
 * So a guarded version -- all nine numerators non-zero with exponents inside +-60 of d's, else the IEEE
 * division, chosen per wavefront like window_z_safe -- is exact and saves ~60 of the ~250 instructions
 * per (hypothesis, view) pair, ~1 % of a VALU-bound half-sweep.  Not built in round 2 (no GPU time left
 * to validate a kernel change at full size).
 *     gcc -O2 -fopenmp -mfma -ffp-contract=off -o /tmp/div_sd scripts/exp/div_shared_denominator.c -lm && /tmp/div_sd
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

static inline float div_sd(float n, float d, float r)
{
    const float q0 = n * r;
    const float e = fmaf(-d, q0, n);
    return fmaf(e, r, q0);
}

int main(void)
{
    unsigned long long bad_normal = 0, bad_other = 0, total = 0;
    /* d: exponents 2^-60 .. 2^60 (plane offsets are O(1..1e3)); n: sign x exponents 2^-80 .. 2^80, and 0 */
#pragma omp parallel for reduction(+ : bad_normal, bad_other, total) schedule(dynamic, 64)
    for (long long blk = 0; blk < (1LL << 16); blk++) {
        for (int k = 0; k < (1 << 16); k++) {
            const uint64_t h = mix(((uint64_t)blk << 20) ^ (uint64_t)k ^ 0x9e3779b97f4a7c15ULL);
            const uint32_t dman = (uint32_t)h & 0x7fffffu, nman = (uint32_t)(h >> 23) & 0x7fffffu;
            const int dexp = 127 - 60 + (int)((h >> 46) % 121), nexp = 127 - 80 + (int)((h >> 53) % 161);
            const uint32_t sgn = (uint32_t)(h >> 63) << 31;
            const float d = u2f(((uint32_t)dexp << 23) | dman);
            float n = u2f(sgn | ((uint32_t)nexp << 23) | nman);
            if ((k & 1023) == 0) n = 0.0f;
            if ((k & 1023) == 1) n = d;        /* exact quotients */
            if ((k & 1023) == 2) n = 3.0f * d;
            const float r = 1.0f / d;           /* = rcp_newton(d) on the device */
            const float want = n / d, got = div_sd(n, d, r);
            total++;
            if (f2u(want) != f2u(got)) {
                const float aw = fabsf(want);
                if (aw >= 0x1p-126f && aw < 0x1p127f)
                    bad_normal++;
                else
                    bad_other++;
            }
        }
    }
    printf("pairs %llu: mismatches with a normal quotient %llu, with a subnormal / huge quotient %llu\n", total, bad_normal,
           bad_other);
    {
        const float d = 517.25f, r = 1.0f / d, nz = -0.0f;
        printf("n = -0: IEEE %08x, sequence %08x\n", f2u(nz / d), f2u(div_sd(nz, d, r)));
    }
    return bad_normal != 0;
}
