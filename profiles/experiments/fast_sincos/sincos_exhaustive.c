/* sincos_exhaustive.c -- every binary32 in [0, 2 pi] through the product's sincos_fast (mygpuraytracer_amd/csrc/pt_sincos_fast.h,
 * compiled here for the host) against the checker's binary64 routine (oracle/pt_oracle.c: o_own_sincosf): wherever the fast
 * routine accepts its result, the two must agree bit for bit.  Built and run by tests/test_own_libm.py.
 *   sincos_exhaustive [stride]     prints: inputs accepted mismatches max_abs_err(hi+lo vs binary64 value)                    */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mygpuraytracer_amd/csrc/pt_sincos_fast.h"

void o_own_sincosf(float x, float *s, float *c);
void o_own_sincos_value(double x, double *s, double *c);

int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    const float top = 0x1.921fb6p+2f;
    uint32_t last;
    memcpy(&last, &top, 4);
    long long n = 0, acc = 0, bad = 0;
    double worst = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : n, acc, bad) reduction(max : worst)
    for (long long u = 0; u <= (long long)last; u += stride) {
        const uint32_t bits = (uint32_t)u;
        float x, s, c, rs, rc, dbg[4];
        memcpy(&x, &bits, 4);
        const int ok = sincos_fast(x, &s, &c, dbg);
        o_own_sincosf(x, &rs, &rc);
        double vs, vc;
        o_own_sincos_value((double)x, &vs, &vc);
        const double es = fabs(((double)dbg[0] + (double)dbg[1]) - vs), ec = fabs(((double)dbg[2] + (double)dbg[3]) - vc);
        if (es > worst) worst = es;
        if (ec > worst) worst = ec;
        n++;
        if (ok) {
            acc++;
            if (memcmp(&s, &rs, 4) || memcmp(&c, &rc, 4)) bad++;
        }
    }
    printf("%lld %lld %lld %.6g\n", n, acc, bad, worst);
    /* how often the sampler's arguments are accepted: around = u * TWO_PI (src/interactions.h:16) for u = j / 2^24 */
    long long m = 0, macc = 0;
#pragma omp parallel for schedule(static) reduction(+ : m, macc)
    for (long long j = 0; j < (1 << 24); j += 3) {
        const float u = (float)j / 16777216.0f, x = u * 6.2831853071795864769252867665590057683943f;
        float s, c;
        m++;
        macc += sincos_fast(x, &s, &c, 0);
    }
    printf("sampler arguments: %lld of %lld accepted\n", macc, m);
    return bad ? 1 : 0;
}
