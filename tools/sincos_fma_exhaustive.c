// exhaustive: FMA form of the portable sin/cos against the separate-multiply-add form, every binary32 with |x| <= 6.2831860
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline void old_sc(double x, double *s, double *c) {
    static const double INVPIO2 = 0x1.45f306dc9c883p-1, PIO2_1 = 0x1.921fb54400000p+0, PIO2_1T = 0x1.0b4611a626331p-34;
    double y = x * INVPIO2;
    int k = (int)(y + (y >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = (x - kd * PIO2_1) - kd * PIO2_1T;
    double z = r * r;
    double ps = -0x1.ae7f3e733b81fp-41 + z * 0x1.952c77030ad4ap-49;
    ps = 0x1.6124613a86d09p-33 + z * ps; ps = -0x1.ae64567f544e4p-26 + z * ps; ps = 0x1.71de3a556c734p-19 + z * ps;
    ps = -0x1.a01a01a01a01ap-13 + z * ps; ps = 0x1.1111111111111p-7 + z * ps; ps = -0x1.5555555555555p-3 + z * ps;
    double sr = r + r * (z * ps);
    double pc = -0x1.93974a8c07c9dp-37 + z * 0x1.ae7f3e733b81fp-45;
    pc = 0x1.1eed8eff8d898p-29 + z * pc; pc = -0x1.27e4fb7789f5cp-22 + z * pc; pc = 0x1.a01a01a01a01ap-16 + z * pc;
    pc = -0x1.6c16c16c16c17p-10 + z * pc; pc = 0x1.5555555555555p-5 + z * pc; pc = -0x1.0000000000000p-1 + z * pc;
    double cr = 1.0 + z * pc;
    switch (k & 3) { case 0: *s = sr; *c = cr; break; case 1: *s = cr; *c = -sr; break; case 2: *s = -sr; *c = -cr; break; default: *s = -cr; *c = sr; }
}
static inline void new_sc(double x, double *s, double *c) {
    static const double INVPIO2 = 0x1.45f306dc9c883p-1, PIO2_1 = 0x1.921fb54400000p+0, PIO2_1T = 0x1.0b4611a626331p-34;
    double y = x * INVPIO2;
    int k = (int)(y + (y >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = fma(-kd, PIO2_1T, fma(-kd, PIO2_1, x));
    double z = r * r;
    double ps = -0x1.ae7f3e733b81fp-41 + z * 0x1.952c77030ad4ap-49;
    ps = fma(z, ps, 0x1.6124613a86d09p-33); ps = fma(z, ps, -0x1.ae64567f544e4p-26); ps = fma(z, ps, 0x1.71de3a556c734p-19);
    ps = fma(z, ps, -0x1.a01a01a01a01ap-13); ps = fma(z, ps, 0x1.1111111111111p-7); ps = fma(z, ps, -0x1.5555555555555p-3);
    double sr = fma(r, z * ps, r);
    double pc = -0x1.93974a8c07c9dp-37 + z * 0x1.ae7f3e733b81fp-45;
    pc = fma(z, pc, 0x1.1eed8eff8d898p-29); pc = fma(z, pc, -0x1.27e4fb7789f5cp-22); pc = fma(z, pc, 0x1.a01a01a01a01ap-16);
    pc = fma(z, pc, -0x1.6c16c16c16c17p-10); pc = fma(z, pc, 0x1.5555555555555p-5); pc = fma(z, pc, -0x1.0000000000000p-1);
    double cr = fma(z, pc, 1.0);
    switch (k & 3) { case 0: *s = sr; *c = cr; break; case 1: *s = cr; *c = -sr; break; case 2: *s = -sr; *c = -cr; break; default: *s = -cr; *c = sr; }
}
int main(void) {
    const float top = 6.2831860f; uint32_t ut; memcpy(&ut, &top, 4);
    unsigned long long bad = 0, n = 0;
#pragma omp parallel for reduction(+:bad,n) schedule(dynamic, 1 << 20)
    for (long long u = 0; u <= (long long)ut; u++) {
        for (int sg = 0; sg < 2; sg++) {
            uint32_t b = (uint32_t)u | (sg ? 0x80000000u : 0u); float xf; memcpy(&xf, &b, 4);
            double s0, c0, s1, c1; old_sc((double)xf, &s0, &c0); new_sc((double)xf, &s1, &c1);
            float a0 = (float)s0, a1 = (float)s1, b0 = (float)c0, b1 = (float)c1;
            if (memcmp(&a0, &a1, 4) || memcmp(&b0, &b1, 4)) {
                bad++;
                if (bad < 20) printf("x=%a old s=%a c=%a new s=%a c=%a (double old %a %a new %a %a)\n", xf, a0, b0, a1, b1, s0, c0, s1, c1);
            }
            n++;
        }
    }
    printf("%llu arguments, %llu differ\n", n, bad);
    return bad != 0;
}
