// pt_sincos_fast.h -- sin and cos of a binary32 argument in [0, 2 pi] WITHOUT binary64 arithmetic, with a certificate.
//
// The hemisphere sampler (calculateRandomDirectionInHemisphere, src/interactions.h:11-43) takes sin and cos of
// around = u * TWO_PI.  The path tracer's results must be those of sincos_own (pt_device.h: Cody-Waite + Taylor in binary64,
// one rounding to binary32 -- what the CPU checker runs), and on gfx950 that routine is ~155 binary64 instructions per wave,
// each issuing at 2.7 cycles against 1.7 for a binary32 one: a sixth of the diffuse bounce.
//
// sincos_fast computes hi + lo ~ sin x (and cos x) in binary32 pairs, about 2^-41 absolute:
//   * k = round(x * 64/pi), d = x - k * pi/64 as a pair (pi/64 in three parts; k * P1, k * P2 exact; fused multiply-adds keep
//     the rounding errors), |d| <= pi/128;
//   * sin / cos of k * pi/64 from a table of pairs (48 bits; exactly 0 and +-1 at multiples of pi/2, so results near a zero of
//     sin or cos keep their RELATIVE accuracy);
//   * sin x = S cos d + C sin d, cos x = C cos d - S sin d with cos d - 1 and sin d - d by short polynomials, the two large
//     products by exact two-products, the sums by exact two-sums;
//   * ROUNDING TEST (Ziv): the result is accepted only if hi + (lo + E) and hi + (lo - E) round to the SAME binary32 -- then every
//     value within E of hi + lo rounds there, so does the binary64 routine's value, and the two routines agree.  Otherwise the
//     function says "not certain" and the caller runs sincos_own (a wave does that for well under 1 % of its tiles).
// E = 2^-39 is 2.25 x the largest |hi + lo - binary64 value| over all of them (8.1e-13); and the claim that matters -- accepted results are bit-identical to
// sincos_own's for EVERY binary32 in [0, 2 pi] -- is checked exhaustively, all 1 086 918 620 of them, by
// tests/test_own_libm.py::test_fast_sincos_exhaustive (this very header compiled for the host, against the checker's routine).
// The same IEEE operations (add, multiply, fused multiply-add, conversions; no contraction, no reassociation) run on the device.
#pragma once

#ifndef PT_SC_FN
#define PT_SC_FN static inline
#endif
#ifndef PT_SC_TAB_DECL
#define PT_SC_TAB_DECL static const
#endif
#ifndef PT_SC_E
#define PT_SC_E 0x1.0p-39f
#endif
#include "pt_sincos_tab.h"

// returns 1 and the correctly decided results, or 0 (outside [0, 2 pi], NaN, or a rounding too close to call)
// dbg (host checks only, may be null): hi and lo of sin and cos, whether accepted or not
PT_SC_FN int sincos_fast(float x, float *s, float *c, float *dbg) {
    if (!(x >= 0.0f && x <= 0x1.921fb6p+2f)) return 0;
    const float kf = (float)(int)(x * PT_SC_64_OVER_PI + 0.5f);
    const int k = (int)kf;
    // d = x - k * pi/64 = dh + dl
    const float d1 = __builtin_fmaf(-kf, PT_SC_P1, x);              // exact
    const float dh = __builtin_fmaf(-kf, PT_SC_P2, d1);
    const float dr = __builtin_fmaf(-kf, PT_SC_P2, d1 - dh);         // the rounding error of dh, exact
    const float dl = __builtin_fmaf(-kf, PT_SC_P3, dr);
    const float Sh = pt_sc_tab[k][0], Sl = pt_sc_tab[k][1], Ch = pt_sc_tab[k][2], Cl = pt_sc_tab[k][3];
    // d^2 as a pair; sin d - d; cos d - 1 as a pair
    const float d2h = dh * dh;
    float d2l = __builtin_fmaf(dh, dh, -d2h);
    d2l = __builtin_fmaf(dh + dh, dl, d2l);
    const float sd3 = (d2h * dh) * __builtin_fmaf(d2h, 0x1.111112p-7f, -0x1.555556p-3f);
    const float cmh = -0.5f * d2h;
    const float cml = __builtin_fmaf(-0.5f, d2l, (d2h * d2h) * __builtin_fmaf(d2h, -0x1.6c16c2p-10f, 0x1.555556p-5f));
    // sin x = Sh + [C d + S (cos d - 1) + small terms]
    const float p1 = Ch * dh, e1 = __builtin_fmaf(Ch, dh, -p1);
    const float p2 = Sh * cmh, e2 = __builtin_fmaf(Sh, cmh, -p2);
    float sm = (e1 + e2) + Sl;
    sm = __builtin_fmaf(Sh, cml, sm);
    sm = __builtin_fmaf(Sl, cmh, sm);
    sm = __builtin_fmaf(Ch, dl, sm);
    sm = __builtin_fmaf(Cl, dh, sm);
    sm = __builtin_fmaf(Ch, sd3, sm);
    const float a = Sh + p1, ea = p1 - (a - Sh);                     // exact: |Sh| >= |p1| or Sh == 0
    const float b = a + p2, eb = p2 - (b - a);
    const float slo = (ea + eb) + sm;
    // cos x = Ch + [-S d + C (cos d - 1) + small terms]
    const float q1 = -(Sh * dh), f1 = __builtin_fmaf(-Sh, dh, -q1);
    const float q2 = Ch * cmh, f2 = __builtin_fmaf(Ch, cmh, -q2);
    float cm = (f1 + f2) + Cl;
    cm = __builtin_fmaf(Ch, cml, cm);
    cm = __builtin_fmaf(Cl, cmh, cm);
    cm = __builtin_fmaf(-Sh, dl, cm);
    cm = __builtin_fmaf(-Sl, dh, cm);
    cm = __builtin_fmaf(-Sh, sd3, cm);
    const float a2 = Ch + q1, ea2 = q1 - (a2 - Ch);
    const float b2 = a2 + q2, eb2 = q2 - (b2 - a2);
    const float clo = (ea2 + eb2) + cm;
    if (dbg) { dbg[0] = b; dbg[1] = slo; dbg[2] = b2; dbg[3] = clo; }
    const float s1 = b + (slo + PT_SC_E), s2 = b + (slo - PT_SC_E);
    const float c1 = b2 + (clo + PT_SC_E), c2 = b2 + (clo - PT_SC_E);
    *s = s1;
    *c = c1;
    return s1 == s2 && c1 == c2;
}
