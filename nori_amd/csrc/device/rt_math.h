/*
 * rt_math.h -- the three transcendental functions of the render path, with SPECIFIED results.
 *
 * The reference calls std::sin / std::cos (Warp::squareTo*, via sincosf), std::log and std::exp (Beckmann) --
 * whose last bit C++ does not define: glibc, ocml and a hand-vectorised libm all differ by an ulp here and
 * there, and one ulp in a sampled direction is enough to flip a hit / miss decision a few bounces later.
 * To make device and oracle radiance comparable bit for bit, both evaluate these functions by the same
 * specification, each with its own implementation of it (this file for the device and
 * the emulation harness, oracle/oracle_libm.h for the oracle):
 *
 *   all arithmetic in IEEE binary64, round-to-nearest, NO fused multiply-add, in the order written here;
 *   the binary64 result (accurate to ~1e-16 relative) is rounded once to binary32.
 *
 *   sincos(x), x >= 0 :  k = floor(x * 2/pi + 1/2);  r = (x - k * PIO2_HI) - k * PIO2_LO   (Cody-Waite, |r| <= pi/4)
 *                        sin r = r + r^3 (S1 + r^2 (S2 + ... S6)),  cos r = 1 - r^2/2 + r^4 (C1 + r^2 (C2 + ... C6))
 *                        (the fdlibm kernel polynomials), quadrant k mod 4 selects / negates
 *   log(x),  x > 0    :  x = m 2^e, m in [sqrt(1/2), sqrt(2));  s = (m - 1) / (m + 1);
 *                        log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^22/23);  log x = e LN2_HI + (log m + e LN2_LO)
 *   exp(x)            :  k = floor(x / ln 2 + 1/2);  r = (x - k LN2_HI) - k LN2_LO;
 *                        exp r = sum_{n <= 13} r^n / n! (Horner);  exp x = exp r * 2^k;  x < -104 -> 0, x > 88.8 -> inf
 *
 * A result differs from the correctly rounded one only where the true value lies within ~1e-9 ulp of a
 * rounding boundary; against glibc it differs in the last bit for a few percent of the arguments.
 */
#pragma once
#include "rt_types.h"

namespace nrt {

NORI_HD double dm_from_bits(uint64_t u) { return __builtin_bit_cast(double, u); }
NORI_HD uint64_t dm_bits(double d) { return __builtin_bit_cast(uint64_t, d); }

NORI_HD void det_sincosf(float xf, float *sn, float *cs) {
    const double x = (double) xf;
    const double kd = __builtin_floor(x * 0.63661977236758134308 + 0.5);
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double s = r + (r * z) * ps;
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double c = (1.0 - 0.5 * z) + (z * z) * pc;
    const int q = (int) ((long long) kd & 3ll);
    const double so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    *sn = (float) ((q & 2) ? -so : so);
    *cs = (float) (((q + 1) & 2) ? -co : co);
}

NORI_HD float det_logf(float xf) {
    if (!(xf > 0.0f)) return xf == 0.0f ? -kInf : (xf != xf ? xf : u2f(0x7fc00000u));
    if (!(xf < kInf)) return xf;
    const double x = (double) xf;                       /* exact; every positive float is a normal double */
    const uint64_t b = dm_bits(x);
    int e = (int) ((b >> 52) & 0x7ffull) - 1023;
    double m = dm_from_bits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);      /* [1, 2) */
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = 1.0 / 21.0 + z * p; p = 1.0 / 19.0 + z * p; p = 1.0 / 17.0 + z * p; p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p; p = 1.0 / 11.0 + z * p; p = 1.0 / 9.0 + z * p; p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p; p = 1.0 / 3.0 + z * p; p = 1.0 + z * p;
    const double lm = (2.0 * s) * p;
    const double ed = (double) e;
    return (float) (ed * 6.93147180369123816490e-01 + (lm + ed * 1.90821492927058770002e-10));
}

NORI_HD float det_expf(float xf) {
    if (xf != xf) return xf;
    if (xf > 88.8f) return kInf;
    if (xf < -104.0f) return 0.0f;
    const double x = (double) xf;
    const double kd = __builtin_floor(x * 1.44269504088896338700e+00 + 0.5);
    const double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;                     /* 1/13! */
    p = 1.0 / 479001600.0 + r * p; p = 1.0 / 39916800.0 + r * p; p = 1.0 / 3628800.0 + r * p; p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p; p = 1.0 / 5040.0 + r * p; p = 1.0 / 720.0 + r * p; p = 1.0 / 120.0 + r * p;
    p = 1.0 / 24.0 + r * p; p = 1.0 / 6.0 + r * p; p = 0.5 + r * p; p = 1.0 + r * p; p = 1.0 + r * p;
    const int k = (int) kd;                            /* -151 .. 129: 2^k is a normal double */
    return (float) (p * dm_from_bits((uint64_t) (k + 1023) << 52));
}

} // namespace nrt
