/*
 * oracle_libm.h -- TEST INFRASTRUCTURE (part of the CPU oracle; never linked into the product).
 *
 * sin/cos, log and exp as the oracle evaluates them.  The reference calls std::sin / std::cos / std::log /
 * std::exp (src/warp.cpp stubs, src/common.cpp:225-235, the Beckmann formulas of SURVEY 8c), whose last bit
 * is unspecified by C++ and varies between libm builds (glibc selects FMA / non-FMA variants per CPU).  The
 * oracle pins them to ONE specification so that its radiance can be compared bit for bit:
 *
 *   IEEE binary64 arithmetic, round-to-nearest, no fused multiply-add, operations in the order written;
 *   binary64 result -> one rounding to binary32.
 *   sincos : k = floor(x * 2/pi + 1/2); r = (x - k PIO2_HI) - k PIO2_LO; fdlibm kernel polynomials S1..S6 / C1..C6
 *   log    : x = m 2^e with m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1); log m = 2 s sum_{n<=11} s^(2n) / (2n+1)
 *   exp    : k = floor(x / ln2 + 1/2); r = (x - k LN2_HI) - k LN2_LO; exp r = sum_{n<=13} r^n / n!; scale by 2^k
 *
 * The device implements the same specification on its own (nori_amd/csrc/device/rt_math.h); this file includes
 * nothing from the device tree.  tests/test_oracle_goldens.py checks these against glibc (<= 1 ulp) and the
 * reference's warp / microfacet goldens still pin the oracle that uses them.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#if defined(ORACLE_LIBM_GLIBC)
/* liboracle_glibc.so: the reference's literal calls (std::sin / std::cos / std::log / std::exp of the host's libm).  Not bit
   comparable with anything -- it is the independent witness: renders of the pinned-specification oracle and of the device
   must agree with it within the image tolerance (tests: test_specified_libm_vs_host_libm_renders, test_gpu_parity), so
   that a mistake shared by the two implementations of the specification cannot pass unnoticed. */
namespace oracle_libm {
inline void sincos(float x, float *s, float *c) { *s = std::sin(x); *c = std::cos(x); }
inline float log(float x) { return std::log(x); }
inline float exp(float x) { return std::exp(x); }
} // namespace oracle_libm
#else
namespace oracle_libm {

inline double horner(const double *c, int n, double z) {      /* c[0] + z (c[1] + z (... c[n-1])) */
    double p = c[n - 1];
    for (int i = n - 2; i >= 0; --i) p = c[i] + z * p;
    return p;
}

inline void sincos(float xf, float *sinOut, float *cosOut) {
    static const double S[6] = {-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                                2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10};
    static const double C[6] = {4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                                -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
    const double x = xf;
    const double k = std::floor(x * 0.63661977236758134308 + 0.5);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double sinr = r + (r * z) * horner(S, 6, z);
    const double cosr = (1.0 - 0.5 * z) + (z * z) * horner(C, 6, z);
    double s, c;
    switch ((int) ((long long) k & 3ll)) {
    case 0: s = sinr; c = cosr; break;
    case 1: s = cosr; c = -sinr; break;
    case 2: s = -sinr; c = -cosr; break;
    default: s = -cosr; c = sinr; break;
    }
    *sinOut = (float) s; *cosOut = (float) c;
}

inline float log(float xf) {
    if (!(xf > 0.0f)) return xf == 0.0f ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::quiet_NaN();
    if (std::isinf(xf)) return xf;
    static const double L[12] = {1.0, 1.0 / 3.0, 1.0 / 5.0, 1.0 / 7.0, 1.0 / 9.0, 1.0 / 11.0, 1.0 / 13.0, 1.0 / 15.0,
                                 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0, 1.0 / 23.0};
    const double x = xf;
    uint64_t bits; std::memcpy(&bits, &x, 8);
    int e = (int) ((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m; std::memcpy(&m, &bits, 8);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double logm = (2.0 * s) * horner(L, 12, s * s);
    const double ed = e;
    return (float) (ed * 6.93147180369123816490e-01 + (logm + ed * 1.90821492927058770002e-10));
}

inline float exp(float xf) {
    if (std::isnan(xf)) return xf;
    if (xf > 88.8f) return std::numeric_limits<float>::infinity();
    if (xf < -104.0f) return 0.0f;
    static const double F[14] = {1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0,
                                 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0};
    const double x = xf;
    const double k = std::floor(x * 1.44269504088896338700e+00 + 0.5);
    const double r = (x - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    const uint64_t bits = (uint64_t) ((int) k + 1023) << 52;
    double scale; std::memcpy(&scale, &bits, 8);
    return (float) (horner(F, 14, r) * scale);
}

} // namespace oracle_libm
#endif
