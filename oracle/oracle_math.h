/*
 * oracle_math.h -- value types of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by
 * or executed from the product (nori_amd/, libnori_hip.so, libnori_host.so);
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-float restatement of the reference's Eigen-based value types.  The
 * evaluation ORDER of every multi-term expression follows what Eigen 3.3+
 * generates for fixed-size float vectors without FMA contraction (the
 * reference's CMake build has no -march=native), because the order decides
 * the rounding:
 *   dot / squaredNorm of 3-vectors : a0*b0 + (a1*b1 + a2*b2)
 *       (Eigen redux_novec_unroller<.., 0, 3>: halves 1 | 2)
 *   cross                          : (a.y*b.z - a.z*b.y, ...)
 *   Matrix4f * Vector4f            : ((c0*x + c1*y) + c2*z) + c3*w per row
 *   normalized()                   : v / sqrt(squaredNorm) component-wise
 * Build with -ffp-contract=off.
 *
 * Reference: include/nori/vector.h, ray.h, bbox.h, frame.h, color.h,
 *            transform.h, dpdf.h, src/common.cpp.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "oracle_libm.h"

namespace oracle {

/* include/nori/common.h:38-48 */
static constexpr float Epsilon = 1e-4f;
static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInvTwoPi = 0.15915494309189533577f;
static constexpr float kInvFourPi = 0.07957747154594766788f;

struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float v) : x(v), y(v), z(v) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct Vec2 {
    float x, y;
    Vec2() : x(0), y(0) {}
    Vec2(float x_, float y_) : x(x_), y(y_) {}
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline Vec3 &operator+=(Vec3 &a, Vec3 b) { a = a + b; return a; }
inline Vec3 &operator*=(Vec3 &a, Vec3 b) { a = a * b; return a; }
inline Vec3 &operator*=(Vec3 &a, float s) { a = a * s; return a; }
inline Vec3 &operator/=(Vec3 &a, float s) { a = a / s; return a; }

inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float squaredNorm(Vec3 a) { return dot(a, a); }
inline float norm(Vec3 a) { return std::sqrt(squaredNorm(a)); }
/* Eigen MatrixBase::normalized(): divide by sqrt(squaredNorm) if > 0 */
inline Vec3 normalized(Vec3 a) {
    float z = squaredNorm(a);
    if (z > 0.0f) return a / std::sqrt(z);
    return a;
}
inline float maxCoeff(Vec3 a) { return std::max(a.x, std::max(a.y, a.z)); }

typedef Vec3 Color3;

/* src/common.cpp:198-209 */
inline bool isValidColor(Color3 c) {
    for (int i = 0; i < 3; ++i) {
        float v = c[i];
        if (v < 0 || !std::isfinite(v)) return false;
    }
    return true;
}
inline float luminance(Color3 c) {
    return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f;
}

inline float degToRad(float v) { return v * (kPi / 180.0f); }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* include/nori/ray.h:25-68 */
struct Ray {
    Vec3 o, d, dRcp;
    float mint, maxt;
    Ray() : mint(Epsilon), maxt(std::numeric_limits<float>::infinity()) {}
    Ray(Vec3 o_, Vec3 d_) : o(o_), d(d_), mint(Epsilon),
        maxt(std::numeric_limits<float>::infinity()) { update(); }
    Ray(Vec3 o_, Vec3 d_, float mint_, float maxt_) : o(o_), d(d_), mint(mint_), maxt(maxt_) { update(); }
    void update() { dRcp = Vec3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
};

/* include/nori/bbox.h */
struct BBox {
    Vec3 min, max;
    BBox() { reset(); }
    void reset() {
        min = Vec3(std::numeric_limits<float>::infinity());
        max = Vec3(-std::numeric_limits<float>::infinity());
    }
    void expandBy(Vec3 p) {
        min = Vec3(std::min(min.x, p.x), std::min(min.y, p.y), std::min(min.z, p.z));
        max = Vec3(std::max(max.x, p.x), std::max(max.y, p.y), std::max(max.z, p.z));
    }
    void expandBy(const BBox &b) { expandBy(b.min); expandBy(b.max); }
    Vec3 center() const { return (max + min) * 0.5f; }
    Vec3 extents() const { return max - min; }
    /* bbox.h:74-84 */
    float surfaceArea() const {
        Vec3 d = max - min;
        float result = 0;
        for (int i = 0; i < 3; ++i) {
            float term = 1;
            for (int j = 0; j < 3; ++j) {
                if (i == j) continue;
                term *= d[j];
            }
            result += term;
        }
        return 2.0f * result;
    }
    int largestAxis() const {
        Vec3 e = max - min;
        if (e.x >= e.y && e.x >= e.z) return 0;
        else if (e.y >= e.x && e.y >= e.z) return 1;
        return 2;
    }
    /* bbox.h:323-350, literal */
    bool rayIntersect(const Ray &ray) const {
        float nearT = -std::numeric_limits<float>::infinity();
        float farT = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 3; i++) {
            float origin = ray.o[i];
            float minVal = min[i], maxVal = max[i];
            if (ray.d[i] == 0) {
                if (origin < minVal || origin > maxVal) return false;
            } else {
                float t1 = (minVal - origin) * ray.dRcp[i];
                float t2 = (maxVal - origin) * ray.dRcp[i];
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return ray.mint <= farT && nearT <= ray.maxt;
    }
    /* bbox.h:353-380 */
    bool rayIntersect(const Ray &ray, float &nearT, float &farT) const {
        nearT = -std::numeric_limits<float>::infinity();
        farT = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 3; i++) {
            float origin = ray.o[i];
            float minVal = min[i], maxVal = max[i];
            if (ray.d[i] == 0) {
                if (origin < minVal || origin > maxVal) return false;
            } else {
                float t1 = (minVal - origin) * ray.dRcp[i];
                float t2 = (maxVal - origin) * ray.dRcp[i];
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return true;
    }
};

/* src/common.cpp:248-257 */
inline void coordinateSystem(const Vec3 &a, Vec3 &b, Vec3 &c) {
    if (std::abs(a.x) > std::abs(a.y)) {
        float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        c = Vec3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        c = Vec3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

/* include/nori/frame.h */
struct Frame {
    Vec3 s, t, n;
    Frame() {}
    explicit Frame(const Vec3 &n_) : n(n_) { coordinateSystem(n, s, t); }
    Vec3 toLocal(const Vec3 &v) const { return Vec3(dot(v, s), dot(v, t), dot(v, n)); }
    /* s * v.x + t * v.y + n * v.z, left-assoc */
    Vec3 toWorld(const Vec3 &v) const { return (s * v.x + t * v.y) + n * v.z; }
    static float cosTheta(const Vec3 &v) { return v.z; }
    static float sinTheta2(const Vec3 &v) { return 1.0f - v.z * v.z; }
    static float sinTheta(const Vec3 &v) {
        float temp = sinTheta2(v);
        if (temp <= 0.0f) return 0.0f;
        return std::sqrt(temp);
    }
    static float tanTheta(const Vec3 &v) {
        float temp = 1 - v.z * v.z;
        if (temp <= 0.0f) return 0.0f;
        return std::sqrt(temp) / v.z;
    }
};

/* src/common.cpp:225-235 */
inline Vec3 sphericalDirection(float theta, float phi) {
    float sinTheta, cosTheta, sinPhi, cosPhi;
    oracle_libm::sincos(theta, &sinTheta, &cosTheta);
    oracle_libm::sincos(phi, &sinPhi, &cosPhi);
    return Vec3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

/* src/common.cpp:259-288, literal */
inline float fresnel(float cosThetaI, float extIOR, float intIOR) {
    float etaI = extIOR, etaT = intIOR;
    if (extIOR == intIOR) return 0.0f;
    if (cosThetaI < 0.0f) {
        std::swap(etaI, etaT);
        cosThetaI = -cosThetaI;
    }
    float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1 - cosThetaI * cosThetaI);
    if (sinThetaTSqr > 1.0f) return 1.0f;
    float cosThetaT = std::sqrt(1.0f - sinThetaTSqr);
    float Rs = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    float Rp = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    return (Rs * Rs + Rp * Rp) / 2.0f;
}

/* Row-major 4x4; include/nori/transform.h:55-76 semantics */
struct Mat4 {
    float m[4][4];
    static Mat4 identity() {
        Mat4 r; std::memset(r.m, 0, sizeof(r.m));
        for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
        return r;
    }
    /* Point: Matrix4f * Vector4f(p,1) then divide by w (transform.h:65-68) */
    Vec3 point(const Vec3 &p) const {
        float r[4];
        for (int i = 0; i < 4; ++i)
            r[i] = ((m[i][0] * p.x + m[i][1] * p.y) + m[i][2] * p.z) + m[i][3] * 1.0f;
        return Vec3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
    }
    /* Vector: topLeftCorner<3,3>() * v, coefficient based (transform.h:55-57) */
    Vec3 vector(const Vec3 &v) const {
        return Vec3(m[0][0] * v.x + (m[0][1] * v.y + m[0][2] * v.z),
                    m[1][0] * v.x + (m[1][1] * v.y + m[1][2] * v.z),
                    m[2][0] * v.x + (m[2][1] * v.y + m[2][2] * v.z));
    }
};
inline Mat4 matmul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
/* Gauss-Jordan in double, rounded to float once (Eigen uses a cofactor SIMD
 * routine for Matrix4f; both are within an ulp or two -- parity unpinned). */
inline Mat4 inverse(const Mat4 &a) {
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { w[i][j] = a.m[i][j]; w[i][j + 4] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[piv][j]);
        double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = w[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    Mat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (float) w[i][j + 4];
    return r;
}

/* pcg32 -- ext/pcg32 is an empty submodule in the reference snapshot
 * (.gitmodules:16-18, wjakob/pcg32, unpinned).  Restated from the published
 * PCG-XSH-RR 64/32 algorithm (O'Neill 2014) as implemented by wjakob/pcg32:
 * call sites src/independent.cpp:37-53. */
struct Pcg32 {
    uint64_t state, inc;
    Pcg32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    void seed(uint64_t initstate, uint64_t initseq = 1) {
        state = 0U;
        inc = (initseq << 1u) | 1u;
        nextUInt();
        state += initstate;
        nextUInt();
    }
    uint32_t nextUInt() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t) (((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t) (oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    float nextFloat() {
        union { uint32_t u; float f; } x;
        x.u = (nextUInt() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
};

/* include/nori/dpdf.h, literal subset */
struct DiscretePDF {
    std::vector<float> cdf;
    float sum = 0, normalization = 0;
    bool normalizedFlag = false;
    DiscretePDF() { clear(); }
    void clear() { cdf.clear(); cdf.push_back(0.0f); normalizedFlag = false; }
    void append(float v) { cdf.push_back(cdf[cdf.size() - 1] + v); }
    size_t size() const { return cdf.size() - 1; }
    float operator[](size_t e) const { return cdf[e + 1] - cdf[e]; }
    float normalize() {
        sum = cdf[cdf.size() - 1];
        if (sum > 0) {
            normalization = 1.0f / sum;
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
            cdf[cdf.size() - 1] = 1.0f;
            normalizedFlag = true;
        } else {
            normalization = 0.0f;
        }
        return sum;
    }
    size_t sample(float v) const {
        std::vector<float>::const_iterator entry = std::lower_bound(cdf.begin(), cdf.end(), v);
        size_t index = (size_t) std::max((ptrdiff_t) 0, entry - cdf.begin() - 1);
        return std::min(index, cdf.size() - 2);
    }
};

} // namespace oracle
