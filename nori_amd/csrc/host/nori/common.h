/*
 * nori/common.h -- host-side basics: exception, string helpers, small value
 * types.  Same names and semantics as the reference's include/nori/common.h,
 * vector.h, color.h, transform.h, ray.h -- re-authored without Eigen /
 * tinyformat (both are empty submodules in the reference snapshot).
 *
 * The host never evaluates anything per sample: these types only carry scene
 * parameters to the device (via nori_scene_desc) and results back.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define NORI_NAMESPACE_BEGIN namespace nori {
#define NORI_NAMESPACE_END }

/* include/nori/common.h:38-48 */
#define Epsilon 1e-4f
#undef M_PI
#define M_PI 3.14159265358979323846f
#define INV_PI 0.31830988618379067154f
#define INV_TWOPI 0.15915494309189533577f
#define INV_FOURPI 0.07957747154594766788f

NORI_NAMESPACE_BEGIN

using std::cerr;
using std::cout;
using std::endl;

/* ---- printf-style formatting with %s working for any streamable type
   (stands in for tinyformat, which NoriException used) ---- */
namespace detail {
inline void formatImpl(std::ostringstream &os, const char *fmt) {
    for (; *fmt; ++fmt) {
        if (fmt[0] == '%' && fmt[1] == '%') ++fmt;
        os << *fmt;
    }
}
template <typename T, typename... Rest>
void formatImpl(std::ostringstream &os, const char *fmt, const T &value, const Rest &...rest) {
    for (; *fmt; ++fmt) {
        if (*fmt != '%') { os << *fmt; continue; }
        if (fmt[1] == '%') { os << '%'; ++fmt; continue; }
        ++fmt;
        int precision = -1;
        while (*fmt && std::strchr("-+ #0123456789", *fmt)) ++fmt;
        if (*fmt == '.') { ++fmt; precision = 0; while (*fmt >= '0' && *fmt <= '9') precision = precision * 10 + (*fmt++ - '0'); }
        while (*fmt && std::strchr("lhzjt", *fmt)) ++fmt;
        const char spec = *fmt;
        std::ios::fmtflags flags = os.flags();
        std::streamsize oldPrec = os.precision();
        if (spec == 'f' || spec == 'e' || spec == 'g') {
            if (spec == 'f') os << std::fixed;
            os.precision(precision >= 0 ? precision : 6);
        }
        os << value;
        os.flags(flags); os.precision(oldPrec);
        if (*fmt) ++fmt;
        formatImpl(os, fmt, rest...);
        return;
    }
}
} // namespace detail

template <typename... Args> std::string format(const char *fmt, const Args &...args) {
    std::ostringstream os;
    detail::formatImpl(os, fmt, args...);
    return os.str();
}

/* include/nori/common.h:135-140 */
class NoriException : public std::runtime_error {
public:
    template <typename... Args> NoriException(const char *fmt, const Args &...args)
        : std::runtime_error(format(fmt, args...)) {}
};

/* src/common.cpp:27-122 */
std::string indent(const std::string &string, int amount = 2);
bool endsWith(const std::string &value, const std::string &ending);
std::string toLower(const std::string &value);
bool toBool(const std::string &str);
int toInt(const std::string &str);
unsigned int toUInt(const std::string &str);
float toFloat(const std::string &str);
std::vector<std::string> tokenize(const std::string &s, const std::string &delim = ", ", bool includeEmpty = false);
std::string timeString(double time, bool precise = false);
std::string memString(size_t size, bool precise = false);

inline float degToRad(float value) { return value * (M_PI / 180.0f); }
inline float radToDeg(float value) { return value * (180.0f / M_PI); }
inline float clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- value types ---- */
struct Vector3f {
    float v[3];
    Vector3f() : v{0, 0, 0} {}
    explicit Vector3f(float s) : v{s, s, s} {}
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
    float &x() { return v[0]; } float &y() { return v[1]; } float &z() { return v[2]; }
    float operator[](int i) const { return v[i]; } float &operator[](int i) { return v[i]; }
    Vector3f operator+(const Vector3f &o) const { return Vector3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vector3f operator-(const Vector3f &o) const { return Vector3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vector3f operator-() const { return Vector3f(-v[0], -v[1], -v[2]); }
    Vector3f operator*(float s) const { return Vector3f(v[0] * s, v[1] * s, v[2] * s); }
    Vector3f operator/(float s) const { return Vector3f(v[0] / s, v[1] / s, v[2] / s); }
    float dot(const Vector3f &o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
    Vector3f cross(const Vector3f &o) const {
        return Vector3f(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
    float squaredNorm() const { return dot(*this); }
    float norm() const { return std::sqrt(squaredNorm()); }
    Vector3f normalized() const { float z = squaredNorm(); return z > 0 ? *this / std::sqrt(z) : *this; }
    float maxCoeff() const { return std::max(v[0], std::max(v[1], v[2])); }
    std::string toString() const { return format("[%f, %f, %f]", v[0], v[1], v[2]); }
};
typedef Vector3f Point3f;
typedef Vector3f Normal3f;

struct Point2f {
    float v[2];
    Point2f() : v{0, 0} {}
    Point2f(float x, float y) : v{x, y} {}
    float x() const { return v[0]; } float y() const { return v[1]; }
    float &x() { return v[0]; } float &y() { return v[1]; }
    std::string toString() const { return format("[%f, %f]", v[0], v[1]); }
};
typedef Point2f Vector2f;

struct Vector2i {
    int v[2];
    Vector2i() : v{0, 0} {}
    Vector2i(int x, int y) : v{x, y} {}
    int x() const { return v[0]; } int y() const { return v[1]; }
    int &x() { return v[0]; } int &y() { return v[1]; }
    std::string toString() const { return format("[%i, %i]", v[0], v[1]); }
};
typedef Vector2i Point2i;

/* include/nori/color.h:16-70, src/common.cpp:166-209 */
struct Color3f {
    float v[3];
    Color3f(float value = 0.f) : v{value, value, value} {}
    Color3f(float r, float g, float b) : v{r, g, b} {}
    float r() const { return v[0]; } float g() const { return v[1]; } float b() const { return v[2]; }
    float operator[](int i) const { return v[i]; } float &operator[](int i) { return v[i]; }
    float maxCoeff() const { return std::max(v[0], std::max(v[1], v[2])); }
    Color3f operator*(const Color3f &o) const { return Color3f(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
    Color3f operator*(float s) const { return Color3f(v[0] * s, v[1] * s, v[2] * s); }
    Color3f operator+(const Color3f &o) const { return Color3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    bool isValid() const;
    Color3f toLinearRGB() const;
    Color3f toSRGB() const;
    float getLuminance() const;
    std::string toString() const { return format("[%f, %f, %f]", v[0], v[1], v[2]); }
};

/* include/nori/transform.h:22-86: a matrix and its inverse, row-major */
struct Transform {
    float m[16], inv[16];
    Transform();
    explicit Transform(const float *trafo);                  /* inverse computed */
    Transform(const float *trafo, const float *inverse);
    Transform inverse() const { return Transform(inv, m); }
    Transform operator*(const Transform &t) const;
    Vector3f applyVector(const Vector3f &v) const;           /* operator*(Vector3f) */
    Normal3f applyNormal(const Normal3f &n) const;           /* operator*(Normal3f) */
    Point3f applyPoint(const Point3f &p) const;              /* operator*(Point3f)  */
    std::string toString() const;
};
void mat4Identity(float *m);
void mat4Mul(const float *a, const float *b, float *out);
bool mat4Inverse(const float *a, float *out);

/* include/nori/ray.h:25-68 */
struct Ray3f {
    Point3f o;
    Vector3f d, dRcp;
    float mint, maxt;
    Ray3f() : mint(Epsilon), maxt(std::numeric_limits<float>::infinity()) {}
    Ray3f(const Point3f &o_, const Vector3f &d_) : o(o_), d(d_), mint(Epsilon),
        maxt(std::numeric_limits<float>::infinity()) { update(); }
    Ray3f(const Point3f &o_, const Vector3f &d_, float mint_, float maxt_) : o(o_), d(d_), mint(mint_), maxt(maxt_) { update(); }
    void update() { dRcp = Vector3f(1.0f / d.x(), 1.0f / d.y(), 1.0f / d.z()); }
    Point3f operator()(float t) const { return o + d * t; }
};

/* include/nori/common.h:179-183 */
enum EMeasure { EUnknownMeasure = 0, ESolidAngle, EDiscrete };

/* include/nori/frame.h -- the three axes; built on the device */
struct Frame {
    Vector3f s, t;
    Normal3f n;
    Vector3f toLocal(const Vector3f &v) const { return Vector3f(v.dot(s), v.dot(t), v.dot(n)); }
    Vector3f toWorld(const Vector3f &v) const { return (s * v.x() + t * v.y()) + n * v.z(); }
    static float cosTheta(const Vector3f &v) { return v.z(); }
};

/* src/common.cpp:225-235: spherical direction from angles */
Vector3f sphericalDirection(float theta, float phi);

/* Resource resolution: stands in for filesystem::resolver (src/common.cpp:160-163,
   src/main.cpp:190) -- search paths for files referenced by a scene. */
class FileResolver {
public:
    void prepend(const std::string &dir) { m_paths.insert(m_paths.begin(), dir); }
    std::string resolve(const std::string &name) const;
private:
    std::vector<std::string> m_paths;
};
FileResolver *getFileResolver();

/* include/nori/timer.h */
class Timer {
public:
    Timer();
    void reset();
    double elapsed() const;                 /* ms */
    std::string elapsedString(bool precise = false) const { return timeString(elapsed(), precise); }
private:
    double m_start;
};

NORI_NAMESPACE_END
