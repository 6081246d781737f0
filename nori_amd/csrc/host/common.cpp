/*
 * common.cpp -- string helpers, colour conversions, transforms
 * (behaviour of the reference's src/common.cpp:27-246; re-authored).
 */
#include <nori/common.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iomanip>

NORI_NAMESPACE_BEGIN

std::string indent(const std::string &string, int amount) {
    std::string out, pad(amount, ' ');
    size_t start = 0;
    bool first = true;
    while (start <= string.size()) {
        size_t nl = string.find('\n', start);
        std::string line = string.substr(start, nl == std::string::npos ? std::string::npos : nl - start);
        if (nl == std::string::npos && line.empty() && !first) break;
        if (!first) out += pad;
        out += line;
        if (nl == std::string::npos) break;
        out += '\n';
        start = nl + 1;
        first = false;
    }
    return out;
}

bool endsWith(const std::string &value, const std::string &ending) {
    return ending.size() <= value.size() && std::equal(ending.rbegin(), ending.rend(), value.rbegin());
}

std::string toLower(const std::string &value) {
    std::string r(value);
    std::transform(r.begin(), r.end(), r.begin(), ::tolower);
    return r;
}

bool toBool(const std::string &str) {
    std::string v = toLower(str);
    if (v == "false") return false;
    if (v == "true") return true;
    throw NoriException("Could not parse boolean value \"%s\"", str);
}

int toInt(const std::string &str) {
    char *end = nullptr;
    int r = (int) strtol(str.c_str(), &end, 10);
    if (*end != '\0') throw NoriException("Could not parse integer value \"%s\"", str);
    return r;
}

unsigned int toUInt(const std::string &str) {
    char *end = nullptr;
    unsigned int r = (int) strtoul(str.c_str(), &end, 10);
    if (*end != '\0') throw NoriException("Could not parse integer value \"%s\"", str);
    return r;
}

float toFloat(const std::string &str) {
    char *end = nullptr;
    float r = (float) strtof(str.c_str(), &end);
    if (*end != '\0') throw NoriException("Could not parse floating point value \"%s\"", str);
    return r;
}

std::vector<std::string> tokenize(const std::string &s, const std::string &delim, bool includeEmpty) {
    std::vector<std::string> tokens;
    std::string::size_type last = 0, pos = s.find_first_of(delim, last);
    while (last != std::string::npos) {
        if (pos != last || includeEmpty) tokens.push_back(s.substr(last, pos - last));
        last = pos;
        if (last != std::string::npos) {
            last += 1;
            pos = s.find_first_of(delim, last);
        }
    }
    return tokens;
}

std::string timeString(double time, bool precise) {
    if (std::isnan(time) || std::isinf(time)) return "inf";
    std::string suffix = "ms";
    if (time > 1000) {
        time /= 1000; suffix = "s";
        if (time > 60) {
            time /= 60; suffix = "m";
            if (time > 60) {
                time /= 60; suffix = "h";
                if (time > 12) { time /= 12; suffix = "d"; }
            }
        }
    }
    std::ostringstream os;
    os << std::setprecision(precise ? 4 : 1) << std::fixed << time << suffix;
    return os.str();
}

std::string memString(size_t size, bool precise) {
    double value = (double) size;
    const char *suffixes[] = {"B", "KiB", "MiB", "GiB", "TiB", "PiB"};
    int suffix = 0;
    while (suffix < 5 && value > 1024.0f) { value /= 1024.0f; ++suffix; }
    std::ostringstream os;
    os << std::setprecision(suffix == 0 ? 0 : (precise ? 4 : 1)) << std::fixed << value << " " << suffixes[suffix];
    return os.str();
}

/* ---- colours: src/common.cpp:166-209 ---- */
Color3f Color3f::toSRGB() const {
    Color3f result;
    for (int i = 0; i < 3; ++i) {
        float value = v[i];
        if (value <= 0.0031308f) result[i] = 12.92f * value;
        else result[i] = (1.0f + 0.055f) * std::pow(value, 1.0f / 2.4f) - 0.055f;
    }
    return result;
}

Color3f Color3f::toLinearRGB() const {
    Color3f result;
    for (int i = 0; i < 3; ++i) {
        float value = v[i];
        if (value <= 0.04045f) result[i] = value * (1.0f / 12.92f);
        else result[i] = std::pow((value + 0.055f) * (1.0f / 1.055f), 2.4f);
    }
    return result;
}

bool Color3f::isValid() const {
    for (int i = 0; i < 3; ++i)
        if (v[i] < 0 || !std::isfinite(v[i])) return false;
    return true;
}

float Color3f::getLuminance() const { return v[0] * 0.212671f + v[1] * 0.715160f + v[2] * 0.072169f; }

/* ---- 4x4 helpers (row-major) ---- */
void mat4Identity(float *m) {
    std::memset(m, 0, sizeof(float) * 16);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

void mat4Mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[4 * i + j] = ((a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j];
    std::memcpy(out, r, sizeof(r));
}

bool mat4Inverse(const float *a, float *out) {
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { w[i][j] = a[4 * i + j]; w[i][j + 4] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[piv][j]);
        double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = w[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (float) w[i][j + 4];
    return true;
}

Transform::Transform() { mat4Identity(m); mat4Identity(inv); }
Transform::Transform(const float *trafo) {
    std::memcpy(m, trafo, sizeof(m));
    if (!mat4Inverse(m, inv)) throw NoriException("Transform: matrix is not invertible");
}
Transform::Transform(const float *trafo, const float *inverse) {
    std::memcpy(m, trafo, sizeof(m));
    std::memcpy(inv, inverse, sizeof(inv));
}
/* src/common.cpp:219-222 */
Transform Transform::operator*(const Transform &t) const {
    Transform r;
    mat4Mul(m, t.m, r.m);
    mat4Mul(t.inv, inv, r.inv);
    return r;
}
/* include/nori/transform.h:55-68 */
Vector3f Transform::applyVector(const Vector3f &v) const {
    return Vector3f(m[0] * v.x() + (m[1] * v.y() + m[2] * v.z()),
                    m[4] * v.x() + (m[5] * v.y() + m[6] * v.z()),
                    m[8] * v.x() + (m[9] * v.y() + m[10] * v.z()));
}
Normal3f Transform::applyNormal(const Normal3f &n) const {
    /* inverse.topLeftCorner<3,3>().transpose() * n */
    return Normal3f(inv[0] * n.x() + (inv[4] * n.y() + inv[8] * n.z()),
                    inv[1] * n.x() + (inv[5] * n.y() + inv[9] * n.z()),
                    inv[2] * n.x() + (inv[6] * n.y() + inv[10] * n.z()));
}
Point3f Transform::applyPoint(const Point3f &p) const {
    float r[4];
    for (int i = 0; i < 4; ++i)
        r[i] = ((m[4 * i] * p.x() + m[4 * i + 1] * p.y()) + m[4 * i + 2] * p.z()) + m[4 * i + 3] * 1.0f;
    return Point3f(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}
std::string Transform::toString() const {
    std::ostringstream os;
    os << "[";
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) { os << std::setprecision(4) << m[4 * i + j]; if (j < 3) os << ", "; }
        if (i < 3) os << ";\n";
    }
    os << "]";
    return os.str();
}

Vector3f sphericalDirection(float theta, float phi) {
    float st, ct, sp, cp;
    sincosf(theta, &st, &ct);
    sincosf(phi, &sp, &cp);
    return Vector3f(st * cp, st * sp, ct);
}

std::string FileResolver::resolve(const std::string &name) const {
    if (!name.empty() && name[0] == '/') return name;
    for (const std::string &dir : m_paths) {
        std::string candidate = dir.empty() ? name : dir + "/" + name;
        std::ifstream f(candidate);
        if (f.good()) return candidate;
    }
    return name;
}

FileResolver *getFileResolver() {
    static FileResolver *resolver = new FileResolver();
    return resolver;
}

Timer::Timer() { reset(); }
void Timer::reset() {
    m_start = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
double Timer::elapsed() const {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - m_start;
}

NORI_NAMESPACE_END
