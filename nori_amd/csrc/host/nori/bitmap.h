/*
 * nori/bitmap.h -- RGB float image with OpenEXR / PNG output (the reference's
 * include/nori/bitmap.h:19-40 on top of OpenEXR + stb, both absent here).
 * Output side of the path ("next" row 1 of SURVEY.md §8f).
 */
#pragma once
#include <nori/common.h>

NORI_NAMESPACE_BEGIN

class Bitmap {
public:
    Bitmap(const Vector2i &size = Vector2i(0, 0)) : m_size(size), m_data((size_t) size.x() * size.y() * 3, 0.0f) {}
    /* Load an OpenEXR file: scanline, FLOAT or HALF R/G/B channels (matched as in
       src/bitmap.cpp:32-49), compression NONE / ZIPS / ZIP */
    explicit Bitmap(const std::string &filename);
    int cols() const { return m_size.x(); }
    int rows() const { return m_size.y(); }
    float *data() { return m_data.data(); }
    const float *data() const { return m_data.data(); }
    Color3f coeff(int y, int x) const { const float *p = &m_data[((size_t) y * cols() + x) * 3]; return Color3f(p[0], p[1], p[2]); }
    void set(int y, int x, const Color3f &c) { float *p = &m_data[((size_t) y * cols() + x) * 3]; p[0] = c[0]; p[1] = c[1]; p[2] = c[2]; }
    /* `filename` without extension, as in src/bitmap.cpp:69,98 */
    void saveEXR(const std::string &filename);
    void savePNG(const std::string &filename);
private:
    Vector2i m_size;
    std::vector<float> m_data;
};

NORI_NAMESPACE_END
