/*
 * obj.cpp -- Wavefront OBJ loader, plugin name "obj", built for multi-million-triangle files.
 *
 * Same result as the reference's loader (src/obj.cpp:20-158): `v / vt / vn / f` records, triangles and
 * quads (a quad a,b,c,d becomes abc + dac), vertices de-duplicated by their (position, texcoord,
 * normal) index triple in first-use order, `toWorld` applied to positions (homogeneous point) and
 * normals (inverse transpose, renormalised) at load time, floats parsed with correct rounding.
 *
 * Different machinery: the file is read in one piece, cut at line ends into one slice per worker
 * thread, and the slices are scanned in parallel with pointer-based tokenising and std::from_chars
 * (no stream objects, no per-line allocations).  The slices' records are stitched together in file
 * order; the first-use de-duplication then runs once over the face corners -- through a direct
 * position-index table when the file has no normals / texcoords, an open-addressing table otherwise.
 * (the reference's getline/istringstream/unordered_map loop is the
 * wall-clock bottleneck of the 10 M-triangle configuration once rendering is fast.)
 */
#include <nori/plugins.h>

#include <charconv>
#include <cstring>
#include <fstream>
#include <thread>

NORI_NAMESPACE_BEGIN

namespace {

constexpr uint32_t kAbsent = 0xffffffffu;

struct Corner { uint32_t p, uv, n; };       /* 1-based OBJ indices, kAbsent where the corner has none */

struct Slice {
    const char *begin = nullptr, *end = nullptr;
    std::vector<float> pos, nrm, uv;        /* transformed positions / normals, raw texcoords */
    std::vector<Corner> corners;            /* 3 per emitted triangle */
    std::string error;
};

inline bool isBlank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

/* next whitespace-delimited token of the line [p, e) */
inline bool nextToken(const char *&p, const char *e, const char *&tb, const char *&te) {
    while (p < e && isBlank(*p)) ++p;
    if (p >= e) return false;
    tb = p;
    while (p < e && !isBlank(*p)) ++p;
    te = p;
    return true;
}

/* `stream >> float`: correctly rounded; a missing or malformed field reads as 0 */
inline float readFloat(const char *&p, const char *e) {
    const char *tb, *te;
    if (!nextToken(p, e, tb, te)) return 0.0f;
    if (*tb == '+') ++tb;
    float v = 0.0f;
    const auto r = std::from_chars(tb, te, v);
    if (r.ec != std::errc()) return 0.0f;
    return v;
}

/* one face corner "p", "p/t", "p//n", "p/t/n" (reference: tokenize on '/' keeping empty parts, toUInt) */
bool readCorner(const char *tb, const char *te, Corner &c, std::string &error) {
    uint32_t v[3] = {kAbsent, kAbsent, kAbsent};
    int part = 0;
    const char *q = tb;
    while (true) {
        const char *s = q;
        while (q < te && *q != '/') ++q;
        if (part > 2) { error = format("Invalid vertex data: \"%s\"", std::string(tb, te)); return false; }
        if (q > s) {
            unsigned long x = 0;
            const auto r = std::from_chars(s, q, x);
            if (r.ec != std::errc() || r.ptr != q) { error = format("Could not parse integer value \"%s\"", std::string(s, q)); return false; }
            v[part] = (uint32_t) x;
        } else if (part == 0) {
            error = format("Could not parse integer value \"%s\"", std::string()); return false;
        }
        ++part;
        if (q >= te) break;
        ++q;                                    /* skip '/' */
        if (q >= te) { ++part; break; }         /* trailing '/': an empty last part */
    }
    if (part > 3) { error = format("Invalid vertex data: \"%s\"", std::string(tb, te)); return false; }
    c.p = v[0]; c.uv = v[1]; c.n = v[2];
    return true;
}

void scanSlice(Slice &sl, const Transform &trafo) {
    const char *p = sl.begin;
    while (p < sl.end) {
        const char *e = (const char *) memchr(p, '\n', (size_t) (sl.end - p));
        if (!e) e = sl.end;
        const char *q = p, *tb, *te;
        if (nextToken(q, e, tb, te)) {
            const size_t len = (size_t) (te - tb);
            if (len == 1 && tb[0] == 'v') {
                Point3f pt;
                pt.x() = readFloat(q, e); pt.y() = readFloat(q, e); pt.z() = readFloat(q, e);
                const Point3f w = trafo.applyPoint(pt);
                sl.pos.push_back(w.x()); sl.pos.push_back(w.y()); sl.pos.push_back(w.z());
            } else if (len == 2 && tb[0] == 'v' && tb[1] == 't') {
                sl.uv.push_back(readFloat(q, e)); sl.uv.push_back(readFloat(q, e));
            } else if (len == 2 && tb[0] == 'v' && tb[1] == 'n') {
                Normal3f n;
                n.x() = readFloat(q, e); n.y() = readFloat(q, e); n.z() = readFloat(q, e);
                const Vector3f w = trafo.applyNormal(n).normalized();
                sl.nrm.push_back(w.x()); sl.nrm.push_back(w.y()); sl.nrm.push_back(w.z());
            } else if (len == 1 && tb[0] == 'f') {
                Corner c[4];
                int n = 0;
                while (n < 4 && nextToken(q, e, tb, te)) {
                    if (!readCorner(tb, te, c[n], sl.error)) return;
                    ++n;
                }
                if (n < 3) { sl.error = format("Could not parse integer value \"%s\"", std::string()); return; }
                sl.corners.push_back(c[0]); sl.corners.push_back(c[1]); sl.corners.push_back(c[2]);
                if (n == 4) { sl.corners.push_back(c[3]); sl.corners.push_back(c[0]); sl.corners.push_back(c[2]); }
            }
        }
        p = e < sl.end ? e + 1 : sl.end;
    }
}

/* first-use ids of (p, uv, n) triples: open addressing, linear probing, grows by doubling */
class CornerTable {
public:
    explicit CornerTable(size_t expected) { rehash(std::max<size_t>(1024, expected)); }
    /* returns the id of c, assigning `next` when c is new (then *isNew = true) */
    uint32_t findOrInsert(const Corner &c, uint32_t next, bool *isNew) {
        if ((m_used + 1) * 10 > m_slots.size() * 7) rehash(m_slots.size() * 2);
        size_t i = hash(c) & m_mask;
        while (true) {
            Slot &s = m_slots[i];
            if (s.id == kAbsent) { s.c = c; s.id = next; ++m_used; *isNew = true; return next; }
            if (s.c.p == c.p && s.c.uv == c.uv && s.c.n == c.n) { *isNew = false; return s.id; }
            i = (i + 1) & m_mask;
        }
    }
private:
    struct Slot { Corner c; uint32_t id; };
    static size_t hash(const Corner &c) {
        uint64_t h = (uint64_t) c.p * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t) c.uv + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= ((uint64_t) c.n + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
        return (size_t) (h ^ (h >> 29));
    }
    void rehash(size_t want) {
        size_t cap = 1024;
        while (cap < want) cap <<= 1;
        std::vector<Slot> old;
        old.swap(m_slots);
        m_slots.assign(cap, Slot{Corner{0, 0, 0}, kAbsent});
        m_mask = cap - 1; m_used = 0;
        for (const Slot &s : old)
            if (s.id != kAbsent) {
                size_t i = hash(s.c) & m_mask;
                while (m_slots[i].id != kAbsent) i = (i + 1) & m_mask;
                m_slots[i] = s; ++m_used;
            }
    }
    std::vector<Slot> m_slots;
    size_t m_mask = 0, m_used = 0;
};

} // namespace

class WavefrontOBJ : public Mesh {
public:
    WavefrontOBJ(const PropertyList &propList) {
        const std::string filename = getFileResolver()->resolve(propList.getString("filename"));
        std::ifstream is(filename, std::ios::binary | std::ios::ate);
        if (is.fail()) throw NoriException("Unable to open OBJ file \"%s\"!", filename);
        const Transform trafo = propList.getTransform("toWorld", Transform());
        const bool verbose = Scene::s_verbose;
        if (verbose) { cout << "Loading \"" << filename << "\" .. "; cout.flush(); }
        Timer timer;

        /* the whole file in memory */
        const std::streamoff size = is.tellg();
        std::string text((size_t) std::max<std::streamoff>(size, 0), '\0');
        is.seekg(0);
        if (size > 0) is.read(&text[0], size);
        if (is.fail() && !is.eof()) throw NoriException("Unable to read OBJ file \"%s\"!", filename);

        /* one slice per worker, cut after a line end */
        const size_t nWorkers = std::max<size_t>(1, std::min<size_t>({(size_t) std::max(1u, std::thread::hardware_concurrency()), (size_t) 32,
                                                                      text.size() / (4u << 20) + 1}));
        std::vector<Slice> slices(nWorkers);
        {
            const char *b = text.data(), *e = b + text.size(), *cut = b;
            for (size_t k = 0; k < nWorkers; ++k) {
                slices[k].begin = cut;
                const char *want = k + 1 == nWorkers ? e : b + text.size() * (k + 1) / nWorkers;
                if (want < cut) want = cut;
                if (want < e) { const char *nl = (const char *) memchr(want, '\n', (size_t) (e - want)); want = nl ? nl + 1 : e; }
                slices[k].end = cut = want;
            }
        }
        {
            std::vector<std::thread> pool;
            for (size_t k = 1; k < nWorkers; ++k) pool.emplace_back([&, k] { scanSlice(slices[k], trafo); });
            scanSlice(slices[0], trafo);
            for (std::thread &t : pool) t.join();
        }
        for (const Slice &s : slices)           /* the first error in file order, as a serial reader would hit it */
            if (!s.error.empty()) throw NoriException("%s", s.error);

        /* stitch the records together in file order */
        size_t nPos = 0, nNrm = 0, nUV = 0, nCorners = 0;
        for (const Slice &s : slices) { nPos += s.pos.size() / 3; nNrm += s.nrm.size() / 3; nUV += s.uv.size() / 2; nCorners += s.corners.size(); }
        std::vector<float> positions(3 * nPos), normals(3 * nNrm), texcoords(2 * nUV);
        std::vector<Corner> corners(nCorners);
        {
            size_t oP = 0, oN = 0, oT = 0, oC = 0;
            for (Slice &s : slices) {
                std::copy(s.pos.begin(), s.pos.end(), positions.begin() + oP); oP += s.pos.size();
                std::copy(s.nrm.begin(), s.nrm.end(), normals.begin() + oN); oN += s.nrm.size();
                std::copy(s.uv.begin(), s.uv.end(), texcoords.begin() + oT); oT += s.uv.size();
                std::copy(s.corners.begin(), s.corners.end(), corners.begin() + oC); oC += s.corners.size();
                Slice().pos.swap(s.pos); Slice().nrm.swap(s.nrm); Slice().uv.swap(s.uv); std::vector<Corner>().swap(s.corners);
            }
        }
        text.clear(); text.shrink_to_fit();

        /* de-duplicate the corners in first-use order */
        m_F.resize(nCorners);
        std::vector<Corner> vertices;
        bool positionsOnly = true;
        for (const Corner &c : corners) if (c.uv != kAbsent || c.n != kAbsent) { positionsOnly = false; break; }
        if (positionsOnly) {
            std::vector<uint32_t> idOfPosition(nPos + 1, kAbsent);
            CornerTable outOfRange(16);         /* indices beyond the `v` records: reported below like the reference does */
            for (size_t i = 0; i < nCorners; ++i) {
                const Corner &c = corners[i];
                if (c.p <= nPos) {
                    uint32_t &id = idOfPosition[c.p];
                    if (id == kAbsent) { id = (uint32_t) vertices.size(); vertices.push_back(c); }
                    m_F[i] = id;
                } else {
                    bool isNew;
                    m_F[i] = outOfRange.findOrInsert(c, (uint32_t) vertices.size(), &isNew);
                    if (isNew) vertices.push_back(c);
                }
            }
        } else {
            CornerTable table(nCorners / 4);
            for (size_t i = 0; i < nCorners; ++i) {
                bool isNew;
                m_F[i] = table.findOrInsert(corners[i], (uint32_t) vertices.size(), &isNew);
                if (isNew) vertices.push_back(corners[i]);
            }
        }
        std::vector<Corner>().swap(corners);

        auto record = [](const std::vector<float> &a, uint32_t index1, size_t width) -> const float * {
            if (index1 == 0 || (size_t) index1 * width > a.size())       /* std::vector::at in the reference */
                throw std::out_of_range("vector::_M_range_check: OBJ index out of range");
            return &a[(size_t) (index1 - 1) * width];
        };
        m_V.resize(3 * vertices.size());
        for (size_t i = 0; i < vertices.size(); ++i) std::memcpy(&m_V[3 * i], record(positions, vertices[i].p, 3), 3 * sizeof(float));
        if (!normals.empty()) {
            m_N.resize(3 * vertices.size());
            for (size_t i = 0; i < vertices.size(); ++i) std::memcpy(&m_N[3 * i], record(normals, vertices[i].n, 3), 3 * sizeof(float));
        }
        if (!texcoords.empty()) {
            m_UV.resize(2 * vertices.size());
            for (size_t i = 0; i < vertices.size(); ++i) std::memcpy(&m_UV[2 * i], record(texcoords, vertices[i].uv, 2), 2 * sizeof(float));
        }
        m_name = filename;
        if (verbose)
            cout << "done. (V=" << getVertexCount() << ", F=" << getTriangleCount() << ", took " << timer.elapsedString()
                 << " and " << memString(m_F.size() * sizeof(uint32_t) + sizeof(float) * (m_V.size() + m_N.size() + m_UV.size()))
                 << ")" << endl;
    }
};

NORI_REGISTER_CLASS(WavefrontOBJ, "obj");
NORI_NAMESPACE_END
