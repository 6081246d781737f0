/*
 * obj.cpp -- Wavefront OBJ loader, plugin name "obj".
 * Semantics of the reference's src/obj.cpp:20-158: `v/vt/vn/f` records,
 * triangles and quads (a quad a,b,c,d becomes abc + acd), vertices de-duplicated
 * by their (position, texcoord, normal) index triple in first-use order,
 * `toWorld` applied to positions (homogeneous point) and normals (inverse
 * transpose, renormalised) at load time.  Load-time host work: not on the
 * accelerated path.
 */
#include <nori/plugins.h>

#include <fstream>
#include <unordered_map>

NORI_NAMESPACE_BEGIN

namespace {
struct VertexKey {
    uint32_t p = (uint32_t) -1, n = (uint32_t) -1, uv = (uint32_t) -1;
    bool operator==(const VertexKey &o) const { return p == o.p && n == o.n && uv == o.uv; }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey &k) const {
        size_t h = std::hash<uint32_t>()(k.p);
        h = h * 37 + std::hash<uint32_t>()(k.uv);
        h = h * 37 + std::hash<uint32_t>()(k.n);
        return h;
    }
};
VertexKey parseVertex(const std::string &token) {
    std::vector<std::string> parts = tokenize(token, "/", true);
    if (parts.size() < 1 || parts.size() > 3) throw NoriException("Invalid vertex data: \"%s\"", token);
    VertexKey k;
    k.p = toUInt(parts[0]);
    if (parts.size() >= 2 && !parts[1].empty()) k.uv = toUInt(parts[1]);
    if (parts.size() >= 3 && !parts[2].empty()) k.n = toUInt(parts[2]);
    return k;
}
} // namespace

class WavefrontOBJ : public Mesh {
public:
    WavefrontOBJ(const PropertyList &propList) {
        std::string filename = getFileResolver()->resolve(propList.getString("filename"));
        std::ifstream is(filename);
        if (is.fail()) throw NoriException("Unable to open OBJ file \"%s\"!", filename);
        Transform trafo = propList.getTransform("toWorld", Transform());
        const bool verbose = Scene::s_verbose;
        if (verbose) { cout << "Loading \"" << filename << "\" .. "; cout.flush(); }
        Timer timer;

        std::vector<Vector3f> positions, normals;
        std::vector<Point2f> texcoords;
        std::vector<VertexKey> vertices;
        std::unordered_map<VertexKey, uint32_t, VertexKeyHash> vertexMap;

        std::string line_str;
        while (std::getline(is, line_str)) {
            std::istringstream line(line_str);
            std::string prefix;
            line >> prefix;
            if (prefix == "v") {
                Point3f p;
                line >> p.x() >> p.y() >> p.z();
                positions.push_back(trafo.applyPoint(p));
            } else if (prefix == "vt") {
                Point2f tc;
                line >> tc.x() >> tc.y();
                texcoords.push_back(tc);
            } else if (prefix == "vn") {
                Normal3f n;
                line >> n.x() >> n.y() >> n.z();
                normals.push_back(trafo.applyNormal(n).normalized());
            } else if (prefix == "f") {
                std::string v[4];
                line >> v[0] >> v[1] >> v[2] >> v[3];
                VertexKey verts[6];
                int nVertices = 3;
                for (int i = 0; i < 3; ++i) verts[i] = parseVertex(v[i]);
                if (!v[3].empty()) {
                    verts[3] = parseVertex(v[3]); verts[4] = verts[0]; verts[5] = verts[2];
                    nVertices = 6;
                }
                /* emitted order for a quad: (0,1,2) and (3,0,2) as in obj.cpp:70-79 */
                for (int i = 0; i < nVertices; ++i) {
                    auto it = vertexMap.find(verts[i]);
                    if (it == vertexMap.end()) {
                        uint32_t id = (uint32_t) vertices.size();
                        vertexMap[verts[i]] = id;
                        m_F.push_back(id);
                        vertices.push_back(verts[i]);
                    } else {
                        m_F.push_back(it->second);
                    }
                }
            }
        }
        m_V.resize(3 * vertices.size());
        for (size_t i = 0; i < vertices.size(); ++i) {
            const Vector3f &p = positions.at(vertices[i].p - 1);
            m_V[3 * i] = p.x(); m_V[3 * i + 1] = p.y(); m_V[3 * i + 2] = p.z();
        }
        if (!normals.empty()) {
            m_N.resize(3 * vertices.size());
            for (size_t i = 0; i < vertices.size(); ++i) {
                const Vector3f &n = normals.at(vertices[i].n - 1);
                m_N[3 * i] = n.x(); m_N[3 * i + 1] = n.y(); m_N[3 * i + 2] = n.z();
            }
        }
        if (!texcoords.empty()) {
            m_UV.resize(2 * vertices.size());
            for (size_t i = 0; i < vertices.size(); ++i) {
                const Point2f &t = texcoords.at(vertices[i].uv - 1);
                m_UV[2 * i] = t.x(); m_UV[2 * i + 1] = t.y();
            }
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
