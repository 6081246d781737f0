/*
 * plugins.cpp -- host side of the plugin surface.
 *
 * Registered names (what the XML `type=` attribute selects), with the
 * reference's parameter names and defaults:
 *   bsdf        diffuse (src/diffuse.cpp:18-20), mirror (src/mirror.cpp:15),
 *               dielectric (src/dielectric.cpp:17-23), microfacet (src/microfacet.cpp:17-36)
 *   emitter     area  [radiance]                          (absent from the reference)
 *   integrator  normals, ao, simple [position, energy], whitted, path_mats,
 *               path_ems, path_mis                        (absent from the reference)
 *   sampler     independent [sampleCount]                 (src/independent.cpp:23-25)
 *   camera      perspective                               (src/perspective.cpp:22-39)
 *   rfilter     gaussian, mitchell, tent, box             (src/rfilter.cpp)
 * None of these classes computes radiance, samples or intersections on the
 * CPU: the virtuals forward to libnori_hip.
 */
#include <nori/plugins.h>

#include <algorithm>

NORI_NAMESPACE_BEGIN

/* =================================================================== Warp */
void Warp::warpBatch(nori_warp_type type, float param, const float *samples, size_t n, float *out) {
    Device &d = Device::shared();
    d.check(nori_hip_warp(d.ctx(), type, param, samples, n, out), "nori_hip_warp");
}
void Warp::pdfBatch(nori_warp_type type, float param, const float *points, size_t n, float *pdf) {
    Device &d = Device::shared();
    d.check(nori_hip_warp_pdf(d.ctx(), type, param, points, n, pdf), "nori_hip_warp_pdf");
}
static Vector3f warp1(nori_warp_type t, float param, const Point2f &s) {
    float in[2] = {s.x(), s.y()}, out[3];
    Warp::warpBatch(t, param, in, 1, out);
    return Vector3f(out[0], out[1], out[2]);
}
static float pdf1(nori_warp_type t, float param, const Vector3f &v) {
    float in[3] = {v.x(), v.y(), v.z()}, out;
    Warp::pdfBatch(t, param, in, 1, &out);
    return out;
}
Point2f Warp::squareToUniformSquare(const Point2f &s) { Vector3f r = warp1(NORI_WARP_SQUARE, 0, s); return Point2f(r.x(), r.y()); }
float Warp::squareToUniformSquarePdf(const Point2f &p) { return pdf1(NORI_WARP_SQUARE, 0, Vector3f(p.x(), p.y(), 0)); }
Point2f Warp::squareToTent(const Point2f &s) { Vector3f r = warp1(NORI_WARP_TENT, 0, s); return Point2f(r.x(), r.y()); }
float Warp::squareToTentPdf(const Point2f &p) { return pdf1(NORI_WARP_TENT, 0, Vector3f(p.x(), p.y(), 0)); }
Point2f Warp::squareToUniformDisk(const Point2f &s) { Vector3f r = warp1(NORI_WARP_DISK, 0, s); return Point2f(r.x(), r.y()); }
float Warp::squareToUniformDiskPdf(const Point2f &p) { return pdf1(NORI_WARP_DISK, 0, Vector3f(p.x(), p.y(), 0)); }
Vector3f Warp::squareToUniformSphere(const Point2f &s) { return warp1(NORI_WARP_UNIFORM_SPHERE, 0, s); }
float Warp::squareToUniformSpherePdf(const Vector3f &v) { return pdf1(NORI_WARP_UNIFORM_SPHERE, 0, v); }
Vector3f Warp::squareToUniformHemisphere(const Point2f &s) { return warp1(NORI_WARP_UNIFORM_HEMISPHERE, 0, s); }
float Warp::squareToUniformHemispherePdf(const Vector3f &v) { return pdf1(NORI_WARP_UNIFORM_HEMISPHERE, 0, v); }
Vector3f Warp::squareToCosineHemisphere(const Point2f &s) { return warp1(NORI_WARP_COSINE_HEMISPHERE, 0, s); }
float Warp::squareToCosineHemispherePdf(const Vector3f &v) { return pdf1(NORI_WARP_COSINE_HEMISPHERE, 0, v); }
Vector3f Warp::squareToBeckmann(const Point2f &s, float alpha) { return warp1(NORI_WARP_BECKMANN, alpha, s); }
float Warp::squareToBeckmannPdf(const Vector3f &m, float alpha) { return pdf1(NORI_WARP_BECKMANN, alpha, m); }

/* =================================================================== BSDF */
void BSDF::sampleBatch(const float *wi, const float *sample, size_t n, float *wo, float *weight, float *eta, int32_t *measure) const {
    nori_bsdf_desc d; fill(d);
    Device &dev = Device::shared();
    dev.check(nori_hip_bsdf_sample(dev.ctx(), &d, wi, sample, n, wo, weight, eta, measure), "nori_hip_bsdf_sample");
}
void BSDF::evalBatch(const float *wi, const float *wo, size_t n, float *value) const {
    nori_bsdf_desc d; fill(d);
    Device &dev = Device::shared();
    dev.check(nori_hip_bsdf_eval(dev.ctx(), &d, wi, wo, n, value), "nori_hip_bsdf_eval");
}
void BSDF::pdfBatch(const float *wi, const float *wo, size_t n, float *pdf) const {
    nori_bsdf_desc d; fill(d);
    Device &dev = Device::shared();
    dev.check(nori_hip_bsdf_pdf(dev.ctx(), &d, wi, wo, n, pdf), "nori_hip_bsdf_pdf");
}
Color3f BSDF::sample(BSDFQueryRecord &bRec, const Point2f &s) const {
    float wo[3], w[3], eta; int32_t m;
    sampleBatch(bRec.wi.v, s.v, 1, wo, w, &eta, &m);
    bRec.wo = Vector3f(wo[0], wo[1], wo[2]); bRec.eta = eta; bRec.measure = (EMeasure) m;
    return Color3f(w[0], w[1], w[2]);
}
Color3f BSDF::eval(const BSDFQueryRecord &bRec) const {
    if (bRec.measure != ESolidAngle) return Color3f(0.0f);       /* src/diffuse.cpp:26 */
    float v[3];
    evalBatch(bRec.wi.v, bRec.wo.v, 1, v);
    return Color3f(v[0], v[1], v[2]);
}
float BSDF::pdf(const BSDFQueryRecord &bRec) const {
    if (bRec.measure != ESolidAngle) return 0.0f;
    float p;
    pdfBatch(bRec.wi.v, bRec.wo.v, 1, &p);
    return p;
}

static void clearDesc(nori_bsdf_desc &d, int type) { std::memset(&d, 0, sizeof(d)); d.type = type; }

class Diffuse : public BSDF {
public:
    Diffuse(const PropertyList &propList) { m_albedo = propList.getColor("albedo", Color3f(0.5f)); }
    bool isDiffuse() const { return true; }
    void fill(nori_bsdf_desc &d) const { clearDesc(d, NORI_BSDF_DIFFUSE); for (int i = 0; i < 3; ++i) d.albedo[i] = m_albedo[i]; }
    std::string toString() const { return format("Diffuse[\n  albedo = %s\n]", m_albedo.toString()); }
private:
    Color3f m_albedo;
};

class Mirror : public BSDF {
public:
    Mirror(const PropertyList &) {}
    void fill(nori_bsdf_desc &d) const { clearDesc(d, NORI_BSDF_MIRROR); }
    std::string toString() const { return "Mirror[]"; }
};

class Dielectric : public BSDF {
public:
    Dielectric(const PropertyList &propList) {
        m_intIOR = propList.getFloat("intIOR", 1.5046f);     /* BK7 */
        m_extIOR = propList.getFloat("extIOR", 1.000277f);   /* air */
    }
    void fill(nori_bsdf_desc &d) const { clearDesc(d, NORI_BSDF_DIELECTRIC); d.int_ior = m_intIOR; d.ext_ior = m_extIOR; }
    std::string toString() const { return format("Dielectric[\n  intIOR = %f,\n  extIOR = %f\n]", m_intIOR, m_extIOR); }
private:
    float m_intIOR, m_extIOR;
};

class Microfacet : public BSDF {
public:
    Microfacet(const PropertyList &propList) {
        m_alpha = propList.getFloat("alpha", 0.1f);
        m_intIOR = propList.getFloat("intIOR", 1.5046f);
        m_extIOR = propList.getFloat("extIOR", 1.000277f);
        m_kd = propList.getColor("kd", Color3f(0.5f));
        m_ks = 1 - m_kd.maxCoeff();                           /* src/microfacet.cpp:35 */
    }
    bool isDiffuse() const { return true; }                   /* src/microfacet.cpp:59-64 */
    void fill(nori_bsdf_desc &d) const {
        clearDesc(d, NORI_BSDF_MICROFACET);
        for (int i = 0; i < 3; ++i) d.albedo[i] = m_kd[i];
        d.alpha = m_alpha; d.int_ior = m_intIOR; d.ext_ior = m_extIOR; d.ks = m_ks;
    }
    std::string toString() const {
        return format("Microfacet[\n  alpha = %f,\n  intIOR = %f,\n  extIOR = %f,\n  kd = %s,\n  ks = %f\n]",
                      m_alpha, m_intIOR, m_extIOR, m_kd.toString(), m_ks);
    }
private:
    float m_alpha, m_intIOR, m_extIOR, m_ks;
    Color3f m_kd;
};

NORI_REGISTER_CLASS(Diffuse, "diffuse");
NORI_REGISTER_CLASS(Mirror, "mirror");
NORI_REGISTER_CLASS(Dielectric, "dielectric");
NORI_REGISTER_CLASS(Microfacet, "microfacet");

/* ================================================================ Emitter */
class AreaLight : public Emitter {
public:
    AreaLight(const PropertyList &propList) { m_radiance = propList.getColor("radiance"); }
    Color3f getRadiance() const { return m_radiance; }
    std::string toString() const { return format("AreaLight[\n  radiance = %s\n]", m_radiance.toString()); }
private:
    Color3f m_radiance;
};
NORI_REGISTER_CLASS(AreaLight, "area");

/* ================================================================ Sampler */
/* src/independent.cpp:21-67.  The pcg32 arithmetic runs on the device
   (nori_hip_pcg32_floats_at); the host keeps the stream key, how many numbers of the stream it has
   handed out, and a buffer.  prepare(block) = m_random.seed(offset.x, offset.y)
   (independent.cpp:36-41): next1D / next2D then walk exactly the reference's pcg32 sequence, however the
   buffer is refilled and whatever blocks were rendered before. */
class Independent : public Sampler {
public:
    Independent(const PropertyList &propList) { m_sampleCount = (size_t) propList.getInteger("sampleCount", 1); }
    std::unique_ptr<Sampler> clone() const {
        std::unique_ptr<Independent> c(new Independent());
        /* the reference copies m_random, i.e. the position in the stream too (independent.cpp:30-34) */
        c->m_sampleCount = m_sampleCount; c->m_state = m_state; c->m_seq = m_seq; c->m_forks = m_forks;
        c->m_drawn = m_drawn - (m_buffer.size() - m_pos);
        return std::move(c);
    }
    void prepare(const ImageBlock &block) {
        m_state = (uint64_t) block.getOffset().x(); m_seq = (uint64_t) block.getOffset().y();
        m_buffer.clear(); m_pos = 0; m_forks = 0; m_drawn = 0;
    }
    void generate() {}
    void advance() {}
    float next1D() {
        if (m_pos >= m_buffer.size()) refill();
        return m_buffer[m_pos++];
    }
    Point2f next2D() { float a = next1D(); float b = next1D(); return Point2f(a, b); }
    void forkStream(uint64_t &state, uint64_t &seq) {
        /* distinct pcg32 streams per path: same initstate, initseq advanced in a
           range disjoint from the refill streams */
        state = m_state; seq = (m_seq << 32) + 0x80000000ull + m_forks++;      /* never equal to the block's own initseq (< 2^31) */
    }
    std::string toString() const { return format("Independent[sampleCount=%i]", (int) m_sampleCount); }
protected:
    Independent() {}
private:
    void refill() {
        const uint32_t kChunk = 4096;
        m_buffer.resize(kChunk);
        uint64_t st = m_state, sq = m_seq;
        Device &d = Device::shared();
        d.check(nori_hip_pcg32_floats_at(d.ctx(), &st, &sq, m_drawn, 1, kChunk, m_buffer.data()), "nori_hip_pcg32_floats_at");
        m_drawn += kChunk;        /* numbers of the stream fetched so far */
        m_pos = 0;
    }
    uint64_t m_state = 0x853c49e6748fea9bULL, m_seq = 0xda3e39cb94b95bdbULL >> 1, m_forks = 0, m_drawn = 0;
    std::vector<float> m_buffer;
    size_t m_pos = 0;
};
NORI_REGISTER_CLASS(Independent, "independent");

/* ================================================================ Filters */
class GaussianFilter : public ReconstructionFilter {
public:
    GaussianFilter(const PropertyList &propList) {
        m_radius = propList.getFloat("radius", 2.0f);
        m_stddev = propList.getFloat("stddev", 0.5f);
    }
    float eval(float x) const {
        float alpha = -1.0f / (2.0f * m_stddev * m_stddev);
        return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * m_radius * m_radius));
    }
    void fill(nori_rfilter_desc &d) const { std::memset(&d, 0, sizeof(d)); d.type = NORI_RFILTER_GAUSSIAN; d.radius = m_radius; d.stddev = m_stddev; }
    std::string toString() const { return format("GaussianFilter[radius=%f, stddev=%f]", m_radius, m_stddev); }
protected:
    float m_stddev;
};

class MitchellNetravaliFilter : public ReconstructionFilter {
public:
    MitchellNetravaliFilter(const PropertyList &propList) {
        m_radius = propList.getFloat("radius", 2.0f);
        m_B = propList.getFloat("B", 1.0f / 3.0f);
        m_C = propList.getFloat("C", 1.0f / 3.0f);
    }
    float eval(float x) const {
        x = std::abs(2.0f * x / m_radius);
        float x2 = x * x, x3 = x2 * x;
        if (x < 1) return 1.0f / 6.0f * ((12 - 9 * m_B - 6 * m_C) * x3 + (-18 + 12 * m_B + 6 * m_C) * x2 + (6 - 2 * m_B));
        if (x < 2) return 1.0f / 6.0f * ((-m_B - 6 * m_C) * x3 + (6 * m_B + 30 * m_C) * x2 + (-12 * m_B - 48 * m_C) * x + (8 * m_B + 24 * m_C));
        return 0.0f;
    }
    void fill(nori_rfilter_desc &d) const { std::memset(&d, 0, sizeof(d)); d.type = NORI_RFILTER_MITCHELL; d.radius = m_radius; d.B = m_B; d.C = m_C; }
    std::string toString() const { return format("MitchellNetravaliFilter[radius=%f, B=%f, C=%f]", m_radius, m_B, m_C); }
protected:
    float m_B, m_C;
};

class TentFilter : public ReconstructionFilter {
public:
    TentFilter(const PropertyList &) { m_radius = 1.0f; }
    float eval(float x) const { return std::max(0.0f, 1.0f - std::abs(x)); }
    void fill(nori_rfilter_desc &d) const { std::memset(&d, 0, sizeof(d)); d.type = NORI_RFILTER_TENT; d.radius = m_radius; }
    std::string toString() const { return "TentFilter[]"; }
};

class BoxFilter : public ReconstructionFilter {
public:
    BoxFilter(const PropertyList &) { m_radius = 0.5f; }
    float eval(float) const { return 1.0f; }
    void fill(nori_rfilter_desc &d) const { std::memset(&d, 0, sizeof(d)); d.type = NORI_RFILTER_BOX; d.radius = m_radius; }
    std::string toString() const { return "BoxFilter[]"; }
};
NORI_REGISTER_CLASS(GaussianFilter, "gaussian");
NORI_REGISTER_CLASS(MitchellNetravaliFilter, "mitchell");
NORI_REGISTER_CLASS(TentFilter, "tent");
NORI_REGISTER_CLASS(BoxFilter, "box");

/* ================================================================= Camera */
void Camera::setParent(NoriObject *parent) {
    if (parent && parent->getClassType() == EScene) m_scene = static_cast<Scene *>(parent);
}

class PerspectiveCamera : public Camera {
public:
    PerspectiveCamera(const PropertyList &propList) {
        m_outputSize.x() = propList.getInteger("width", 1280);
        m_outputSize.y() = propList.getInteger("height", 720);
        m_cameraToWorld = propList.getTransform("toWorld", Transform());
        m_fov = propList.getFloat("fov", 30.0f);
        m_nearClip = propList.getFloat("nearClip", 1e-4f);
        m_farClip = propList.getFloat("farClip", 1e4f);
        m_rfilter = NULL;
    }
    ~PerspectiveCamera() { delete m_rfilter; }
    void activate() {
        /* projection matrices are derived on the device side of the boundary
           (nori_hip_upload_scene, src/perspective.cpp:41-68); only the default
           filter is instantiated here (src/perspective.cpp:71-73) */
        if (!m_rfilter)
            m_rfilter = static_cast<ReconstructionFilter *>(NoriObjectFactory::createInstance("gaussian", PropertyList()));
    }
    Color3f sampleRay(Ray3f &ray, const Point2f &samplePosition, const Point2f &) const;
    void addChild(NoriObject *obj) {
        switch (obj->getClassType()) {
        case EReconstructionFilter:
            if (m_rfilter) throw NoriException("Camera: tried to register multiple reconstruction filters!");
            m_rfilter = static_cast<ReconstructionFilter *>(obj);
            break;
        default:
            throw NoriException("Camera::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()));
        }
    }
    void fill(nori_camera_desc &d) const {
        d.width = m_outputSize.x(); d.height = m_outputSize.y();
        d.fov = m_fov; d.near_clip = m_nearClip; d.far_clip = m_farClip;
        std::memcpy(d.to_world, m_cameraToWorld.m, sizeof(float) * 16);
    }
    std::string toString() const {
        return format("PerspectiveCamera[\n  cameraToWorld = %s,\n  outputSize = %s,\n  fov = %f,\n  clip = [%f, %f],\n  rfilter = %s\n]",
                      indent(m_cameraToWorld.toString(), 18), m_outputSize.toString(), m_fov, m_nearClip, m_farClip,
                      indent(m_rfilter->toString()));
    }
private:
    Transform m_cameraToWorld;
    float m_fov, m_nearClip, m_farClip;
};
NORI_REGISTER_CLASS(PerspectiveCamera, "perspective");

/* ============================================================ Integrators */
void Integrator::LiBatch(const Scene *scene, const nori_ray *rays, size_t n, const uint64_t *state, const uint64_t *seq, float *rgb) const {
    Device &d = scene->device();
    d.check(nori_hip_li(d.ctx(), rays, n, state, seq, rgb), "nori_hip_li");
}
Color3f Integrator::Li(const Scene *scene, Sampler *sampler, const Ray3f &ray) const {
    nori_ray r;
    for (int i = 0; i < 3; ++i) { r.o[i] = ray.o[i]; r.d[i] = ray.d[i]; }
    r.mint = ray.mint; r.maxt = ray.maxt;
    uint64_t st, sq; sampler->forkStream(st, sq);
    float rgb[3];
    LiBatch(scene, &r, 1, &st, &sq, rgb);
    return Color3f(rgb[0], rgb[1], rgb[2]);
}

#define NORI_SIMPLE_INTEGRATOR(Cls, Enum, Label)                                                       \
    class Cls : public Integrator {                                                                    \
    public:                                                                                            \
        Cls(const PropertyList &) {}                                                                   \
        void fill(nori_integrator_desc &d) const { std::memset(&d, 0, sizeof(d)); d.type = Enum; }    \
        std::string toString() const { return Label "[]"; }                                            \
    };
NORI_SIMPLE_INTEGRATOR(NormalIntegrator, NORI_INTEGRATOR_NORMALS, "NormalIntegrator")
NORI_SIMPLE_INTEGRATOR(AOIntegrator, NORI_INTEGRATOR_AO, "AmbientOcclusion")
NORI_SIMPLE_INTEGRATOR(WhittedIntegrator, NORI_INTEGRATOR_WHITTED, "WhittedIntegrator")
NORI_SIMPLE_INTEGRATOR(PathMatsIntegrator, NORI_INTEGRATOR_PATH_MATS, "PathMatsIntegrator")
NORI_SIMPLE_INTEGRATOR(PathEmsIntegrator, NORI_INTEGRATOR_PATH_EMS, "PathEmsIntegrator")
NORI_SIMPLE_INTEGRATOR(PathMisIntegrator, NORI_INTEGRATOR_PATH_MIS, "PathMisIntegrator")

class SimpleIntegrator : public Integrator {
public:
    SimpleIntegrator(const PropertyList &propList) {
        m_position = propList.getPoint("position");
        m_energy = propList.getColor("energy");
    }
    void fill(nori_integrator_desc &d) const {
        std::memset(&d, 0, sizeof(d)); d.type = NORI_INTEGRATOR_SIMPLE;
        for (int i = 0; i < 3; ++i) { d.position[i] = m_position[i]; d.energy[i] = m_energy[i]; }
    }
    std::string toString() const { return format("SimpleIntegrator[\n  position = %s,\n  energy = %s\n]", m_position.toString(), m_energy.toString()); }
private:
    Point3f m_position;
    Color3f m_energy;
};
NORI_REGISTER_CLASS(NormalIntegrator, "normals");
NORI_REGISTER_CLASS(AOIntegrator, "ao");
NORI_REGISTER_CLASS(SimpleIntegrator, "simple");
NORI_REGISTER_CLASS(WhittedIntegrator, "whitted");
NORI_REGISTER_CLASS(PathMatsIntegrator, "path_mats");
NORI_REGISTER_CLASS(PathEmsIntegrator, "path_ems");
NORI_REGISTER_CLASS(PathMisIntegrator, "path_mis");

/* =================================================================== Mesh */
Mesh::Mesh() {}
Mesh::~Mesh() { delete m_bsdf; delete m_emitter; }

void Mesh::activate() {
    if (!m_bsdf)   /* src/mesh.cpp:23-29: default diffuse BRDF */
        m_bsdf = static_cast<BSDF *>(NoriObjectFactory::createInstance("diffuse", PropertyList()));
}

void Mesh::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
    case EBSDF:
        if (m_bsdf) throw NoriException("Mesh: tried to register multiple BSDF instances!");
        m_bsdf = static_cast<BSDF *>(obj);
        break;
    case EEmitter:
        if (m_emitter) throw NoriException("Mesh: tried to register multiple Emitter instances!");
        m_emitter = static_cast<Emitter *>(obj);
        break;
    default:
        throw NoriException("Mesh::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()));
    }
}

std::string Mesh::toString() const {
    return format("Mesh[\n  name = \"%s\",\n  vertexCount = %i,\n  triangleCount = %i,\n  bsdf = %s,\n  emitter = %s\n]",
                  m_name, (int) getVertexCount(), (int) getTriangleCount(),
                  m_bsdf ? indent(m_bsdf->toString()) : std::string("null"),
                  m_emitter ? indent(m_emitter->toString()) : std::string("null"));
}

void Mesh::fill(nori_mesh_desc &d) const {
    std::memset(&d, 0, sizeof(d));
    d.n_vertices = getVertexCount(); d.n_triangles = getTriangleCount();
    d.positions = m_V.data();
    d.normals = m_N.empty() ? nullptr : m_N.data();
    d.texcoords = m_UV.empty() ? nullptr : m_UV.data();
    d.indices = m_F.data();
    m_bsdf->fill(d.bsdf);
    d.is_emitter = m_emitter ? 1 : 0;
    if (m_emitter) { Color3f r = m_emitter->getRadiance(); for (int i = 0; i < 3; ++i) d.radiance[i] = r[i]; }
}

std::string Intersection::toString() const {
    if (!mesh) return "Intersection[invalid]";
    return format("Intersection[\n  p = %s,\n  t = %f,\n  uv = %s,\n  mesh = %s\n]", p.toString(), t, uv.toString(), mesh->getName());
}

/* ================================================================== Accel */
void Accel::addMesh(Mesh *mesh) { m_meshes.push_back(mesh); }

void Accel::build() {
    /* Accel::build of the reference is a no-op (src/accel.cpp:19-21).  Here it
       is where the BVH comes from -- but the build runs on the far side of the
       C ABI, the first time the scene's device context is requested
       (Scene::device), so that a scene can be parsed and flattened on a
       machine without a GPU. */
}

static void toRay(const Ray3f &ray, nori_ray &r) {
    for (int i = 0; i < 3; ++i) { r.o[i] = ray.o[i]; r.d[i] = ray.d[i]; }
    r.mint = ray.mint; r.maxt = ray.maxt;
}

void Accel::rayIntersectBatch(const nori_ray *rays, nori_intersection *its, size_t n, bool shadowRay) const {
    Device &d = m_scene->device();
    d.check(nori_hip_intersect(d.ctx(), rays, its, n, shadowRay ? 1 : 0), "nori_hip_intersect");
}

bool Accel::rayIntersect(const Ray3f &ray, Intersection &its, bool shadowRay) const {
    nori_ray r; toRay(ray, r);
    nori_intersection o;
    rayIntersectBatch(&r, &o, 1, shadowRay);
    if (o.mesh == NORI_NO_HIT) return false;
    if (shadowRay) return true;
    its.p = Point3f(o.p[0], o.p[1], o.p[2]); its.t = o.t; its.uv = Point2f(o.uv[0], o.uv[1]);
    its.shFrame.s = Vector3f(o.sh_s[0], o.sh_s[1], o.sh_s[2]); its.shFrame.t = Vector3f(o.sh_t[0], o.sh_t[1], o.sh_t[2]);
    its.shFrame.n = Vector3f(o.sh_n[0], o.sh_n[1], o.sh_n[2]);
    its.geoFrame.s = Vector3f(o.geo_s[0], o.geo_s[1], o.geo_s[2]); its.geoFrame.t = Vector3f(o.geo_t[0], o.geo_t[1], o.geo_t[2]);
    its.geoFrame.n = Vector3f(o.geo_n[0], o.geo_n[1], o.geo_n[2]);
    its.mesh = m_meshes[o.mesh]; its.tri = o.tri;
    return true;
}

Color3f PerspectiveCamera::sampleRay(Ray3f &ray, const Point2f &samplePosition, const Point2f &) const {
    if (!m_scene) throw NoriException("PerspectiveCamera::sampleRay(): the camera is not attached to a scene");
    Device &d = m_scene->device();
    nori_ray r;
    d.check(nori_hip_sample_rays(d.ctx(), samplePosition.v, 1, &r), "nori_hip_sample_rays");
    ray.o = Point3f(r.o[0], r.o[1], r.o[2]); ray.d = Vector3f(r.d[0], r.d[1], r.d[2]);
    ray.mint = r.mint; ray.maxt = r.maxt; ray.update();
    return Color3f(1.0f);
}

/* ================================================================== Scene */
bool Scene::s_verbose = true;

Scene::Scene(const PropertyList &) { m_accel = new Accel(this); std::memset(&m_desc, 0, sizeof(m_desc)); }

Scene::~Scene() {
    delete m_accel; delete m_sampler; delete m_camera; delete m_integrator;
    for (Mesh *m : m_meshes) delete m;       /* the reference leaks these (src/scene.cpp:20-25) */
}

void Scene::activate() {
    m_accel->build();
    if (!m_integrator) throw NoriException("No integrator was specified!");
    if (!m_camera) throw NoriException("No camera was specified!");
    if (!m_sampler)
        m_sampler = static_cast<Sampler *>(NoriObjectFactory::createInstance("independent", PropertyList()));
    if (s_verbose) {
        cout << endl;
        cout << "Configuration: " << toString() << endl;
        cout << endl;
    }
}

void Scene::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
    case EMesh: {
        Mesh *mesh = static_cast<Mesh *>(obj);
        m_accel->addMesh(mesh);
        m_meshes.push_back(mesh);
    } break;
    case EEmitter:
        /* src/scene.cpp:55-59 throws here too: emitters are attached to meshes */
        throw NoriException("Scene::addChild(): emitters must be children of a <mesh> (area lights)");
    case ESampler:
        if (m_sampler) throw NoriException("There can only be one sampler per scene!");
        m_sampler = static_cast<Sampler *>(obj);
        break;
    case ECamera:
        if (m_camera) throw NoriException("There can only be one camera per scene!");
        m_camera = static_cast<Camera *>(obj);
        break;
    case EIntegrator:
        if (m_integrator) throw NoriException("There can only be one integrator per scene!");
        m_integrator = static_cast<Integrator *>(obj);
        break;
    default:
        throw NoriException("Scene::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()));
    }
    m_descValid = false;
}

std::string Scene::toString() const {
    std::string meshes;
    for (size_t i = 0; i < m_meshes.size(); ++i) {
        meshes += std::string("  ") + indent(m_meshes[i]->toString(), 2);
        if (i + 1 < m_meshes.size()) meshes += ",";
        meshes += "\n";
    }
    return format("Scene[\n  integrator = %s,\n  sampler = %s\n  camera = %s,\n  meshes = {\n  %s  }\n]",
                  indent(m_integrator->toString()), indent(m_sampler->toString()), indent(m_camera->toString()), indent(meshes, 2));
}

const nori_scene_desc &Scene::getDesc() const {
    if (!m_descValid) {
        m_meshDescs.resize(m_meshes.size());
        for (size_t i = 0; i < m_meshes.size(); ++i) m_meshes[i]->fill(m_meshDescs[i]);
        std::memset(&m_desc, 0, sizeof(m_desc));
        m_desc.n_meshes = (uint32_t) m_meshes.size();
        m_desc.meshes = m_meshDescs.data();
        m_camera->fill(m_desc.camera);
        m_camera->getReconstructionFilter()->fill(m_desc.rfilter);
        m_integrator->fill(m_desc.integrator);
        m_desc.sample_count = (int32_t) m_sampler->getSampleCount();
        m_descValid = true;
    }
    return m_desc;
}

Device &Scene::device() const {
    if (!m_device) {
        std::unique_ptr<Device> d(new Device());
        const nori_scene_desc &desc = getDesc();
        d->check(nori_hip_upload_scene(d->ctx(), &desc), "nori_hip_upload_scene");
        d->check(nori_hip_build_accel(d->ctx(), NORI_ACCEL_AUTO), "nori_hip_build_accel");
        m_device = std::move(d);
    }
    return *m_device;
}
DeviceGroup &Scene::deviceGroup(int n) const {
    if (!m_group || m_group->size() != n) {
        std::unique_ptr<DeviceGroup> g(new DeviceGroup(n));
        const nori_scene_desc &desc = getDesc();
        g->check(nori_hip_group_upload_scene(g->group(), &desc, NORI_ACCEL_AUTO), "nori_hip_group_upload_scene");
        m_group = std::move(g);
    }
    return *m_group;
}
NORI_REGISTER_CLASS(Scene, "scene");

NORI_NAMESPACE_END
