/*
 * nori/plugins.h -- the plugin interfaces of the hot path, as the reference
 * declares them: BSDF (include/nori/bsdf.h), Emitter (emitter.h), Sampler
 * (sampler.h), Camera (camera.h), ReconstructionFilter (rfilter.h), Integrator
 * (integrator.h), Mesh + Intersection (mesh.h), Accel (accel.h), Scene
 * (scene.h), ImageBlock (block.h), Warp (warp.h).
 *
 * Same names, signatures, parameter names, defaults and error behaviour; the
 * difference is WHERE the work happens.  Constructors parse the PropertyList
 * on the host; every per-sample method forwards to the gfx950 kernels through
 * the C ABI (nori/device.h), singly or -- the form the render loop uses -- in
 * batches.  Each class also knows how to write itself into the POD
 * nori_scene_desc that crosses the boundary (`fill`).
 */
#pragma once
#include <memory>
#include <nori/device.h>
#include <nori/object.h>

NORI_NAMESPACE_BEGIN

class Scene;
class Mesh;
class ImageBlock;

/* ------------------------------------------------------------------ Warp */
/* include/nori/warp.h:18-57 */
class Warp {
public:
    static Point2f squareToUniformSquare(const Point2f &sample);
    static float squareToUniformSquarePdf(const Point2f &p);
    static Point2f squareToTent(const Point2f &sample);
    static float squareToTentPdf(const Point2f &p);
    static Point2f squareToUniformDisk(const Point2f &sample);
    static float squareToUniformDiskPdf(const Point2f &p);
    static Vector3f squareToUniformSphere(const Point2f &sample);
    static float squareToUniformSpherePdf(const Vector3f &v);
    static Vector3f squareToUniformHemisphere(const Point2f &sample);
    static float squareToUniformHemispherePdf(const Vector3f &v);
    static Vector3f squareToCosineHemisphere(const Point2f &sample);
    static float squareToCosineHemispherePdf(const Vector3f &v);
    static Vector3f squareToBeckmann(const Point2f &sample, float alpha);
    static float squareToBeckmannPdf(const Vector3f &m, float alpha);
    /* batched forms: samples 2n -> out 3n; points 3n -> pdf n */
    static void warpBatch(nori_warp_type type, float param, const float *samples, size_t n, float *out);
    static void pdfBatch(nori_warp_type type, float param, const float *points, size_t n, float *pdf);
};

/* ------------------------------------------------------------------ BSDF */
/* include/nori/bsdf.h:17-38 */
struct BSDFQueryRecord {
    Vector3f wi, wo;
    float eta;
    EMeasure measure;
    BSDFQueryRecord(const Vector3f &wi_) : wi(wi_), eta(1.f), measure(EUnknownMeasure) {}
    BSDFQueryRecord(const Vector3f &wi_, const Vector3f &wo_, EMeasure m) : wi(wi_), wo(wo_), eta(1.f), measure(m) {}
};

/* include/nori/bsdf.h:43-101 */
class BSDF : public NoriObject {
public:
    virtual Color3f sample(BSDFQueryRecord &bRec, const Point2f &sample) const;
    virtual Color3f eval(const BSDFQueryRecord &bRec) const;
    virtual float pdf(const BSDFQueryRecord &bRec) const;
    EClassType getClassType() const { return EBSDF; }
    virtual bool isDiffuse() const { return false; }
    /* batched forms (wi/wo 3n, sample 2n, weight 3n) */
    void sampleBatch(const float *wi, const float *sample, size_t n, float *wo, float *weight, float *eta, int32_t *measure) const;
    void evalBatch(const float *wi, const float *wo, size_t n, float *value) const;
    void pdfBatch(const float *wi, const float *wo, size_t n, float *pdf) const;
    /* flatten into the C-ABI record */
    virtual void fill(nori_bsdf_desc &d) const = 0;
};

/* ---------------------------------------------------------------- Emitter */
/* include/nori/emitter.h:16-24; `area` (param "radiance") is authored */
class Emitter : public NoriObject {
public:
    EClassType getClassType() const { return EEmitter; }
    virtual Color3f getRadiance() const = 0;
};

/* ---------------------------------------------------------------- Sampler */
/* include/nori/sampler.h:56-95 */
class Sampler : public NoriObject {
public:
    virtual ~Sampler() {}
    virtual std::unique_ptr<Sampler> clone() const = 0;
    virtual void prepare(const ImageBlock &block) = 0;
    virtual void generate() = 0;
    virtual void advance() = 0;
    virtual float next1D() = 0;
    virtual Point2f next2D() = 0;
    virtual size_t getSampleCount() const { return m_sampleCount; }
    /* a fresh (initstate, initseq) pcg32 stream key for one device-side path */
    virtual void forkStream(uint64_t &state, uint64_t &seq) = 0;
    EClassType getClassType() const { return ESampler; }
protected:
    size_t m_sampleCount;
};

/* ------------------------------------------------------ ReconstructionFilter */
/* include/nori/rfilter.h:23-43 */
class ReconstructionFilter : public NoriObject {
public:
    float getRadius() const { return m_radius; }
    /* evaluated on the host only while tabulating / printing; the device
       receives the parameters and tabulates itself (src/block.cpp:18-27) */
    virtual float eval(float x) const = 0;
    virtual void fill(nori_rfilter_desc &d) const = 0;
    EClassType getClassType() const { return EReconstructionFilter; }
protected:
    float m_radius;
};

/* ----------------------------------------------------------------- Camera */
/* include/nori/camera.h:22-62 */
class Camera : public NoriObject {
public:
    virtual Color3f sampleRay(Ray3f &ray, const Point2f &samplePosition, const Point2f &apertureSample) const = 0;
    const Vector2i &getOutputSize() const { return m_outputSize; }
    const ReconstructionFilter *getReconstructionFilter() const { return m_rfilter; }
    virtual void fill(nori_camera_desc &d) const = 0;
    virtual void setParent(NoriObject *parent);
    EClassType getClassType() const { return ECamera; }
protected:
    Vector2i m_outputSize;
    ReconstructionFilter *m_rfilter = nullptr;
    Scene *m_scene = nullptr;
};

/* ------------------------------------------------------------- Integrator */
/* include/nori/integrator.h:20-49 */
class Integrator : public NoriObject {
public:
    virtual ~Integrator() {}
    virtual void preprocess(const Scene *) {}
    /* one radiance estimate, on the device (n = 1 batch) */
    virtual Color3f Li(const Scene *scene, Sampler *sampler, const Ray3f &ray) const;
    /* n estimates; stream k draws from pcg32.seed(state[k], seq[k]) */
    void LiBatch(const Scene *scene, const nori_ray *rays, size_t n, const uint64_t *state, const uint64_t *seq, float *rgb) const;
    virtual void fill(nori_integrator_desc &d) const = 0;
    EClassType getClassType() const { return EIntegrator; }
};

/* ------------------------------------------------------ Mesh / Intersection */
/* include/nori/mesh.h:23-52 */
struct Intersection {
    Point3f p;
    float t;
    Point2f uv;
    Frame shFrame, geoFrame;
    const Mesh *mesh;
    uint32_t tri;
    Intersection() : t(0), mesh(nullptr), tri(0) {}
    Vector3f toLocal(const Vector3f &d) const { return shFrame.toLocal(d); }
    Vector3f toWorld(const Vector3f &d) const { return shFrame.toWorld(d); }
    std::string toString() const;
};

/* include/nori/mesh.h:62-167: buffers are xyz-interleaved (== Eigen's
   column-major 3xN) */
class Mesh : public NoriObject {
public:
    virtual ~Mesh();
    virtual void activate();
    uint32_t getTriangleCount() const { return (uint32_t) (m_F.size() / 3); }
    uint32_t getVertexCount() const { return (uint32_t) (m_V.size() / 3); }
    const std::vector<float> &getVertexPositions() const { return m_V; }
    const std::vector<float> &getVertexNormals() const { return m_N; }
    const std::vector<float> &getVertexTexCoords() const { return m_UV; }
    const std::vector<uint32_t> &getIndices() const { return m_F; }
    bool isEmitter() const { return m_emitter != nullptr; }
    Emitter *getEmitter() { return m_emitter; }
    const Emitter *getEmitter() const { return m_emitter; }
    const BSDF *getBSDF() const { return m_bsdf; }
    virtual void addChild(NoriObject *child);
    const std::string &getName() const { return m_name; }
    virtual std::string toString() const;
    EClassType getClassType() const { return EMesh; }
    void fill(nori_mesh_desc &d) const;
protected:
    Mesh();
    std::string m_name;
    std::vector<float> m_V, m_N, m_UV;
    std::vector<uint32_t> m_F;
    BSDF *m_bsdf = nullptr;
    Emitter *m_emitter = nullptr;
};

/* ------------------------------------------------------------------ Accel */
/* include/nori/accel.h:20-59.  addMesh accepts any number of meshes (the
   reference throws on the second, src/accel.cpp:13-14); build() flattens them
   and the BVH is built on the device side of the C ABI when first needed. */
class Accel {
public:
    explicit Accel(Scene *scene) : m_scene(scene) {}
    void addMesh(Mesh *mesh);
    void build();
    bool rayIntersect(const Ray3f &ray, Intersection &its, bool shadowRay) const;
    void rayIntersectBatch(const nori_ray *rays, nori_intersection *its, size_t n, bool shadowRay) const;
    const std::vector<Mesh *> &getMeshes() const { return m_meshes; }
private:
    Scene *m_scene;
    std::vector<Mesh *> m_meshes;
};

/* ------------------------------------------------------------------ Scene */
/* include/nori/scene.h:20-113 */
class Scene : public NoriObject {
public:
    Scene(const PropertyList &);
    virtual ~Scene();
    const Accel *getAccel() const { return m_accel; }
    const Integrator *getIntegrator() const { return m_integrator; }
    Integrator *getIntegrator() { return m_integrator; }
    const Camera *getCamera() const { return m_camera; }
    const Sampler *getSampler() const { return m_sampler; }
    Sampler *getSampler() { return m_sampler; }
    const std::vector<Mesh *> &getMeshes() const { return m_meshes; }
    bool rayIntersect(const Ray3f &ray, Intersection &its) const { return m_accel->rayIntersect(ray, its, false); }
    bool rayIntersect(const Ray3f &ray) const { Intersection its; return m_accel->rayIntersect(ray, its, true); }
    virtual void activate();
    virtual void addChild(NoriObject *obj);
    virtual std::string toString() const;
    EClassType getClassType() const { return EScene; }

    /* --- boundary --- */
    /* POD view of the whole scene (pointers stay valid while the Scene lives) */
    const nori_scene_desc &getDesc() const;
    /* context with this scene uploaded and its BVH built (created on first use) */
    Device &device() const;
    /* the same on the first n GPUs of the node, for renders shared over them (created on first use) */
    DeviceGroup &deviceGroup(int n) const;
    /* printing the scene summary on activate (src/scene.cpp:41-43) can be silenced */
    static bool s_verbose;
private:
    std::vector<Mesh *> m_meshes;
    Integrator *m_integrator = nullptr;
    Sampler *m_sampler = nullptr;
    Camera *m_camera = nullptr;
    Accel *m_accel = nullptr;
    mutable nori_scene_desc m_desc;
    mutable std::vector<nori_mesh_desc> m_meshDescs;
    mutable bool m_descValid = false;
    mutable std::unique_ptr<Device> m_device;
    mutable std::unique_ptr<DeviceGroup> m_group;
};

/* ------------------------------------------------------------- ImageBlock */
/* include/nori/block.h:40-108: RGBW accumulators with a filter border.  The
   full-frame block lives in HBM while rendering; this class is its host view
   (download / merge / normalise). */
class Bitmap;
class ImageBlock {
public:
    ImageBlock(const Vector2i &size, const ReconstructionFilter *filter);
    void setOffset(const Point2i &o) { m_offset = o; }
    const Point2i &getOffset() const { return m_offset; }
    void setSize(const Vector2i &s) { m_size = s; }
    const Vector2i &getSize() const { return m_size; }
    int getBorderSize() const { return m_borderSize; }
    int rows() const { return m_size.y() + 2 * m_borderSize; }
    int cols() const { return m_size.x() + 2 * m_borderSize; }
    void clear() { std::fill(m_data.begin(), m_data.end(), 0.0f); }
    float *data() { return m_data.data(); }
    const float *data() const { return m_data.data(); }
    /* src/block.cpp:45-51 */
    Bitmap *toBitmap() const;
    /* src/block.cpp:93-102 (blocks of equal size: sum of RGBW) */
    void put(const ImageBlock &b);
    std::string toString() const;
private:
    Point2i m_offset;
    Vector2i m_size;
    int m_borderSize = 0;
    std::vector<float> m_data;    /* rows x cols x 4 */
};

/* src/parser.cpp:16 */
NoriObject *loadFromXML(const std::string &filename);

/* render() of src/main.cpp:58-148: renders on the device, returns the
   full-frame block; fills `stats` if given. */
std::unique_ptr<ImageBlock> renderScene(Scene *scene, nori_render_stats *stats = nullptr);

NORI_NAMESPACE_END
