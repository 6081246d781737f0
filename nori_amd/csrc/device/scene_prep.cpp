/*
 * scene_prep.cpp -- load-time host code of libnori_hip (see scene_prep.h).
 *
 * Reference behaviour reproduced here (all load-time, float, no FMA):
 *   ImageBlock filter tabulation     src/block.cpp:18-27
 *   PerspectiveCamera::activate      src/perspective.cpp:41-74
 *   Mesh::surfaceArea + DiscretePDF  src/mesh.cpp:31-37, include/nori/dpdf.h:42-97
 *   per-triangle bbox / centroid     src/mesh.cpp:78-90 (inputs of the BVH build)
 */
#include "scene_prep.h"
#include "rt_wide.h"

#include <algorithm>
#include <thread>
#include <functional>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace nrt {

float rfilter_eval(const nori_rfilter_desc &d, float radius, float x) {
    switch (d.type) {
    case NORI_RFILTER_GAUSSIAN: {
        float alpha = -1.0f / (2.0f * d.stddev * d.stddev);
        return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
    }
    case NORI_RFILTER_MITCHELL: {
        const float B = d.B, C = d.C;
        x = std::abs(2.0f * x / radius);
        float x2 = x * x, x3 = x2 * x;
        if (x < 1)
            return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
        else if (x < 2)
            return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
        return 0.0f;
    }
    case NORI_RFILTER_TENT:
        return std::max(0.0f, 1.0f - std::abs(x));
    default:
        return 1.0f;
    }
}

static void mat4_mul(const float *a, const float *b, float *r) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[4 * i + j] = ((a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j];
}

static bool mat4_inverse(const float *a, float *out) {
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { w[i][j] = a[4 * i + j]; w[i][j + 4] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[piv][j]);
        const double dv = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= dv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = w[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (float) w[i][j + 4];
    return true;
}

std::string prepare_scene(const nori_scene_desc &desc, HostScene &out) {
    out = HostScene();
    if (desc.n_meshes > 0 && !desc.meshes) return "scene has meshes == NULL";
    if (desc.camera.width <= 0 || desc.camera.height <= 0) return "camera has a non-positive output size";
    if (desc.sample_count < 0) return "negative sample count";
    out.sample_count = desc.sample_count;

    uint64_t nV = 0, nT = 0;
    for (uint32_t i = 0; i < desc.n_meshes; ++i) { nV += desc.meshes[i].n_vertices; nT += desc.meshes[i].n_triangles; }
    if (nT >= (1ull << 28)) return "too many triangles (limit 2^28)";
    out.positions.resize(nV); out.normals.resize(nV); out.texcoords.resize(nV);
    out.indices.resize(3 * nT); out.tri_mesh.resize(nT); out.shade_tris.resize(nT * kShadeQuads);

    uint32_t vOff = 0, tOff = 0;
    for (uint32_t mi = 0; mi < desc.n_meshes; ++mi) {
        const nori_mesh_desc &m = desc.meshes[mi];
        if ((m.n_vertices && !m.positions) || (m.n_triangles && !m.indices)) return "mesh with NULL buffers";
        MeshRec rec; std::memset(&rec, 0, sizeof(rec));
        rec.tri_offset = tOff; rec.vtx_offset = vOff; rec.n_triangles = m.n_triangles;
        rec.flags = (m.normals ? kMeshHasNormals : 0u) | (m.texcoords ? kMeshHasUV : 0u) | (m.is_emitter ? kMeshEmitter : 0u);
        if (m.texcoords) out.has_uv = true;
        rec.bsdf_type = m.bsdf.type;
        if (rec.bsdf_type < 0 || rec.bsdf_type > 3) return "unknown BSDF type";
        for (int k = 0; k < 3; ++k) { rec.albedo[k] = m.bsdf.albedo[k]; rec.radiance[k] = m.radiance[k]; }
        rec.alpha = m.bsdf.alpha; rec.int_ior = m.bsdf.int_ior; rec.ext_ior = m.bsdf.ext_ior; rec.ks = m.bsdf.ks;
        for (uint32_t v = 0; v < m.n_vertices; ++v) {
            f4 p; p.x = m.positions[3 * v]; p.y = m.positions[3 * v + 1]; p.z = m.positions[3 * v + 2]; p.w = 0.0f;
            out.positions[vOff + v] = p;
            f4 n; n.x = n.y = n.z = n.w = 0.0f;
            if (m.normals) { n.x = m.normals[3 * v]; n.y = m.normals[3 * v + 1]; n.z = m.normals[3 * v + 2]; }
            out.normals[vOff + v] = n;
            f2 t; t.x = t.y = 0.0f;
            if (m.texcoords) { t.x = m.texcoords[2 * v]; t.y = m.texcoords[2 * v + 1]; }
            out.texcoords[vOff + v] = t;
        }
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            for (int k = 0; k < 3; ++k) {
                uint32_t id = m.indices[3 * t + k];
                if (id >= m.n_vertices) return "triangle index out of range";
                out.indices[3 * (size_t) (tOff + t) + k] = vOff + id;
            }
            out.tri_mesh[tOff + t] = mi;
            f4 *q = &out.shade_tris[(size_t) (tOff + t) * kShadeQuads];
            for (int k = 0; k < 3; ++k) {
                const uint32_t g = out.indices[3 * (size_t) (tOff + t) + k];
                q[k] = out.positions[g]; q[k].w = 0.0f;
                q[3 + k] = out.normals[g]; q[3 + k].w = 0.0f;
            }
            q[0].w = u2f(mi);
        }
        if (m.is_emitter) {
            if (m.n_triangles == 0) return "area emitter attached to an empty mesh";
            /* DiscretePDF over Mesh::surfaceArea: append, then normalize */
            rec.cdf_offset = (uint32_t) out.emitter_cdf.size();
            out.emitter_cdf.push_back(0.0f);
            for (uint32_t t = 0; t < m.n_triangles; ++t) {
                const uint32_t *id = &out.indices[3 * (size_t) (tOff + t)];
                const f3 p0 = xyz(out.positions[id[0]]), p1 = xyz(out.positions[id[1]]), p2 = xyz(out.positions[id[2]]);
                const f3 c = cross(p1 - p0, p2 - p0);
                const float area = 0.5f * sqrtf(dot(c, c));
                out.emitter_cdf.push_back(out.emitter_cdf.back() + area);
            }
            const float sum = out.emitter_cdf.back();
            /* a light without area cannot be sampled (pdf 1 / area): fail at load time instead of feeding inf / NaN
               emitter samples to the film's isValid() guard */
            if (!(sum > 0.0f) || !(sum < kInf)) return "area emitter attached to a mesh of zero (or non-finite) surface area";
            float normalization = 0.0f;
            {
                normalization = 1.0f / sum;
                for (uint32_t t = 1; t <= m.n_triangles; ++t) out.emitter_cdf[rec.cdf_offset + t] *= normalization;
                out.emitter_cdf[rec.cdf_offset + m.n_triangles] = 1.0f;
            }
            rec.inv_area = normalization;
            out.emitters.push_back(mi);
        }
        out.meshes.push_back(rec);
        vOff += m.n_vertices; tOff += m.n_triangles;
    }

    /* camera, src/perspective.cpp:41-74 */
    const nori_camera_desc &c = desc.camera;
    CameraRec &cam = out.camera;
    cam.width = c.width; cam.height = c.height;
    cam.inv_w = 1.0f / (float) c.width; cam.inv_h = 1.0f / (float) c.height;
    cam.near_clip = c.near_clip; cam.far_clip = c.far_clip;
    std::memcpy(cam.camera_to_world, c.to_world, sizeof(float) * 16);
    {
        const float aspect = c.width / (float) c.height;
        const float recip = 1.0f / (c.far_clip - c.near_clip);
        const float cot = 1.0f / std::tan((c.fov / 2.0f) * (kPi / 180.0f));
        float persp[16] = {0};
        persp[0] = cot; persp[5] = cot;
        persp[10] = c.far_clip * recip; persp[11] = -c.near_clip * c.far_clip * recip;
        persp[14] = 1.0f;
        /* DiagonalMatrix(-0.5, -0.5 aspect, 1) * Translation(-1, -1/aspect, 0) */
        const float sx = -0.5f, sy = -0.5f * aspect, sz = 1.0f;
        const float tx = -1.0f, ty = -1.0f / aspect, tz = 0.0f;
        float A[16] = {0};
        A[0] = sx; A[5] = sy; A[10] = sz; A[15] = 1.0f;
        A[3] = sx * tx; A[7] = sy * ty; A[11] = sz * tz;
        float M[16];
        mat4_mul(A, persp, M);
        if (!mat4_inverse(M, cam.sample_to_camera)) return "singular camera projection";
    }

    /* filter, src/block.cpp:18-27 */
    const nori_rfilter_desc &rf = desc.rfilter;
    if (rf.type < 0 || rf.type > 3) return "unknown reconstruction filter";
    FilterRec &fl = out.filter;
    fl.radius = rf.type == NORI_RFILTER_TENT ? 1.0f : (rf.type == NORI_RFILTER_BOX ? 0.5f : rf.radius);
    if (!(fl.radius > 0.0f)) return "reconstruction filter radius must be positive";
    fl.border = (int) std::ceil(fl.radius - 0.5f);
    for (int i = 0; i < kFilterRes; ++i) {
        const float pos = (fl.radius * i) / kFilterRes;
        fl.table[i] = rfilter_eval(rf, fl.radius, pos);
    }
    fl.table[kFilterRes] = 0.0f;
    fl.lookup_factor = kFilterRes / fl.radius;

    out.integrator.type = desc.integrator.type;
    if (out.integrator.type < 0 || out.integrator.type > 6) return "unknown integrator type";
    for (int k = 0; k < 3; ++k) { out.integrator.position[k] = desc.integrator.position[k]; out.integrator.energy[k] = desc.integrator.energy[k]; }
    return std::string();
}

/* ------------------------------------------------------------- SAH builder */
namespace {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int i = 0; i < 3; ++i) { mn[i] = std::numeric_limits<float>::infinity(); mx[i] = -std::numeric_limits<float>::infinity(); } }
    void grow(const Box &b) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], b.mn[i]); mx[i] = std::max(mx[i], b.mx[i]); } }
    void grow(const float *p) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
    float area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct BuildNode {
    Box box;
    int32_t left = -1, right = -1;     /* build-node ids; -1 for leaves */
    uint32_t first = 0, count = 0;
    uint32_t depth = 0;
};

constexpr int kBins = 32;
static uint32_t kLeafTarget = 4;      /* NORI_HIP_SAH_LEAF overrides (experiments) */
constexpr float kCostNode = 1.0f;
constexpr float kSbvhMargin = 0.99f;            /* a spatial split must beat the object split's SAH estimate by this factor: the estimate is greedy, and a split
                                                   that wins by a fraction of a percent pays for its duplicated references with a worse tree below */
constexpr int kReinsertPasses = 10;             /* batches of optimize_tree_reinsertion (2 % of the inner nodes each) */
constexpr float kSbvhBudgetDefault = 0.3f;      /* spatial splits: references beyond one per triangle, as a fraction of the triangle count (build_tree_spatial) */
static float kCostTri = 1.0f;      /* relative cost of one triangle test; NORI_HIP_SAH_TRI_COST overrides (experiments) */

/* ---- spatial splits (Stich, Friedrich, Dietrich: "Spatial Splits in Bounding Volume Hierarchies", HPG 2009)
   A reference = a triangle and the box of the part of it a subtree is responsible for.  A spatial split cuts the references that
   straddle its plane in two -- the triangle then hangs in two leaves.  That is safe for Accel::rayIntersect (src/accel.cpp:23-43):
   a leaf step tests the WHOLE triangle (Mesh::rayIntersect, src/mesh.cpp:39-76, same operands, same t, u, v from either leaf), and
   the tie rule of the scan -- equal t: the larger index wins -- replaces a hit by itself.  What a reference's box must hold: every
   point of the triangle inside the cut, padded as the whole triangle's box is (tri_box_pad: the slab test and the triangle test
   round differently). */
struct Ref { Box box; uint32_t tri; };

/* the part of polygon `in` (n vertices) on one side of the plane x[axis] = pos; returns the number of vertices in `out` */
static int clip_poly(const double (*in)[3], int n, int axis, double pos, bool keep_above, double (*out)[3]) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const double *a = in[i], *b = in[(i + 1) % n];
        const double da = keep_above ? a[axis] - pos : pos - a[axis], db = keep_above ? b[axis] - pos : pos - b[axis];
        if (da >= 0.0) { for (int k = 0; k < 3; ++k) out[m][k] = a[k]; ++m; }
        if ((da > 0.0 && db < 0.0) || (da < 0.0 && db > 0.0)) {
            const double t = da / (da - db);
            for (int k = 0; k < 3; ++k) out[m][k] = a[k] + t * (b[k] - a[k]);
            out[m][axis] = pos;
            ++m;
        }
    }
    return m;
}

static float round_down(double v) { float f = (float) v; if ((double) f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
static float round_up(double v) { float f = (float) v; if ((double) f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }

/* box of (triangle within lo <= x[axis] <= hi), padded by `pad`, within `within`; false: nothing of the triangle lies there */
static bool clip_tri_box(const double tri[3][3], int axis, double lo, double hi, float pad, const Box &within, Box &out) {
    double a[8][3], b[8][3];
    int n = clip_poly(tri, 3, axis, lo, true, a);
    if (n > 0) n = clip_poly(a, n, axis, hi, false, b);
    if (n <= 0) return false;
    double mn[3] = {b[0][0], b[0][1], b[0][2]}, mx[3] = {b[0][0], b[0][1], b[0][2]};
    for (int i = 1; i < n; ++i) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], b[i][k]); mx[k] = std::max(mx[k], b[i][k]); }
    for (int k = 0; k < 3; ++k) {
        out.mn[k] = std::max(within.mn[k], round_down(mn[k]) - pad);
        out.mx[k] = std::min(within.mx[k], round_up(mx[k]) + pad);
        if (!(out.mn[k] <= out.mx[k])) return false;
    }
    return true;
}

} // namespace

/* Top-down SAH with object AND spatial splits (single thread: the scenes it serves are small).  Fills `bn` (node 0 = the root)
   and `prim` = the triangle of every leaf reference, leaf after leaf in depth-first order -- what the flattening below reads.
   budget: references the tree may hold beyond one per triangle, as a fraction of the triangle count. */
static void build_tree_spatial(const HostScene &sc, const std::vector<Box> &boxes, const std::vector<float> &padOf, uint32_t max_depth_limit,
                               float budget, std::vector<BuildNode> &bn, std::vector<uint32_t> &prim, uint32_t &n_spatial) {
    const uint32_t n = (uint32_t) boxes.size();
    auto log2ceil = [](uint32_t v) { uint32_t r = 0; while ((1u << r) < v) ++r; return r; };
    auto triangle = [&](uint32_t g, double t[3][3]) {
        for (int k = 0; k < 3; ++k) { const f4 &p = sc.positions[sc.indices[3 * (size_t) g + k]]; t[k][0] = p.x; t[k][1] = p.y; t[k][2] = p.z; }
    };
    auto centre = [](const Box &b, int ax) { return 0.5f * (b.mn[ax] + b.mx[ax]); };
    struct Task { uint32_t node; std::vector<Ref> refs; };
    std::vector<Task> todo;
    bn.clear(); prim.clear(); n_spatial = 0;
    bn.emplace_back();
    { Task t; t.node = 0; t.refs.resize(n); for (uint32_t g = 0; g < n; ++g) { t.refs[g].box = boxes[g]; t.refs[g].tri = g; } todo.push_back(std::move(t)); }
    long long spare = (long long) (budget * (float) n);       /* references still to be handed out */
    float rootArea = 0.0f;
    const float margin = std::getenv("NORI_HIP_SBVH_MARGIN") ? (float) std::atof(std::getenv("NORI_HIP_SBVH_MARGIN")) : kSbvhMargin;
    constexpr float kMinOverlap = 0.01f;                     /* ... and its extent along the object split's axis, relative to the node's */
    constexpr float kAlpha = 1e-5f;                          /* overlap of the object split's children, relative to the root: below it no spatial split is tried */

    while (!todo.empty()) {
        Task task = std::move(todo.back()); todo.pop_back();
        std::vector<Ref> &refs = task.refs;
        const uint32_t count = (uint32_t) refs.size(), depth = bn[task.node].depth;
        Box nb, cb; nb.reset(); cb.reset();
        for (const Ref &r : refs) { nb.grow(r.box); const float c[3] = {centre(r.box, 0), centre(r.box, 1), centre(r.box, 2)}; cb.grow(c); }
        bn[task.node].box = nb;
        if (task.node == 0) rootArea = std::max(nb.area(), 1e-30f);
        auto makeLeaf = [&] { bn[task.node].first = (uint32_t) prim.size(); bn[task.node].count = count; for (const Ref &r : refs) prim.push_back(r.tri); };
        if (count <= 1) { makeLeaf(); continue; }
        const uint32_t needBalanced = log2ceil((count + kLeafTarget - 1) / kLeafTarget) + 1;
        const bool forceMedian = depth + needBalanced + 1 >= max_depth_limit;

        /* object split: bins over the centres of the references' boxes */
        float objCost = std::numeric_limits<float>::infinity(); int objAxis = -1, objBin = -1; Box objL, objR;
        float scale[3] = {0, 0, 0};
        if (!forceMedian)
            for (int ax = 0; ax < 3; ++ax) {
                if (!(cb.mx[ax] > cb.mn[ax])) continue;
                scale[ax] = kBins / (cb.mx[ax] - cb.mn[ax]);
                Box bb[kBins]; uint32_t bc[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
                for (const Ref &r : refs) {
                    int b = (int) ((centre(r.box, ax) - cb.mn[ax]) * scale[ax]);
                    b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
                    bb[b].grow(r.box); bc[b]++;
                }
                Box rb[kBins]; uint32_t rc[kBins]; Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += bc[b]; rb[b] = acc; rc[b] = c; }
                Box la; la.reset(); uint32_t lc = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    la.grow(bb[b]); lc += bc[b];
                    if (lc == 0 || rc[b + 1] == 0) continue;
                    const float cost = la.area() * (float) lc + rb[b + 1].area() * (float) rc[b + 1];
                    if (cost < objCost) { objCost = cost; objAxis = ax; objBin = b; objL = la; objR = rb[b + 1]; }
                }
            }

        /* spatial split: bins over the node's box; a reference enters the bin its box starts in, leaves the bin it ends in, and
           grows every bin it touches by the part of its triangle inside */
        float spCost = std::numeric_limits<float>::infinity(); int spAxis = -1; float spPos = 0.0f; uint32_t spL = 0, spR = 0;
        bool trySpatial = !forceMedian && spare > 0 && objAxis >= 0;
        if (trySpatial) {
            Box ov;
            for (int k = 0; k < 3; ++k) { ov.mn[k] = std::max(objL.mn[k], objR.mn[k]); ov.mx[k] = std::min(objL.mx[k], objR.mx[k]); }
            /* (the second test: boxes are padded, so the children of ANY object split overlap by a sliver whose faces have the area
               of the node's cross-section -- Stich's surface-area criterion alone sends every node through the clipping below) */
            trySpatial = ov.mn[0] <= ov.mx[0] && ov.mn[1] <= ov.mx[1] && ov.mn[2] <= ov.mx[2] && ov.area() / rootArea > kAlpha &&
                         ov.mx[objAxis] - ov.mn[objAxis] > kMinOverlap * (nb.mx[objAxis] - nb.mn[objAxis]);
        }
        if (trySpatial)
            for (int ax = 0; ax < 3; ++ax) {
                const float lo = nb.mn[ax], ext = nb.mx[ax] - nb.mn[ax];
                if (!(ext > 0.0f)) continue;
                if (ax != objAxis) continue;      /* the axis the object split chose: on the pa5 table scene the trees of all three axes are no better
                                                     (node / triangle tests per ray 11.01 / 5.60 against 11.01 / 5.58) and take twice as long to build */
                const float inv = kBins / ext;
                Box bb[kBins]; uint32_t enter[kBins], leave[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); enter[b] = leave[b] = 0; }
                auto binOf = [&](float v) { int b = (int) ((v - lo) * inv); return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b); };
                auto plane = [&](int b) { return lo + ext * ((float) b / (float) kBins); };      /* lower plane of bin b */
                for (const Ref &r : refs) {
                    const int b0 = binOf(r.box.mn[ax]), b1 = binOf(r.box.mx[ax]);
                    enter[b0]++; leave[b1]++;
                    if (b0 == b1) { bb[b0].grow(r.box); continue; }
                    /* (the triangle clipped per bin, not its box cut per bin: 11.79 / 7.36 tests per ray with boxes) */
                    double t[3][3]; triangle(r.tri, t);
                    for (int b = b0; b <= b1; ++b) {
                        Box part;
                        const double plo = b == b0 ? -1e300 : (double) plane(b), phi = b == b1 ? 1e300 : (double) plane(b + 1);
                        if (clip_tri_box(t, ax, plo, phi, padOf[r.tri], r.box, part)) bb[b].grow(part);
                    }
                }
                Box rb[kBins]; uint32_t rc[kBins]; Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += leave[b]; rb[b] = acc; rc[b] = c; }
                Box la; la.reset(); uint32_t lc = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    la.grow(bb[b]); lc += enter[b];
                    if (lc == 0 || rc[b + 1] == 0 || lc >= count || rc[b + 1] >= count) continue;      /* a split must shrink both sides */
                    const float cost = la.area() * (float) lc + rb[b + 1].area() * (float) rc[b + 1];
                    if (cost < spCost) { spCost = cost; spAxis = ax; spPos = plane(b + 1); spL = lc; spR = rc[b + 1]; }
                }
            }

        const float area = nb.area();
        const bool spatial = spAxis >= 0 && spCost < margin * objCost && (long long) (spL + spR) - (long long) count <= spare;
        const float bestCost = spatial ? spCost : objCost;
        const bool haveSplit = spatial || objAxis >= 0;
        if (!forceMedian && haveSplit && count <= kLeafTarget && !(kCostNode * area + kCostTri * bestCost < kCostTri * area * (float) count)) { makeLeaf(); continue; }

        std::vector<Ref> L, R;
        if (spatial) {
            L.reserve(spL); R.reserve(spR);
            for (const Ref &r : refs) {
                if (r.box.mx[spAxis] <= spPos) { L.push_back(r); continue; }
                if (r.box.mn[spAxis] >= spPos) { R.push_back(r); continue; }
                double t[3][3]; triangle(r.tri, t);
                Ref a = r, b = r;
                const bool inL = clip_tri_box(t, spAxis, -1e300, (double) spPos, padOf[r.tri], r.box, a.box);
                const bool inR = clip_tri_box(t, spAxis, (double) spPos, 1e300, padOf[r.tri], r.box, b.box);
                if (inL) L.push_back(a);
                if (inR) R.push_back(b);
                if (!inL && !inR) L.push_back(r);      /* (cannot happen: the triangle lies somewhere) */
            }
            if (L.empty() || R.empty() || L.size() >= count || R.size() >= count) { L.clear(); R.clear(); }      /* fall through to the object split */
            else { spare -= (long long) (L.size() + R.size()) - (long long) count; ++n_spatial;
                   if (std::getenv("NORI_HIP_SBVH_DEBUG")) fprintf(stderr, "[sbvh] depth %u count %u: axis %d pos %g, %zu + %zu refs, cost %g (object %g on axis %d), node area %g\n", depth, count, spAxis, spPos, L.size(), R.size(), spCost, objCost, objAxis, area); }
        }
        if (L.empty() && !forceMedian && objAxis >= 0) {
            for (const Ref &r : refs) {
                int b = (int) ((centre(r.box, objAxis) - cb.mn[objAxis]) * scale[objAxis]);
                b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
                (b <= objBin ? L : R).push_back(r);
            }
            if (L.empty() || R.empty()) { L.clear(); R.clear(); }
        }
        if (L.empty()) {
            if (count <= kLeafTarget) { makeLeaf(); continue; }
            int axis = 0;
            { const float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2]; axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2); }
            const size_t mid = count / 2;
            std::nth_element(refs.begin(), refs.begin() + mid, refs.end(), [&](const Ref &a, const Ref &b) { return centre(a.box, axis) < centre(b.box, axis); });
            L.assign(refs.begin(), refs.begin() + mid); R.assign(refs.begin() + mid, refs.end());
        }
        const int32_t l = (int32_t) bn.size(); bn.emplace_back();
        const int32_t r = (int32_t) bn.size(); bn.emplace_back();
        bn[task.node].left = l; bn[task.node].right = r;
        bn[l].depth = bn[r].depth = depth + 1;
        { Task t; t.node = (uint32_t) r; t.refs = std::move(R); todo.push_back(std::move(t)); }
        { Task t; t.node = (uint32_t) l; t.refs = std::move(L); todo.push_back(std::move(t)); }      /* left first: leaves in depth-first order */
    }
}

/* Insertion-based optimisation of a finished tree (Bittner, Hapala, Havran: "Fast Insertion-Based Optimization of Bounding Volume
   Hierarchies", CGF 2013): top-down SAH is greedy -- a split is never revisited.  Here the inner nodes that cost the most for what they
   hold are taken out of the tree one batch at a time and their two subtrees re-inserted where the tree's SAH cost grows least
   (branch-and-bound search from the root).  Leaves stay as they are (their triangles, their order); node 0 stays the root.
   depth_limit: no leaf may end up deeper (the traversal stack; for small trees the LDS-only stack of wf_extend). */
/* Stack entries a walk of the WIDE form of this tree can need (build_bvh_sah's collapse: every node adopts grandchildren -- the
   inner child of largest area first -- until it has four children; three pushes per wide level).  The same adoption rule as the
   flattening below, depths only: what decides whether an optimised tree is still walkable before it replaces the one that was. */
static uint32_t wide_stack_depth(const std::vector<BuildNode> &bn) {
    if (bn.empty() || bn[0].left < 0) return 0;
    uint32_t wideDepth = 0;
    std::vector<std::pair<int32_t, uint32_t>> todo;      /* (build node, wide depth) */
    todo.emplace_back(0, 0u);
    while (!todo.empty()) {
        const int32_t b = todo.back().first; const uint32_t depth = todo.back().second; todo.pop_back();
        int32_t kid[4] = {bn[b].left, bn[b].right, -1, -1}; int n = 2;
        while (n < 4) {
            int best = -1; float bestArea = -1.0f;
            for (int k = 0; k < n; ++k) {
                if (bn[kid[k]].left < 0) continue;
                float a = bn[kid[k]].box.area();
                if (!(a < kInf)) a = kInf;
                if (a > bestArea) { bestArea = a; best = k; }
            }
            if (best < 0) break;
            const int32_t c = kid[best];
            kid[best] = bn[c].left; kid[n++] = bn[c].right;
        }
        for (int k = 0; k < n; ++k) {
            if (bn[kid[k]].left < 0) wideDepth = std::max(wideDepth, depth + 1);
            else todo.emplace_back(kid[k], depth + 1);
        }
    }
    return 3 * wideDepth;
}

static void optimize_tree_reinsertion(std::vector<BuildNode> &bn, uint32_t depth_limit, int passes, float batch_frac) {
    const int n = (int) bn.size();
    if (n < 7) return;
    std::vector<int32_t> parent((size_t) n, -1);
    std::vector<uint32_t> height((size_t) n, 0);
    auto isLeaf = [&](int b) { return bn[(size_t) b].left < 0; };
    auto areaOf = [&](int b) { const float a = bn[(size_t) b].box.area(); return std::isfinite(a) ? a : 0.0f; };
    for (int b = 0; b < n; ++b) if (!isLeaf(b)) { parent[(size_t) bn[(size_t) b].left] = b; parent[(size_t) bn[(size_t) b].right] = b; }
    {
        std::vector<int> order; order.reserve((size_t) n); order.push_back(0);
        for (size_t i = 0; i < order.size(); ++i) { const int b = order[i]; if (!isLeaf(b)) { order.push_back(bn[(size_t) b].left); order.push_back(bn[(size_t) b].right); } }
        for (size_t i = order.size(); i-- > 0; ) { const int b = order[i]; if (!isLeaf(b)) height[(size_t) b] = 1 + std::max(height[(size_t) bn[(size_t) b].left], height[(size_t) bn[(size_t) b].right]); }
    }
    auto depthOf = [&](int b) { uint32_t d = 0; while (parent[(size_t) b] >= 0) { b = parent[(size_t) b]; ++d; } return d; };
    /* boxes and heights from b up; returns by how much the inner nodes' areas grew (the tree's SAH cost in units of area) */
    auto refit = [&](int b) {
        double delta = 0.0;
        while (b >= 0) {
            BuildNode &nd = bn[(size_t) b];
            const float before = areaOf(b);
            Box nb = bn[(size_t) nd.left].box; nb.grow(bn[(size_t) nd.right].box);
            nd.box = nb;
            height[(size_t) b] = 1 + std::max(height[(size_t) nd.left], height[(size_t) nd.right]);
            delta += (double) areaOf(b) - (double) before;
            b = parent[(size_t) b];
        }
        return delta;
    };
    /* x and its parent leave the tree: the sibling moves up.  Returns the freed slot (x's old parent); `delta` += the change of cost */
    auto detach = [&](int x, double &delta) {
        const int Q = parent[(size_t) x], BP = parent[(size_t) Q];
        const int B = bn[(size_t) Q].left == x ? bn[(size_t) Q].right : bn[(size_t) Q].left;
        (bn[(size_t) BP].left == Q ? bn[(size_t) BP].left : bn[(size_t) BP].right) = B;
        parent[(size_t) B] = BP; parent[(size_t) x] = -1; parent[(size_t) Q] = -1;
        delta -= (double) areaOf(Q);
        delta += refit(BP);
        return Q;
    };
    /* slot Q takes B's place and holds (B, x) */
    auto attach = [&](int x, int B, int Q, double &delta) {
        const int BP = parent[(size_t) B];
        (bn[(size_t) BP].left == B ? bn[(size_t) BP].left : bn[(size_t) BP].right) = Q;
        parent[(size_t) Q] = BP;
        bn[(size_t) Q].left = B; bn[(size_t) Q].right = x;
        parent[(size_t) B] = Q; parent[(size_t) x] = Q;
        bn[(size_t) Q].box = bn[(size_t) B].box;      /* (so that refit's "before" of Q is B's area: Q's own area is added below) */
        delta += (double) areaOf(B);
        delta += refit(Q);
    };
    auto unionArea = [&](const Box &a, const Box &b) { Box u = a; u.grow(b); const float x = u.area(); return std::isfinite(x) ? x : std::numeric_limits<float>::infinity(); };

    struct Cand { float induced; int node; bool operator<(const Cand &o) const { return induced > o.induced; } };      /* min-heap on the induced cost */
    std::vector<Cand> heap;
    /* best sibling for the detached subtree x: minimise (area added to the ancestors) + area of the new parent */
    auto bestSibling = [&](int x) {
        const float ax = areaOf(x);
        float best = std::numeric_limits<float>::infinity(); int bestNode = -1;
        heap.clear();
        heap.push_back({0.0f, bn[0].left}); std::push_heap(heap.begin(), heap.end());
        heap.push_back({0.0f, bn[0].right}); std::push_heap(heap.begin(), heap.end());
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const Cand c = heap.back(); heap.pop_back();
            if (c.induced + ax >= best) break;      /* the heap is ordered by induced cost: nothing left can beat the best */
            const float direct = unionArea(bn[(size_t) c.node].box, bn[(size_t) x].box);
            const float total = c.induced + direct;
            if (total < best && depthOf(c.node) + 1 + std::max(height[(size_t) c.node], height[(size_t) x]) <= depth_limit) { best = total; bestNode = c.node; }
            const float below = total - areaOf(c.node);
            if (!isLeaf(c.node) && below + ax < best) {
                heap.push_back({below, bn[(size_t) c.node].left}); std::push_heap(heap.begin(), heap.end());
                heap.push_back({below, bn[(size_t) c.node].right}); std::push_heap(heap.begin(), heap.end());
            }
        }
        return bestNode;
    };

    unsigned long long lcg = 0x9e3779b97f4a7c15ull;
    int idle = 0;
    for (int pass = 0; pass < passes; ++pass) {
        /* the batch: inner nodes (not the root, not its children -- the root's slot stays where it is) by what they cost for what they hold */
        std::vector<std::pair<float, int>> rank;
        for (int b = 1; b < n; ++b) {
            if (isLeaf(b) || parent[(size_t) b] <= 0) continue;
            const float a = areaOf(b), al = areaOf(bn[(size_t) b].left), ar = areaOf(bn[(size_t) b].right);
            if (!(a > 0.0f)) continue;
            const float m = a * (a / std::max(0.5f * (al + ar), 1e-30f)) * (a / std::max(std::min(al, ar), 1e-30f));
            rank.emplace_back(-m, b);
        }
        if (rank.empty()) break;
        const size_t take = std::min(rank.size(), std::max<size_t>(1, (size_t) (batch_frac * (float) rank.size())));
        if (pass & 1) {      /* every other pass a random batch (a fixed sequence: the tree does not depend on the run): the ranking alone
                                keeps proposing the nodes it could not improve */
            for (size_t k = 0; k < take; ++k) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(rank[k], rank[k + (size_t) ((lcg >> 33) % (rank.size() - k))]);
            }
        } else std::partial_sort(rank.begin(), rank.begin() + (std::ptrdiff_t) take, rank.end());
        size_t improved = 0;
        for (size_t k = 0; k < take; ++k) {
            const int N = rank[k].second;
            const int P = parent[(size_t) N];
            if (P <= 0 || isLeaf(N) || parent[(size_t) P] < 0) continue;      /* an earlier re-insertion of this pass moved it next to the root */
            const int S = bn[(size_t) P].left == N ? bn[(size_t) P].right : bn[(size_t) P].left;
            int X[2] = {bn[(size_t) N].left, bn[(size_t) N].right};
            if (areaOf(X[0]) < areaOf(X[1])) std::swap(X[0], X[1]);
            double delta = 0.0;
            const int slotP = detach(N, delta);            /* = P; S has moved up */
            delta -= (double) areaOf(N);                    /* N dissolves: its two subtrees are free */
            parent[(size_t) X[0]] = parent[(size_t) X[1]] = -1;
            const int B0 = bestSibling(X[0]);
            bool ok = B0 >= 0;
            if (ok) attach(X[0], B0, slotP, delta);
            const int B1 = ok ? bestSibling(X[1]) : -1;
            ok = ok && B1 >= 0;
            if (ok) attach(X[1], B1, N, delta);
            if (ok && delta < -1e-7 * (double) areaOf(0)) { ++improved; continue; }
            /* no better (or no place within the depth limit): back to where they were */
            double undo = 0.0;
            if (B1 >= 0 && B0 >= 0) (void) detach(X[1], undo);      /* frees slot N */
            if (B0 >= 0) (void) detach(X[0], undo);                 /* frees slot P */
            bn[(size_t) N].left = X[0]; bn[(size_t) N].right = X[1];
            parent[(size_t) X[0]] = parent[(size_t) X[1]] = N;
            { Box nb = bn[(size_t) X[0]].box; nb.grow(bn[(size_t) X[1]].box); bn[(size_t) N].box = nb; height[(size_t) N] = 1 + std::max(height[(size_t) X[0]], height[(size_t) X[1]]); }
            attach(N, S, slotP, undo);
        }
        idle = improved == 0 ? idle + 1 : 0;
        if (idle >= 4) break;      /* two ranked and two random batches in a row found nothing */
    }
    /* depths for the flattening */
    std::vector<int> st; st.push_back(0); bn[0].depth = 0;
    while (!st.empty()) {
        const int b = st.back(); st.pop_back();
        if (isLeaf(b)) continue;
        bn[(size_t) bn[(size_t) b].left].depth = bn[(size_t) bn[(size_t) b].right].depth = bn[(size_t) b].depth + 1;
        st.push_back(bn[(size_t) b].left); st.push_back(bn[(size_t) b].right);
    }
}

std::string build_bvh_sah(const HostScene &sc, uint32_t max_depth_limit, HostBvh &out, bool wide) {
    if (const char *e = std::getenv("NORI_HIP_SAH_TRI_COST")) kCostTri = std::max(0.1f, (float) std::atof(e));
    if (const char *e = std::getenv("NORI_HIP_SAH_LEAF")) kLeafTarget = (uint32_t) std::min(kMaxLeafTris, std::max(1, std::atoi(e)));
    const auto t0 = std::chrono::steady_clock::now();
    out = HostBvh();
    const uint32_t n = (uint32_t) sc.tri_mesh.size();
    if (n == 0) {
        /* traversal returns before touching nodes when there are no triangles */
        out.nodes.resize(kNodeQuads); out.tris.resize(kPairQuads);
        std::memset(out.nodes.data(), 0, sizeof(f4) * kNodeQuads);
        std::memset(out.tris.data(), 0, sizeof(f4) * kPairQuads);
        out.root = 0; out.n_nodes = 1;
        return std::string();
    }

    std::vector<Box> boxes(n);
    std::vector<uint8_t> isUnbounded(n, 0);      /* numerically collinear triangles: rt_types.h, tri_box_pad */
    uint32_t nUnbounded = 0;
    std::vector<float> cent(3 * (size_t) n), padOf(n);
    Box sceneBox; sceneBox.reset();
    for (uint32_t t = 0; t < n; ++t) {
        Box b; b.reset();
        for (int k = 0; k < 3; ++k) { const f4 &p = sc.positions[sc.indices[3 * (size_t) t + k]]; const float q[3] = {p.x, p.y, p.z}; b.grow(q); }
        boxes[t] = b; sceneBox.grow(b);
    }
    /* Pad every triangle box: the slab test must never cull a leaf whose
       Moeller-Trumbore test (rounded differently) accepts the ray. */
    {
        const float dx = sceneBox.mx[0] - sceneBox.mn[0], dy = sceneBox.mx[1] - sceneBox.mn[1], dz = sceneBox.mx[2] - sceneBox.mn[2];
        const float pad = box_pad_rel() * std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-30f;
        for (uint32_t t = 0; t < n; ++t) {
            const uint32_t *id = &sc.indices[3 * (size_t) t];
            const f3 p0 = xyz(sc.positions[id[0]]), p1 = xyz(sc.positions[id[1]]), p2 = xyz(sc.positions[id[2]]);
            bool unbounded;
            const float padT = tri_box_pad(p1 - p0, p2 - p0, pad, unbounded);      /* slivers: rt_types.h */
            isUnbounded[t] = unbounded ? 1 : 0; padOf[t] = padT;
            nUnbounded += unbounded ? 1u : 0u;
            for (int k = 0; k < 3; ++k) {
                if (unbounded) { boxes[t].mn[k] = -kBoxInf; boxes[t].mx[k] = kBoxInf; }
                else { boxes[t].mn[k] -= padT; boxes[t].mx[k] += padT; }
                cent[3 * (size_t) t + k] = 0.5f * (boxes[t].mn[k] + boxes[t].mx[k]);
            }
        }
    }

    /* bounded triangles first: the root splits them from the unbounded ones, whose subtree (all boxes
       infinite, median splits) every ray walks completely */
    std::vector<uint32_t> prim(n);
    {
        uint32_t a = 0, b = n - nUnbounded;
        for (uint32_t t = 0; t < n; ++t) { if (isUnbounded[t]) prim[b++] = t; else prim[a++] = t; }
    }
    const uint32_t nBounded = n - nUnbounded;
    const bool timing = std::getenv("NORI_HIP_BUILD_TIMING") != nullptr;
    auto lap = [&](const char *what) { if (timing) fprintf(stderr, "[sah] %-10s %.0f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count()); };
    lap("boxes");
    /* Top-down binned SAH.  Subtrees are independent once their triangle ranges are disjoint, so the
       build runs on all host cores: the few nodes above kTaskTris triangles are split one after the
       other with the binning itself spread over the threads, everything below becomes a task for the
       pool.  Every reduction is exact (min / max / counts), so the tree does not depend on the number of
       threads. */
    const uint32_t kTaskTris = 1u << 16;
    uint32_t nThreads = n < (1u << 18) ? 1u : std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (const char *e = std::getenv("NORI_HIP_BUILD_THREADS")) nThreads = (uint32_t) std::max(1, std::min(64, std::atoi(e)));

    auto log2ceil = [](uint32_t v) { uint32_t r = 0; while ((1u << r) < v) ++r; return r; };

    /* run fn(slice_begin, slice_end, slice_index) over [first, first + count) on `threads` threads */
    auto parallelFor = [](uint32_t first, uint32_t count, uint32_t threads, auto &&fn) {
        if (threads <= 1 || count < (1u << 16)) { fn(first, first + count, 0u); return; }
        std::vector<std::thread> pool;
        for (uint32_t k = 1; k < threads; ++k)
            pool.emplace_back([&fn, first, count, threads, k] { fn(first + (uint32_t) ((uint64_t) count * k / threads), first + (uint32_t) ((uint64_t) count * (k + 1) / threads), k); });
        fn(first, first + (uint32_t) ((uint64_t) count / threads), 0u);
        for (std::thread &t : pool) t.join();
    };

    struct Bins { Box bb[3][kBins]; uint32_t bc[3][kBins]; };

    /* box of a node and its split position: returns false when the node stays a leaf */
    auto splitNode = [&](BuildNode &node, bool isRoot, uint32_t threads, uint32_t &mid) -> bool {
        const uint32_t first = node.first, count = node.count, depth = node.depth;
        const uint32_t slices = (threads <= 1 || count < (1u << 16)) ? 1u : threads;
        Box nbs[64], cbs[64];
        parallelFor(first, count, threads, [&](uint32_t lo, uint32_t hi, uint32_t k) {
            Box nb, cb; nb.reset(); cb.reset();
            for (uint32_t i = lo; i < hi; ++i) { nb.grow(boxes[prim[i]]); cb.grow(&cent[3 * (size_t) prim[i]]); }
            nbs[k] = nb; cbs[k] = cb;
        });
        Box nb = nbs[0], cb = cbs[0];
        for (uint32_t k = 1; k < slices; ++k) { nb.grow(nbs[k]); cb.grow(cbs[k]); }
        node.box = nb;
        if (count <= 1) return false;
        if (isRoot && nUnbounded > 0 && nBounded > 0) { mid = nBounded; return true; }

        /* depth budget: once the remaining levels only just suffice for a
           balanced split, stop using SAH */
        const uint32_t needBalanced = log2ceil((count + kLeafTarget - 1) / kLeafTarget) + 1;
        const bool forceMedian = depth + needBalanced + 1 >= max_depth_limit;

        int axis = 0;
        { float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
          axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2); }
        mid = first;
        bool split = false;

        if (!forceMedian) {
            float scale[3]; bool usable[3];
            for (int ax = 0; ax < 3; ++ax) { usable[ax] = cb.mx[ax] > cb.mn[ax]; scale[ax] = usable[ax] ? kBins / (cb.mx[ax] - cb.mn[ax]) : 0.0f; }
            Bins one; std::vector<Bins> many; if (slices > 1) many.resize(slices);
            Bins *part = slices > 1 ? many.data() : &one;
            parallelFor(first, count, threads, [&](uint32_t lo, uint32_t hi, uint32_t k) {
                Bins &B = part[k];
                for (int ax = 0; ax < 3; ++ax) for (int b = 0; b < kBins; ++b) { B.bb[ax][b].reset(); B.bc[ax][b] = 0; }
                for (uint32_t i = lo; i < hi; ++i) {
                    const uint32_t g = prim[i];
                    for (int ax = 0; ax < 3; ++ax) {
                        if (!usable[ax]) continue;
                        int b = (int) ((cent[3 * (size_t) g + ax] - cb.mn[ax]) * scale[ax]);
                        b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
                        B.bb[ax][b].grow(boxes[g]); B.bc[ax][b]++;
                    }
                }
            });
            for (uint32_t k = 1; k < slices; ++k)
                for (int ax = 0; ax < 3; ++ax) for (int b = 0; b < kBins; ++b) { part[0].bb[ax][b].grow(part[k].bb[ax][b]); part[0].bc[ax][b] += part[k].bc[ax][b]; }
            float bestCost = std::numeric_limits<float>::infinity(); int bestAxis = -1, bestBin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                if (!usable[ax]) continue;
                const Box *bb = part[0].bb[ax]; const uint32_t *bc = part[0].bc[ax];
                float ra[kBins]; uint32_t rc[kBins]; Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += bc[b]; ra[b] = acc.area(); rc[b] = c; }
                Box la; la.reset(); uint32_t lc = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    la.grow(bb[b]); lc += bc[b];
                    if (lc == 0 || rc[b + 1] == 0) continue;
                    const float cost = la.area() * (float) lc + ra[b + 1] * (float) rc[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = b; }
                }
            }
            if (bestAxis >= 0) {
                const float area = nb.area();
                const float splitCost = kCostNode * area + kCostTri * bestCost;
                const float leafCost = kCostTri * area * (float) count;
                if (count > kLeafTarget || splitCost < leafCost) {
                    const float cmin = cb.mn[bestAxis], sc2 = scale[bestAxis];
                    auto it = std::partition(prim.begin() + first, prim.begin() + first + count, [&](uint32_t g) {
                        int b = (int) ((cent[3 * (size_t) g + bestAxis] - cmin) * sc2);
                        b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
                        return b <= bestBin;
                    });
                    mid = (uint32_t) (it - prim.begin());
                    split = mid > first && mid < first + count;
                }
            }
        }
        if (!split) {
            if (count <= kLeafTarget) return false;            /* stays a leaf */
            mid = first + count / 2;
            std::nth_element(prim.begin() + first, prim.begin() + mid, prim.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return cent[3 * (size_t) a + axis] < cent[3 * (size_t) b + axis]; });
        }
        return true;
    };

    /* the subtree below nodes[root], depth first, appended to `nodes` */
    auto buildSubtree = [&](std::vector<BuildNode> &nodes, uint32_t root, bool rootIsTreeRoot) {
        std::vector<uint32_t> todo; todo.push_back(root);
        while (!todo.empty()) {
            const uint32_t id = todo.back(); todo.pop_back();
            uint32_t mid = 0;
            BuildNode cur = nodes[id];
            const bool split = splitNode(cur, rootIsTreeRoot && id == root, 1, mid);
            nodes[id] = cur;
            if (!split) continue;
            const int32_t l = (int32_t) nodes.size(); nodes.emplace_back();
            const int32_t r = (int32_t) nodes.size(); nodes.emplace_back();
            nodes[id].left = l; nodes[id].right = r;
            nodes[l].first = cur.first; nodes[l].count = mid - cur.first; nodes[l].depth = cur.depth + 1;
            nodes[r].first = mid; nodes[r].count = cur.first + cur.count - mid; nodes[r].depth = cur.depth + 1;
            todo.push_back((uint32_t) r); todo.push_back((uint32_t) l);
        }
    };

    std::vector<BuildNode> bn;
    bn.reserve((size_t) n / 2 + 16);
    /* spatial splits: scenes the single-threaded builder serves (below 2^18 triangles), none of whose triangles has an unbounded
       box; NORI_HIP_SBVH = references the tree may hold beyond one per triangle, as a fraction of the triangle count (0: object
       splits only) */
    float sbvhBudget = kSbvhBudgetDefault;
    if (const char *e = std::getenv("NORI_HIP_SBVH")) sbvhBudget = std::max(0.0f, (float) std::atof(e));
    uint32_t nSpatial = 0;
    if (sbvhBudget > 0.0f && nThreads <= 1 && nUnbounded == 0 && n > kLeafTarget) {
        build_tree_spatial(sc, boxes, padOf, max_depth_limit, sbvhBudget, bn, prim, nSpatial);
        if (timing) fprintf(stderr, "[sah] spatial splits: %u, references %zu for %u triangles\n", nSpatial, prim.size(), n);
    } else {
    bn.emplace_back();
    bn[0].first = 0; bn[0].count = n; bn[0].depth = 0;
    if (nThreads <= 1) {
        buildSubtree(bn, 0, true);
    } else {
        /* top of the tree: big nodes one at a time, binning on all threads */
        std::vector<uint32_t> big, tasks;
        big.push_back(0);
        for (size_t q = 0; q < big.size(); ++q) {
            const uint32_t id = big[q];
            uint32_t mid = 0;
            BuildNode cur = bn[id];
            const bool split = splitNode(cur, id == 0, nThreads, mid);
            bn[id] = cur;
            if (!split) continue;
            const int32_t l = (int32_t) bn.size(); bn.emplace_back();
            const int32_t r = (int32_t) bn.size(); bn.emplace_back();
            bn[id].left = l; bn[id].right = r;
            bn[l].first = cur.first; bn[l].count = mid - cur.first; bn[l].depth = cur.depth + 1;
            bn[r].first = mid; bn[r].count = cur.first + cur.count - mid; bn[r].depth = cur.depth + 1;
            for (int32_t c : {l, r}) (bn[c].count > kTaskTris ? big : tasks).push_back((uint32_t) c);
        }
        lap("top");
        /* the subtrees, one task each; results are stitched in task order */
        std::vector<std::vector<BuildNode>> sub(tasks.size());
        std::atomic<size_t> next(0);
        auto worker = [&] {
            for (size_t k; (k = next.fetch_add(1)) < tasks.size(); ) {
                sub[k].reserve((size_t) bn[tasks[k]].count / 2 + 4);
                sub[k].push_back(bn[tasks[k]]);
                buildSubtree(sub[k], 0, false);
            }
        };
        {
            std::vector<std::thread> pool;
            for (uint32_t k = 1; k < nThreads; ++k) pool.emplace_back(worker);
            worker();
            for (std::thread &t : pool) t.join();
        }
        lap("subtrees");
        for (size_t k = 0; k < tasks.size(); ++k) {
            const int32_t base = (int32_t) bn.size() - 1;            /* local node i >= 1 -> base + i */
            auto remap = [&](BuildNode nd) { if (nd.left >= 0) { nd.left += base; nd.right += base; } return nd; };
            bn[tasks[k]] = remap(sub[k][0]);
            for (size_t i = 1; i < sub[k].size(); ++i) bn.push_back(remap(sub[k][i]));
            std::vector<BuildNode>().swap(sub[k]);
        }
    }

    }
    lap("tree");
    /* the tree optimised by re-insertion (optimize_tree_reinsertion): trees of the single-threaded builder (below 2^18 triangles)
       without unbounded boxes; kept only if the inner nodes' summed area -- the part of the SAH cost it can change -- drops by 2 %.
       (pa5 table: SAH cost 5.37 -> 3.87, node tests per ray 11.0 -> 9.6 -- the top-down builder's first splits bin the centres of a
       scene whose ground planes are ten times its objects' extent; veach 5.27 -> 4.77 / 8.35 -> 7.72.  The pa4 Cornell box: 16.31 ->
       16.29 with 3.5 % MORE node tests for the rays a closed room really sees -- below the threshold, its tree stays.)
       NORI_HIP_REINSERT = passes (0: off). */
    {
        int passes = kReinsertPasses;
        if (const char *e = std::getenv("NORI_HIP_REINSERT")) passes = std::max(0, std::atoi(e));
        if (passes > 0 && nThreads <= 1 && nUnbounded == 0 && bn.size() >= 7) {
            uint32_t deepest = 0;
            auto innerArea = [&] { double a = 0.0; for (const BuildNode &nd : bn) if (nd.left >= 0) { const float x = nd.box.area(); if (std::isfinite(x)) a += x; } return a; };
            for (const BuildNode &nd : bn) if (nd.left < 0) deepest = std::max(deepest, nd.depth);
            const double before = innerArea();
            std::vector<BuildNode> kept = bn;
            /* depth: what the tree has, at least the 15 levels wf_extend's LDS-only stack holds, never beyond the traversal stack.
               A wide (BVH4) layout is walked with three pushes per wide level, and which BVH2 levels disappear in the collapse
               follows the boxes' areas: the optimised tree is kept only if its wide form is still within the stack -- a tree
               that was walkable before the optimisation stays walkable (the one that was is kept otherwise). */
            optimize_tree_reinsertion(bn, std::min(std::max(deepest, 15u), max_depth_limit - 1), passes, 0.02f);
            const double after = innerArea();
            const bool too_deep = wide && wide_stack_depth(bn) + 1 > max_depth_limit && wide_stack_depth(kept) + 1 <= max_depth_limit;
            if (timing) fprintf(stderr, "[sah] re-insertion: inner area %.6g -> %.6g (%+.1f %%)%s\n", before, after, 100.0 * (after / before - 1.0),
                                too_deep ? " -- its wide form is deeper than the traversal stack: not kept" : after <= 0.98 * before ? "" : " -- not kept");
            if (!(after <= 0.98 * before) || too_deep) bn.swap(kept);
            lap("reinsert");
        }
    }
    const uint32_t nRefs = (uint32_t) prim.size();      /* = n unless spatial splits duplicated references */
    if (nRefs >= (1u << 28)) return "too many triangle references (limit 2^28)";
    /* leaf triangle records: pairs, leaf by leaf in prim order (rt_types.h) */
    auto isLeaf = [&](int32_t b) { return bn[b].left < 0; };
    std::vector<uint32_t> firstPair(bn.size(), 0);
    {
        /* leaves in prim order: the leaf that starts at prim position p, if any */
        std::vector<int32_t> leafAt(nRefs, -1), leaves;
        for (size_t b = 0; b < bn.size(); ++b) if (isLeaf((int32_t) b)) leafAt[bn[b].first] = (int32_t) b;
        leaves.reserve(bn.size() / 2 + 1);
        uint32_t nPairs = wide ? 1u : 0u;        /* wide trees reserve pair 0 as the all-zero pair unused child slots point to */
        for (uint32_t p = 0; p < nRefs; ++p)
            if (leafAt[p] >= 0) { const int32_t b = leafAt[p]; leaves.push_back(b); firstPair[b] = nPairs; nPairs += (bn[b].count + 1) / 2; }
        out.tris.assign((size_t) std::max<uint32_t>(nPairs, 1) * kPairQuads, f4{0.0f, 0.0f, 0.0f, 0.0f});
        parallelFor(0, (uint32_t) leaves.size(), nThreads, [&](uint32_t lo, uint32_t hi, uint32_t) {
            for (uint32_t li = lo; li < hi; ++li) {
                const int32_t b = leaves[li];
                for (uint32_t k = 0; k < ((bn[b].count + 1) / 2) * 2; ++k) {
                    f4 *q = &out.tris[(size_t) (firstPair[b] + k / 2) * kPairQuads];
                    if (k >= bn[b].count) {      /* padding: all-zero triangle, never hit */
                        const float z[3] = {0.0f, 0.0f, 0.0f};
                        pair_pack(q, (int) (k & 1u), z, z, z, kNoTriangle, kNoTriangle);
                        continue;
                    }
                    const uint32_t g = prim[bn[b].first + k];
                    const uint32_t *id = &sc.indices[3 * (size_t) g];
                    const f3 p0 = xyz(sc.positions[id[0]]), p1 = xyz(sc.positions[id[1]]), p2 = xyz(sc.positions[id[2]]);
                    const f3 e1 = p1 - p0, e2 = p2 - p0;
                    const float a0[3] = {p0.x, p0.y, p0.z}, a1[3] = {e1.x, e1.y, e1.z}, a2[3] = {e2.x, e2.y, e2.z};
                    pair_pack(q, (int) (k & 1u), a0, a1, a2, g, sc.tri_mesh[g]);
                }
            }
        });
        out.n_pairs = nPairs;
    }

    lap("pairs");
    /* flatten: one device node per inner build node, DFS pre-order */
    auto leafCode = [&](int32_t b) -> int32_t {
        return (int32_t) ~((firstPair[b] << 3) | ((bn[b].count + 1) / 2 - 1));
    };
    for (size_t b = 0; b < bn.size(); ++b)
        if (isLeaf((int32_t) b) && bn[b].count > (uint32_t) kMaxLeafTris) return "internal: leaf exceeds kMaxLeafTris";

    const int32_t sahRoot = (nUnbounded > 0 && nBounded > 0) ? bn[0].left : 0;      /* statistic over the spatial hierarchy only */
    const float rootArea = std::max(bn[sahRoot].box.area(), 1e-30f);
    double sah = 0.0;
    uint32_t maxDepth = 0, nLeaves = 0;
    if (wide && !isLeaf(0)) {
        /* WIDE nodes: every node adopts grandchildren -- the inner child of largest surface area first -- until it has
           four children; the BVH2 nodes in between disappear */
        struct WNode { int32_t kid[4]; int n; uint32_t depth; };
        std::vector<WNode> wn;
        std::vector<std::pair<int32_t, uint32_t>> st;               /* (build node, wide id) */
        auto collapse = [&](int32_t b, uint32_t depth) {
            WNode w; w.n = 2; w.kid[0] = bn[b].left; w.kid[1] = bn[b].right; w.depth = depth;
            while (w.n < 4) {
                int best = -1; float bestArea = -1.0f;
                for (int k = 0; k < w.n; ++k) {
                    if (isLeaf(w.kid[k])) continue;
                    float a = bn[w.kid[k]].box.area();
                    if (!(a < kInf)) a = kInf;
                    if (a > bestArea) { bestArea = a; best = k; }
                }
                if (best < 0) break;
                const int32_t c = w.kid[best];
                w.kid[best] = bn[c].left; w.kid[w.n++] = bn[c].right;
            }
            return w;
        };
        wn.push_back(collapse(0, 0));
        for (size_t i = 0; i < wn.size(); ++i)                      /* build order (breadth first); the device order is chosen below */
            for (int k = 0; k < wn[i].n; ++k)
                if (!isLeaf(wn[i].kid[k])) { const int32_t c = wn[i].kid[k]; wn[i].kid[k] = -2 - (int32_t) wn.size(); wn.push_back(collapse(c, wn[i].depth + 1)); st.emplace_back(c, (uint32_t) wn.size() - 1); }
        /* kid < -1: inner child, wide id = -2 - kid; kid >= 0: leaf build node */
        std::vector<int32_t> innerBuild(wn.size(), 0);
        for (auto &pr : st) innerBuild[pr.second] = pr.first;
        /* device order = depth-first pre-order: a subtree is one contiguous range of node records, so the many steps a
           ray spends near the leaves stay within a few pages / cache lines (breadth-first numbering, which scatters every
           step of a path over the whole 100+ MB array, measured 2.3x slower on the 10 M-triangle terrain) */
        std::vector<int32_t> devId(wn.size(), -1);
        {
            std::vector<int32_t> todo; todo.push_back(0);
            int32_t next = 0;
            while (!todo.empty()) {
                const int32_t w = todo.back(); todo.pop_back();
                devId[(size_t) w] = next++;
                for (int k = wn[(size_t) w].n - 1; k >= 0; --k) if (wn[(size_t) w].kid[k] < -1) todo.push_back(-2 - wn[(size_t) w].kid[k]);
            }
        }
        out.nodes.resize(wn.size() * kNodeQuads);
        uint32_t wideDepth = 0;
        for (size_t i = 0; i < wn.size(); ++i) {
            float mn[4][3], mx[4][3]; int32_t link[4];
            for (int k = 0; k < wn[i].n; ++k) {
                const int32_t kid = wn[i].kid[k];
                const int32_t b = kid < -1 ? innerBuild[(size_t) (-2 - kid)] : kid;
                for (int a = 0; a < 3; ++a) { mn[k][a] = bn[b].box.mn[a]; mx[k][a] = bn[b].box.mx[a]; }
                link[k] = kid < -1 ? devId[(size_t) (-2 - kid)] : leafCode(kid);
                if (kid >= 0) { nLeaves++; wideDepth = std::max(wideDepth, wn[i].depth + 1); sah += kCostTri * (std::isfinite(bn[b].box.area()) ? bn[b].box.area() / rootArea : 0.0f) * bn[b].count; }
            }
            const float area = i == 0 ? bn[0].box.area() : bn[innerBuild[i]].box.area();
            sah += kCostNode * (std::isfinite(area) ? area / rootArea : 0.0f);
            wide_pack(wn[i].n, mn, mx, link, &out.nodes[(size_t) devId[i] * kNodeQuads]);
        }
        out.root = 0;
        out.n_nodes = (uint32_t) wn.size();
        out.wide = true;
        maxDepth = 3 * wideDepth;            /* stack entries a walk can need: three pushes per level */
    } else
    if (isLeaf(0)) {
        out.root = leafCode(0);
        out.nodes.resize(kNodeQuads);
        std::memset(out.nodes.data(), 0, sizeof(f4) * kNodeQuads);
        out.n_nodes = 0; nLeaves = 1;
        sah = kCostTri * bn[0].count;
    } else {
        std::vector<int32_t> devId(bn.size(), -1);
        uint32_t nInner = 0;
        { std::vector<int32_t> st; st.push_back(0);
          while (!st.empty()) { int32_t b = st.back(); st.pop_back(); if (isLeaf(b)) continue; devId[b] = (int32_t) nInner++; st.push_back(bn[b].right); st.push_back(bn[b].left); } }
        out.nodes.resize((size_t) nInner * kNodeQuads);
        for (size_t b = 0; b < bn.size(); ++b) {
            const float area = bn[b].box.area();
            const float rel = std::isfinite(area) ? area / rootArea : 0.0f;
            if (isLeaf((int32_t) b)) {
                nLeaves++; maxDepth = std::max(maxDepth, bn[b].depth);
                sah += kCostTri * rel * bn[b].count;
                continue;
            }
            sah += kCostNode * rel;
            const Box &L = bn[bn[b].left].box, &R = bn[bn[b].right].box;
            const int32_t cl = isLeaf(bn[b].left) ? leafCode(bn[b].left) : devId[bn[b].left];
            const int32_t cr = isLeaf(bn[b].right) ? leafCode(bn[b].right) : devId[bn[b].right];
            node_pack(L.mn, L.mx, R.mn, R.mx, cl, cr, &out.nodes[(size_t) devId[b] * kNodeQuads]);
        }
        out.root = 0;
        out.n_nodes = nInner;
    }
    lap("flatten");
    out.n_leaves = nLeaves;
    out.max_depth = maxDepth;
    out.sah_cost = (float) sah;
    if (maxDepth + 1 > max_depth_limit) return "internal: BVH deeper than the traversal stack";
    out.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return std::string();
}

} // namespace nrt
