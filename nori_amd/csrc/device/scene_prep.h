/*
 * scene_prep.h -- load-time host code of libnori_hip: flattens a
 * nori_scene_desc into the arrays the kernels read (rt_types.h layouts) and
 * builds the SAH BVH.  Runs once per scene; not on the per-sample path.
 */
#pragma once
#include <string>
#include <vector>

#include "../../../include/nori_hip.h"
#include "rt_types.h"

namespace nrt {

struct HostScene {
    std::vector<f4> positions, normals;
    std::vector<f2> texcoords;
    std::vector<uint32_t> indices;      /* 3 per triangle, global vertex ids */
    std::vector<uint32_t> tri_mesh;     /* mesh id per global triangle       */
    std::vector<f4> shade_tris;         /* kShadeQuads per global triangle   */
    std::vector<MeshRec> meshes;
    std::vector<float> emitter_cdf;
    std::vector<uint32_t> emitters;
    CameraRec camera;
    FilterRec filter;
    IntegratorRec integrator;
    int32_t sample_count = 1;
    bool has_uv = false;
};

struct HostBvh {
    bool wide = false;                  /* nodes are WIDE nodes (BVH4, quantised child boxes; rt_types.h) */
    std::vector<f4> nodes;              /* kNodeQuads per node   */
    std::vector<f4> tris;               /* kPairQuads per triangle pair, leaf order */
    uint32_t n_pairs = 0;
    int32_t root = 0;
    uint32_t n_nodes = 0, n_leaves = 0, max_depth = 0;
    float sah_cost = 0.0f;
    float build_ms = 0.0f;
};

/* Returns an empty string on success, else the error message. */
std::string prepare_scene(const nori_scene_desc &desc, HostScene &out);

/* Binned-SAH BVH2 over all triangles of the scene (Accel::build,
 * src/accel.cpp:19-21).  max_depth_limit bounds the traversal stack.
 * wide: emit the tree as WIDE nodes -- every BVH2 node adopts grandchildren (largest surface area first) until it has
 * four children, their boxes quantised to 8 bits against the node's own box (rt_types.h). */
std::string build_bvh_sah(const HostScene &scene, uint32_t max_depth_limit, HostBvh &out, bool wide = false);

/* Filter evaluation, src/rfilter.cpp:25-29,56-70,85-103 */
float rfilter_eval(const nori_rfilter_desc &d, float radius, float x);

} // namespace nrt
