/* lbvh.h -- GPU LBVH builder (see lbvh.hip). */
#pragma once
#include <algorithm>
#include <cstdlib>
#include <string>

#include "rt_types.h"

namespace nrt {

struct LbvhDeviceResult {
    f4 *d_nodes = nullptr;     /* hipMalloc'ed; ownership passes to the caller */
    f4 *d_tris = nullptr;
    int32_t root = 0;
    uint32_t n_nodes = 0, n_leaves = 0, max_depth = 0, n_pairs = 0;
    float build_ms = 0.0f;
    bool wide = false;         /* d_nodes holds WIDE nodes (BVH4, quantised child boxes) */
    uint32_t ploc_iterations = 0;
    uint32_t reinserted = 0;   /* subtrees moved by the re-insertion iterations */
};

/* What follows PLOC (lbvh_steps.h): `sweeps` treelet sweeps, `iterations` rounds of parallel re-insertion -- round i over the candidates
   whose slot is i modulo `stride` -- and `sweeps_after` more sweeps.  Small batches move more of their candidates (fewer of them want
   the same nodes) and the later ones search the improved tree: 16 rounds at stride 8 cost the 10 M-triangle terrain what 4 at stride 1
   do and gain twice as much.  Measured there (profiles/r6_16_c5_build_matrix.txt): a sweep 25 - 30 ms, a round at stride 1 / 8: 13 / 3.5
   ms (search 9.5 ms per 20 M candidates, refit 2.2 ms); PLOC + 2 sweeps: 87 ms, wf_extend +7.6 % against the host's SAH tree (built in
   2.4 s); 1 sweep + 16 rounds at stride 8: 142 ms, +1.5 %.  Small scenes build in a few ms whatever runs: there the tree converges (32
   rounds at stride 4, tools/reinsert_probe.py).  Host-side knobs, shared by the device builder and the CPU harness that runs the same steps. */
struct BuildTuning { int sweeps; int iterations; uint32_t stride; int sweeps_after; };
inline BuildTuning build_tuning(uint32_t n_triangles) {
    BuildTuning t;
    const bool big = n_triangles > (1u << 20);
    t.sweeps = big ? 1 : 2; t.iterations = big ? 16 : 32; t.stride = big ? 8u : 4u; t.sweeps_after = big ? 0 : 1;
    if (const char *e = std::getenv("NORI_HIP_TREELET_SWEEPS")) t.sweeps = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("NORI_HIP_REINSERT_ITERS")) t.iterations = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("NORI_HIP_REINSERT_STRIDE")) t.stride = (uint32_t) std::max(1, std::atoi(e));
    if (const char *e = std::getenv("NORI_HIP_REINSERT_SWEEPS_AFTER")) t.sweeps_after = std::max(0, std::atoi(e));
    if (n_triangles < 8u) t.iterations = 0;      /* (nothing to move: the root and its children stay) */
    return t;
}

/* `dev` must have positions / indices / n_triangles set (device pointers);
 * d_tri_mesh: mesh id per global triangle (device).  Returns "" or an error. */
/* wide: emit WIDE nodes (rt_types.h) instead of BVH2 nodes; max_depth is then the stack entries a walk can need. */
/* ploc_radius: 0 = radix tree over the Morton codes (Karras 2012); > 0 = PLOC with that search radius (lbvh_steps.h). */
std::string build_bvh_lbvh_device(const DevScene &dev, const uint32_t *d_tri_mesh, LbvhDeviceResult &out, bool wide = false, uint32_t ploc_radius = 0);

} // namespace nrt
