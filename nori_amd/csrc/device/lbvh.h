/* lbvh.h -- GPU LBVH builder (see lbvh.hip). */
#pragma once
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
};

/* `dev` must have positions / indices / n_triangles set (device pointers);
 * d_tri_mesh: mesh id per global triangle (device).  Returns "" or an error. */
/* wide: emit WIDE nodes (rt_types.h) instead of BVH2 nodes; max_depth is then the stack entries a walk can need. */
/* ploc_radius: 0 = radix tree over the Morton codes (Karras 2012); > 0 = PLOC with that search radius (lbvh_steps.h). */
std::string build_bvh_lbvh_device(const DevScene &dev, const uint32_t *d_tri_mesh, LbvhDeviceResult &out, bool wide = false, uint32_t ploc_radius = 0);

} // namespace nrt
