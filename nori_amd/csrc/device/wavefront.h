/* wavefront.h -- the wavefront render engine (see wavefront.hip). */
#pragma once
#include <string>

#include "../../../include/nori_hip.h"
#include "rt_types.h"

namespace nrt {

struct WfLaunch {
    uint32_t spp_begin, spp_count;
    uint32_t tile_mod, tile_rem;
    uint32_t tiles_x, tiles_y, n_sel_tiles;
    int32_t tile_w;
    int stack_depth;            /* traversal stack entries the tree needs: max_depth + 1 */
    bool count_traversal;
    bool time_kernels;          /* HIP events around every launch (ktimer.h) */
    size_t max_paths;           /* paths in flight per batch */
};

struct WfStats {
    unsigned long long n_camera = 0, n_closest = 0, n_shadow = 0, n_nodes = 0, n_tris = 0, n_invalid = 0;
    uint32_t n_batches = 0, n_iterations = 0, n_launches = 0;
    size_t state_bytes = 0;
    float class_ms[3] = {0.0f, 0.0f, 0.0f};      /* KernelClass: trace, shade, film */
    uint32_t class_launches[3] = {0, 0, 0};
};

/* Renders the selected tiles / samples into d_rgbw (accumulating), on `stream`.
 * Synchronises the stream.  Returns "" or an error message. */
std::string wavefront_render(const DevScene &sc, const float *d_filter_table, const WfLaunch &launch, float *d_rgbw,
                             void *stream, WfStats &stats);

/* frees the cached device buffers of this process (called from nori_hip_destroy) */
void wavefront_release();

} // namespace nrt
