/* wavefront.h -- the wavefront render engine (see wavefront.hip). */
#pragma once
#include <string>

#include "../../../include/nori_hip.h"
#include "film.h"
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
    size_t max_paths;           /* paths in flight: records of the state pool */
    size_t max_samples;         /* camera samples per batch: what the film's sample store holds at a time (a batch bigger than max_paths
                                   starts its samples pass by pass -- regeneration, wavefront.hip) */
    bool film_reference;        /* add the samples in the reference's order (film.h): needs the whole frame in ONE batch */
    const FilmBlockRows *film_share = nullptr;      /* reference order: the selected tiles are these block rows (tile_mod 1, tile_rem = their first tile) */
};

struct WfStats {
    unsigned long long n_camera = 0, n_closest = 0, n_shadow = 0, n_nodes = 0, n_tris = 0, n_invalid = 0;
    uint32_t n_batches = 0, n_iterations = 0, n_launches = 0;
    size_t state_bytes = 0;
    float class_ms[4] = {0.0f, 0.0f, 0.0f, 0.0f};      /* KernelClass: trace, shade, film, overlapped tails */
    uint32_t class_launches[4] = {0, 0, 0, 0};
    uint32_t trace_cus = 0;                      /* CUs wf_extend's stream owned (fewer than the device has: shading, or the tails of earlier batches, ran beside it) */
    uint32_t tail_cus = 0;                       /* CUs set aside for the tails of the batches (0: tails run on the bulk's stream) */
};

/* The engine's per-context resources (path-state pool, pipe streams / events, device properties);
 * created on the context's device, owned by nori_hip_ctx, never shared between contexts. */
struct WfEngine;
WfEngine *wavefront_create();
void wavefront_destroy(WfEngine *);
/* HBM bytes one path in flight costs (two state copies + hit record) and one camera sample of a batch costs in the film's
   sample store (both halves of it): size max_paths and max_samples */
size_t wavefront_bytes_per_path();
size_t wavefront_bytes_per_sample();
/* gives the engine's pool back to the device (an out-of-memory retry starts from nothing) */
void wavefront_release_pool(WfEngine *engine);
/* device bytes this context already holds for rendering (they are reusable, so they count as free) */
size_t wavefront_held_bytes(const WfEngine *engine, const FilmStore &film);

/* Renders the selected tiles / samples into d_rgbw (accumulating), on `stream`, with the context's
 * engine resources and film store.  Synchronises the stream.  Returns "" or an error message. */
std::string wavefront_render(WfEngine &engine, FilmStore &film_store, const DevScene &sc, const float *d_filter_table,
                             const WfLaunch &launch, float *d_rgbw, void *stream, WfStats &stats);

/* node records of the LDS image (rt_top.h) that fit next to wf_extend's traversal stacks, per node layout */
int wf_top_capacity(bool wide_nodes, bool records_32b, bool deeper_than_lds_stack);

/* -DNORI_COUNT_EXCURSIONS builds: adds this translation unit's excursion counters (rt_types.h) to out[4], optionally
   resetting them; false in the product build */
bool wavefront_excursions(unsigned long long out[4], bool reset);

} // namespace nrt
