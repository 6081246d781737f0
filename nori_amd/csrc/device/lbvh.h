/* lbvh.h -- GPU LBVH builder (see lbvh.hip). */
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
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
    uint32_t n_refs = 0;       /* references the tree holds: the triangles, plus the parts of those that were cut */
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

/* Triangle splitting in front of the builders (lbvh_steps.h, "references"): `budget` = references beyond one per triangle as a fraction
   of the triangle count, `cap` = cuts one triangle may take.  split_bisect: the largest scale D of the priorities whose counts stay
   within the budget (total(D) is a sum over all triangles: a loop on the CPU, a reduction kernel on the device). */
struct SplitTuning { float budget; uint32_t cap; float scale; uint32_t inside; };      /* inside: other triangles a box must hold per cut (0: not asked) */
inline SplitTuning split_tuning(uint32_t n_triangles) {
    /* pa5 table (wf_extend ms at 128 spp, host tree 81.0): no splitting 111.4; scale 4 / 3 / 2 / 1.5: 84.1 / 83.1 / 81.5 / 81.5; without the
       look at what else lies in the box 85.4 -- and the Cornell box loses 5 % (node tests per ray 8.33 -> 8.80, the steps on the CPU):
       profiles/r6_18_builders_split.txt, r6_19_split_scale.txt */
    SplitTuning t; t.budget = 0.3f; t.cap = 15u; t.scale = 2.0f; t.inside = 4u;
    if (const char *e = std::getenv("NORI_HIP_SPLIT_INSIDE")) t.inside = (uint32_t) std::max(0, std::atoi(e));
    if (const char *e = std::getenv("NORI_HIP_SPLIT_BUDGET")) t.budget = std::max(0.0f, (float) std::atof(e));
    if (const char *e = std::getenv("NORI_HIP_SPLIT_CAP")) t.cap = (uint32_t) std::min(63, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("NORI_HIP_SPLIT_SCALE")) t.scale = std::max(0.0f, (float) std::atof(e));
    if (n_triangles < 8u) t.budget = 0.0f;
    return t;
}
/* The scale of the priorities: a triangle takes one cut per `scale` TYPICAL priorities of the scene -- typical = the mean of the
   priorities' bit patterns (an integer sum: the same whatever order a reduction adds in; as a float that is close to their geometric
   mean) over the triangles that have one -- so that a mesh of like-sized triangles is left alone however tilted they are and only what is
   several times larger than what surrounds it is cut.  scale = 0: the budget alone decides. */
inline float split_scale_D(unsigned long long sum_bits, unsigned long long count, float scale) {
    if (!(scale > 0.0f)) return 1e30f;
    if (count == 0ull) return 0.0f;
    const uint32_t bits = (uint32_t) (sum_bits / count);
    float typical; std::memcpy(&typical, &bits, 4);
    return typical > 0.0f ? 1.0f / (scale * typical) : 0.0f;
}
/* The scale the cuts are counted with: split_scale_D's, unless the counts then exceed the budget -- then the largest scale below it
   that keeps them within (total(D) = the sum of the triangles' counts: a loop on the CPU, a reduction kernel on the device). */
template <class Total> inline float split_choose_D(Total &&total, unsigned long long want, float D_scale) {
    if (want == 0ull || !(D_scale > 0.0f)) return 0.0f;
    float lo = 0.0f, hi = D_scale;
    if (D_scale < 1e29f) { if (total(D_scale) <= want) return D_scale; }
    else { hi = 1.0f; for (int k = 0; k < 60 && total(hi) <= want; ++k) { lo = hi; hi *= 4.0f; if (!(hi < 1e29f)) return lo; } }
    for (int k = 0; k < 40; ++k) { const float mid = 0.5f * (lo + hi); if (total(mid) <= want) lo = mid; else hi = mid; }
    return lo;
}
/* summed-volume table of the kSplitGrid^3 cell counts (lbvh_steps.h, split_inside): 275 k additions, on the host either way */
inline void split_sat(const uint32_t *cells, uint32_t *sat) {
    const int G = 64, G1 = 65;      /* kSplitGrid (lbvh_steps.h) */
    for (size_t i = 0; i < (size_t) G1 * G1 * G1; ++i) sat[i] = 0u;
    for (int z = 1; z <= G; ++z) for (int y = 1; y <= G; ++y) for (int x = 1; x <= G; ++x)
        sat[((size_t) z * G1 + y) * G1 + x] = cells[((size_t) (z - 1) * G + (y - 1)) * G + (x - 1)]
            + sat[((size_t) (z - 1) * G1 + y) * G1 + x] + sat[((size_t) z * G1 + (y - 1)) * G1 + x] + sat[((size_t) z * G1 + y) * G1 + (x - 1)]
            - sat[((size_t) (z - 1) * G1 + (y - 1)) * G1 + x] - sat[((size_t) (z - 1) * G1 + y) * G1 + (x - 1)] - sat[((size_t) z * G1 + (y - 1)) * G1 + (x - 1)]
            + sat[((size_t) (z - 1) * G1 + (y - 1)) * G1 + (x - 1)];
}

/* `dev` must have positions / indices / n_triangles set (device pointers);
 * d_tri_mesh: mesh id per global triangle (device).  Returns "" or an error. */
/* wide: emit WIDE nodes (rt_types.h) instead of BVH2 nodes; max_depth is then the stack entries a walk can need. */
/* ploc_radius: 0 = radix tree over the Morton codes (Karras 2012); > 0 = PLOC with that search radius (lbvh_steps.h). */
std::string build_bvh_lbvh_device(const DevScene &dev, const uint32_t *d_tri_mesh, LbvhDeviceResult &out, bool wide = false, uint32_t ploc_radius = 0);

} // namespace nrt
