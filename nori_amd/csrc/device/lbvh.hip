/*
 * lbvh.hip -- Accel::build on the GPU, emitted directly in the traversal layout of rt_types.h.  Since round 6 this is what
 * NORI_ACCEL_AUTO builds: every benchmark line is rendered through a tree these kernels made.
 *
 * The reference's Accel::build is a no-op (src/accel.cpp:19-21) and its rayIntersect scans every triangle; a BVH is this
 * project's replacement.  The kernels are one thread (or wave) per element over the steps of lbvh_steps.h, which the CPU test
 * harness runs as loops (tests/emu/emu_builder.h) -- same steps, same arithmetic, the same tree:
 *
 *   1. k_scene_bounds   per-triangle boxes (src/mesh.cpp:78-83) -> scene box (float atomics on order-preserving integer keys)
 *   2. references       k_split_priority / _grid / _limit / _total / _emit: the tree is built over (triangle, box) references; a
 *                       triangle whose box is several times the scene's typical one, has volume and holds other geometry enters as
 *                       several clipped parts (the host builder's spatial splits, decided up front).  Every reference gets the
 *                       63-bit Morton code of its box centre (src/mesh.cpp:85-90 uses the vertex centroid; any centre orders as well)
 *   3. hipcub radix sort of (code, reference) pairs; k_refs_gather: triangle and padded box by position
 *   4. k_hierarchy      radix tree (Karras 2012: one thread per internal node finds its key range and split with clz(key_i ^ key_j);
 *                       equal codes are told apart by their sorted position), or
 *   4'. k_ploc_*        PLOC (Meister & Bittner 2018): clusters in Morton order merge with their nearest neighbour by area, then
 *       k_treelet_wave  treelet restructuring (Karras & Aila 2013), a wave per treelet, and
 *       k_reins_*       parallel re-insertion (Meister & Bittner 2018): every subtree searches the same tree for the place where the
 *                       inner nodes' areas shrink most, marks the nodes its move touches by a 64-bit maximum, winners relink, refit;
 *                       k_ploc_leaf_positions / k_ploc_finish: the tree's leaves left to right = the order everything below uses
 *   5. k_leaf_boxes / k_tree_level   the references' boxes in that order and a min/max segment tree over them (one launch per
 *                       level): every node covers a CONTIGUOUS range, so its box is a range query -- no bottom-up atomics
 *   6. k_collapse       which subtrees of <= 4 references become one leaf (surface-area heuristic)
 *   7. k_mark_leaves / scans / k_emit_pairs / k_emit_nodes   the surviving nodes renumbered densely as 64-B two-child-box
 *                       records; the leaves' triangles as 96-B de-indexed pair records, leaf by leaf
 *   8. k_depth          levels of emitted nodes above the deepest leaf (sizes the traversal stack)
 *   wide = true (scenes beyond the caches): instead of 7's two-box records the tree is emitted as WIDE nodes (BVH4,
 *       quantised child boxes; rt_types.h).  k_wide_expand walks the kept nodes top-down, one launch per level
 *       of the wide tree: a wide node takes its node's two children and twice replaces the inner child of largest
 *       surface area by that child's children; the inner children that remain are the wide nodes of the next level.
 *       k_emit_wide quantises the (<= 4) child boxes against their union (rt_wide.h) -- the nodes in between vanish.
 *
 * Numerically collinear triangles (rt_types.h, tri_box_pad) get Morton bit 63: the root separates them
 * from the spatial hierarchy, their boxes are infinite; they are never cut and never moved.
 *
 * Any valid BVH returns the same hits as the linear scan (conservative node test + tie rule in rt_trace.h);
 * tests/test_gpu_parity.py checks these trees against the oracle's brute force and against the host builder's (scene_prep.cpp:
 * binned SAH + spatial splits + re-insertion on CPU threads -- the fall-back, and the yardstick: wf_extend with the device tree
 * runs within 1 % of it on the Cornell box, the pa5 table and the AO scene, within 3 % on the 10 M-triangle terrain).
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lbvh.h"
#include "lbvh_steps.h"

using namespace nrt;

namespace {

__device__ __forceinline__ unsigned int float_key(float f) {      /* order preserving */
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float key_float(unsigned int k) {
    unsigned int u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}

/* The kernels are one thread per element over the steps of lbvh_steps.h (which the CPU harness runs as loops). */

__global__ void k_scene_bounds(const f4 *pos, const uint32_t *idx, uint32_t n, unsigned int *bounds) {
    __shared__ unsigned int s[6];
    if (threadIdx.x < 3) s[threadIdx.x] = 0xffffffffu; else if (threadIdx.x < 6) s[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        f3 mn, mx; tri_box(pos, idx, t, mn, mx);
        atomicMin(&s[0], float_key(mn.x)); atomicMin(&s[1], float_key(mn.y)); atomicMin(&s[2], float_key(mn.z));
        atomicMax(&s[3], float_key(mx.x)); atomicMax(&s[4], float_key(mx.y)); atomicMax(&s[5], float_key(mx.z));
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s[threadIdx.x]);
}

/* ---- references (lbvh_steps.h): which triangles enter the tree as several parts, and the parts ---- */
/* priorities; their bit patterns summed (the scene's typical priority: lbvh.h, split_scale_D) */
__global__ void k_split_priority(const f4 *pos, const uint32_t *idx, uint32_t n, f3 smin, f3 sinv, float *prio, unsigned long long *sums) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const float p = t < n ? split_priority(pos, idx, t, smin, sinv) : 0.0f;
    if (t < n) prio[t] = p;
    unsigned long long bits = p > 0.0f ? (unsigned long long) __float_as_uint(p) : 0ull, have = p > 0.0f ? 1ull : 0ull;
    for (int off = 32; off > 0; off >>= 1) { bits += __shfl_down(bits, off); have += __shfl_down(have, off); }
    if ((threadIdx.x & 63u) == 0u && have) { atomicAdd(&sums[0], bits); atomicAdd(&sums[1], have); }
}
__global__ void k_split_grid(const f4 *pos, const uint32_t *idx, uint32_t n, f3 smin, f3 sinv, uint32_t *cells) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || tri_unbounded(pos, idx, t)) return;
    f3 mn, mx; tri_box(pos, idx, t, mn, mx);
    atomicAdd(&cells[split_grid_cell(mn, mx, smin, sinv)], 1u);
}
/* cuts a triangle may take at most: what else lies in its box (sat: the summed-volume table of k_split_grid's counts) */
__global__ void k_split_limit(const f4 *pos, const uint32_t *idx, uint32_t n, f3 smin, f3 sinv, const float *prio, const uint32_t *sat, uint32_t cap, uint32_t inside, uint32_t *limit) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t l = cap;
    if (inside > 0u && prio[t] > 0.0f) { f3 mn, mx; tri_box(pos, idx, t, mn, mx); l = min(cap, split_inside(sat, mn, mx, smin, sinv) / inside); }
    limit[t] = l;
}
__global__ void k_split_total(const float *prio, const uint32_t *limit, uint32_t n, float D, unsigned long long *total) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = t < n ? (unsigned long long) split_count(prio[t], D, limit[t]) : 0ull;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63u) == 0u && c) atomicAdd(total, c);
}
/* references per triangle (write = false) / the references themselves at their place ref_first[t] */
__global__ void k_split_emit(const f4 *pos, const uint32_t *idx, uint32_t n, const float *prio, const uint32_t *limit, float D, float pad0, f3 smin, f3 sinv,
                             bool write, uint32_t *count, const uint32_t *ref_first, RefOut out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t cuts = prio ? split_count(prio[t], D, limit[t]) : 0u;
    const uint32_t c = split_emit(pos, idx, t, cuts, pad0, smin, sinv, write, out, write ? ref_first[t] : 0u);
    if (!write) count[t] = c;
}
/* the sorted references: triangle and padded box by position */
__global__ void k_refs_gather(const uint32_t *sorted_ref, uint32_t n, RefOut refs, uint32_t *tri, f4 *lmn, f4 *lmx) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t r = sorted_ref[k];
    tri[k] = refs.tri[r]; lmn[k] = refs.mn[r]; lmx[k] = refs.mx[r];
}
__global__ void k_iota(uint32_t *v, uint32_t n) { const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; if (k < n) v[k] = k; }

__global__ void k_hierarchy(const unsigned long long *keys, int n, RadixNode *nodes, uint32_t *parent_inner, uint32_t *parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    nodes[i] = radix_node(keys, n, i, parent_inner, parent_leaf);
}

/* ---- PLOC ---- */
__global__ void k_ploc_init(const f4 *lmn, const f4 *lmx, uint32_t n, PlocClusters c) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    f4 mn = lmn[k], mx = lmx[k];
    mn.w = __uint_as_float(kLeafBit | k); mx.w = __uint_as_float(1u);
    c.mn[k] = mn; c.mx[k] = mx;
}
__global__ void k_ploc_nearest(PlocClusters c, uint32_t m, uint32_t radius, uint32_t *nearest) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) nearest[i] = ploc_nearest(c, m, i, radius);
}
/* flags for ONE scan: stays in the low word, lead in the high word */
__global__ void k_ploc_decide(const uint32_t *nearest, uint32_t m, unsigned long long *flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t lead, stays; ploc_decide(nearest, i, lead, stays);
    flags[i] = (unsigned long long) stays | ((unsigned long long) lead << 32);
}
__global__ void k_ploc_apply(PlocClusters in, PlocClusters out, PlocNodes nodes, const uint32_t *nearest, uint32_t m,
                             const unsigned long long *flags, const unsigned long long *incl, uint32_t node_base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i], r = incl[i] - f;      /* exclusive ranks */
    ploc_apply(in, out, nodes, nearest, i, (uint32_t) (f >> 32), (uint32_t) f & 1u, (uint32_t) (r >> 32), (uint32_t) r, node_base);
}
__global__ void k_treelet(PlocNodes nodes, TreeletData td, TreeletParams tp, uint32_t *visits, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) treelet_climb(nodes, td, tp, visits, n - 2u, k);
}

/* ---- a WAVE per treelet (Karras & Aila's layout, for 64 lanes).  One thread per treelet spends a sweep in 1.1 KB of scratch per
   lane and a dynamic programme that 63 lanes of its wave wait for: 340 ms per sweep on 10 M triangles.  Here the lanes of a wave
   still climb from 64 triangles, but every treelet one of them arrives at as the SECOND visitor is optimised by the whole wave:
     lane 0      treelet_form (lbvh_steps.h) into the wave's LDS
     all lanes   the areas of the 2^k - 1 subsets, two per lane
     all lanes   the dynamic programme, subset size by subset size: the (subset, partition) pairs of a size -- 21, 105, 245, 315, 217, 63
                 for seven leaves -- are spread over the lanes from a table; a pair's cost goes into the subset's slot by a 64-bit LDS
                 minimum on (cost bits, rank of the partition in the serial order, partition): 17 passes instead of 966 steps, and the
                 same winner as the serial loop, which keeps the FIRST of equally cheap partitions
     lane 0      treelet_rewire
   Same steps, same arithmetic, same tie rule as treelet_optimize: the tree is the one the CPU harness builds. */
struct TreeletPair { unsigned char S, P, rank, pad; };
struct TreeletTable { const TreeletPair *pairs; uint32_t start[8][9]; };      /* pairs of k leaves, subset size s: [start[k][s], start[k][s + 1]) */
struct TreeletShared {
    TreeletWork w;
    float area[1 << kTreeletLeaves], copt[1 << kTreeletLeaves];
    unsigned long long best[1 << kTreeletLeaves];
    unsigned char part[1 << kTreeletLeaves];
    int go;
};

__device__ void treelet_optimize_wave(const PlocNodes &nodes, const TreeletData &td, TreeletParams tp, const TreeletTable &tab, uint32_t id, TreeletShared &sh, int lane) {
    if (lane == 0) sh.go = treelet_form(nodes, td, tp, id, sh.w) ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (!sh.go) return;                                    /* wave-uniform */
    const int k = sh.w.k, full = (1 << k) - 1;
    for (int S = lane + 1; S <= full; S += 64) {
        sh.area[S] = treelet_subset_area(sh.w, S);
        sh.best[S] = ~0ull;
        if ((S & (S - 1)) == 0) { int i = 0; while (i < kTreeletLeaves - 1 && !(S & (1 << i))) ++i; sh.copt[S] = sh.w.lcost[i]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int s = 2; s <= k; ++s) {
        for (uint32_t q = tab.start[k][s] + (uint32_t) lane; q < tab.start[k][s + 1]; q += 64u) {
            const TreeletPair pr = tab.pairs[q];
            const float c = sh.copt[pr.P] + sh.copt[pr.S ^ pr.P];          /* >= 0: its bits order like the value */
            atomicMin(&sh.best[pr.S], ((unsigned long long) __float_as_uint(c) << 32) | ((unsigned long long) pr.rank << 8) | pr.P);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int S = lane + 1; S <= full; S += 64)
            if (__popc((unsigned) S) == s) {
                const unsigned long long b = sh.best[S];
                sh.copt[S] = tp.c_node * sh.area[S] + __uint_as_float((uint32_t) (b >> 32));
                sh.part[S] = (unsigned char) (b & 0xffu);
            }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && sh.copt[full] < coh_ld(&td.cost[id]) * 0.99999f) treelet_rewire(nodes, td, id, sh.w, sh.copt, sh.part);      /* else: leave the subtree as it is */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* by the whole wave: lane 0's (agent-scope) stores are acknowledged before any lane of it arrives at the parent */
}

constexpr int kTreeletWaves = 4;      /* per workgroup */
__global__ __launch_bounds__(64 * kTreeletWaves) void k_treelet_wave(PlocNodes nodes, TreeletData td, TreeletParams tp, TreeletTable tab, uint32_t *visits, uint32_t n) {
    __shared__ TreeletShared s_sh[kTreeletWaves];
    TreeletShared &sh = s_sh[threadIdx.x >> 6];
    const int lane = (int) (threadIdx.x & 63u);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, root_id = n - 2u;
    bool active = k < n;
    uint32_t p = active ? nodes.parent_prim[k] : 0u;
    for (uint32_t guard = 0; guard < 4096u; ++guard) {
        /* arrive: the first of a node's two visitors stops, the second -- both subtrees below are finished -- has it optimised.
           What the other subtree's waves wrote is read at agent scope (coh_ld, lbvh_steps.h) and this wave's own stores are
           acknowledged before it counts itself in: no cache-wide fence */
        bool second = false;
        if (active) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t before = __hip_atomic_fetch_add(&visits[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::: "memory");
            if (before == 0u) active = false; else second = true;
        }
        for (unsigned long long pend = __ballot(second); pend != 0ull; pend &= pend - 1ull) {
            const uint32_t id = (uint32_t) __shfl((int) p, __ffsll((long long) pend) - 1);
            treelet_optimize_wave(nodes, td, tp, tab, id, sh, lane);
        }
        if (active) { if (p == root_id) active = false; else p = coh_ld(&nodes.parent_node[p]); }      /* (the node's own parent is not touched by its treelet) */
        if (__ballot(active) == 0ull) break;
    }
}
/* ---- parallel re-insertion (lbvh_steps.h): one thread per candidate slot of the iteration's residue class ---- */
__global__ void k_reins_search(PlocNodes nodes, TreeletData td, ReinsData rd, uint32_t n_inner, uint32_t n_slots, uint32_t phase, uint32_t stride) {
    const uint32_t slot = phase + (blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (slot >= n_slots) return;
    reins_search(nodes, td, rd, n_inner, slot);
    (void) reins_locks(nodes, rd, n_inner, slot, true);      /* (marks do not change what the others' searches read) */
}
__global__ void k_reins_check(PlocNodes nodes, ReinsData rd, uint32_t n_inner, uint32_t n_slots, uint32_t phase, uint32_t stride, uint32_t *moved) {
    const uint32_t slot = phase + (blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (slot >= n_slots) return;
    const bool win = reins_locks(nodes, rd, n_inner, slot, false);
    rd.win[slot] = win ? 1u : 0u;
    if (win) atomicAdd(moved, 1u);
}
__global__ void k_reins_apply(PlocNodes nodes, ReinsData rd, uint32_t n_inner, uint32_t n_slots, uint32_t phase, uint32_t stride) {
    const uint32_t slot = phase + (blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (slot < n_slots && rd.win[slot]) reins_apply(nodes, rd, n_inner, slot);
}
/* bottom-up from every triangle; the second arrival at a node computes it (the hand-over of k_treelet_wave: agent-scope accesses,
   the arriving thread's stores acknowledged before it counts itself in) */
__global__ void k_reins_refit(PlocNodes nodes, TreeletData td, ReinsData rd, TreeletParams tp, uint32_t *visits, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t root_id = n - 2u;
    uint32_t p = nodes.parent_prim[k];
    for (uint32_t guard = 0; guard < 4096u; ++guard) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t before = __hip_atomic_fetch_add(&visits[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
        if (before == 0u) return;
        reins_refit_node(nodes, td, rd, tp, p);
        if (p == root_id) return;
        p = nodes.parent_node[p];
    }
}

__global__ void k_ploc_leaf_positions(PlocNodes nodes, uint32_t n, const uint32_t *order, const f4 *lmn, const f4 *lmx, uint32_t *leaf_pos, uint32_t *order_out, f4 *lmn_out, f4 *lmx_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = ploc_first_position(nodes, n - 1u, kLeafBit | k);
    leaf_pos[k] = p; order_out[p] = order[k]; lmn_out[p] = lmn[k]; lmx_out[p] = lmx[k];
}
__global__ void k_ploc_finish(PlocNodes nodes, uint32_t n_nodes, const uint32_t *leaf_pos, RadixNode *out, uint32_t *parent_inner, uint32_t *parent_leaf) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_nodes) return;
    out[n_nodes - 1u - id] = ploc_finish(nodes, n_nodes, id, leaf_pos, parent_inner, parent_leaf);
}

/* segment tree over the padded triangle boxes in the builder's order: entries [N + k] */
__global__ void k_leaf_boxes(const f4 *lmn, const f4 *lmx, uint32_t n, uint32_t N, f4 *tmin, f4 *tmax) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    f4 mn4, mx4;
    if (k < n) { mn4 = lmn[k]; mx4 = lmx[k]; mn4.w = mx4.w = 0.0f; }
    else { mn4.x = mn4.y = mn4.z = kInf; mx4.x = mx4.y = mx4.z = -kInf; mn4.w = mx4.w = 0.0f; }
    tmin[N + k] = mn4; tmax[N + k] = mx4;
}
__global__ void k_tree_level(uint32_t first, uint32_t count, f4 *tmin, f4 *tmax) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) seg_tree_combine(first + k, tmin, tmax);
}

__global__ void k_collapse(const RadixNode *nodes, uint32_t n_inner, const f4 *tmin, const f4 *tmax, uint32_t N, CollapseParams cp, uint32_t *collapse) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) collapse[i] = collapse_decide(nodes, i, tmin, tmax, N, cp);
}
__global__ void k_mark_leaves(const RadixNode *nodes, uint32_t n_inner, const uint32_t *collapse, const uint32_t *parent_inner,
                              uint32_t *leaf_cnt, uint32_t *leaf_pairs, uint32_t *keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) keep[i] = mark_leaves(nodes, i, collapse, parent_inner, leaf_cnt, leaf_pairs);
}

/* ---- WIDE emission ---- */
/* one thread per wide node of the current level: choose its children, queue the inner ones for the next level */
__global__ void k_wide_expand(const RadixNode *nodes, const uint32_t *collapse, const f4 *tmin, const f4 *tmax, uint32_t N,
                              const uint32_t *frontier, uint32_t count, uint32_t *next, uint32_t *next_count,
                              uint32_t *kids, uint32_t *n_kids, uint32_t *is_wide) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t i = frontier[t];
    uint32_t kid[4];
    const int n = wide_children(nodes, collapse, tmin, tmax, N, i, kid);
    n_kids[i] = (uint32_t) n;
    for (int k = 0; k < n; ++k) {
        kids[4 * (size_t) i + k] = kid[k];
        uint32_t lo, hi;
        if (!child_range(nodes, collapse, kid[k], lo, hi)) { is_wide[kid[k]] = 1u; next[atomicAdd(next_count, 1u)] = kid[k]; }
    }
}
__global__ void k_emit_wide(const RadixNode *nodes, uint32_t n_inner, const f4 *tmin, const f4 *tmax, uint32_t N, const uint32_t *collapse,
                            const uint32_t *pair_start, const uint32_t *is_wide, const uint32_t *wide_index, const uint32_t *kids,
                            const uint32_t *n_kids, f4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !is_wide[i]) return;
    f4 q[4];
    emit_wide_node(nodes, tmin, tmax, N, collapse, pair_start, wide_index, kids + 4 * (size_t) i, (int) n_kids[i], q);
    f4 *dst = out + (size_t) wide_index[i] * kNodeQuads;
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
}

__global__ void k_emit_nodes(const RadixNode *nodes, uint32_t n_inner, const f4 *tmin, const f4 *tmax, uint32_t N,
                             const uint32_t *collapse, const uint32_t *pair_start, const uint32_t *keep, const uint32_t *node_index, f4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !keep[i]) return;
    f4 q[4];
    emit_node(nodes, i, tmin, tmax, N, collapse, pair_start, node_index, q);
    f4 *dst = out + (size_t) node_index[i] * kNodeQuads;      /* dense: only the surviving nodes, in tree-array order */
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
}

/* one thread per leaf start: the leaf's triangles, de-indexed, as pair records */
__global__ void k_emit_pairs(const f4 *pos, const uint32_t *idx, const uint32_t *tri_mesh, const uint32_t *order, uint32_t n,
                             const uint32_t *leaf_cnt, const uint32_t *pair_start, f4 *out, uint32_t pair_base) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t cnt = leaf_cnt[k];
    if (cnt) emit_leaf_pairs(pos, idx, tri_mesh, order, k, cnt, out + (size_t) (pair_start[k] + pair_base) * kPairQuads);
}

__global__ void k_depth(const uint32_t *parent_inner, const uint32_t *parent_leaf, const uint32_t *keep, uint32_t n, unsigned int *max_depth) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) atomicMax(max_depth, leaf_depth(parent_inner, parent_leaf, keep, k));
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes < 16 ? 16 : bytes); }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

#define LB_TRY(expr)                                                                    \
    do { hipError_t e__ = (expr); if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); } while (0)

} // namespace

namespace nrt {

std::string build_bvh_lbvh_device(const DevScene &dev, const uint32_t *d_tri_mesh, LbvhDeviceResult &out, bool wide, uint32_t ploc_radius) {
    out = LbvhDeviceResult();
    if (dev.n_triangles <= 4) wide = false;                  /* a single leaf: no nodes at all */
    const uint32_t pair_base = wide ? 1u : 0u;               /* wide trees reserve pair 0 as the all-zero pair of unused slots */
    const uint32_t n_tris = dev.n_triangles;
    uint32_t n = n_tris;                                     /* from step 2 on: the REFERENCES the tree is built over */
    struct Events {      /* released on every return path */
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void) hipEventDestroy(a); if (b) (void) hipEventDestroy(b); }
    } ev;
    LB_TRY(hipEventCreate(&ev.a)); LB_TRY(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    LB_TRY(hipEventRecord(e0, 0));

    if (n == 0) return "lbvh: empty scene";
    /* NORI_HIP_BUILD_TIMING: milliseconds per phase on stderr (a device synchronisation per lap: not for timed builds) */
    const bool timing = getenv("NORI_HIP_BUILD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void) hipDeviceSynchronize();
        const std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        fprintf(stderr, "[lbvh] %-28s %8.2f ms\n", what, std::chrono::duration<float, std::milli>(t - t_lap).count());
        t_lap = t;
    };
    const int B = 256;
    uint32_t gridN = (n + B - 1) / B;

    /* 1. scene bounds */
    Buf bounds; LB_TRY(bounds.alloc(6 * sizeof(unsigned int)));
    const unsigned int init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    LB_TRY(hipMemcpy(bounds.p, init, sizeof(init), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scene_bounds, dim3(std::min<uint32_t>(gridN, 4096)), dim3(B), 0, 0, dev.positions, dev.indices, n, bounds.as<unsigned int>());
    unsigned int hb[6];
    LB_TRY(hipMemcpy(hb, bounds.p, sizeof(hb), hipMemcpyDeviceToHost));
    f3 smin = mk3(key_float(hb[0]), key_float(hb[1]), key_float(hb[2])), smax = mk3(key_float(hb[3]), key_float(hb[4]), key_float(hb[5]));
    const float ex = smax.x - smin.x, ey = smax.y - smin.y, ez = smax.z - smin.z;
    const float pad = box_pad_rel() * sqrtf(ex * ex + ey * ey + ez * ez) + 1e-30f;    /* as scene_prep.cpp */
    const f3 sinv = mk3(ex > 0 ? 1.0f / ex : 0.0f, ey > 0 ? 1.0f / ey : 0.0f, ez > 0 ? 1.0f / ez : 0.0f);

    /* 2. references (lbvh_steps.h): every triangle once, or -- where a box is several times what is typical for the scene and holds other
       geometry -- as several parts */
    const SplitTuning stn = split_tuning(n_tris);
    Buf s_prio, s_limit, s_count, s_first, r_tri, r_mn, r_mx, r_key;
    float split_D = 0.0f;
    if (stn.budget > 0.0f && ploc_radius != 0u) {      /* (the radix tree splits space by the keys' bits: parts of one triangle would only deepen it) */
        Buf sums, cells, sat;
        LB_TRY(s_prio.alloc((size_t) n_tris * 4)); LB_TRY(s_limit.alloc((size_t) n_tris * 4));
        LB_TRY(sums.alloc(3 * 8)); LB_TRY(hipMemset(sums.p, 0, 3 * 8));
        hipLaunchKernelGGL(k_split_priority, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, n_tris, smin, sinv, s_prio.as<float>(), sums.as<unsigned long long>());
        unsigned long long hs[2] = {0ull, 0ull};
        LB_TRY(hipMemcpy(hs, sums.p, 16, hipMemcpyDeviceToHost));
        const float D_scale = split_scale_D(hs[0], hs[1], stn.scale);
        if (hs[1] > 0ull && D_scale > 0.0f) {
            if (stn.inside > 0u) {
                const int G = kSplitGrid, G1 = kSplitGrid + 1;
                std::vector<uint32_t> h_cells((size_t) G * G * G), h_sat((size_t) G1 * G1 * G1, 0u);
                LB_TRY(cells.alloc(h_cells.size() * 4)); LB_TRY(hipMemset(cells.p, 0, h_cells.size() * 4));
                hipLaunchKernelGGL(k_split_grid, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, n_tris, smin, sinv, cells.as<uint32_t>());
                LB_TRY(hipMemcpy(h_cells.data(), cells.p, h_cells.size() * 4, hipMemcpyDeviceToHost));
                split_sat(h_cells.data(), h_sat.data());
                LB_TRY(sat.alloc(h_sat.size() * 4));
                LB_TRY(hipMemcpy(sat.p, h_sat.data(), h_sat.size() * 4, hipMemcpyHostToDevice));
            }
            hipLaunchKernelGGL(k_split_limit, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, n_tris, smin, sinv, s_prio.as<float>(), sat.as<uint32_t>(), stn.cap, stn.inside, s_limit.as<uint32_t>());
            std::string total_err;
            auto total = [&](float D) -> unsigned long long {
                unsigned long long h = 0ull;
                if (hipMemsetAsync(sums.as<unsigned long long>() + 2, 0, 8, 0) != hipSuccess) total_err = "lbvh: split total";
                hipLaunchKernelGGL(k_split_total, dim3(gridN), dim3(B), 0, 0, s_prio.as<float>(), s_limit.as<uint32_t>(), n_tris, D, sums.as<unsigned long long>() + 2);
                if (hipMemcpy(&h, sums.as<unsigned long long>() + 2, 8, hipMemcpyDeviceToHost) != hipSuccess) total_err = "lbvh: split total";
                return h;
            };
            split_D = split_choose_D(total, (unsigned long long) ((double) stn.budget * (double) n_tris), D_scale);
            if (split_D > 0.0f && total(split_D) == 0ull) split_D = 0.0f;      /* nothing to cut (a mesh of like-sized triangles): the references are the triangles */
            if (!total_err.empty()) return total_err;
        }
    }
    {
        const bool cutting = split_D > 0.0f;
        RefOut none{nullptr, nullptr, nullptr, nullptr};
        uint32_t m = n_tris;
        if (cutting) {
            LB_TRY(s_count.alloc((size_t) n_tris * 4)); LB_TRY(s_first.alloc((size_t) n_tris * 4));
            hipLaunchKernelGGL(k_split_emit, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, n_tris, s_prio.as<float>(), s_limit.as<uint32_t>(), split_D, pad, smin, sinv,
                               false, s_count.as<uint32_t>(), (const uint32_t *) nullptr, none);
            size_t scan_bytes = 0;
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, s_count.as<uint32_t>(), s_first.as<uint32_t>(), (int) n_tris));
            Buf scan_tmp; LB_TRY(scan_tmp.alloc(scan_bytes));
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, s_count.as<uint32_t>(), s_first.as<uint32_t>(), (int) n_tris));
            uint32_t last[2] = {0u, 0u};
            LB_TRY(hipMemcpy(&last[0], s_first.as<uint32_t>() + (n_tris - 1), 4, hipMemcpyDeviceToHost));
            LB_TRY(hipMemcpy(&last[1], s_count.as<uint32_t>() + (n_tris - 1), 4, hipMemcpyDeviceToHost));
            m = last[0] + last[1];
        } else {
            LB_TRY(s_first.alloc((size_t) n_tris * 4));
            hipLaunchKernelGGL(k_iota, dim3(gridN), dim3(B), 0, 0, s_first.as<uint32_t>(), n_tris);
        }
        LB_TRY(r_tri.alloc((size_t) m * 4)); LB_TRY(r_mn.alloc((size_t) m * 16)); LB_TRY(r_mx.alloc((size_t) m * 16)); LB_TRY(r_key.alloc((size_t) m * 8));
        const RefOut refs{r_tri.as<uint32_t>(), r_mn.as<f4>(), r_mx.as<f4>(), r_key.as<unsigned long long>()};
        hipLaunchKernelGGL(k_split_emit, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, n_tris, cutting ? s_prio.as<float>() : (const float *) nullptr, s_limit.as<uint32_t>(), split_D, pad, smin, sinv,
                           true, (uint32_t *) nullptr, s_first.as<uint32_t>(), refs);
        LB_TRY(hipGetLastError());
        n = m; gridN = (n + B - 1) / B;
        out.n_refs = m;
    }
    lap("references");

    /* 3. sorted by Morton key; then by position: the reference's triangle (order) and padded box (lmn / lmx) */
    Buf keys_b, vals_a, vals_b, o_tri, o_tri2, l_mn, l_mx, l_mn2, l_mx2;
    LB_TRY(keys_b.alloc((size_t) n * 8));
    LB_TRY(vals_a.alloc((size_t) n * 4)); LB_TRY(vals_b.alloc((size_t) n * 4));
    LB_TRY(o_tri.alloc((size_t) n * 4)); LB_TRY(l_mn.alloc((size_t) n * 16)); LB_TRY(l_mx.alloc((size_t) n * 16));
    hipLaunchKernelGGL(k_iota, dim3(gridN), dim3(B), 0, 0, vals_a.as<uint32_t>(), n);
    size_t temp_bytes = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, r_key.as<unsigned long long>(), keys_b.as<unsigned long long>(),
                                              vals_a.as<uint32_t>(), vals_b.as<uint32_t>(), (int) n, 0, 64));
    Buf temp; LB_TRY(temp.alloc(temp_bytes));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, r_key.as<unsigned long long>(), keys_b.as<unsigned long long>(),
                                              vals_a.as<uint32_t>(), vals_b.as<uint32_t>(), (int) n, 0, 64));
    hipLaunchKernelGGL(k_refs_gather, dim3(gridN), dim3(B), 0, 0, vals_b.as<uint32_t>(), n, RefOut{r_tri.as<uint32_t>(), r_mn.as<f4>(), r_mx.as<f4>(), r_key.as<unsigned long long>()},
                       o_tri.as<uint32_t>(), l_mn.as<f4>(), l_mx.as<f4>());
    const unsigned long long *keys = keys_b.as<unsigned long long>();
    const uint32_t *order = o_tri.as<uint32_t>();      /* position in the builder's order -> global triangle */
    const f4 *lmn = l_mn.as<f4>(), *lmx = l_mx.as<f4>();

    /* leaves: start positions, triangle counts, pair counts -> first pair of every leaf */
    Buf leaf_cnt, leaf_pairs, pair_start, rnodes, pin, plf, keep, node_index, collapse, tmin, tmax;
    uint32_t N = 1;
    LB_TRY(leaf_cnt.alloc((size_t) n * 4)); LB_TRY(leaf_pairs.alloc((size_t) n * 4)); LB_TRY(pair_start.alloc((size_t) n * 4));
    LB_TRY(hipMemset(leaf_cnt.p, 0, (size_t) n * 4)); LB_TRY(hipMemset(leaf_pairs.p, 0, (size_t) n * 4));
    if (n <= 4) {
        const uint32_t one[2] = {n, (n + 1) / 2};
        LB_TRY(hipMemcpy(leaf_cnt.p, &one[0], 4, hipMemcpyHostToDevice));
        LB_TRY(hipMemcpy(leaf_pairs.p, &one[1], 4, hipMemcpyHostToDevice));
    } else {
        LB_TRY(rnodes.alloc((size_t) (n - 1) * sizeof(RadixNode)));
        LB_TRY(pin.alloc((size_t) n * 4)); LB_TRY(plf.alloc((size_t) n * 4));
        if (ploc_radius == 0u) {
            /* 4. radix tree */
            hipLaunchKernelGGL(k_hierarchy, dim3(gridN), dim3(B), 0, 0, keys, (int) n, rnodes.as<RadixNode>(), pin.as<uint32_t>(), plf.as<uint32_t>());
        } else {
            /* 4'. PLOC (lbvh_steps.h): clusters in Morton order merge with their nearest neighbour until one is left */
            Buf ca_mn, ca_mx, cb_mn, cb_mx, nearest, flags, incl, scan_tmp, nl, nr, nc, npn, npp, leaf_pos;
            LB_TRY(ca_mn.alloc((size_t) n * 16)); LB_TRY(ca_mx.alloc((size_t) n * 16)); LB_TRY(cb_mn.alloc((size_t) n * 16)); LB_TRY(cb_mx.alloc((size_t) n * 16));
            LB_TRY(nearest.alloc((size_t) n * 4)); LB_TRY(flags.alloc((size_t) n * 8)); LB_TRY(incl.alloc((size_t) n * 8));
            LB_TRY(nl.alloc((size_t) n * 4)); LB_TRY(nr.alloc((size_t) n * 4)); LB_TRY(nc.alloc((size_t) n * 4));
            LB_TRY(npn.alloc((size_t) n * 4)); LB_TRY(npp.alloc((size_t) n * 4)); LB_TRY(leaf_pos.alloc((size_t) n * 4));
            size_t scan_bytes = 0;
            LB_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, flags.as<unsigned long long>(), incl.as<unsigned long long>(), (int) n));
            LB_TRY(scan_tmp.alloc(scan_bytes));
            PlocClusters ca{ca_mn.as<f4>(), ca_mx.as<f4>()}, cb{cb_mn.as<f4>(), cb_mx.as<f4>()};
            PlocNodes pn{nl.as<uint32_t>(), nr.as<uint32_t>(), nc.as<uint32_t>(), npn.as<uint32_t>(), npp.as<uint32_t>()};
            hipLaunchKernelGGL(k_ploc_init, dim3(gridN), dim3(B), 0, 0, lmn, lmx, n, ca);
            uint32_t m = n, node_base = 0u, iterations = 0u;
            while (m > 1u) {
                const dim3 g((m + B - 1) / B);
                hipLaunchKernelGGL(k_ploc_nearest, g, dim3(B), 0, 0, ca, m, ploc_radius, nearest.as<uint32_t>());
                hipLaunchKernelGGL(k_ploc_decide, g, dim3(B), 0, 0, nearest.as<uint32_t>(), m, flags.as<unsigned long long>());
                size_t sb = scan_bytes;
                LB_TRY(hipcub::DeviceScan::InclusiveSum(scan_tmp.p, sb, flags.as<unsigned long long>(), incl.as<unsigned long long>(), (int) m));
                hipLaunchKernelGGL(k_ploc_apply, g, dim3(B), 0, 0, ca, cb, pn, nearest.as<uint32_t>(), m, flags.as<unsigned long long>(),
                                   incl.as<unsigned long long>(), node_base);
                unsigned long long total = 0ull;
                LB_TRY(hipMemcpy(&total, incl.as<unsigned long long>() + (m - 1), 8, hipMemcpyDeviceToHost));
                const uint32_t merged = (uint32_t) (total >> 32), left = (uint32_t) total;
                if (merged == 0u || left + merged != m) return "lbvh: a PLOC iteration merged nothing";
                node_base += merged; m = left;
                std::swap(ca, cb);
                if (++iterations > 100000u) return "lbvh: PLOC did not terminate";
            }
            if (node_base != n - 1u) return "lbvh: PLOC node count";
            out.ploc_iterations = iterations;
            lap("sort + PLOC");
            {   /* treelet restructuring (lbvh_steps.h): one launch per sweep, a thread per triangle climbing the tree */
                /* two sweeps (NORI_HIP_TREELET_SWEEPS overrides).  Round 3 ran none above 2^20 triangles: a sweep cost 340 ms on the
                   10 M-triangle terrain.  Round 4 (profiles/r4_09_treelet_wave.txt): a wave per treelet took that to 198 ms -- and showed
                   that the dynamic programme had not been the cost: every arrival at a node was bracketed by __threadfence(), i.e. a
                   write-back of the L2 and an invalidation of the L1 (15 M per sweep).  With the tree data read and written at agent
                   scope instead (coh_ld / coh_st, lbvh_steps.h) and no cache-wide fence: **20 - 28 ms per sweep**, same trees.
                   10 M triangles, PLOC + 0 / 1 / 2 / 3 / 5 sweeps: build 29 / 49 / 78 / 106 / 162 ms, wf_extend (64 spp) 33.4 / 33.0 / 32.5 /
                   32.4 / 32.5 ms (host SAH tree, built in 2.4 s: 29.4).  328 k-triangle AO scene: wf_extend 12.0 ms against the host
                   tree's 11.8; Cornell box, BVH2: node tests per ray 9.31 -> 8.49 (host SAH: 8.77). */
                /* [r6] then parallel re-insertion (lbvh_steps.h); how many of each: build_tuning (lbvh.h).  Node tests per ray of a small
                   render, PLOC + 2 sweeps against + 8 iterations + 1 sweep (tools/reinsert_probe.py, the steps on the CPU): Cornell box 8.56 ->
                   8.32 (host SAH tree 8.79), pa5 table 11.46 -> 10.89 (9.63, with spatial splits), terrain of 200 k triangles 27.8 -> 26.0 (24.6) */
                const BuildTuning rp = build_tuning(n);
                const int sweeps = rp.sweeps;
                if (sweeps > 0 || rp.iterations > 0) {
                    Buf t_mn, t_mx, t_cost, t_visits;
                    LB_TRY(t_mn.alloc((size_t) n * 16)); LB_TRY(t_mx.alloc((size_t) n * 16)); LB_TRY(t_cost.alloc((size_t) n * 4)); LB_TRY(t_visits.alloc((size_t) n * 4));
                    TreeletData td{t_mn.as<f4>(), t_mx.as<f4>(), t_cost.as<float>(), lmn, lmx};
                    TreeletParams tp; tp.c_node = 1.0f; tp.c_tri = 1.0f;
                    /* the (subset, partition) pairs of the dynamic programme for 3 .. 7 leaves, by subset size, partitions in the serial
                       loop's order (treelet_first_partition / treelet_next_partition): 1 388 pairs */
                    std::vector<TreeletPair> pairs;
                    TreeletTable tab; std::memset(&tab, 0, sizeof(tab));
                    for (int k = 3; k <= kTreeletLeaves; ++k)
                        for (int sz = 2; sz <= k + 1; ++sz) {
                            tab.start[k][sz] = (uint32_t) pairs.size();
                            if (sz > k) break;
                            for (int S = 3; S < (1 << k); ++S) {
                                if (__builtin_popcount((unsigned) S) != sz) continue;
                                int rank = 0;
                                for (int P = treelet_first_partition(S); P != 0; P = treelet_next_partition(S, P), ++rank) {
                                    TreeletPair pr; pr.S = (unsigned char) S; pr.P = (unsigned char) P; pr.rank = (unsigned char) rank; pr.pad = 0;
                                    pairs.push_back(pr);
                                }
                            }
                        }
                    Buf t_pairs;
                    LB_TRY(t_pairs.alloc(pairs.size() * sizeof(TreeletPair)));
                    LB_TRY(hipMemcpy(t_pairs.p, pairs.data(), pairs.size() * sizeof(TreeletPair), hipMemcpyHostToDevice));
                    tab.pairs = t_pairs.as<TreeletPair>();
                    const bool serial = getenv("NORI_HIP_TREELET_SERIAL") != nullptr && atoi(getenv("NORI_HIP_TREELET_SERIAL")) != 0;      /* one thread per treelet (A/B, tests) */
                    auto sweep = [&]() -> hipError_t {
                        hipError_t e = hipMemsetAsync(t_visits.p, 0, (size_t) n * 4, 0);
                        if (e != hipSuccess) return e;
                        if (serial) hipLaunchKernelGGL(k_treelet, dim3((n + 63) / 64), dim3(64), 0, 0, pn, td, tp, t_visits.as<uint32_t>(), n);
                        else hipLaunchKernelGGL(k_treelet_wave, dim3((n + 64 * kTreeletWaves - 1) / (64 * kTreeletWaves)), dim3(64 * kTreeletWaves), 0, 0, pn, td, tp, tab, t_visits.as<uint32_t>(), n);
                        return hipGetLastError();
                    };
                    for (int sw = 0; sw < sweeps; ++sw) LB_TRY(sweep());
                    lap("treelet sweeps");
                    if (rp.iterations > 0) {
                        const uint32_t n_inner = n - 1u, n_slots = 2u * n - 1u;
                        Buf r_lock, r_rkey, r_target, r_pivot, r_win, r_moved;
                        LB_TRY(r_lock.alloc((size_t) n_slots * 8)); LB_TRY(r_rkey.alloc((size_t) n_slots * 8));
                        LB_TRY(r_target.alloc((size_t) n_slots * 4)); LB_TRY(r_pivot.alloc((size_t) n_slots * 4)); LB_TRY(r_win.alloc((size_t) n_slots * 4));
                        LB_TRY(r_moved.alloc(4)); LB_TRY(hipMemsetAsync(r_moved.p, 0, 4, 0));
                        ReinsData rd{r_lock.as<unsigned long long>(), r_rkey.as<unsigned long long>(), r_target.as<uint32_t>(), r_pivot.as<uint32_t>(), r_win.as<uint32_t>()};
                        auto refit = [&]() -> hipError_t {
                            hipError_t e = hipMemsetAsync(t_visits.p, 0, (size_t) n * 4, 0);
                            if (e != hipSuccess) return e;
                            hipLaunchKernelGGL(k_reins_refit, dim3(gridN), dim3(B), 0, 0, pn, td, rd, tp, t_visits.as<uint32_t>(), n);
                            return hipGetLastError();
                        };
                        LB_TRY(refit());      /* (boxes, counts and costs of ALL nodes: with no sweep before, nothing has computed them yet) */
                        lap("re-insertion: boxes + refit");
                        for (int it = 0; it < rp.iterations; ++it) {
                            const uint32_t phase = (uint32_t) it % rp.stride, cands = (n_slots - phase + rp.stride - 1u) / rp.stride;
                            const dim3 g((cands + 63u) / 64u);      /* one wave per workgroup: a long search holds up 63 others, not 255 */
                            LB_TRY(hipMemsetAsync(r_lock.p, 0, (size_t) n_slots * 8, 0));
                            hipLaunchKernelGGL(k_reins_search, g, dim3(64), 0, 0, pn, td, rd, n_inner, n_slots, phase, rp.stride);
                            lap("re-insertion: search + mark");
                            hipLaunchKernelGGL(k_reins_check, g, dim3(64), 0, 0, pn, rd, n_inner, n_slots, phase, rp.stride, r_moved.as<uint32_t>());
                            hipLaunchKernelGGL(k_reins_apply, g, dim3(64), 0, 0, pn, rd, n_inner, n_slots, phase, rp.stride);
                            lap("re-insertion: check + move");
                            LB_TRY(refit());
                            lap("re-insertion: refit");
                        }
                        LB_TRY(hipMemcpy(&out.reinserted, r_moved.p, 4, hipMemcpyDeviceToHost));
                        for (int sw = 0; sw < rp.sweeps_after; ++sw) LB_TRY(sweep());
                        lap("sweeps after");
                    }
                    LB_TRY(hipGetLastError());
                    LB_TRY(hipDeviceSynchronize());      /* the buffers go out of scope here */
                }
            }
            /* the order of the tree's leaves, left to right: every node covers a contiguous range of it */
            LB_TRY(o_tri2.alloc((size_t) n * 4)); LB_TRY(l_mn2.alloc((size_t) n * 16)); LB_TRY(l_mx2.alloc((size_t) n * 16));
            hipLaunchKernelGGL(k_ploc_leaf_positions, dim3(gridN), dim3(B), 0, 0, pn, n, order, lmn, lmx, leaf_pos.as<uint32_t>(), o_tri2.as<uint32_t>(), l_mn2.as<f4>(), l_mx2.as<f4>());
            hipLaunchKernelGGL(k_ploc_finish, dim3(gridN), dim3(B), 0, 0, pn, n - 1u, leaf_pos.as<uint32_t>(), rnodes.as<RadixNode>(), pin.as<uint32_t>(), plf.as<uint32_t>());
            LB_TRY(hipDeviceSynchronize());
            order = o_tri2.as<uint32_t>(); lmn = l_mn2.as<f4>(); lmx = l_mx2.as<f4>();
        }
        /* 5. segment tree of boxes */
        N = 1; while (N < n) N <<= 1;
        LB_TRY(tmin.alloc((size_t) 2 * N * sizeof(f4))); LB_TRY(tmax.alloc((size_t) 2 * N * sizeof(f4)));
        hipLaunchKernelGGL(k_leaf_boxes, dim3((N + B - 1) / B), dim3(B), 0, 0, lmn, lmx, n, N, tmin.as<f4>(), tmax.as<f4>());
        for (uint32_t first = N >> 1; first >= 1; first >>= 1) {
            hipLaunchKernelGGL(k_tree_level, dim3((first + B - 1) / B), dim3(B), 0, 0, first, first, tmin.as<f4>(), tmax.as<f4>());
            if (first == 1) break;
        }
        /* which small subtrees become one leaf */
        CollapseParams cp; cp.c_pair = 1.5f; cp.c_node = 1.0f;      /* measured on the 10 M-triangle terrain: 1.5 .. 4 trace alike, 1.5 gives the smaller tree */
        if (const char *e = getenv("NORI_HIP_LBVH_PAIR_COST")) cp.c_pair = std::max(0.01f, (float) atof(e));
        LB_TRY(collapse.alloc((size_t) n * 4));
        hipLaunchKernelGGL(k_collapse, dim3(gridN), dim3(B), 0, 0, rnodes.as<RadixNode>(), n - 1, tmin.as<f4>(), tmax.as<f4>(), N, cp, collapse.as<uint32_t>());
        LB_TRY(keep.alloc((size_t) n * 4)); LB_TRY(node_index.alloc((size_t) n * 4));
        hipLaunchKernelGGL(k_mark_leaves, dim3(gridN), dim3(B), 0, 0, rnodes.as<RadixNode>(), n - 1, collapse.as<uint32_t>(), pin.as<uint32_t>(),
                           leaf_cnt.as<uint32_t>(), leaf_pairs.as<uint32_t>(), keep.as<uint32_t>());
    }
    {
        size_t scan_bytes = 0;
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, leaf_pairs.as<uint32_t>(), pair_start.as<uint32_t>(), (int) n));
        Buf scan_tmp; LB_TRY(scan_tmp.alloc(scan_bytes));
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, leaf_pairs.as<uint32_t>(), pair_start.as<uint32_t>(), (int) n));
    }
    uint32_t last[2] = {0, 0};
    LB_TRY(hipMemcpy(&last[0], pair_start.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(&last[1], leaf_pairs.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost));
    const uint32_t n_pairs = last[0] + last[1] + pair_base;
    out.n_pairs = n_pairs;

    /* 7. pair records */
    f4 *d_tris = nullptr;
    LB_TRY(hipMalloc((void **) &d_tris, (size_t) std::max<uint32_t>(n_pairs, 1) * kPairQuads * sizeof(f4)));
    out.d_tris = d_tris;
    if (pair_base) LB_TRY(hipMemset(d_tris, 0, kPairQuads * sizeof(f4)));
    hipLaunchKernelGGL(k_emit_pairs, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, d_tri_mesh, order, n,
                       leaf_cnt.as<uint32_t>(), pair_start.as<uint32_t>(), d_tris, pair_base);

    if (n <= 4) {
        f4 *d_nodes = nullptr;
        LB_TRY(hipMalloc((void **) &d_nodes, kNodeQuads * sizeof(f4)));
        LB_TRY(hipMemset(d_nodes, 0, kNodeQuads * sizeof(f4)));
        out.d_nodes = d_nodes; out.root = (int32_t) ~((0u << 3) | ((n + 1u) / 2u - 1u));
        out.n_nodes = 0; out.n_leaves = 1; out.max_depth = 0;
    } else if (wide) {
        /* WIDE nodes: level by level from the root (radix node 0) */
        Buf fa, fb, cnt, kids, n_kids, is_wide, wide_index;
        LB_TRY(fa.alloc((size_t) n * 4)); LB_TRY(fb.alloc((size_t) n * 4)); LB_TRY(cnt.alloc(4));
        LB_TRY(kids.alloc((size_t) n * 16)); LB_TRY(n_kids.alloc((size_t) n * 4));
        LB_TRY(is_wide.alloc((size_t) n * 4)); LB_TRY(wide_index.alloc((size_t) n * 4));
        LB_TRY(hipMemset(is_wide.p, 0, (size_t) n * 4)); LB_TRY(hipMemset(n_kids.p, 0, (size_t) n * 4));
        const uint32_t one = 1u, zero = 0u;
        LB_TRY(hipMemcpy(is_wide.p, &one, 4, hipMemcpyHostToDevice));            /* radix node 0 = the root */
        LB_TRY(hipMemcpy(fa.p, &zero, 4, hipMemcpyHostToDevice));
        uint32_t count = 1, levels = 0;
        Buf *cur = &fa, *nxt = &fb;
        while (count > 0) {
            LB_TRY(hipMemset(cnt.p, 0, 4));
            hipLaunchKernelGGL(k_wide_expand, dim3((count + B - 1) / B), dim3(B), 0, 0, rnodes.as<RadixNode>(), collapse.as<uint32_t>(), tmin.as<f4>(), tmax.as<f4>(), N,
                               cur->as<uint32_t>(), count, nxt->as<uint32_t>(), cnt.as<uint32_t>(), kids.as<uint32_t>(), n_kids.as<uint32_t>(), is_wide.as<uint32_t>());
            LB_TRY(hipMemcpy(&count, cnt.p, 4, hipMemcpyDeviceToHost));
            std::swap(cur, nxt);
            if (++levels > 4096) return "lbvh: wide levels did not terminate";
        }
        uint32_t n_wide = 0;
        {
            size_t scan_bytes = 0;
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_wide.as<uint32_t>(), wide_index.as<uint32_t>(), (int) (n - 1)));
            Buf scan_tmp; LB_TRY(scan_tmp.alloc(scan_bytes));
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, is_wide.as<uint32_t>(), wide_index.as<uint32_t>(), (int) (n - 1)));
            uint32_t tail[2] = {0, 0};
            LB_TRY(hipMemcpy(&tail[0], wide_index.as<uint32_t>() + (n - 2), 4, hipMemcpyDeviceToHost));
            LB_TRY(hipMemcpy(&tail[1], is_wide.as<uint32_t>() + (n - 2), 4, hipMemcpyDeviceToHost));
            n_wide = tail[0] + tail[1];
        }
        f4 *d_nodes = nullptr;
        LB_TRY(hipMalloc((void **) &d_nodes, (size_t) std::max<uint32_t>(n_wide, 1) * kNodeQuads * sizeof(f4)));
        out.d_nodes = d_nodes;
        hipLaunchKernelGGL(k_emit_wide, dim3(gridN), dim3(B), 0, 0, rnodes.as<RadixNode>(), n - 1, tmin.as<f4>(), tmax.as<f4>(), N, collapse.as<uint32_t>(),
                           pair_start.as<uint32_t>(), is_wide.as<uint32_t>(), wide_index.as<uint32_t>(), kids.as<uint32_t>(), n_kids.as<uint32_t>(), d_nodes);
        out.root = 0; out.n_nodes = n_wide; out.n_leaves = 0; out.max_depth = 3 * levels; out.wide = true;
        LB_TRY(hipGetLastError());
    } else {
        /* 6. nodes: the radix nodes that are not inside a collapsed subtree, renumbered densely (root stays 0) */
        uint32_t n_nodes = 0;
        {
            size_t scan_bytes = 0;
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep.as<uint32_t>(), node_index.as<uint32_t>(), (int) (n - 1)));
            Buf scan_tmp; LB_TRY(scan_tmp.alloc(scan_bytes));
            LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, keep.as<uint32_t>(), node_index.as<uint32_t>(), (int) (n - 1)));
            uint32_t tail[2] = {0, 0};
            LB_TRY(hipMemcpy(&tail[0], node_index.as<uint32_t>() + (n - 2), 4, hipMemcpyDeviceToHost));
            LB_TRY(hipMemcpy(&tail[1], keep.as<uint32_t>() + (n - 2), 4, hipMemcpyDeviceToHost));
            n_nodes = tail[0] + tail[1];
        }
        f4 *d_nodes = nullptr;
        LB_TRY(hipMalloc((void **) &d_nodes, (size_t) std::max<uint32_t>(n_nodes, 1) * kNodeQuads * sizeof(f4)));
        out.d_nodes = d_nodes;
        hipLaunchKernelGGL(k_emit_nodes, dim3(gridN), dim3(B), 0, 0, rnodes.as<RadixNode>(), n - 1, tmin.as<f4>(), tmax.as<f4>(), N,
                           collapse.as<uint32_t>(), pair_start.as<uint32_t>(), keep.as<uint32_t>(), node_index.as<uint32_t>(), d_nodes);
        /* 8. depth */
        Buf md; LB_TRY(md.alloc(4)); LB_TRY(hipMemset(md.p, 0, 4));
        hipLaunchKernelGGL(k_depth, dim3(gridN), dim3(B), 0, 0, pin.as<uint32_t>(), plf.as<uint32_t>(), keep.as<uint32_t>(), n, md.as<unsigned int>());
        unsigned int depth = 0;
        LB_TRY(hipMemcpy(&depth, md.p, 4, hipMemcpyDeviceToHost));
        out.root = 0; out.n_nodes = n_nodes; out.n_leaves = 0; out.max_depth = depth;
        LB_TRY(hipGetLastError());
    }
    lap("leaves, boxes, emission");
    LB_TRY(hipEventRecord(e1, 0));
    LB_TRY(hipEventSynchronize(e1));
    LB_TRY(hipEventElapsedTime(&out.build_ms, e0, e1));
    return std::string();
}

} // namespace nrt
