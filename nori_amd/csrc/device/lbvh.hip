/*
 * lbvh.hip -- Accel::build on the GPU: a linear BVH (Morton order + radix
 * tree) emitted directly in the traversal layout of rt_types.h.
 *
 * The reference's Accel::build is a no-op (src/accel.cpp:19-21) and its
 * rayIntersect scans every triangle; a BVH is this project's replacement, and
 * for the 10-M-triangle configuration (BASELINE config 5) a host build would
 * dominate wall-clock, so the whole build runs in HIP kernels:
 *
 *   1. k_scene_bounds   per-triangle boxes (src/mesh.cpp:78-83) -> scene box
 *                       (float atomics on order-preserving integer keys)
 *   2. k_morton         63-bit Morton code of the box centre (src/mesh.cpp:85-90
 *                       uses the vertex centroid; any centre orders as well)
 *   3. hipcub radix sort of (code, triangle) pairs
 *   4. k_hierarchy      Karras 2012: one thread per internal node finds its key
 *                       range and split with clz(key_i ^ key_j); equal codes are
 *                       told apart by their sorted position
 *   5. k_leaf_boxes / k_tree_level   padded triangle boxes in sorted order and
 *                       a min/max segment tree over them (one launch per level):
 *                       every node covers a CONTIGUOUS sorted range, so its box
 *                       is a range query -- no bottom-up atomics, no
 *                       inter-workgroup visibility hazards
 *   6. k_collapse       which subtrees of <= 4 triangles become one leaf (surface-area heuristic)
 *   7. k_mark_leaves / scans / k_emit_pairs / k_emit_nodes   the surviving radix nodes renumbered
 *                       densely as 64-B two-child-box records; the leaves' triangles as 96-B
 *                       de-indexed pair records, leaf by leaf
 *   8. k_depth          longest root-to-leaf chain (sizes the LDS stack)
 *   wide = true (scenes beyond the caches): instead of 7's two-box records the tree is emitted as WIDE nodes (BVH4,
 *       quantised child boxes; rt_types.h).  k_wide_expand walks the kept radix nodes top-down, one launch per level
 *       of the wide tree: a wide node takes its radix node's two children and twice replaces the inner child of largest
 *       surface area by that child's children; the inner children that remain are the wide nodes of the next level.
 *       k_emit_wide quantises the (<= 4) child boxes against their union (rt_wide.h) -- the radix nodes in between vanish.
 *
 * Numerically collinear triangles (rt_types.h, tri_box_pad) get Morton bit 63: the root separates them
 * from the spatial hierarchy, their boxes are infinite.
 *
 * Any valid BVH returns the same hits as the linear scan (conservative node
 * test + tie rule in rt_trace.h); tests/test_gpu_parity.py checks this builder
 * against the oracle's brute force as well.  The SAH builder (scene_prep.cpp)
 * gives trees that traverse faster and stays the default for small scenes.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <string>

#include "lbvh.h"
#include "rt_wide.h"

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

__device__ __forceinline__ void tri_box(const f4 *pos, const uint32_t *idx, uint32_t t, f3 &mn, f3 &mx) {
    const f3 a = xyz(pos[idx[3 * (size_t) t]]), b = xyz(pos[idx[3 * (size_t) t + 1]]), c = xyz(pos[idx[3 * (size_t) t + 2]]);
    mn = mk3(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)));
    mx = mk3(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)));
}

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

/* spread the low 21 bits of v to every third bit */
__device__ __forceinline__ unsigned long long expand21(unsigned long long v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

/* 63-bit Morton code of the triangle's box centre (21 bits per axis: a 2M^3 grid keeps the
   triangles of multi-million-triangle meshes in distinct cells; with 10 bits per axis whole
   clusters shared a code and were split in index order) */
__global__ void k_morton(const f4 *pos, const uint32_t *idx, uint32_t n, f3 smin, f3 sinv, unsigned long long *keys, uint32_t *vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    f3 mn, mx; tri_box(pos, idx, t, mn, mx);
    {   /* numerically collinear triangles (rt_types.h, tri_box_pad) sort behind everything else: bit 63 makes
           the root split them from the spatial hierarchy */
        const f3 p0 = xyz(pos[idx[3 * (size_t) t]]), p1 = xyz(pos[idx[3 * (size_t) t + 1]]), p2 = xyz(pos[idx[3 * (size_t) t + 2]]);
        bool unbounded;
        (void) tri_box_pad(p1 - p0, p2 - p0, 0.0f, unbounded);
        if (unbounded) { keys[t] = 0x8000000000000000ull; vals[t] = t; return; }
    }
    const float cx = (0.5f * (mn.x + mx.x) - smin.x) * sinv.x, cy = (0.5f * (mn.y + mx.y) - smin.y) * sinv.y,
                cz = (0.5f * (mn.z + mx.z) - smin.z) * sinv.z;
    const float S = 2097152.0f, M = 2097151.0f;
    const unsigned long long ix = (unsigned long long) fminf(fmaxf(cx * S, 0.0f), M);
    const unsigned long long iy = (unsigned long long) fminf(fmaxf(cy * S, 0.0f), M);
    const unsigned long long iz = (unsigned long long) fminf(fmaxf(cz * S, 0.0f), M);
    keys[t] = (expand21(ix) << 2) | (expand21(iy) << 1) | expand21(iz);
    vals[t] = t;
}

/* internal node i: children (bit 31 set = leaf primitive position), key range, parent links */
struct RadixNode { uint32_t left, right, lo, hi; };
constexpr uint32_t kLeafBit = 0x80000000u;

/* common-prefix length of the (key, position) pairs i and j: equal codes are told apart by their
   position in the sorted order, so every pair has a distinct prefix length */
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long) x) : 64 + __clz(i ^ j);
}

__global__ void k_hierarchy(const unsigned long long *keys, int n, RadixNode *nodes, uint32_t *parent_inner, uint32_t *parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    RadixNode nd;
    nd.lo = (uint32_t) lo; nd.hi = (uint32_t) hi;
    if (lo == gamma) { nd.left = kLeafBit | (uint32_t) gamma; parent_leaf[gamma] = (uint32_t) i; }
    else { nd.left = (uint32_t) gamma; parent_inner[gamma] = (uint32_t) i; }
    if (hi == gamma + 1) { nd.right = kLeafBit | (uint32_t) (gamma + 1); parent_leaf[gamma + 1] = (uint32_t) i; }
    else { nd.right = (uint32_t) (gamma + 1); parent_inner[gamma + 1] = (uint32_t) i; }
    nodes[i] = nd;
    if (i == 0) parent_inner[0] = 0xffffffffu;
}

/* segment tree over the sorted, padded triangle boxes: entries [N + k] */
__global__ void k_leaf_boxes(const f4 *pos, const uint32_t *idx, const uint32_t *order, uint32_t n, uint32_t N, float pad0,
                             f3 smin, f3 smax, f4 *tmin, f4 *tmax) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    f4 mn4, mx4;
    if (k < n) {
        const uint32_t g = order[k];
        f3 mn, mx; tri_box(pos, idx, g, mn, mx);
        const f3 p0 = xyz(pos[idx[3 * (size_t) g]]), p1 = xyz(pos[idx[3 * (size_t) g + 1]]), p2 = xyz(pos[idx[3 * (size_t) g + 2]]);
        bool unbounded;
        const float pad = tri_box_pad(p1 - p0, p2 - p0, pad0, unbounded);      /* slivers: rt_types.h */
        if (unbounded) { mn = mk3(-kBoxInf); mx = mk3(kBoxInf); }
        mn4.x = mn.x - pad; mn4.y = mn.y - pad; mn4.z = mn.z - pad; mx4.x = mx.x + pad; mx4.y = mx.y + pad; mx4.z = mx.z + pad;
    } else {
        mn4.x = mn4.y = mn4.z = kInf; mx4.x = mx4.y = mx4.z = -kInf;
    }
    mn4.w = mx4.w = 0.0f;
    tmin[N + k] = mn4; tmax[N + k] = mx4;
}

__global__ void k_tree_level(uint32_t first, uint32_t count, f4 *tmin, f4 *tmax) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const uint32_t i = first + k;
    const f4 a = tmin[2 * i], b = tmin[2 * i + 1], c = tmax[2 * i], d = tmax[2 * i + 1];
    f4 mn, mx;
    mn.x = fminf(a.x, b.x); mn.y = fminf(a.y, b.y); mn.z = fminf(a.z, b.z); mn.w = 0.0f;
    mx.x = fmaxf(c.x, d.x); mx.y = fmaxf(c.y, d.y); mx.z = fmaxf(c.z, d.z); mx.w = 0.0f;
    tmin[i] = mn; tmax[i] = mx;
}

__device__ __forceinline__ void range_box(const f4 *tmin, const f4 *tmax, uint32_t N, uint32_t lo, uint32_t hi, f3 &mn, f3 &mx) {
    mn = mk3(kInf); mx = mk3(-kInf);
    uint32_t l = lo + N, r = hi + N + 1;
    while (l < r) {
        if (l & 1u) { const f4 a = tmin[l], b = tmax[l]; ++l;
            mn = mk3(fminf(mn.x, a.x), fminf(mn.y, a.y), fminf(mn.z, a.z)); mx = mk3(fmaxf(mx.x, b.x), fmaxf(mx.y, b.y), fmaxf(mx.z, b.z)); }
        if (r & 1u) { --r; const f4 a = tmin[r], b = tmax[r];
            mn = mk3(fminf(mn.x, a.x), fminf(mn.y, a.y), fminf(mn.z, a.z)); mx = mk3(fmaxf(mx.x, b.x), fmaxf(mx.y, b.y), fmaxf(mx.z, b.z)); }
        l >>= 1; r >>= 1;
    }
}

/* A subtree of <= 4 triangles (contiguous in sorted order) can become ONE leaf (its triangles stored as
   pairs, rt_types.h) or stay split.  k_collapse decides per radix node with the surface-area heuristic:
       leaf : A(node) * pairs * Cpair          split : A(node) * Cnode + best(left) + best(right)
   (a leaf step tests a pair of triangles, 1.5 - 2.5x the work of a node step).  collapse[i] = 1: node i is a
   leaf wherever it is reached.  Nodes above 4 triangles are always inner nodes. */
struct CollapseParams { float c_pair, c_node; };

__device__ __forceinline__ float box_area(f3 mn, f3 mx) {
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

__device__ float best_cost(const RadixNode *nodes, const f4 *tmin, const f4 *tmax, uint32_t N, uint32_t child, CollapseParams cp, bool *collapse_out) {
    f3 mn, mx;
    if (child & kLeafBit) {
        const uint32_t k = child & ~kLeafBit;
        range_box(tmin, tmax, N, k, k, mn, mx);
        return box_area(mn, mx) * cp.c_pair;
    }
    const RadixNode nd = nodes[child];
    range_box(tmin, tmax, N, nd.lo, nd.hi, mn, mx);
    const float area = box_area(mn, mx);
    const float leaf = area * (float) ((nd.hi - nd.lo + 2u) / 2u) * cp.c_pair;
    const float split = area * cp.c_node + best_cost(nodes, tmin, tmax, N, nd.left, cp, nullptr) + best_cost(nodes, tmin, tmax, N, nd.right, cp, nullptr);
    if (collapse_out) *collapse_out = leaf <= split;
    return fminf(leaf, split);
}

__global__ void k_collapse(const RadixNode *nodes, uint32_t n_inner, const f4 *tmin, const f4 *tmax, uint32_t N, CollapseParams cp, uint32_t *collapse) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    const RadixNode nd = nodes[i];
    bool c = false;
    if (nd.hi - nd.lo + 1 <= 4u) (void) best_cost(nodes, tmin, tmax, N, i, cp, &c);      /* subtree of <= 3 inner nodes */
    collapse[i] = c ? 1u : 0u;
}

/* is `child` a leaf (primitive or collapsed subtree)?  [lo, hi] = its sorted range */
__device__ __forceinline__ bool child_range(const RadixNode *nodes, const uint32_t *collapse, uint32_t child, uint32_t &lo, uint32_t &hi) {
    if (child & kLeafBit) { lo = hi = child & ~kLeafBit; return true; }
    lo = nodes[child].lo; hi = nodes[child].hi;
    return collapse[child] != 0u;
}

/* the first pair of the leaf starting at sorted position lo is pair_start[lo] (exclusive scan of the
   per-leaf pair counts) */
__device__ __forceinline__ int32_t child_link(const RadixNode *nodes, const uint32_t *collapse, const uint32_t *pair_start, const uint32_t *node_index,
                                              uint32_t child, uint32_t &lo, uint32_t &hi, uint32_t pair_base = 0u) {
    if (!child_range(nodes, collapse, child, lo, hi)) return (int32_t) node_index[child];
    const uint32_t cnt = hi - lo + 1;
    return (int32_t) ~(((pair_start[lo] + pair_base) << 3) | ((cnt + 1u) / 2u - 1u));
}

/* ---- WIDE emission ---- */
/* one thread per wide node of the current level: choose its children, queue the inner ones for the next level */
__global__ void k_wide_expand(const RadixNode *nodes, const uint32_t *collapse, const f4 *tmin, const f4 *tmax, uint32_t N,
                              const uint32_t *frontier, uint32_t count, uint32_t *next, uint32_t *next_count,
                              uint32_t *kids, uint32_t *n_kids, uint32_t *is_wide) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t i = frontier[t];
    const RadixNode nd = nodes[i];
    uint32_t kid[4] = {nd.left, nd.right, 0u, 0u};
    int n = 2;
    while (n < 4) {
        int best = -1; float bestArea = -1.0f;
        for (int k = 0; k < n; ++k) {
            uint32_t lo, hi;
            if (child_range(nodes, collapse, kid[k], lo, hi)) continue;      /* a leaf stays */
            f3 mn, mx; range_box(tmin, tmax, N, lo, hi, mn, mx);
            float a = box_area(mn, mx);
            if (!(a < kInf)) a = kInf;                                        /* unbounded subtree first */
            if (a > bestArea) { bestArea = a; best = k; }
        }
        if (best < 0) break;
        const RadixNode c = nodes[kid[best]];
        kid[best] = c.left; kid[n++] = c.right;
    }
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
    float mn[4][3], mx[4][3]; int32_t link[4];
    const int n = (int) n_kids[i];
    for (int k = 0; k < n; ++k) {
        uint32_t lo, hi;
        link[k] = child_link(nodes, collapse, pair_start, wide_index, kids[4 * (size_t) i + k], lo, hi, 1u);      /* pair 0 = the null pair */
        f3 a, b; range_box(tmin, tmax, N, lo, hi, a, b);
        mn[k][0] = a.x; mn[k][1] = a.y; mn[k][2] = a.z; mx[k][0] = b.x; mx[k][1] = b.y; mx[k][2] = b.z;
    }
    f4 q[4];
    wide_pack(n, mn, mx, link, q);
    f4 *dst = out + (size_t) wide_index[i] * kNodeQuads;
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
}

/* leaf_cnt[lo] = triangles of the leaf that starts at sorted position lo, leaf_pairs[lo] = its pairs;
   keep[i] = 1 for the radix nodes that survive as BVH nodes: not collapsed and not below a collapsed node */
__global__ void k_mark_leaves(const RadixNode *nodes, uint32_t n_inner, const uint32_t *collapse, const uint32_t *parent_inner,
                              uint32_t *leaf_cnt, uint32_t *leaf_pairs, uint32_t *keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    const RadixNode nd = nodes[i];
    bool reachable = collapse[i] == 0u;
    for (uint32_t p = i; reachable && p != 0u && nodes[p].hi - nodes[p].lo + 1 <= 4u; ) {      /* ancestors that might have collapsed */
        p = parent_inner[p];
        if (p == 0xffffffffu) break;
        if (collapse[p]) reachable = false;
    }
    keep[i] = reachable ? 1u : 0u;
    if (!reachable) return;
    uint32_t lo, hi;
    if (child_range(nodes, collapse, nd.left, lo, hi)) { leaf_cnt[lo] = hi - lo + 1; leaf_pairs[lo] = (hi - lo + 2) / 2; }
    if (child_range(nodes, collapse, nd.right, lo, hi)) { leaf_cnt[lo] = hi - lo + 1; leaf_pairs[lo] = (hi - lo + 2) / 2; }
}

__global__ void k_emit_nodes(const RadixNode *nodes, uint32_t n_inner, const f4 *tmin, const f4 *tmax, uint32_t N,
                             const uint32_t *collapse, const uint32_t *pair_start, const uint32_t *keep, const uint32_t *node_index, f4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !keep[i]) return;
    const RadixNode nd = nodes[i];
    uint32_t llo, lhi, rlo, rhi;
    const int32_t cl = child_link(nodes, collapse, pair_start, node_index, nd.left, llo, lhi), cr = child_link(nodes, collapse, pair_start, node_index, nd.right, rlo, rhi);
    f3 lmn, lmx, rmn, rmx;
    range_box(tmin, tmax, N, llo, lhi, lmn, lmx);
    range_box(tmin, tmax, N, rlo, rhi, rmn, rmx);
    const float a0[3] = {lmn.x, lmn.y, lmn.z}, a1[3] = {lmx.x, lmx.y, lmx.z}, b0[3] = {rmn.x, rmn.y, rmn.z}, b1[3] = {rmx.x, rmx.y, rmx.z};
    f4 q[4];
    node_pack(a0, a1, b0, b1, cl, cr, q);
    f4 *dst = out + (size_t) node_index[i] * kNodeQuads;      /* dense: only the surviving nodes, in radix-tree order */
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
}

/* one thread per leaf start: the leaf's triangles, de-indexed, as pair records */
__global__ void k_emit_pairs(const f4 *pos, const uint32_t *idx, const uint32_t *tri_mesh, const uint32_t *order, uint32_t n,
                             const uint32_t *leaf_cnt, const uint32_t *pair_start, f4 *out, uint32_t pair_base) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t cnt = leaf_cnt[k];
    if (cnt == 0u) return;
    for (uint32_t t = 0; t < ((cnt + 1u) / 2u) * 2u; ++t) {
        f4 q[kPairQuads];
        f4 *dst = out + (size_t) (pair_start[k] + pair_base + t / 2u) * kPairQuads;
        if ((t & 1u) == 0u) for (int j = 0; j < kPairQuads; ++j) q[j].x = q[j].y = q[j].z = q[j].w = 0.0f;
        else for (int j = 0; j < kPairQuads; ++j) q[j] = dst[j];
        if (t < cnt) {
            const uint32_t g = order[k + t];
            const f3 p0 = xyz(pos[idx[3 * (size_t) g]]), p1 = xyz(pos[idx[3 * (size_t) g + 1]]), p2 = xyz(pos[idx[3 * (size_t) g + 2]]);
            const f3 e1 = p1 - p0, e2 = p2 - p0;      /* the subtraction mesh.cpp:43 performs per ray */
            const float a0[3] = {p0.x, p0.y, p0.z}, a1[3] = {e1.x, e1.y, e1.z}, a2[3] = {e2.x, e2.y, e2.z};
            pair_pack(q, (int) (t & 1u), a0, a1, a2, g, tri_mesh[g]);
        } else {
            const float z[3] = {0.0f, 0.0f, 0.0f};
            pair_pack(q, (int) (t & 1u), z, z, z, kNoTriangle, kNoTriangle);
        }
        for (int j = 0; j < kPairQuads; ++j) dst[j] = q[j];
    }
}

__global__ void k_depth(const uint32_t *parent_inner, const uint32_t *parent_leaf, uint32_t n, unsigned int *max_depth) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    unsigned int depth = 1;
    uint32_t p = parent_leaf[k];
    while (p != 0u && p != 0xffffffffu && depth < 4096u) { p = parent_inner[p]; ++depth; }
    atomicMax(max_depth, depth);
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

std::string build_bvh_lbvh_device(const DevScene &dev, const uint32_t *d_tri_mesh, LbvhDeviceResult &out, bool wide) {
    out = LbvhDeviceResult();
    if (dev.n_triangles <= 4) wide = false;                  /* a single leaf: no nodes at all */
    const uint32_t pair_base = wide ? 1u : 0u;               /* wide trees reserve pair 0 as the all-zero pair of unused slots */
    const uint32_t n = dev.n_triangles;
    hipEvent_t e0, e1;
    LB_TRY(hipEventCreate(&e0)); LB_TRY(hipEventCreate(&e1));
    LB_TRY(hipEventRecord(e0, 0));

    if (n == 0) return "lbvh: empty scene";
    const int B = 256;
    const uint32_t gridN = (n + B - 1) / B;

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

    /* 7 first: the triangle records only need the sorted order; tiny scenes are one leaf */
    Buf keys_a, keys_b, vals_a, vals_b;
    LB_TRY(keys_a.alloc((size_t) n * 8)); LB_TRY(keys_b.alloc((size_t) n * 8));
    LB_TRY(vals_a.alloc((size_t) n * 4)); LB_TRY(vals_b.alloc((size_t) n * 4));
    hipLaunchKernelGGL(k_morton, dim3(gridN), dim3(B), 0, 0, dev.positions, dev.indices, n, smin, sinv, keys_a.as<unsigned long long>(), vals_a.as<uint32_t>());
    size_t temp_bytes = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_a.as<unsigned long long>(), keys_b.as<unsigned long long>(),
                                              vals_a.as<uint32_t>(), vals_b.as<uint32_t>(), (int) n, 0, 64));
    Buf temp; LB_TRY(temp.alloc(temp_bytes));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.as<unsigned long long>(), keys_b.as<unsigned long long>(),
                                              vals_a.as<uint32_t>(), vals_b.as<uint32_t>(), (int) n, 0, 64));
    const unsigned long long *keys = keys_b.as<unsigned long long>();
    const uint32_t *order = vals_b.as<uint32_t>();      /* sorted position -> global triangle */

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
        /* 4. radix tree */
        LB_TRY(rnodes.alloc((size_t) (n - 1) * sizeof(RadixNode)));
        LB_TRY(pin.alloc((size_t) n * 4)); LB_TRY(plf.alloc((size_t) n * 4));
        hipLaunchKernelGGL(k_hierarchy, dim3(gridN), dim3(B), 0, 0, keys, (int) n, rnodes.as<RadixNode>(), pin.as<uint32_t>(), plf.as<uint32_t>());
        /* 5. segment tree of boxes */
        N = 1; while (N < n) N <<= 1;
        LB_TRY(tmin.alloc((size_t) 2 * N * sizeof(f4))); LB_TRY(tmax.alloc((size_t) 2 * N * sizeof(f4)));
        hipLaunchKernelGGL(k_leaf_boxes, dim3((N + B - 1) / B), dim3(B), 0, 0, dev.positions, dev.indices, order, n, N, pad, smin, smax, tmin.as<f4>(), tmax.as<f4>());
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
        hipLaunchKernelGGL(k_depth, dim3(gridN), dim3(B), 0, 0, pin.as<uint32_t>(), plf.as<uint32_t>(), n, md.as<unsigned int>());
        unsigned int depth = 0;
        LB_TRY(hipMemcpy(&depth, md.p, 4, hipMemcpyDeviceToHost));
        out.root = 0; out.n_nodes = n_nodes; out.n_leaves = 0; out.max_depth = depth;
        LB_TRY(hipGetLastError());
    }
    LB_TRY(hipEventRecord(e1, 0));
    LB_TRY(hipEventSynchronize(e1));
    LB_TRY(hipEventElapsedTime(&out.build_ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    return std::string();
}

} // namespace nrt
