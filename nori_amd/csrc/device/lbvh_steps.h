/*
 * lbvh_steps.h -- the per-element steps of the device BVH builders (lbvh.hip), written as plain functions of an
 * index so that the kernels are one-line wrappers and the CPU test harness (tests/emu) can run the same steps as
 * loops: Morton keys, the radix tree of Karras 2012, and PLOC (parallel locally-ordered clustering, Meister &
 * Bittner 2018) -- the builder for scenes where the tree's quality decides the render time.
 *
 * PLOC, per iteration over the m clusters that are left (initially the triangles in Morton order):
 *   ploc_nearest   every cluster looks R places to the left and right for the neighbour whose union with it has the
 *                  smallest surface area (ties: the position with the longest common prefix)
 *   ploc_decide    two clusters that chose EACH OTHER merge; the lower position leads.  Exclusive scans of the
 *                  `lead` and `stays` flags give the new node's id and the cluster's place in the next iteration
 *   ploc_apply     the leader writes the new node (children, triangle count, parent links) and the merged cluster
 * Node ids grow in creation order, so the last node is the root; ploc_finish renumbers them root = 0.
 * Every merge is local in Morton order, but the union is chosen by AREA, not by the next Morton bit: where the radix
 * tree must split a cell along the axis whose bit comes next -- e.g. by HEIGHT across a patch of terrain -- PLOC pairs
 * what is actually close.
 *
 * The rest of the pipeline (segment tree of boxes, leaf collapse, pair / node emission) needs every node to cover a
 * CONTIGUOUS range of an ordering of the triangles.  Any binary tree has one: the order of its leaves, left to right.
 * ploc_first_position computes it by walking up the parent links (offset = triangles in the left siblings passed on
 * the way), ploc_finish writes the nodes with their ranges, and the builder permutes the triangles into that order.
 *
 * Numerically collinear triangles (rt_types.h, tri_box_pad: unbounded boxes) only cluster with each other, and with the
 * rest at the very end: cost 0 among themselves, kPlocMixed with a bounded cluster.
 */
#pragma once
#include "rt_types.h"
#include "rt_wide.h"

namespace nrt {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoParent = 0xffffffffu;
constexpr float kPlocMixed = 1e38f;

/* internal node: children (bit 31 set = leaf, position of its triangle in the builder's order), covered range */
struct RadixNode { uint32_t left, right, lo, hi; };

/* ---- Morton keys ---- */
/* spread the low 21 bits of v to every third bit */
NORI_HD unsigned long long expand21(unsigned long long v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
/* 63-bit Morton code of a box centre (21 bits per axis: a 2M^3 grid keeps the triangles of multi-million-triangle
   meshes in distinct cells) */
NORI_HD unsigned long long morton63(f3 mn, f3 mx, f3 smin, f3 sinv) {
    const float cx = (0.5f * (mn.x + mx.x) - smin.x) * sinv.x, cy = (0.5f * (mn.y + mx.y) - smin.y) * sinv.y,
                cz = (0.5f * (mn.z + mx.z) - smin.z) * sinv.z;
    const float S = 2097152.0f, M = 2097151.0f;
    const unsigned long long ix = (unsigned long long) fminf(fmaxf(cx * S, 0.0f), M);
    const unsigned long long iy = (unsigned long long) fminf(fmaxf(cy * S, 0.0f), M);
    const unsigned long long iz = (unsigned long long) fminf(fmaxf(cz * S, 0.0f), M);
    return (expand21(ix) << 2) | (expand21(iy) << 1) | expand21(iz);
}

/* ---- radix tree (Karras 2012) ---- */
NORI_HD int clz64(unsigned long long x) { return __builtin_clzll(x); }
NORI_HD int clz32(uint32_t x) { return __builtin_clz(x); }
/* common-prefix length of the (key, position) pairs i and j: equal codes are told apart by their
   position in the sorted order, so every pair has a distinct prefix length */
NORI_HD int radix_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? clz64(x) : 64 + clz32((uint32_t) (i ^ j));
}
/* internal node i of the radix tree over n sorted keys; writes the parent links of its children */
NORI_HD RadixNode radix_node(const unsigned long long *keys, int n, int i, uint32_t *parent_inner, uint32_t *parent_leaf) {
    const int d = (radix_delta(keys, n, i, i + 1) - radix_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = radix_delta(keys, n, i, i - d);
    int lmax = 2;
    while (radix_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (radix_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = radix_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (radix_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    RadixNode nd;
    nd.lo = (uint32_t) lo; nd.hi = (uint32_t) hi;
    if (lo == gamma) { nd.left = kLeafBit | (uint32_t) gamma; parent_leaf[gamma] = (uint32_t) i; }
    else { nd.left = (uint32_t) gamma; parent_inner[gamma] = (uint32_t) i; }
    if (hi == gamma + 1) { nd.right = kLeafBit | (uint32_t) (gamma + 1); parent_leaf[gamma + 1] = (uint32_t) i; }
    else { nd.right = (uint32_t) (gamma + 1); parent_inner[gamma + 1] = (uint32_t) i; }
    if (i == 0) parent_inner[0] = kNoParent;
    return nd;
}

/* ---- PLOC ---- */
/* a cluster: box, id (leaf: kLeafBit | sorted position; else node id in creation order), triangle count; the box of
   an unbounded cluster is (-kBoxInf, kBoxInf)^3 */
struct PlocClusters {
    f4 *mn;      /* xyz = lower corner, w = id bits */
    f4 *mx;      /* xyz = upper corner, w = count bits */
};
/* a node in creation order */
struct PlocNodes {
    uint32_t *left, *right, *count;       /* children ids (as cluster ids), triangles below */
    uint32_t *parent_node, *parent_prim;  /* by node id / by sorted position */
};

NORI_HD float half_area(f3 mn, f3 mx) {
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return dx * dy + (dy * dz + dz * dx);
}
NORI_HD bool ploc_unbounded(f4 mx) { return !(mx.x < kBoxInf); }
NORI_HD float ploc_cost(f4 amn, f4 amx, f4 bmn, f4 bmx) {
    const bool ua = ploc_unbounded(amx), ub = ploc_unbounded(bmx);
    if (ua || ub) return (ua && ub) ? 0.0f : kPlocMixed;
    return half_area(mk3(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z)),
                     mk3(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z)));
}
/* position of cluster i's nearest neighbour among the m clusters, searching `radius` places each way */
NORI_HD uint32_t ploc_nearest(const PlocClusters &c, uint32_t m, uint32_t i, uint32_t radius) {
    const f4 amn = c.mn[i], amx = c.mx[i];
    const uint32_t j0 = i > radius ? i - radius : 0u, j1 = (i + radius < m - 1u) ? i + radius : m - 1u;
    float best = kInf; uint32_t arg = i;
    for (uint32_t j = j0; j <= j1; ++j) {
        if (j == i) continue;
        const float cost = ploc_cost(amn, amx, c.mn[j], c.mx[j]);
        /* equal costs (duplicated triangles, sheets of equal boxes): the position sharing the longest prefix with i, so that
           whole runs pair up (2k, 2k + 1) at once instead of one pair per iteration at the run's end; the key (cost, i ^ j) is
           symmetric in i and j, so the best pair overall is always mutual and every iteration merges something */
        if (cost < best || (cost == best && (i ^ j) < (i ^ arg))) { best = cost; arg = j; }
    }
    return arg;
}
/* flags of cluster i for the scans: lead = it creates a node (mutual choice, lower position), stays = it has a place in
   the next iteration (everything but the partner that is merged away) */
NORI_HD void ploc_decide(const uint32_t *nearest, uint32_t i, uint32_t &lead, uint32_t &stays) {
    const uint32_t j = nearest[i];
    const bool mutual = j != i && nearest[j] == i;
    lead = (mutual && i < j) ? 1u : 0u;
    stays = (!mutual || i < j) ? 1u : 0u;
}
NORI_HD uint32_t ploc_count_of(f4 mx) { return f2u(mx.w); }
/* cluster i moves to (or, leading a merge, creates node node_base + lead_rank and moves as that node to) place
   stays_rank of the next iteration */
NORI_HD void ploc_apply(const PlocClusters &in, const PlocClusters &out, const PlocNodes &nodes, const uint32_t *nearest, uint32_t i,
                        uint32_t lead, uint32_t stays, uint32_t lead_rank, uint32_t stays_rank, uint32_t node_base) {
    if (!stays) return;
    f4 mn = in.mn[i], mx = in.mx[i];
    if (lead) {
        const uint32_t j = nearest[i];
        const f4 bmn = in.mn[j], bmx = in.mx[j];
        const uint32_t id = node_base + lead_rank, a = f2u(mn.w), b = f2u(bmn.w);
        const uint32_t cnt = ploc_count_of(mx) + ploc_count_of(bmx);
        nodes.left[id] = a; nodes.right[id] = b; nodes.count[id] = cnt;
        if (a & kLeafBit) nodes.parent_prim[a & ~kLeafBit] = id; else nodes.parent_node[a] = id;
        if (b & kLeafBit) nodes.parent_prim[b & ~kLeafBit] = id; else nodes.parent_node[b] = id;
        mn.x = fminf(mn.x, bmn.x); mn.y = fminf(mn.y, bmn.y); mn.z = fminf(mn.z, bmn.z); mn.w = u2f(id);
        mx.x = fmaxf(mx.x, bmx.x); mx.y = fmaxf(mx.y, bmx.y); mx.z = fmaxf(mx.z, bmx.z); mx.w = u2f(cnt);
    }
    out.mn[stays_rank] = mn; out.mx[stays_rank] = mx;
}

/* triangles below a child id */
NORI_HD uint32_t ploc_child_count(const PlocNodes &nodes, uint32_t child) { return (child & kLeafBit) ? 1u : nodes.count[child]; }
/* first position, in left-to-right leaf order, of the subtree `id` (a node id, or kLeafBit | sorted position):
   the triangles of the left siblings passed on the way up to the root (node n_nodes - 1) */
NORI_HD uint32_t ploc_first_position(const PlocNodes &nodes, uint32_t n_nodes, uint32_t id) {
    uint32_t pos = 0u, c = id;
    while (c != n_nodes - 1u) {
        const uint32_t p = (c & kLeafBit) ? nodes.parent_prim[c & ~kLeafBit] : nodes.parent_node[c];
        if (nodes.right[p] == c) pos += ploc_child_count(nodes, nodes.left[p]);
        c = p;
    }
    return pos;
}
/* node `id` (creation order) as the radix-tree record the emission kernels read; its index there is n_nodes - 1 - id (root = 0).
   leaf_pos[k] = ploc_first_position of the triangle at sorted position k. */
NORI_HD RadixNode ploc_finish(const PlocNodes &nodes, uint32_t n_nodes, uint32_t id, const uint32_t *leaf_pos,
                              uint32_t *parent_inner, uint32_t *parent_leaf) {
    const uint32_t self = n_nodes - 1u - id;
    RadixNode nd;
    nd.lo = ploc_first_position(nodes, n_nodes, id);
    nd.hi = nd.lo + nodes.count[id] - 1u;
    const uint32_t a = nodes.left[id], b = nodes.right[id];
    if (a & kLeafBit) { nd.left = kLeafBit | leaf_pos[a & ~kLeafBit]; parent_leaf[leaf_pos[a & ~kLeafBit]] = self; }
    else { nd.left = n_nodes - 1u - a; parent_inner[n_nodes - 1u - a] = self; }
    if (b & kLeafBit) { nd.right = kLeafBit | leaf_pos[b & ~kLeafBit]; parent_leaf[leaf_pos[b & ~kLeafBit]] = self; }
    else { nd.right = n_nodes - 1u - b; parent_inner[n_nodes - 1u - b] = self; }
    if (self == 0u) parent_inner[0] = kNoParent;
    return nd;
}

/* ---- boxes ---- */
NORI_HD void tri_box(const f4 *pos, const uint32_t *idx, uint32_t t, f3 &mn, f3 &mx) {
    const f3 a = xyz(pos[idx[3 * (size_t) t]]), b = xyz(pos[idx[3 * (size_t) t + 1]]), c = xyz(pos[idx[3 * (size_t) t + 2]]);
    mn = mk3(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)));
    mx = mk3(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)));
}
NORI_HD bool tri_unbounded(const f4 *pos, const uint32_t *idx, uint32_t t) {
    const f3 p0 = xyz(pos[idx[3 * (size_t) t]]), p1 = xyz(pos[idx[3 * (size_t) t + 1]]), p2 = xyz(pos[idx[3 * (size_t) t + 2]]);
    bool unbounded;
    (void) tri_box_pad(p1 - p0, p2 - p0, 0.0f, unbounded);
    return unbounded;
}
/* the box a triangle enters the tree with: padded (slivers wider: rt_types.h), unbounded = (-kBoxInf, kBoxInf)^3 */
NORI_HD void tri_leaf_box(const f4 *pos, const uint32_t *idx, uint32_t g, float pad0, f4 &mn4, f4 &mx4) {
    f3 mn, mx; tri_box(pos, idx, g, mn, mx);
    const f3 p0 = xyz(pos[idx[3 * (size_t) g]]), p1 = xyz(pos[idx[3 * (size_t) g + 1]]), p2 = xyz(pos[idx[3 * (size_t) g + 2]]);
    bool unbounded;
    const float pad = tri_box_pad(p1 - p0, p2 - p0, pad0, unbounded);
    if (unbounded) { mn = mk3(-kBoxInf); mx = mk3(kBoxInf); }
    mn4.x = mn.x - pad; mn4.y = mn.y - pad; mn4.z = mn.z - pad; mx4.x = mx.x + pad; mx4.y = mx.y + pad; mx4.z = mx.z + pad;
    mn4.w = mx4.w = 0.0f;
}
/* ---- references: triangles split before the tree is built (after Karras & Aila 2013, section 4) ----
 * The builders cluster REFERENCES -- (triangle, box) -- and a triangle may enter as several, each with the box of the part of the
 * triangle inside one cell; the same triangle then hangs in several leaves (as under the host builder's spatial splits: a leaf whose
 * part the ray misses is not visited, the hit is the scan's either way).  Which triangles, and how often:
 *   split_priority   p = (2^-level V^(2/3))^(1/3), V = the VOLUME of the triangle's box in units of the scene box, level = that of the
 *                    coarsest plane of the Morton grid that crosses the box.  Karras & Aila weigh the box's area beyond what a finely
 *                    split triangle's boxes would have; that also cuts every large axis-parallel triangle -- floors, walls, table tops --
 *                    whose parts are flat boxes the size of many objects: a clustering builder stacks them into a subtree of their own
 *                    that every ray along the floor walks (pa5 table, node tests per ray 10.8 -> 15.1 for triangle tests 8.3 -> 5.1;
 *                    with the volume 9.8 and 5.0: profiles/r6_20_split_cpu_counts.txt).  A flat box overlaps nothing: cutting it
 *                    saves a triangle test at the price of node tests.  A box with volume holds other geometry in its empty part
 *   split_count      min(limit, floor(D p)): D = one cut per `scale` typical priorities of the scene (lbvh.h, split_scale_D), lowered
 *                    if the counts exceed the budget; limit = what split_inside finds in the box
 *   split_emit       the triangle's <= 1 + count references: a part is cut by the coarsest grid plane that crosses its box -- parts of
 *                    different triangles end on the same planes, so the clustering finds them side by side -- and hands its
 *                    remaining cuts to the halves in proportion to their boxes' areas.
 * A part's box is the box of (triangle clipped to the part's cell) evaluated in binary64, rounded outwards, padded as the whole
 * triangle's is (tri_leaf_box): every point of the triangle lies in some cell, hence in some part's box.  A triangle with no cut
 * keeps exactly the box and the Morton key it had without this step: where nothing is cut the builders' trees are what they were. */
struct RefOut { uint32_t *tri; f4 *mn, *mx; unsigned long long *key; };      /* by reference: triangle, padded box, Morton key */
constexpr int kSplitMaxParts = 64;

/* level (0 = the plane through the middle of the scene box) and position of the coarsest Morton-grid plane strictly inside
   (lo, hi) of an axis whose scene extent is [s0, s0 + 1 / sinv]; false: none down to the grid's 21 bits */
NORI_HD bool split_plane_axis(float lo, float hi, float s0, float sinv, int &level, float &pos) {
    if (!(hi > lo) || !(sinv > 0.0f)) return false;
    const double a = ((double) lo - (double) s0) * (double) sinv, b = ((double) hi - (double) s0) * (double) sinv;
    for (int l = 0; l < 21; ++l) {
        const double cells = (double) (2u << l);                 /* planes at odd multiples of 2^-(l+1) */
        const double k = floor(a * cells) + 1.0;                 /* first grid line above a at this resolution */
        /* (lines at even multiples belong to coarser levels: they were tried before) */
        if (k / cells < b) {
            const double x = (double) s0 + (k / cells) / (double) sinv;
            const float xf = (float) x;
            if (xf > lo && xf < hi) { level = l; pos = xf; return true; }
        }
    }
    return false;
}
NORI_HD bool split_plane(f3 mn, f3 mx, f3 smin, f3 sinv, int &axis, int &level, float &pos) {
    const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z}, s0[3] = {smin.x, smin.y, smin.z}, si[3] = {sinv.x, sinv.y, sinv.z};
    bool found = false; float bestExtent = 0.0f;
    level = 0;
    for (int ax = 0; ax < 3; ++ax) {
        int l = 0; float p = 0.0f;
        if (!split_plane_axis(lo[ax], hi[ax], s0[ax], si[ax], l, p)) continue;
        const float extent = (hi[ax] - lo[ax]) * si[ax];
        if (!found || l < level || (l == level && extent > bestExtent)) { found = true; axis = ax; level = l; pos = p; bestExtent = extent; }
    }
    return found;
}
/* cube root by additions, multiplications and divisions only (a seed from the bit pattern, four Newton steps): the same bits from the
   host compiler's and the device compiler's arithmetic, which the libraries' cbrtf do not promise -- the CPU harness and the device must
   count the same cuts */
NORI_HD float split_cbrt(float x) {
    if (!(x > 0.0f) || !(x < kInf)) return 0.0f;
    float y = u2f(f2u(x) / 3u + 0x2a514067u);
    for (int k = 0; k < 4; ++k) y = (2.0f * y + x / (y * y)) * (1.0f / 3.0f);
    return y;
}
NORI_HD float split_priority(const f4 *pos, const uint32_t *idx, uint32_t t, f3 smin, f3 sinv) {
    if (tri_unbounded(pos, idx, t)) return 0.0f;
    f3 mn, mx; tri_box(pos, idx, t, mn, mx);
    int axis, level; float where;
    if (!split_plane(mn, mx, smin, sinv, axis, level, where)) return 0.0f;
    /* in units of the scene box, so that the scale of the counts does not depend on the scene's */
    const f3 e = mk3((mx.x - mn.x) * sinv.x, (mx.y - mn.y) * sinv.y, (mx.z - mn.z) * sinv.z);
    const float side = split_cbrt(e.x * e.y * e.z), spare = side * side;      /* (volume)^(2/3): an area again; a flat box has none */
    if (!(spare > 0.0f)) return 0.0f;
    return split_cbrt(spare / (float) (1u << level));
}
NORI_HD uint32_t split_count(float priority, float D, uint32_t cap) {
    const float c = D * priority;
    return !(c >= 1.0f) ? 0u : (c >= (float) cap ? cap : (uint32_t) c);
}
/* What else lies in a triangle's box: a cut only pays where OTHER geometry stands in the empty part of the box -- the tilted face of
   a block in an empty room gains nothing from hugging boxes, only nodes.  The triangles' box centres are counted into a kSplitGrid^3
   grid over the scene box (split_grid_cell), the counts summed into a summed-volume table (entry (x, y, z) = centres in cells
   [0, x) x [0, y) x [0, z), dimension kSplitGrid + 1), and split_inside reads the number of centres in the cells a box touches with
   eight look-ups.  Integer counts: the same on any device, in any order. */
constexpr int kSplitGrid = 64;
NORI_HD int split_grid_coord(float v, float s0, float sinv) {
    const float c = (v - s0) * sinv * (float) kSplitGrid;
    return c >= (float) (kSplitGrid - 1) ? kSplitGrid - 1 : (c > 0.0f ? (int) c : 0);
}
NORI_HD uint32_t split_grid_cell(f3 mn, f3 mx, f3 smin, f3 sinv) {
    const int x = split_grid_coord(0.5f * (mn.x + mx.x), smin.x, sinv.x), y = split_grid_coord(0.5f * (mn.y + mx.y), smin.y, sinv.y), z = split_grid_coord(0.5f * (mn.z + mx.z), smin.z, sinv.z);
    return (uint32_t) ((z * kSplitGrid + y) * kSplitGrid + x);
}
NORI_HD uint32_t split_sat_at(const uint32_t *sat, int x, int y, int z) { return sat[(z * (kSplitGrid + 1) + y) * (kSplitGrid + 1) + x]; }
NORI_HD uint32_t split_inside(const uint32_t *sat, f3 mn, f3 mx, f3 smin, f3 sinv) {
    const int x0 = split_grid_coord(mn.x, smin.x, sinv.x), y0 = split_grid_coord(mn.y, smin.y, sinv.y), z0 = split_grid_coord(mn.z, smin.z, sinv.z);
    const int x1 = split_grid_coord(mx.x, smin.x, sinv.x) + 1, y1 = split_grid_coord(mx.y, smin.y, sinv.y) + 1, z1 = split_grid_coord(mx.z, smin.z, sinv.z) + 1;
    const uint32_t cells = split_sat_at(sat, x1, y1, z1) - split_sat_at(sat, x0, y1, z1) - split_sat_at(sat, x1, y0, z1) - split_sat_at(sat, x1, y1, z0)
                         + split_sat_at(sat, x0, y0, z1) + split_sat_at(sat, x0, y1, z0) + split_sat_at(sat, x1, y0, z0) - split_sat_at(sat, x0, y0, z0);
    /* the box's share of the cells it touches (a thin slab touches a whole layer of cells and holds next to nothing of it) */
    const float fx = fminf(1.0f, (mx.x - mn.x) * sinv.x * (float) kSplitGrid / (float) (x1 - x0)), fy = fminf(1.0f, (mx.y - mn.y) * sinv.y * (float) kSplitGrid / (float) (y1 - y0)),
                fz = fminf(1.0f, (mx.z - mn.z) * sinv.z * (float) kSplitGrid / (float) (z1 - z0));
    return (uint32_t) ((float) cells * (fx * (fy * fz)));
}
/* box of the triangle's part inside the cell [cmn, cmx] (binary64 clipping, rounded outwards); false: nothing there */
NORI_HD bool split_part_box(const double tri[3][3], f3 cmn, f3 cmx, f3 &mn, f3 &mx) {
    double a[10][3], b[10][3];
    int n = 3;
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) a[i][k] = tri[i][k];
    const double lo[3] = {cmn.x, cmn.y, cmn.z}, hi[3] = {cmx.x, cmx.y, cmx.z};
    for (int plane = 0; plane < 6 && n > 0; ++plane) {
        const int axis = plane >> 1; const bool above = (plane & 1) == 0; const double where = above ? lo[axis] : hi[axis];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const double *p = a[i], *q = a[(i + 1) % n];
            const double dp = above ? p[axis] - where : where - p[axis], dq = above ? q[axis] - where : where - q[axis];
            if (dp >= 0.0) { for (int k = 0; k < 3; ++k) b[m][k] = p[k]; ++m; }
            if ((dp > 0.0 && dq < 0.0) || (dp < 0.0 && dq > 0.0)) {
                const double t = dp / (dp - dq);
                for (int k = 0; k < 3; ++k) b[m][k] = p[k] + t * (q[k] - p[k]);
                b[m][axis] = where; ++m;
            }
        }
        n = m < 10 ? m : 10;
        for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) a[i][k] = b[i][k];
    }
    if (n <= 0) return false;
    double dmn[3] = {a[0][0], a[0][1], a[0][2]}, dmx[3] = {a[0][0], a[0][1], a[0][2]};
    for (int i = 1; i < n; ++i) for (int k = 0; k < 3; ++k) { dmn[k] = a[i][k] < dmn[k] ? a[i][k] : dmn[k]; dmx[k] = a[i][k] > dmx[k] ? a[i][k] : dmx[k]; }
    float fmn[3], fmx[3];
    for (int k = 0; k < 3; ++k) {
        float f = (float) dmn[k]; if ((double) f > dmn[k]) f = u2f(f > 0.0f ? f2u(f) - 1u : (f < 0.0f ? f2u(f) + 1u : 0x80000001u)); fmn[k] = f;
        float g = (float) dmx[k]; if ((double) g < dmx[k]) g = u2f(g > 0.0f ? f2u(g) + 1u : (g < 0.0f ? f2u(g) - 1u : 0x00000001u)); fmx[k] = g;
    }
    mn = mk3(fmn[0], fmn[1], fmn[2]); mx = mk3(fmx[0], fmx[1], fmx[2]);
    return true;
}
/* the references of triangle t, at most 1 + cuts (a part no grid plane crosses keeps its cuts unused): counted (write = false: the
   builder's scan places every triangle's run) or written at out[first ...]; returns how many */
NORI_HD uint32_t split_emit(const f4 *pos, const uint32_t *idx, uint32_t t, uint32_t cuts, float pad0, f3 smin, f3 sinv, bool write, const RefOut &out, uint32_t first) {
    f3 tmn, tmx; tri_box(pos, idx, t, tmn, tmx);
    const f3 p0 = xyz(pos[idx[3 * (size_t) t]]), p1 = xyz(pos[idx[3 * (size_t) t + 1]]), p2 = xyz(pos[idx[3 * (size_t) t + 2]]);
    bool unbounded;
    const float pad = tri_box_pad(p1 - p0, p2 - p0, pad0, unbounded);
    if (cuts == 0u || unbounded) {      /* the whole triangle, exactly as tri_leaf_box / the Morton key of its box see it */
        if (write) {
            f4 a, b; tri_leaf_box(pos, idx, t, pad0, a, b);
            out.tri[first] = t; out.mn[first] = a; out.mx[first] = b;
            out.key[first] = unbounded ? 0x8000000000000000ull : morton63(tmn, tmx, smin, sinv);
        }
        return 1u;
    }
    const double tri[3][3] = {{p0.x, p0.y, p0.z}, {p1.x, p1.y, p1.z}, {p2.x, p2.y, p2.z}};
    struct Part { f3 mn, mx; uint32_t cuts; };      /* box of (triangle within the part's cell) = the cell of what it is cut into */
    Part stack[kSplitMaxParts];
    int sp = 0;
    uint32_t written = 0;
    stack[sp].mn = tmn; stack[sp].mx = tmx; stack[sp].cuts = cuts < (uint32_t) kSplitMaxParts - 1u ? cuts : (uint32_t) kSplitMaxParts - 1u; ++sp;
    while (sp > 0) {
        const Part pt = stack[--sp];
        int axis = 0, level = 0; float where = 0.0f;
        bool cut = pt.cuts > 0u && sp + 2 <= kSplitMaxParts && split_plane(pt.mn, pt.mx, smin, sinv, axis, level, where);
        Part lo = pt, hi = pt;
        if (cut) {
            f3 lo_cmx = pt.mx, hi_cmn = pt.mn;
            if (axis == 0) { lo_cmx.x = where; hi_cmn.x = where; } else if (axis == 1) { lo_cmx.y = where; hi_cmn.y = where; } else { lo_cmx.z = where; hi_cmn.z = where; }
            cut = split_part_box(tri, pt.mn, lo_cmx, lo.mn, lo.mx) && split_part_box(tri, hi_cmn, pt.mx, hi.mn, hi.mx);
        }
        if (!cut) {
            if (write) {
                f4 a, b;
                a.x = pt.mn.x - pad; a.y = pt.mn.y - pad; a.z = pt.mn.z - pad; a.w = 0.0f;
                b.x = pt.mx.x + pad; b.y = pt.mx.y + pad; b.z = pt.mx.z + pad; b.w = 0.0f;
                out.tri[first + written] = t; out.mn[first + written] = a; out.mx[first + written] = b;
                out.key[first + written] = morton63(pt.mn, pt.mx, smin, sinv);
            }
            ++written;
            continue;
        }
        const float wl = half_area(lo.mn, lo.mx), wh = half_area(hi.mn, hi.mx);
        const uint32_t rest = pt.cuts - 1u;
        uint32_t cl = (wl + wh) > 0.0f ? (uint32_t) ((float) rest * (wl / (wl + wh)) + 0.5f) : rest / 2u;
        if (cl > rest) cl = rest;
        lo.cuts = cl; hi.cuts = rest - cl;
        stack[sp++] = hi; stack[sp++] = lo;
    }
    return written;
}

/* ---- treelet restructuring (Karras & Aila 2013) of the PLOC tree ----
 * PLOC chooses every merge locally; what it leaves on the table sits inside small subtrees (a
 * top-down SAH rebuild ABOVE the subtrees recovers only a third of the gap to the host's SAH tree).  A treelet is a node
 * with the <= 7 subtrees that hang below it after repeatedly opening the child of largest surface area; for those <= 7
 * leaves the topology of minimal SAH cost is found exactly -- dynamic programming over their 2^7 subsets -- and the
 * treelet's inner nodes are rewired to it.  Bottom-up, whatever shape the tree has by now (a sweep rewires ids, so PLOC's
 * creation order stops being a schedule after the first one): treelet_climb starts at every triangle and walks up the
 * parent links; at a node the FIRST of the two arrivals stops, the second -- both subtrees below are finished --
 * optimizes the node's treelet and climbs on.  Which thread does a node is a matter of timing; what it computes is not.
 * Per node: box (xyz of two f4), SAH cost of its subtree  C(node) = c_node A(node) + C(left) + C(right),  C(triangle) =
 * c_tri A(triangle).  Treelets that contain an unbounded box (numerically collinear triangles) are left alone. */
constexpr int kTreeletLeaves = 7;
struct TreeletData { f4 *nmn, *nmx; float *cost; const f4 *lmn, *lmx; };      /* inner nodes by id: box, subtree cost; leaves by position: the references' padded boxes */
struct TreeletParams { float c_node, c_tri; };

/* How the treelet code reads and writes what OTHER waves of the same launch wrote or will read (links, counts, boxes, costs): on the
   device as relaxed atomics of agent scope -- loads and stores that go to the level at which all CUs of the GPU agree (sc1), past
   this CU's L1 and this XCD's L2.  A sweep hands a subtree from the waves that optimised it to the wave that optimises its parent
   through one counter; with plain accesses that hand-over needs __threadfence() on either side of it, i.e. a write-back of the whole
   L2 and an invalidation of the L1 per arrival (buffer_wbl2 sc1 / buffer_inv sc1: 15 M of them per sweep on 10 M triangles, most of
   the 200 - 340 ms a sweep took).  With every shared datum accessed at agent scope there is nothing dirty to write back and nothing
   stale to invalidate: waiting for the stores' acknowledgements (s_waitcnt vmcnt(0)) before the counter is all the order needed. */
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T coh_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void coh_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ f4 coh_ld4(const f4 *p) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long a = coh_ld(q), b = coh_ld(q + 1);
    f4 r; r.x = __uint_as_float((uint32_t) a); r.y = __uint_as_float((uint32_t) (a >> 32)); r.z = __uint_as_float((uint32_t) b); r.w = __uint_as_float((uint32_t) (b >> 32));
    return r;
}
__device__ __forceinline__ void coh_st4(f4 *p, const f4 &v) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    coh_st(q, (unsigned long long) __float_as_uint(v.x) | ((unsigned long long) __float_as_uint(v.y) << 32));
    coh_st(q + 1, (unsigned long long) __float_as_uint(v.z) | ((unsigned long long) __float_as_uint(v.w) << 32));
}
#else      /* the CPU harness, and the host pass over the device code */
template <class T> NORI_HD T coh_ld(const T *p) { return *p; }
template <class T> NORI_HD void coh_st(T *p, T v) { *p = v; }
NORI_HD f4 coh_ld4(const f4 *p) { return *p; }
NORI_HD void coh_st4(f4 *p, const f4 &v) { *p = v; }
#endif

NORI_HD void treelet_child(const PlocNodes &nodes, const TreeletData &td, TreeletParams tp, uint32_t child, f3 &mn, f3 &mx, float &cost, uint32_t &count) {
    if (child & kLeafBit) {
        mn = xyz(td.lmn[child & ~kLeafBit]); mx = xyz(td.lmx[child & ~kLeafBit]); count = 1u;
        cost = tp.c_tri * half_area(mn, mx);
    } else {
        mn = xyz(coh_ld4(&td.nmn[child])); mx = xyz(coh_ld4(&td.nmx[child])); count = coh_ld(&nodes.count[child]); cost = coh_ld(&td.cost[child]);
    }
}
NORI_HD void treelet_set_parent(const PlocNodes &nodes, uint32_t child, uint32_t parent) {
    if (child & kLeafBit) coh_st(&nodes.parent_prim[child & ~kLeafBit], parent); else coh_st(&nodes.parent_node[child], parent);
}
/* What the optimisation of one treelet works on: its <= 7 leaves (ids, boxes, subtree costs, triangle counts) and the inner node
   ids it owns (inner[0] = the treelet's root).  Lives on the stack of the one thread that optimises a treelet (treelet_optimize)
   or in the LDS of the wave that shares the work (lbvh.hip, treelet_optimize_wave): the same three steps either way --
   treelet_form, the dynamic programme over the subsets of the leaves, treelet_rewire. */
struct TreeletWork {
    uint32_t leaf[kTreeletLeaves], cnt[kTreeletLeaves], inner[kTreeletLeaves];
    f3 lmn[kTreeletLeaves], lmx[kTreeletLeaves]; float lcost[kTreeletLeaves];
    int k, n_inner;
};

/* box and cost of node `id` from its children (first sweep: nothing is known yet), then its treelet: the node's two children, and
   while fewer than seven, the inner one of largest area replaced by its children.  False: nothing to optimise (an unbounded box
   in reach, or two leaves only). */
NORI_HD bool treelet_form(const PlocNodes &nodes, const TreeletData &td, TreeletParams tp, uint32_t id, TreeletWork &w) {
    int k = 2, n_inner = 1;
    w.inner[0] = id;
    w.leaf[0] = coh_ld(&nodes.left[id]); w.leaf[1] = coh_ld(&nodes.right[id]);
    for (int i = 0; i < 2; ++i) treelet_child(nodes, td, tp, w.leaf[i], w.lmn[i], w.lmx[i], w.lcost[i], w.cnt[i]);
    const bool bounded = w.lmx[0].x < kBoxInf && w.lmx[1].x < kBoxInf;
    {   /* this node as it stands */
        const f3 mn = mk3(fminf(w.lmn[0].x, w.lmn[1].x), fminf(w.lmn[0].y, w.lmn[1].y), fminf(w.lmn[0].z, w.lmn[1].z));
        const f3 mx = mk3(fmaxf(w.lmx[0].x, w.lmx[1].x), fmaxf(w.lmx[0].y, w.lmx[1].y), fmaxf(w.lmx[0].z, w.lmx[1].z));
        f4 a, b; a.x = mn.x; a.y = mn.y; a.z = mn.z; a.w = 0.0f; b.x = mx.x; b.y = mx.y; b.z = mx.z; b.w = 0.0f;
        coh_st4(&td.nmn[id], a); coh_st4(&td.nmx[id], b);
        coh_st(&td.cost[id], bounded ? tp.c_node * half_area(mn, mx) + (w.lcost[0] + w.lcost[1]) : 1e30f);
    }
    w.k = k; w.n_inner = n_inner;
    if (!bounded) return false;
    while (k < kTreeletLeaves) {      /* open the inner treelet leaf of largest area */
        int best = -1; float bestArea = -1.0f;
        for (int i = 0; i < k; ++i) {
            if (w.leaf[i] & kLeafBit) continue;
            const float a = half_area(w.lmn[i], w.lmx[i]);
            if (a > bestArea) { bestArea = a; best = i; }
        }
        if (best < 0) break;
        const uint32_t open = w.leaf[best];
        w.inner[n_inner++] = open;
        const uint32_t a = coh_ld(&nodes.left[open]), b = coh_ld(&nodes.right[open]);
        w.leaf[best] = a; w.leaf[k] = b;
        treelet_child(nodes, td, tp, a, w.lmn[best], w.lmx[best], w.lcost[best], w.cnt[best]);
        treelet_child(nodes, td, tp, b, w.lmn[k], w.lmx[k], w.lcost[k], w.cnt[k]);
        if (!(w.lmx[best].x < kBoxInf) || !(w.lmx[k].x < kBoxInf)) return false;
        ++k;
    }
    w.k = k; w.n_inner = n_inner;
    return k >= 3;                      /* two leaves: one topology */
}

/* half the surface area of the union of the leaves in subset S */
NORI_HD float treelet_subset_area(const TreeletWork &w, int S) {
    f3 mn = mk3(kInf), mx = mk3(-kInf);
    for (int i = 0; i < w.k; ++i)
        if (S & (1 << i)) { mn = mk3(fminf(mn.x, w.lmn[i].x), fminf(mn.y, w.lmn[i].y), fminf(mn.z, w.lmn[i].z)); mx = mk3(fmaxf(mx.x, w.lmx[i].x), fmaxf(mx.y, w.lmx[i].y), fmaxf(mx.z, w.lmx[i].z)); }
    return half_area(mn, mx);
}

/* The partitions of subset S into two non-empty halves, each once: the halves P that keep S's lowest bit out, in the order
   first_partition, next_partition, ... until 0.  THE order: among partitions of equal cost the first one wins, here and in the
   wave's table (lbvh.hip), so that both forms build the same tree. */
NORI_HD int treelet_first_partition(int S) { const int delta = (S - 1) & S; return (-delta) & S; }
NORI_HD int treelet_next_partition(int S, int P) { const int delta = (S - 1) & S; return (P - delta) & S; }

/* the treelet's inner ids take the subsets of the optimal topology, top down */
NORI_HD void treelet_rewire(const PlocNodes &nodes, const TreeletData &td, uint32_t id, const TreeletWork &w, const float *copt, const unsigned char *part) {
    const int k = w.k, full = (1 << k) - 1;
    int stack_S[kTreeletLeaves]; uint32_t stack_id[kTreeletLeaves]; int sp = 0, used = 1;
    stack_S[sp] = full; stack_id[sp] = id; ++sp;
    while (sp > 0) {
        --sp;
        const int S = stack_S[sp]; const uint32_t me = stack_id[sp];
        const int halves[2] = {(int) part[S], S ^ (int) part[S]};
        uint32_t child_id[2];
        for (int h = 0; h < 2; ++h) {
            const int T = halves[h];
            if ((T & (T - 1)) == 0) {                        /* one leaf of the treelet */
                int i = 0; while (i < kTreeletLeaves - 1 && !(T & (1 << i))) ++i;
                child_id[h] = w.leaf[i];
            } else {
                child_id[h] = w.inner[used < kTreeletLeaves - 1 ? used : kTreeletLeaves - 1]; ++used;
                if (sp < kTreeletLeaves) { stack_S[sp] = T; stack_id[sp] = child_id[h]; ++sp; }
            }
            treelet_set_parent(nodes, child_id[h], me);
        }
        coh_st(&nodes.left[me], child_id[0]); coh_st(&nodes.right[me], child_id[1]);
        f3 mn = mk3(kInf), mx = mk3(-kInf); uint32_t c = 0u;
        for (int i = 0; i < k; ++i)
            if (S & (1 << i)) { mn = mk3(fminf(mn.x, w.lmn[i].x), fminf(mn.y, w.lmn[i].y), fminf(mn.z, w.lmn[i].z)); mx = mk3(fmaxf(mx.x, w.lmx[i].x), fmaxf(mx.y, w.lmx[i].y), fmaxf(mx.z, w.lmx[i].z)); c += w.cnt[i]; }
        f4 a, b; a.x = mn.x; a.y = mn.y; a.z = mn.z; a.w = 0.0f; b.x = mx.x; b.y = mx.y; b.z = mx.z; b.w = 0.0f;
        coh_st4(&td.nmn[me], a); coh_st4(&td.nmx[me], b); coh_st(&td.cost[me], copt[S]); coh_st(&nodes.count[me], c);
    }
}

/* one thread does it all: the CPU harness, and the device when NORI_HIP_TREELET_SERIAL is set */
NORI_HD void treelet_optimize(const PlocNodes &nodes, const TreeletData &td, TreeletParams tp, uint32_t id) {
    TreeletWork w;
    if (!treelet_form(nodes, td, tp, id, w)) return;
    /* dynamic programming over the subsets of the k leaves */
    const int k = w.k, full = (1 << k) - 1;
    float area[1 << kTreeletLeaves], copt[1 << kTreeletLeaves];
    unsigned char part[1 << kTreeletLeaves];
    for (int S = 1; S <= full; ++S) area[S] = treelet_subset_area(w, S);
    for (int i = 0; i < k; ++i) copt[1 << i] = w.lcost[i];
    for (int S = 3; S <= full; ++S) {
        if ((S & (S - 1)) == 0) continue;                   /* a single leaf */
        float best = kInf; int bestP = 0;
        int P = treelet_first_partition(S);
        do {
            const float c = copt[P] + copt[S ^ P];
            if (c < best) { best = c; bestP = P; }
            P = treelet_next_partition(S, P);
        } while (P != 0);
        copt[S] = tp.c_node * area[S] + best;
        part[S] = (unsigned char) bestP;
    }
    if (!(copt[full] < coh_ld(&td.cost[id]) * 0.99999f)) return;     /* nothing to gain: leave the subtree as PLOC built it */
    treelet_rewire(nodes, td, id, w, copt, part);
}

/* One sweep = treelet_climb for every triangle position k (visits[] zeroed before).  On the device the arrival counter is an
   atomic with a fence on either side: what the other subtree's thread wrote must be visible, and this CU's L1 may still
   hold lines from before it was written (MI355X: the vector L1 is not refreshed by other CUs' stores). */
NORI_HD void treelet_climb(const PlocNodes &nodes, const TreeletData &td, TreeletParams tp, uint32_t *visits, uint32_t root_id, uint32_t k) {
    uint32_t p = nodes.parent_prim[k];
    for (uint32_t guard = 0; guard < 4096u; ++guard) {
#if defined(__HIP_DEVICE_COMPILE__)
        __threadfence();
        const uint32_t before = atomicAdd(&visits[p], 1u);
        __threadfence();
#else
        const uint32_t before = visits[p]++;
#endif
        if (before == 0u) return;
        treelet_optimize(nodes, td, tp, p);
        if (p == root_id) return;
        p = coh_ld(&nodes.parent_node[p]);
    }
}

/* ---- parallel re-insertion (Meister & Bittner 2018) of the PLOC tree ----
 * The treelet sweeps repair what is wrong INSIDE seven-leaf neighbourhoods; what PLOC's Morton-local merges got wrong across
 * the tree -- a subtree that belongs under a node far away in the key order -- they cannot reach.  Re-insertion can: take a
 * subtree x (an inner node or a single triangle) out of the tree together with its parent p -- x's sibling moves up into p's
 * place -- and hang it, under p again, beside the node y where the sum of the inner nodes' areas (the tree's SAH cost) shrinks
 * most.  With p_0 = p, p_1, ... the ancestors of x, s_k the other child of p_k and R_k the union of B(s_0 .. s_k) (= the box of
 * p_k once x is gone; B(p_k) = R_k u B(x) exactly), the gain of moving x beside a node y of subtree(s_k) is
 *     A(p_0) + sum_{0 < i < k} (A(p_i) - A(R_i))  -  A(x u y)  -  sum_{y' above y, up to s_k} (A(y' u x) - A(y'))
 * -- the ancestors from p_k up have x below them before and after -- and  A(p_0) + sum_{0 < i < k} (..) - A(R_k)  beside the
 * shrunken p_k itself.  reins_search climbs from p to the root and walks every subtree(s_k) without a stack (parent links),
 * pruned by what the best place found so far gains.  All candidates of an iteration search the SAME tree; then
 *   reins_lock     a candidate marks every node whose links its move changes or whose position it relies on -- x, p, s_0, p_1, y,
 *                  y's parent and the nodes from y up to below p_k -- with (gain, x) by a 64-bit maximum
 *   reins_check    it wins if all its marks stand.  The winners' node sets are disjoint, and no winner moves into a subtree
 *                  that another winner moves (it would have marked that subtree's root, which the other marks as its x): the
 *                  moves commute and the result is a tree, whatever the order they are applied in
 *   reins_apply    relinks (two child links, three parent links, p's children)
 *   reins_refit    boxes, triangle counts and subtree costs bottom-up, as a sweep's climb does.
 * The root keeps its id (the last node): neither it nor its children move, nothing is hung above it.  Subtrees with an
 * unbounded box (numerically collinear triangles) stay where they are. */
struct ReinsData {
    unsigned long long *lock;      /* [2 n - 1] by slot: inner node id, or n - 1 + position of a triangle */
    unsigned long long *key;       /* [2 n - 1] the candidate's (gain bits << 32 | slot), 0 = no move */
    uint32_t *target, *pivot;      /* [2 n - 1] y and p_k (kNoParent: y is an ancestor of x, the move goes beside the shrunken y) */
    uint32_t *win;                 /* [2 n - 1] */
};
NORI_HD uint32_t reins_slot(uint32_t id, uint32_t n_inner) { return (id & kLeafBit) ? n_inner + (id & ~kLeafBit) : id; }
NORI_HD uint32_t reins_id(uint32_t slot, uint32_t n_inner) { return slot < n_inner ? slot : (kLeafBit | (slot - n_inner)); }
NORI_HD uint32_t reins_parent(const PlocNodes &nd, uint32_t id) { return (id & kLeafBit) ? nd.parent_prim[id & ~kLeafBit] : nd.parent_node[id]; }
NORI_HD void reins_box(const TreeletData &td, const ReinsData &rd, uint32_t id, f3 &mn, f3 &mx) {
    if (id & kLeafBit) { mn = xyz(td.lmn[id & ~kLeafBit]); mx = xyz(td.lmx[id & ~kLeafBit]); }
    else { mn = xyz(td.nmn[id]); mx = xyz(td.nmx[id]); }
}
NORI_HD float reins_union_area(f3 amn, f3 amx, f3 bmn, f3 bmx) {
    return half_area(mk3(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z)), mk3(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z)));
}

/* the best place for the subtree in `slot`: key[slot] = 0 (stay) or (gain, slot), target[slot] = y, pivot[slot] = p_k */
NORI_HD void reins_search(const PlocNodes &nd, const TreeletData &td, const ReinsData &rd, uint32_t n_inner, uint32_t slot) {
    const uint32_t root = n_inner - 1u, x = reins_id(slot, n_inner);
    rd.key[slot] = 0ull;
    if (x == root) return;
    const uint32_t p0 = reins_parent(nd, x);
    if (p0 == root) return;
    f3 xmn, xmx, bmn, bmx;
    reins_box(td, rd, x, xmn, xmx);
    if (!(xmx.x < kBoxInf)) return;
    const float ax = half_area(xmn, xmx);
    reins_box(td, rd, p0, bmn, bmx);
    const float ap0 = half_area(bmn, bmx);
    if (!(ap0 < kInf)) return;
    float base = ap0, best = 1e-6f * ap0, best_base = 0.0f;
    uint32_t best_y = kNoParent, best_pivot = kNoParent;
    uint32_t prev = x, piv = p0;
    f3 rmn = mk3(kInf), rmx = mk3(-kInf);
    for (uint32_t guard = 0; guard < 4096u; ++guard) {
        const uint32_t l = nd.left[piv], s = l == prev ? nd.right[piv] : l;
        {   /* subtree(s), left to right; growth = what the nodes above y (within the subtree) grow by when x hangs below them */
            uint32_t y = s; float growth = 0.0f; bool down = true;
            for (uint32_t steps = 0; steps < (1u << 26); ++steps) {
                if (down) {
                    reins_box(td, rd, y, bmn, bmx);
                    const float direct = reins_union_area(xmn, xmx, bmn, bmx), gain = base - growth - direct;
                    if (gain > best && !(piv == p0 && y == s)) { best = gain; best_base = base; best_y = y; best_pivot = piv; }
                    const float below = growth + (direct - half_area(bmn, bmx));
                    if (!(y & kLeafBit) && base - below - ax > best) { growth = below; y = nd.left[y]; continue; }
                    down = false;
                }
                if (y == s) break;
                const uint32_t par = reins_parent(nd, y);
                if (nd.left[par] == y) { y = nd.right[par]; down = true; }
                else {
                    y = par;
                    reins_box(td, rd, y, bmn, bmx);
                    growth -= reins_union_area(xmn, xmx, bmn, bmx) - half_area(bmn, bmx);
                }
            }
        }
        reins_box(td, rd, s, bmn, bmx);
        rmn = mk3(fminf(rmn.x, bmn.x), fminf(rmn.y, bmn.y), fminf(rmn.z, bmn.z)); rmx = mk3(fmaxf(rmx.x, bmx.x), fmaxf(rmx.y, bmx.y), fmaxf(rmx.z, bmx.z));
        if (piv == root) break;
        if (piv != p0) {
            const float ar = half_area(rmn, rmx), gain = base - ar;      /* beside the shrunken p_k */
            if (gain > best) { best = gain; best_base = base; best_y = piv; best_pivot = kNoParent; }
            reins_box(td, rd, piv, bmn, bmx);
            base += half_area(bmn, bmx) - ar;
        }
        prev = piv; piv = nd.parent_node[piv];
    }
    if (best_y == kNoParent) return;
    if (best_pivot != kNoParent) {      /* the walk's running sum drifts: the gain again, from y upwards */
        reins_box(td, rd, best_y, bmn, bmx);
        float g = best_base - reins_union_area(xmn, xmx, bmn, bmx);
        for (uint32_t c = reins_parent(nd, best_y), guard = 0; c != best_pivot && guard < 4096u; c = nd.parent_node[c], ++guard) {
            reins_box(td, rd, c, bmn, bmx);
            g -= reins_union_area(xmn, xmx, bmn, bmx) - half_area(bmn, bmx);
        }
        best = g;
        if (!(best > 1e-6f * ap0)) return;
    }
    rd.target[slot] = best_y; rd.pivot[slot] = best_pivot;
    rd.key[slot] = ((unsigned long long) f2u(best) << 32) | slot;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void reins_mark(unsigned long long *lock, unsigned long long key) { atomicMax(lock, key); }
#else
NORI_HD void reins_mark(unsigned long long *lock, unsigned long long key) { if (*lock < key) *lock = key; }
#endif
/* the nodes a move depends on: marked (acquire) or checked -- false as soon as one mark is another candidate's */
NORI_HD bool reins_locks(const PlocNodes &nd, const ReinsData &rd, uint32_t n_inner, uint32_t slot, bool acquire) {
    const unsigned long long key = rd.key[slot];
    if (key == 0ull) return false;
    const uint32_t x = reins_id(slot, n_inner), y = rd.target[slot], pivot = rd.pivot[slot];
    const uint32_t p0 = reins_parent(nd, x), g = nd.parent_node[p0];
    const uint32_t s0 = nd.left[p0] == x ? nd.right[p0] : nd.left[p0];
#define NORI_REINS_TOUCH(id) do { unsigned long long *lk = &rd.lock[reins_slot(id, n_inner)]; if (acquire) reins_mark(lk, key); else if (*lk != key) return false; } while (0)
    NORI_REINS_TOUCH(x); NORI_REINS_TOUCH(p0); NORI_REINS_TOUCH(s0); NORI_REINS_TOUCH(g);
    if (pivot == kNoParent) { NORI_REINS_TOUCH(y); NORI_REINS_TOUCH(nd.parent_node[y]); return true; }
    uint32_t c = y;
    for (uint32_t guard = 0; guard < 4096u; ++guard) {
        NORI_REINS_TOUCH(c);
        const uint32_t par = reins_parent(nd, c);
        if (par == pivot) break;
        c = par;
    }
    NORI_REINS_TOUCH(reins_parent(nd, y));
#undef NORI_REINS_TOUCH
    return true;
}
NORI_HD void reins_set_parent(const PlocNodes &nd, uint32_t child, uint32_t parent) {
    if (child & kLeafBit) nd.parent_prim[child & ~kLeafBit] = parent; else nd.parent_node[child] = parent;
}
/* a winner's move: x's sibling takes p's place, p takes y's place and holds (x, y) */
NORI_HD void reins_apply(const PlocNodes &nd, const ReinsData &rd, uint32_t n_inner, uint32_t slot) {
    const uint32_t x = reins_id(slot, n_inner), y = rd.target[slot];
    const uint32_t p0 = reins_parent(nd, x), g = nd.parent_node[p0];
    const uint32_t s0 = nd.left[p0] == x ? nd.right[p0] : nd.left[p0];
    if (nd.left[g] == p0) nd.left[g] = s0; else nd.right[g] = s0;
    reins_set_parent(nd, s0, g);
    const uint32_t q = reins_parent(nd, y);      /* (after the step above: y may be p's parent or its sibling) */
    if (nd.left[q] == y) nd.left[q] = p0; else nd.right[q] = p0;
    nd.parent_node[p0] = q;
    nd.left[p0] = x; nd.right[p0] = y;
    reins_set_parent(nd, y, p0);
}
/* node `id` from its children: box, triangles below, SAH cost of the subtree (as treelet_form / treelet_rewire keep them) */
NORI_HD void reins_refit_node(const PlocNodes &nd, const TreeletData &td, const ReinsData &rd, TreeletParams tp, uint32_t id) {
    const uint32_t c[2] = {coh_ld(&nd.left[id]), coh_ld(&nd.right[id])};
    f3 mn[2], mx[2]; float cost[2]; uint32_t cnt[2];
    for (int i = 0; i < 2; ++i) {
        if (c[i] & kLeafBit) { mn[i] = xyz(td.lmn[c[i] & ~kLeafBit]); mx[i] = xyz(td.lmx[c[i] & ~kLeafBit]); cnt[i] = 1u; cost[i] = tp.c_tri * half_area(mn[i], mx[i]); }
        else { mn[i] = xyz(coh_ld4(&td.nmn[c[i]])); mx[i] = xyz(coh_ld4(&td.nmx[c[i]])); cnt[i] = coh_ld(&nd.count[c[i]]); cost[i] = coh_ld(&td.cost[c[i]]); }
    }
    const f3 bmn = mk3(fminf(mn[0].x, mn[1].x), fminf(mn[0].y, mn[1].y), fminf(mn[0].z, mn[1].z));
    const f3 bmx = mk3(fmaxf(mx[0].x, mx[1].x), fmaxf(mx[0].y, mx[1].y), fmaxf(mx[0].z, mx[1].z));
    f4 a, b; a.x = bmn.x; a.y = bmn.y; a.z = bmn.z; a.w = 0.0f; b.x = bmx.x; b.y = bmx.y; b.z = bmx.z; b.w = 0.0f;
    const bool bounded = bmx.x < kBoxInf;
    coh_st4(&td.nmn[id], a); coh_st4(&td.nmx[id], b);
    coh_st(&td.cost[id], bounded ? tp.c_node * half_area(bmn, bmx) + (cost[0] + cost[1]) : 1e30f);
    coh_st(&nd.count[id], cnt[0] + cnt[1]);
}
/* CPU form of the climb from the triangle at position k (the device's: k_reins_refit, lbvh.hip) */
NORI_HD void reins_refit_climb(const PlocNodes &nd, const TreeletData &td, const ReinsData &rd, TreeletParams tp, uint32_t *visits, uint32_t root_id, uint32_t k) {
    uint32_t p = nd.parent_prim[k];
    for (uint32_t guard = 0; guard < 4096u; ++guard) {
        if (visits[p]++ == 0u) return;
        reins_refit_node(nd, td, rd, tp, p);
        if (p == root_id) return;
        p = nd.parent_node[p];
    }
}

/* min / max segment tree over the boxes in the builder's order: leaves at [N + k], node i = union of 2 i, 2 i + 1 */
NORI_HD void seg_tree_combine(uint32_t i, f4 *tmin, f4 *tmax) {
    const f4 a = tmin[2 * i], b = tmin[2 * i + 1], c = tmax[2 * i], d = tmax[2 * i + 1];
    f4 mn, mx;
    mn.x = fminf(a.x, b.x); mn.y = fminf(a.y, b.y); mn.z = fminf(a.z, b.z); mn.w = 0.0f;
    mx.x = fmaxf(c.x, d.x); mx.y = fmaxf(c.y, d.y); mx.z = fmaxf(c.z, d.z); mx.w = 0.0f;
    tmin[i] = mn; tmax[i] = mx;
}
NORI_HD void range_box(const f4 *tmin, const f4 *tmax, uint32_t N, uint32_t lo, uint32_t hi, f3 &mn, f3 &mx) {
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
NORI_HD float box_area(f3 mn, f3 mx) {
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

/* ---- leaves ---- */
/* A subtree of <= 4 triangles (contiguous in the builder's order) can become ONE leaf (its triangles stored as
   pairs, rt_types.h) or stay split.  collapse_decide chooses per node with the surface-area heuristic:
       leaf : A(node) * pairs * Cpair          split : A(node) * Cnode + best(left) + best(right)
   (a leaf step tests a pair of triangles, 1.5 - 2.5x the work of a node step).  collapse[i] = 1: node i is a
   leaf wherever it is reached.  Nodes above 4 triangles are always inner nodes. */
struct CollapseParams { float c_pair, c_node; };

NORI_HD float best_cost(const RadixNode *nodes, const f4 *tmin, const f4 *tmax, uint32_t N, uint32_t child, CollapseParams cp, bool *collapse_out) {
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
NORI_HD uint32_t collapse_decide(const RadixNode *nodes, uint32_t i, const f4 *tmin, const f4 *tmax, uint32_t N, CollapseParams cp) {
    const RadixNode nd = nodes[i];
    bool c = false;
    if (nd.hi - nd.lo + 1 <= 4u) (void) best_cost(nodes, tmin, tmax, N, i, cp, &c);      /* subtree of <= 3 inner nodes */
    return c ? 1u : 0u;
}
/* is `child` a leaf (primitive or collapsed subtree)?  [lo, hi] = its range */
NORI_HD bool child_range(const RadixNode *nodes, const uint32_t *collapse, uint32_t child, uint32_t &lo, uint32_t &hi) {
    if (child & kLeafBit) { lo = hi = child & ~kLeafBit; return true; }
    lo = nodes[child].lo; hi = nodes[child].hi;
    return collapse[child] != 0u;
}
/* the first pair of the leaf starting at position lo is pair_start[lo] (exclusive scan of the per-leaf pair counts) */
NORI_HD int32_t child_link(const RadixNode *nodes, const uint32_t *collapse, const uint32_t *pair_start, const uint32_t *node_index,
                           uint32_t child, uint32_t &lo, uint32_t &hi, uint32_t pair_base = 0u) {
    if (!child_range(nodes, collapse, child, lo, hi)) return (int32_t) node_index[child];
    const uint32_t cnt = hi - lo + 1;
    return (int32_t) ~(((pair_start[lo] + pair_base) << 3) | ((cnt + 1u) / 2u - 1u));
}
/* leaf_cnt[lo] = triangles of the leaf that starts at position lo, leaf_pairs[lo] = its pairs; returns keep = 1 for the
   nodes that survive as BVH nodes: not collapsed and not below a collapsed node */
NORI_HD uint32_t mark_leaves(const RadixNode *nodes, uint32_t i, const uint32_t *collapse, const uint32_t *parent_inner,
                             uint32_t *leaf_cnt, uint32_t *leaf_pairs) {
    const RadixNode nd = nodes[i];
    bool reachable = collapse[i] == 0u;
    for (uint32_t p = i; reachable && p != 0u && nodes[p].hi - nodes[p].lo + 1 <= 4u; ) {      /* ancestors that might have collapsed */
        p = parent_inner[p];
        if (p == kNoParent) break;
        if (collapse[p]) reachable = false;
    }
    if (!reachable) return 0u;
    uint32_t lo, hi;
    if (child_range(nodes, collapse, nd.left, lo, hi)) { leaf_cnt[lo] = hi - lo + 1; leaf_pairs[lo] = (hi - lo + 2) / 2; }
    if (child_range(nodes, collapse, nd.right, lo, hi)) { leaf_cnt[lo] = hi - lo + 1; leaf_pairs[lo] = (hi - lo + 2) / 2; }
    return 1u;
}

/* ---- emission ---- */
/* node i as a 64-B two-child-box record (rt_types.h) */
NORI_HD void emit_node(const RadixNode *nodes, uint32_t i, const f4 *tmin, const f4 *tmax, uint32_t N, const uint32_t *collapse,
                       const uint32_t *pair_start, const uint32_t *node_index, f4 q[4]) {
    const RadixNode nd = nodes[i];
    uint32_t llo, lhi, rlo, rhi;
    const int32_t cl = child_link(nodes, collapse, pair_start, node_index, nd.left, llo, lhi), cr = child_link(nodes, collapse, pair_start, node_index, nd.right, rlo, rhi);
    f3 lmn, lmx, rmn, rmx;
    range_box(tmin, tmax, N, llo, lhi, lmn, lmx);
    range_box(tmin, tmax, N, rlo, rhi, rmn, rmx);
    const float a0[3] = {lmn.x, lmn.y, lmn.z}, a1[3] = {lmx.x, lmx.y, lmx.z}, b0[3] = {rmn.x, rmn.y, rmn.z}, b1[3] = {rmx.x, rmx.y, rmx.z};
    node_pack(a0, a1, b0, b1, cl, cr, q);
}
/* the leaf that starts at position k (cnt triangles): its triangles, de-indexed, as pair records */
NORI_HD void emit_leaf_pairs(const f4 *pos, const uint32_t *idx, const uint32_t *tri_mesh, const uint32_t *order, uint32_t k, uint32_t cnt,
                             f4 *dst_pairs) {
    for (uint32_t t = 0; t < ((cnt + 1u) / 2u) * 2u; ++t) {
        f4 q[kPairQuads];
        f4 *dst = dst_pairs + (size_t) (t / 2u) * kPairQuads;
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
/* WIDE emission, one wide node: take node i's two children and twice replace the inner child of largest surface area by
   that child's children.  Returns the number of children in kid[]; the inner ones are the wide nodes of the next level. */
NORI_HD int wide_children(const RadixNode *nodes, const uint32_t *collapse, const f4 *tmin, const f4 *tmax, uint32_t N, uint32_t i, uint32_t kid[4]) {
    const RadixNode nd = nodes[i];
    kid[0] = nd.left; kid[1] = nd.right; kid[2] = kid[3] = 0u;
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
    return n;
}
NORI_HD void emit_wide_node(const RadixNode *nodes, const f4 *tmin, const f4 *tmax, uint32_t N, const uint32_t *collapse, const uint32_t *pair_start,
                            const uint32_t *wide_index, const uint32_t *kids, int n, f4 q[4]) {
    float mn[4][3], mx[4][3]; int32_t link[4];
    for (int k = 0; k < n; ++k) {
        uint32_t lo, hi;
        link[k] = child_link(nodes, collapse, pair_start, wide_index, kids[k], lo, hi, 1u);      /* pair 0 = the null pair */
        f3 a, b; range_box(tmin, tmax, N, lo, hi, a, b);
        mn[k][0] = a.x; mn[k][1] = a.y; mn[k][2] = a.z; mx[k][0] = b.x; mx[k][1] = b.y; mx[k][2] = b.z;
    }
    wide_pack(n, mn, mx, link, q);
}
/* nodes of the emitted tree on the way from the triangle at position k to the root (keep[]: mark_leaves -- the nodes inside a
   collapsed subtree are not nodes of the tree: counting them priced the Cornell box's tree two levels deeper than it is) */
NORI_HD uint32_t leaf_depth(const uint32_t *parent_inner, const uint32_t *parent_leaf, const uint32_t *keep, uint32_t k) {
    uint32_t depth = 0u, p = parent_leaf[k];
    for (uint32_t guard = 0; p != kNoParent && guard < 4096u; ++guard) {
        depth += keep[p] ? 1u : 0u;
        if (p == 0u) break;
        p = parent_inner[p];
    }
    return depth;
}

} // namespace nrt
