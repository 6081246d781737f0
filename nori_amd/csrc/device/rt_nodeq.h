/*
 * rt_nodeq.h -- BVH2 nodes in 32 bytes: the form wf_extend's node loop reads.
 *
 * Why: wf_extend is bound by the number of vector-memory INSTRUCTIONS it issues, not by bytes, lanes or arithmetic -- on
 * gfx950 a global_load_dwordx4 occupies the CU's address path for ~16 cycles whether 64 lanes or 4 take part (measured:
 * DESIGN.md section 3.3; adding one such load to the node step costs 6 ms per frame, adding 16 VALU instructions 1.7 ms).
 * The 64-B node of rt_types.h is four loads per node step; this one is two.  The walk still has to find exactly the
 * leaves the 64-B walk finds a hit in -- BoundingBox::rayIntersect (include/nori/bbox.h:323-350) is not part of the
 * reference's arithmetic, it only must never cull a box the ray passes through -- so the record may be lossy as long as
 * it is conservative:
 *
 *   dword 0 / 1   x planes: low bounds (left child in bits 0-15, right child in bits 16-31) / high bounds
 *   dword 2 / 3   y planes      dword 4 / 5   z planes
 *   dword 6 / 7   the links of the left / right child, as in the 64-B node
 *
 * A plane is a 16-bit coordinate on ONE grid for the whole tree: position = mn_a + q scale_a per axis a, the grid spanning
 * the boxes of the root's children (padded).  Low bounds are rounded down to a grid plane and high bounds up (a step is
 * 1 / 65535 of the scene: on the Cornell box 1.6 % more triangle tests and 0.3 % more node tests than the exact boxes).
 * Per ray and axis  t(q) = q A + B  with  A = scale r,  B = (mn - o) r  (r = 1 / d as slab_rcp gives it): one conversion and
 * one fused multiply-add per plane; the sign of r says which of the two dwords holds the near planes.  The rounding of that
 * evaluation is covered as in the wide node (rt_trace.h, trav_wide_step): |t_computed - t_exact| <= S_a = 4e-7 (65535 |A_a| +
 * |B_a|), folded into B per axis -- near planes use B - S, far planes B + S.
 *
 * A tree qualifies if its root is an inner node and no box is unbounded (numerically collinear triangles, rt_types.h);
 * otherwise there are no 32-B records and wf_extend walks the 64-B nodes.  Both forms describe the same tree: same
 * links, same leaves.  The CPU harness walks either (tests/emu), which is how the conservativeness is tested
 * without a GPU: hits identical to the linear scan for every ray.
 */
#pragma once
#include "rt_types.h"

namespace nrt {

constexpr int kNodeqQuads = 2;
constexpr float kNodeqSlackA = 65535.0f * 4e-7f, kNodeqSlackB = 4e-7f;

/* the box of child k of a 64-B node as [lo, hi] per axis (binary64: exact); false if it is unbounded */
NORI_HD bool node_child_box(const f4 q[4], int k, double lo[3], double hi[3]) {
    const float c[3] = {k == 0 ? q[0].x : q[0].z, k == 0 ? q[0].y : q[0].w, k == 0 ? q[1].x : q[1].y};
    const float h[3] = {k == 0 ? q[2].x : q[2].z, k == 0 ? q[2].y : q[2].w, k == 0 ? q[1].z : q[1].w};
    for (int a = 0; a < 3; ++a) {
        if (!(h[a] < 0.5f * kBoxHalfInf) || !(fabsf(c[a]) < 0.5f * kBoxHalfInf)) return false;      /* box_centre_half's unbounded box: c = 0, h = kBoxHalfInf */
        lo[a] = (double) c[a] - (double) h[a]; hi[a] = (double) c[a] + (double) h[a];
    }
    return true;
}

/* the grid of a tree from its root record; false if the tree does not qualify */
NORI_HD bool nodeq_grid(const f4 *nodes, int32_t root, NodeqGrid &g) {
    if (root < 0) return false;
    double l0[3], h0[3], l1[3], h1[3];
    if (!node_child_box(nodes + (size_t) root * kNodeQuads, 0, l0, h0) || !node_child_box(nodes + (size_t) root * kNodeQuads, 1, l1, h1)) return false;
    for (int a = 0; a < 3; ++a) {
        const double lo = l0[a] < l1[a] ? l0[a] : l1[a], hi = h0[a] > h1[a] ? h0[a] : h1[a];
        const double ext = hi - lo, big = fabs(lo) > fabs(hi) ? fabs(lo) : fabs(hi);
        const double pad = 1e-4 * ext + 1e-6 * big + 1e-30;
        const float mn = (float) (lo - pad);                     /* (rounding to nearest moves it by < 6e-8 big: well inside the pad) */
        float sc = (float) ((ext + 2.0 * pad) / 65535.0);
        sc = u2f(f2u(sc) + 2u);                                  /* a little up: mn + 65535 scale >= hi + pad / 2 */
        if (!(sc > 0.0f) || !(sc < 1e30f) || !(fabsf(mn) < 1e30f)) return false;
        g.mn[a] = mn; g.scale[a] = sc;
    }
    return true;
}

/* grid coordinate of a low / high bound: the last plane at or below x / the first at or above it, clamped to the grid (every
   true box lies inside the root's; what sticks out of the grid is the rounding of the stored boxes).  mn + q scale is exact in
   binary64 (16 x 24 bits, plus a 24-bit term of similar magnitude), so the comparison below decides it. */
NORI_HD uint32_t nodeq_lo(double x, float mn, float scale) {
    double q = floor((x - (double) mn) / (double) scale);
    if (!(q >= 0.0)) q = 0.0;
    if (q > 65535.0) q = 65535.0;
    while (q > 0.0 && (double) mn + q * (double) scale > x) q -= 1.0;
    return (uint32_t) q;
}
NORI_HD uint32_t nodeq_hi(double x, float mn, float scale) {
    double q = ceil((x - (double) mn) / (double) scale);
    if (!(q >= 0.0)) q = 0.0;
    if (q > 65535.0) q = 65535.0;
    while (q < 65535.0 && (double) mn + q * (double) scale < x) q += 1.0;
    return (uint32_t) q;
}

/* 64-B node -> 32-B record; false if a child box is unbounded */
NORI_HD bool nodeq_from_node(const f4 q[4], const NodeqGrid &g, f4 out[2]) {
    double l0[3], h0[3], l1[3], h1[3];
    if (!node_child_box(q, 0, l0, h0) || !node_child_box(q, 1, l1, h1)) return false;
    uint32_t w[6];
    for (int a = 0; a < 3; ++a) {
        w[2 * a] = nodeq_lo(l0[a], g.mn[a], g.scale[a]) | (nodeq_lo(l1[a], g.mn[a], g.scale[a]) << 16);
        w[2 * a + 1] = nodeq_hi(h0[a], g.mn[a], g.scale[a]) | (nodeq_hi(h1[a], g.mn[a], g.scale[a]) << 16);
    }
    out[0].x = u2f(w[0]); out[0].y = u2f(w[1]); out[0].z = u2f(w[2]); out[0].w = u2f(w[3]);
    out[1].x = u2f(w[4]); out[1].y = u2f(w[5]); out[1].z = q[3].x; out[1].w = q[3].y;
    return true;
}

/* what a ray needs for the plane evaluation: per axis A, B - S, B + S and whether it travels against the axis.
   The hand-written loop of wf_extend (wavefront.hip) performs exactly these operations, in this order. */
struct NodeqRay { float A[3], Bn[3], Bf[3]; bool neg[3]; };
NORI_HD void nodeq_ray(const NodeqGrid &g, f3 o, f3 rcp, NodeqRay &R) {
    const float oo[3] = {o.x, o.y, o.z}, rr[3] = {rcp.x, rcp.y, rcp.z};
    for (int a = 0; a < 3; ++a) {
        const float A = g.scale[a] * rr[a];
        const float B = (g.mn[a] - oo[a]) * rr[a];
        const float S = __builtin_fmaf(fabsf(A), kNodeqSlackA, fabsf(B) * kNodeqSlackB);
        R.A[a] = A; R.Bn[a] = B - S; R.Bf[a] = B + S; R.neg[a] = (f2u(rr[a]) >> 31) != 0u;
    }
}

/* the slab intervals of both children */
NORI_HD void nodeq_slabs(const f4 &q0, const f4 &q1, const NodeqRay &R, float &nl, float &fl, float &nr, float &fr) {
    const uint32_t w[6] = {f2u(q0.x), f2u(q0.y), f2u(q0.z), f2u(q0.w), f2u(q1.x), f2u(q1.y)};
    float tn0[3], tf0[3], tn1[3], tf1[3];
    for (int a = 0; a < 3; ++a) {
        const uint32_t nw = R.neg[a] ? w[2 * a + 1] : w[2 * a], fw = R.neg[a] ? w[2 * a] : w[2 * a + 1];
        tn0[a] = __builtin_fmaf((float) (nw & 0xffffu), R.A[a], R.Bn[a]); tn1[a] = __builtin_fmaf((float) (nw >> 16), R.A[a], R.Bn[a]);
        tf0[a] = __builtin_fmaf((float) (fw & 0xffffu), R.A[a], R.Bf[a]); tf1[a] = __builtin_fmaf((float) (fw >> 16), R.A[a], R.Bf[a]);
    }
    nl = fmaxf(fmaxf(tn0[0], tn0[1]), tn0[2]); fl = fminf(fminf(tf0[0], tf0[1]), tf0[2]);
    nr = fmaxf(fmaxf(tn1[0], tn1[1]), tn1[2]); fr = fminf(fminf(tf1[0], tf1[1]), tf1[2]);
}

} // namespace nrt
