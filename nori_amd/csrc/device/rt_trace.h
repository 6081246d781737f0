/*
 * rt_trace.h -- per-lane BVH traversal, Moeller-Trumbore, intersection record.
 *
 * Replaces Accel::rayIntersect (src/accel.cpp:23-99): the O(#triangles) scan of
 * accel.cpp:30-40 becomes a BVH2 walk whose leaf test is Mesh::rayIntersect
 * (src/mesh.cpp:39-76) in the same operation order, and whose tie rule
 * reproduces the scan (a later triangle with equal t wins, because mesh.cpp:75
 * accepts t <= maxt after accel.cpp:37 shrank maxt to t).  Node test: the
 * slab test of include/nori/bbox.h:323-350, made conservative (far side
 * widened by 2 ulp-ish) so no leaf whose triangle test would accept is culled.
 *
 * `Stack` is a policy with reset()/push(int)/pop_or(int empty_value)->int: in the
 * HIP kernels it is an LDS column ([depth][lane], bank = lane -> conflict free);
 * the CPU emulation harness passes a plain array.
 */
#pragma once
#include "rt_types.h"
#include "rt_top.h"
#include "rt_nodeq.h"

#if defined(NORI_TRAV_HISTOGRAM) && !defined(__HIP_DEVICE_COMPILE__)
/* CPU harness only (tools/trav_histogram.py): visits per node and per leaf pair record */
extern "C" void nori_trav_hist(int kind, uint32_t index);
#define NORI_HIST(kind, index) nori_trav_hist(kind, index)
#else
#define NORI_HIST(kind, index) ((void) 0)
#endif

namespace nrt {

struct RayIn {
    f3 o, d;
    float mint, maxt;
};

/* exact_rcp_raw (rt_types.h): no fallback branch for the operands outside its domain (it costs the traversal kernel its last
 * free registers): a determinant beyond 8.5e37 needs scene coordinates beyond 1e12, and a denormal one is rejected by
 * |det| < 1e-8 before its reciprocal is used (src/mesh.cpp:52). */

/* Reciprocal direction for the slab tests.  A zero (or denormal) component maps to +-2^60: huge, so that (plane - o) * rcp
 * is astronomically far off the plane and exactly 0 on it -- the containment rule of bbox.h:331-333 (boundary inclusive)
 * -- yet a power of two small enough that every product the node tests form with it (centre and half-extent form of
 * the BVH2 node, quantised-plane form of the wide node) stays finite: no inf - inf, no 0 * inf. */
NORI_HD float slab_rcp(float d) {
    float r = exact_rcp_raw(d);
    if (!(fabsf(r) <= 1.152921504606846976e18f)) r = (f2u(d) >> 31) ? -1.152921504606846976e18f : 1.152921504606846976e18f;
    return r;
}
NORI_HD float slab_rcp_wide(float d) { return slab_rcp(d); }

/* byte k of a dword as float (v_cvt_f32_ubyte0..3 on the device) */
NORI_HD float byte_to_float(uint32_t w, int k) { return (float) ((w >> (8 * k)) & 255u); }

/* Moeller-Trumbore on a pre-gathered leaf record; src/mesh.cpp:39-76 */
NORI_HD bool tri_test(f3 p0, f3 edge1, f3 edge2, f3 o, f3 d, float &u, float &v, float &t) {
    f3 pvec = cross(d, edge2);
    float det = dot(edge1, pvec);
    if (det > -1e-8f && det < 1e-8f) return false;
    float inv_det = exact_rcp_raw(det);
    f3 tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross(tvec, edge1);
    v = dot(d, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(edge2, qvec) * inv_det;
    return true;
}

/* Slab test (include/nori/bbox.h:323-350) of BOTH child boxes of a node, boxes in centre / half-extent form
 * (rt_types.h): per axis  m = (c - o) r,  near = m - h |r|,  far = m + h |r|.  The centres pair up as 2-wide vectors
 * (x, y of a child against (o.x, o.y); the two z's against o.z): 3 v_pk_add_f32 + 3 v_pk_mul_f32; near / far are one
 * v_fma_f32 each with |r| as a source modifier -- 12 per node, full rate -- and one v_max3 / v_min3 per child:
 * 22 VALU instructions where the min / max form of the two-plane test needs 34 (16 of them half rate).
 * The fused multiply-add is explicit (IEEE: same bits on the device and in the CPU twins); the node test is not part
 * of the reference's arithmetic -- it only has to be conservative, see trav_inner_step. */
typedef float v2f __attribute__((vector_size(8)));

NORI_HD void slab_two(const f4 &q0, const f4 &q1, const f4 &q2, f3 o, f3 rcp, float &nl, float &fl, float &nr, float &fr) {
    const v2f oxy = {o.x, o.y}, rxy = {rcp.x, rcp.y}, ozz = {o.z, o.z}, rzz = {rcp.z, rcp.z};
    const v2f lc = {q0.x, q0.y}, rc = {q0.z, q0.w}, cz = {q1.x, q1.y};
    const v2f ml = (lc - oxy) * rxy, mr = (rc - oxy) * rxy, mz = (cz - ozz) * rzz;
    const float ax = fabsf(rcp.x), ay = fabsf(rcp.y), az = fabsf(rcp.z);
    nl = fmaxf(fmaxf(__builtin_fmaf(-ax, q2.x, ml[0]), __builtin_fmaf(-ay, q2.y, ml[1])), __builtin_fmaf(-az, q1.z, mz[0]));
    fl = fminf(fminf(__builtin_fmaf(ax, q2.x, ml[0]), __builtin_fmaf(ay, q2.y, ml[1])), __builtin_fmaf(az, q1.z, mz[0]));
    nr = fmaxf(fmaxf(__builtin_fmaf(-ax, q2.z, mr[0]), __builtin_fmaf(-ay, q2.w, mr[1])), __builtin_fmaf(-az, q1.w, mz[1]));
    fr = fminf(fminf(__builtin_fmaf(ax, q2.z, mr[0]), __builtin_fmaf(ay, q2.w, mr[1])), __builtin_fmaf(az, q1.w, mz[1]));
}

/* Traversal state of one ray, advanced ONE step at a time so that a kernel can
 * interleave traversal with other work (regenerating finished lanes) instead of
 * letting 63 lanes wait for the longest walk of the wave.  The whole control
 * state is ONE register:
 *   node >= 0          next step tests the inner node `node` (both child boxes)
 *   node <  0, != done next step tests ONE leaf triangle: ~node = (tri << 3) | (left - 1),
 *                      i.e. a leaf link is its own cursor; advancing is ~node += 7
 *   node == kTravDone  finished; `hit` is the answer
 * Closest-hit (any = false) or any-hit / shadow (any = true); `any` is run-time
 * state so both kinds of query share one instruction stream.  hit.t doubles as
 * the current far limit of the ray. */
constexpr int kTravDone = (int) 0x80000000u;      /* ~kTravDone is no valid leaf link: first_tri < 2^28 */

struct Trav {
    f3 o, d, rcp;
    float mint;
    int node;
    Hit hit;
    bool any;
};

NORI_HD bool trav_active(const Trav &tv) { return tv.node != kTravDone; }
NORI_HD bool trav_at_inner(const Trav &tv) { return tv.node >= 0; }
NORI_HD bool trav_at_leaf(const Trav &tv) { return (uint32_t) tv.node > 0x80000000u; }
NORI_HD void trav_idle(Trav &tv) { tv.node = kTravDone; tv.any = false; }

/* node layout a kernel is compiled for: BVH2 nodes, WIDE nodes (rt_types.h), or whichever the scene has (run-time
   branch; the batch twins and wf_finish, where registers are not the limit) */
enum { kLayoutBvh2 = 0, kLayoutWide = 1, kLayoutAny = 2 };

template <int LAYOUT = kLayoutBvh2, class Stack>
NORI_HD void trav_begin(const DevScene &sc, const RayIn &ray, bool any, Stack &stack, Trav &tv) {
    tv.o = ray.o; tv.d = ray.d;
    const bool wide = LAYOUT == kLayoutWide || (LAYOUT == kLayoutAny && sc.wide != 0u);
    tv.rcp = wide ? mk3(slab_rcp_wide(ray.d.x), slab_rcp_wide(ray.d.y), slab_rcp_wide(ray.d.z))
                  : mk3(slab_rcp(ray.d.x), slab_rcp(ray.d.y), slab_rcp(ray.d.z));
    tv.mint = ray.mint;
    tv.hit.tri = kNoHit; tv.hit.mesh = kNoHit; tv.hit.t = ray.maxt; tv.hit.u = 0.0f; tv.hit.v = 0.0f;
    tv.any = any;
    stack.reset();
    tv.node = sc.n_triangles != 0 ? sc.root : kTravDone;      /* the root may itself be a leaf link */
}

/* pop the next subtree or finish */
template <class Stack>
NORI_HD void trav_pop(Stack &stack, Trav &tv) {
    tv.node = stack.pop_or(kTravDone);
}

/* The hot part of the tree in LDS: rt_top.h (which records, the image layout, the link codes). */
/* the four quads of node `node`: from the LDS cache (`top` non-null and the link carries kTopBit) or from memory.
   On the device the cache is reached through an LDS-address-space pointer: two exec-masked paths, ds_read_b128 for the
   cached nodes and global_load_dwordx4 for the others.  (With a generic pointer the compiler merges the two paths into one
   flat_load through a selected address -- and a flat load that resolves to LDS still occupies the vector-memory address
   path the cache was meant to relieve.) */
#if defined(__HIP_DEVICE_COMPILE__)
typedef float v4f_lds __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f_lds *TopNodesP;
NORI_HD f4 top_quad(TopNodesP p) { const v4f_lds v = *p; f4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
NORI_HD TopNodesP top_nodes_pointer(const f4 *generic) { return (TopNodesP) generic; }
#else
typedef const f4 *TopNodesP;
NORI_HD f4 top_quad(TopNodesP p) { return *p; }
NORI_HD TopNodesP top_nodes_pointer(const f4 *generic) { return generic; }
#endif

NORI_HD void node_fetch(const DevScene &sc, TopNodesP top, int node, f4 &q0, f4 &q1, f4 &q2, f4 &q3) {
    if (top != nullptr && (node & kTopBit)) {
        const TopNodesP nq = top + (node & kTopQuadMask);      /* the link names the record's quad offset in the image */
        q0 = top_quad(nq); q1 = top_quad(nq + 1); q2 = top_quad(nq + 2); q3 = top_quad(nq + 3);
    } else {
        const f4 *nq = sc.nodes + (size_t) node * kNodeQuads;
        q0 = nq[0]; q1 = nq[1]; q2 = nq[2]; q3 = nq[3];
    }
}

/* one inner-node step: both child boxes from one 64-B record */
template <bool COUNT, class Stack>
NORI_HD void trav_inner_step(const DevScene &sc, Stack &stack, Trav &tv, TraversalCounters &cnt, TopNodesP top = nullptr) {
    f4 q0, q1, q2, q3;
    node_fetch(sc, top, tv.node, q0, q1, q2, q3);
    if (COUNT) cnt.nodes++;
    NORI_HIST(0, (uint32_t) tv.node);
    float nl, fl, nr, fr;
    slab_two(q0, q1, q2, tv.o, tv.rcp, nl, fl, nr, fr);
    /* conservative: m carries 2.5 ulp/2 of relative error, the fma one rounding -- a far side widened by 10 u covers a ray
       that meets the box far from its origin; where the origin is close to a large box the error is absolute,
       <= 4 u (|c - o| + h) |r|, a tenth of the padding every leaf box carries (kBoxPadRel) for origins within the scene */
    fl *= 1.0000006f; fr *= 1.0000006f;
    /* the box interval is clipped against [0, far limit], not [mint, ...]: a ray that leaves a surface at a
       grazing angle can re-hit its own triangle at a t that is pure rounding noise yet >= mint (the
       reference's scan reports it), while the ray is outside the triangle's box by then */
    const bool hl = (nl <= fl) && (fl >= 0.0f) && (nl <= tv.hit.t);
    const bool hr = (nr <= fr) && (fr >= 0.0f) && (nr <= tv.hit.t);
    const int cl = (int) f2u(q3.x), cr = (int) f2u(q3.y);
    if (hl && hr) {
        const bool leftFirst = nl <= nr;
        stack.push(leftFirst ? cr : cl);
        tv.node = leftFirst ? cl : cr;
    } else if (hl) {
        tv.node = cl;
    } else if (hr) {
        tv.node = cr;
    } else {
        trav_pop(stack, tv);
    }
}

/* one inner-node step on the 32-B record of the node (rt_nodeq.h): the C++ statement of what wf_extend's hand-written loop
   does (wavefront.hip, bvh2_node_loop_asm) -- same operations, same order -- and what the CPU harness walks.  `top`: the LDS
   image with 32-B node records (sc.top_image_q). */
template <bool COUNT, class Stack>
NORI_HD void trav_inner_step_q(const DevScene &sc, Stack &stack, Trav &tv, TraversalCounters &cnt, TopNodesP top = nullptr) {
    f4 q0, q1;
    if (top != nullptr && (tv.node & kTopBit)) {
        const TopNodesP nq = top + (tv.node & kTopQuadMask);
        q0 = top_quad(nq); q1 = top_quad(nq + 1);
    } else {
        const f4 *nq = sc.nodes_q + (size_t) tv.node * kNodeqQuads;
        q0 = nq[0]; q1 = nq[1];
    }
    if (COUNT) cnt.nodes++;
    NORI_HIST(0, (uint32_t) tv.node);
    NodeqRay R;
    nodeq_ray(sc.grid, tv.o, tv.rcp, R);
    float nl, fl, nr, fr;
    nodeq_slabs(q0, q1, R, nl, fl, nr, fr);
    const bool hl = (nl <= fl) && (fl >= 0.0f) && (nl <= tv.hit.t);
    const bool hr = (nr <= fr) && (fr >= 0.0f) && (nr <= tv.hit.t);
    const int cl = (int) f2u(q1.z), cr = (int) f2u(q1.w);
    if (hl && hr) {
        const bool leftFirst = nl <= nr;
        stack.push(leftFirst ? cr : cl);
        tv.node = leftFirst ? cl : cr;
    } else if (hl) {
        tv.node = cl;
    } else if (hr) {
        tv.node = cr;
    } else {
        trav_pop(stack, tv);
    }
}

/* One WIDE-node step (rt_types.h): the four quantised child boxes of one 64-B record.
 *   t = q * A + B per plane, A = r 2^e (exact), B = (origin - o) r: one v_cvt_f32_ubyte + one v_fma per plane;
 *   the ray's direction signs pick which plane dword is the near one per axis -- no per-child min / max.
 * Conservative by construction, not by tuned factors.  With u = 2^-24 the computed t of a plane on axis i differs
 * from the exact one by at most 2u |t| + 2u |B_i| <= S_i = 4e-7 (255 |A_i| + |B_i|) (B carries two roundings plus the one of B -+ S, the
 * fma one, r half an ulp).  The slack is applied PER AXIS, folded into B: near planes use B_i - S_i, far planes
 * B_i + S_i, so every computed near / far value lies on the safe side of the exact one and the plain interval test
 * never skips a child whose exact interval overlaps [0, limit].  (A single slack for all axes would let the axis
 * with the smallest direction component -- huge |B_i| -- switch culling off for the whole ray.)
 * Closest child first; the others go to the stack in slot order along the node's axis, reversed for rays that
 * travel against it, so that the nearer ones pop first. */
/* the four planes t = q A + B of one plane dword (child k in byte k), as explicit scalars: no arrays, nothing for
   the compiler to index dynamically */
struct Wide4 { float a, b, c, d; };
NORI_HD Wide4 wide_planes(uint32_t w, float A, float B) {
    /* explicit fused multiply-add (IEEE, one rounding -- the same bits on the device and in the CPU twins); the node
       test is not part of the reference's arithmetic, only its conservativeness matters */
    Wide4 r;
    r.a = __builtin_fmaf(byte_to_float(w, 0), A, B); r.b = __builtin_fmaf(byte_to_float(w, 1), A, B);
    r.c = __builtin_fmaf(byte_to_float(w, 2), A, B); r.d = __builtin_fmaf(byte_to_float(w, 3), A, B);
    return r;
}

#ifndef NORI_LAB_WIDE_STEP
#define NORI_LAB_WIDE_STEP      /* (wf_experiments.h, variant builds: perturbations of the step) */
#endif
template <bool COUNT, class Stack>
NORI_HD void trav_wide_step(const DevScene &sc, Stack &stack, Trav &tv, TraversalCounters &cnt, TopNodesP top = nullptr) {
    f4 q0, q1, q2, q3;
    node_fetch(sc, top, tv.node, q0, q1, q2, q3);
    NORI_LAB_WIDE_STEP
    if (COUNT) cnt.nodes++;
    const uint32_t meta = f2u(q0.w);
    const int l0 = (int) f2u(q3.x), l1 = (int) f2u(q3.y), l2 = (int) f2u(q3.z), l3 = (int) f2u(q3.w);
    bool h0, h1, h2, h3;
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
    const bool nx = tv.rcp.x < 0.0f, ny = tv.rcp.y < 0.0f, nz = tv.rcp.z < 0.0f;
    if (meta & kWideAllHit) {
        h0 = l0 != kWideEmpty; h1 = l1 != kWideEmpty; h2 = l2 != kWideEmpty; h3 = l3 != kWideEmpty;
    } else {
        const float Ax = ldexpf(tv.rcp.x, (int) (meta & 255u) - 128), Ay = ldexpf(tv.rcp.y, (int) ((meta >> 8) & 255u) - 128),
                    Az = ldexpf(tv.rcp.z, (int) ((meta >> 16) & 255u) - 128);
        const float Bx = (q0.x - tv.o.x) * tv.rcp.x, By = (q0.y - tv.o.y) * tv.rcp.y, Bz = (q0.z - tv.o.z) * tv.rcp.z;
        const float Sx = __builtin_fmaf(fabsf(Ax), 255.0f * 4e-7f, fabsf(Bx) * 4e-7f), Sy = __builtin_fmaf(fabsf(Ay), 255.0f * 4e-7f, fabsf(By) * 4e-7f),
                    Sz = __builtin_fmaf(fabsf(Az), 255.0f * 4e-7f, fabsf(Bz) * 4e-7f);
        const float limit = tv.hit.t;
        const uint32_t loX = f2u(q1.x), loY = f2u(q1.y), loZ = f2u(q1.z), hiX = f2u(q1.w), hiY = f2u(q2.x), hiZ = f2u(q2.y);
        const Wide4 nX = wide_planes(nx ? hiX : loX, Ax, Bx - Sx), fX = wide_planes(nx ? loX : hiX, Ax, Bx + Sx);
        const Wide4 nY = wide_planes(ny ? hiY : loY, Ay, By - Sy), fY = wide_planes(ny ? loY : hiY, Ay, By + Sy);
        const Wide4 nZ = wide_planes(nz ? hiZ : loZ, Az, Bz - Sz), fZ = wide_planes(nz ? loZ : hiZ, Az, Bz + Sz);
        n0 = fmaxf(nX.a, fmaxf(nY.a, nZ.a)); n1 = fmaxf(nX.b, fmaxf(nY.b, nZ.b));
        n2 = fmaxf(nX.c, fmaxf(nY.c, nZ.c)); n3 = fmaxf(nX.d, fmaxf(nY.d, nZ.d));
        const float f0 = fminf(fX.a, fminf(fY.a, fZ.a)), f1 = fminf(fX.b, fminf(fY.b, fZ.b));
        const float f2 = fminf(fX.c, fminf(fY.c, fZ.c)), f3 = fminf(fX.d, fminf(fY.d, fZ.d));
        h0 = (n0 <= f0) && (f0 >= 0.0f) && (n0 <= limit); h1 = (n1 <= f1) && (f1 >= 0.0f) && (n1 <= limit);
        h2 = (n2 <= f2) && (f2 >= 0.0f) && (n2 <= limit); h3 = (n3 <= f3) && (f3 >= 0.0f) && (n3 <= limit);
    }
    /* slot order along the node's axis, reversed for a ray that travels against it; walk the slots from the far end:
       whatever is pushed later pops earlier; the nearest child found so far is kept for the next step */
    const uint32_t axis = (meta >> 24) & 3u;
    const bool rev = (axis == 0u && nx) || (axis == 1u && ny) || (axis == 2u && nz);      /* booleans only: selecting a component
                                                                                            of tv.rcp by index would send tv to scratch */
    /* Written as selects (a lambda per slot with branches compiled to ~25 skip branches and four regions with their own copies of
       first / firstKey: 203 vector + 136 scalar instructions per step against 180 + 106 now; terrain wf_extend -0.5 %): slot j of
       the visiting order is slot j for a reversed ray, slot 3 - j otherwise; the first hit child seen becomes `first`; a later one
       either takes its place (key <= firstKey: the old first is pushed) or is pushed itself. */
    const bool ha = rev ? h0 : h3, hb = rev ? h1 : h2, hc = rev ? h2 : h1, hd = rev ? h3 : h0;
    const float ka = rev ? n0 : n3, kb = rev ? n1 : n2, kc = rev ? n2 : n1, kd = rev ? n3 : n0;
    const int la = rev ? l0 : l3, lb = rev ? l1 : l2, lc = rev ? l2 : l1, ld = rev ? l3 : l0;
    int first = la; float firstKey = ka; bool have = ha;
#define NORI_WIDE_VISIT(h, key, lk) { \
        const bool closer = (key) <= firstKey; \
        if ((h) && have) stack.push(closer ? first : (lk)); \
        const bool take = (h) && (!have || closer); \
        first = take ? (lk) : first; firstKey = take ? (key) : firstKey; have = have || (h); }
    NORI_WIDE_VISIT(hb, kb, lb)
    NORI_WIDE_VISIT(hc, kc, lc)
    NORI_WIDE_VISIT(hd, kd, ld)
#undef NORI_WIDE_VISIT
    if (have) tv.node = first;
    else trav_pop(stack, tv);
}

/* Moeller-Trumbore (src/mesh.cpp:39-76) on TWO triangles at once: every quantity is a 2-wide vector
 * (triangle a, triangle b), so the ~60 multiplies / adds of the test run as packed-f32 instructions --
 * two triangles for the issue slots of one.  Each lane of the vector performs the scalar test's
 * operations in the scalar test's order (element-wise IEEE): same bits as tri_test. */
struct TriPairHit { v2f u, v, t; bool ok[2]; };

NORI_HD v2f splat2(float x) { const v2f r = {x, x}; return r; }

NORI_HD void tri_pair_test(const f4 &q0, const f4 &q1, const f4 &q2, const f4 &q3, const f4 &q4, f3 o, f3 d,
                           float mint, float maxt, TriPairHit &r) {
    const v2f p0x = {q0.x, q0.y}, p0y = {q0.z, q0.w}, p0z = {q1.x, q1.y};
    const v2f e1x = {q1.z, q1.w}, e1y = {q2.x, q2.y}, e1z = {q2.z, q2.w};
    const v2f e2x = {q3.x, q3.y}, e2y = {q3.z, q3.w}, e2z = {q4.x, q4.y};
    const v2f dx = splat2(d.x), dy = splat2(d.y), dz = splat2(d.z);
    /* pvec = cross(d, edge2); det = dot(edge1, pvec) */
    const v2f pvx = dy * e2z - dz * e2y, pvy = dz * e2x - dx * e2z, pvz = dx * e2y - dy * e2x;
    const v2f det = e1x * pvx + (e1y * pvy + e1z * pvz);
    v2f inv;
    inv[0] = exact_rcp_raw(det[0]); inv[1] = exact_rcp_raw(det[1]);      /* = 1.0f / det, bit for bit */
    /* tvec = o - p0; u = dot(tvec, pvec) * inv_det */
    const v2f tx = splat2(o.x) - p0x, ty = splat2(o.y) - p0y, tz = splat2(o.z) - p0z;
    r.u = (tx * pvx + (ty * pvy + tz * pvz)) * inv;
    /* qvec = cross(tvec, edge1); v = dot(d, qvec) * inv_det; t = dot(edge2, qvec) * inv_det */
    const v2f qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    r.v = (dx * qx + (dy * qy + dz * qz)) * inv;
    r.t = (e2x * qx + (e2y * qy + e2z * qz)) * inv;
    for (int k = 0; k < 2; ++k) {
        const float dk = det[k], uk = r.u[k], vk = r.v[k], tk = r.t[k];
        r.ok[k] = !(dk > -1e-8f && dk < 1e-8f) && !(uk < 0.0f || uk > 1.0f) && !(vk < 0.0f || uk + vk > 1.0f) &&
                  tk >= mint && tk <= maxt;
    }
}

/* one leaf step: ONE PAIR of triangles, then advance within the leaf.  `top`: the LDS image (rt_top.h) -- a cursor
   with kTopBit names a pair record cached there */
template <bool COUNT, bool MESH = true, class Stack>
NORI_HD void trav_leaf_step(const DevScene &sc, Stack &stack, Trav &tv, TraversalCounters &cnt, TopNodesP top = nullptr) {
    const uint32_t cursor = ~(uint32_t) tv.node;
    f4 q0, q1, q2, q3, q4;
    const bool cached = top != nullptr && (cursor & (uint32_t) kTopBit) != 0u;
    const TopNodesP lq = top + ((cursor >> 3) & (uint32_t) kTopPairMask) * kPairQuads;
    const f4 *tq = sc.tris + (size_t) (cursor >> 3) * kPairQuads;
    if (cached) { q0 = top_quad(lq); q1 = top_quad(lq + 1); q2 = top_quad(lq + 2); q3 = top_quad(lq + 3); q4 = top_quad(lq + 4); }
    else { q0 = tq[0]; q1 = tq[1]; q2 = tq[2]; q3 = tq[3]; q4 = tq[4]; }
    if (COUNT) cnt.tris += 2;
    NORI_HIST(1, cursor >> 3);
    TriPairHit r;
    tri_pair_test(q0, q1, q2, q3, q4, tv.o, tv.d, tv.mint, tv.hit.t, r);
    /* MESH = false (wf_extend, which does not keep hit.mesh: wf_shade fetches the mesh with the shading record, hit_unpack): the
       update of the hit written as selects -- one v_cndmask per field and candidate.  As branches the compiler builds a region per
       outcome, each with its own copies of the four fields: 32 register moves and ten skip branches per leaf step.  Same decisions. */
    if (!MESH) {
        const uint32_t gid0 = f2u(q4.z), gid1 = f2u(q4.w);
        const bool any = tv.any;
        const bool take0 = any ? r.ok[0] : (r.ok[0] && r.t[0] <= tv.hit.t && !(r.t[0] == tv.hit.t && tv.hit.tri != kNoHit && gid0 < tv.hit.tri));
        const float t0 = take0 ? r.t[0] : tv.hit.t, u0 = take0 ? r.u[0] : tv.hit.u, v0 = take0 ? r.v[0] : tv.hit.v;
        const uint32_t tri0 = take0 ? gid0 : tv.hit.tri;
        const bool take1 = any ? (r.ok[1] && !r.ok[0]) : (r.ok[1] && r.t[1] <= t0 && !(r.t[1] == t0 && tri0 != kNoHit && gid1 < tri0));
        tv.hit.t = take1 ? r.t[1] : t0; tv.hit.u = take1 ? r.u[1] : u0; tv.hit.v = take1 ? r.v[1] : v0; tv.hit.tri = take1 ? gid1 : tri0;
        if (any && (r.ok[0] || r.ok[1])) { tv.node = kTravDone; return; }
    } else if (r.ok[0] || r.ok[1]) {
        const f4 q5 = cached ? top_quad(lq + 5) : tq[5];
        if (tv.any) {
            const int k = r.ok[0] ? 0 : 1;
            tv.hit.tri = f2u(k == 0 ? q4.z : q4.w); tv.hit.t = r.t[k]; tv.node = kTravDone;
            return;
        }
        /* candidates in order a, b; tie rule of the linear scan: a later triangle (larger global id) with
           equal t replaces an earlier one */
        for (int k = 0; k < 2; ++k) {
            const float t = r.t[k];
            const uint32_t gid = f2u(k == 0 ? q4.z : q4.w);
            if (r.ok[k] && t <= tv.hit.t && !(t == tv.hit.t && tv.hit.tri != kNoHit && gid < tv.hit.tri)) {
                tv.hit.t = t; tv.hit.u = r.u[k]; tv.hit.v = r.v[k]; tv.hit.tri = gid; tv.hit.mesh = f2u(k == 0 ? q5.x : q5.y);
            }
        }
    }
    if ((cursor & 7u) == 0u) trav_pop(stack, tv);
    else tv.node = (int) ~(cursor + 7u);          /* next pair, one fewer left */
}

/* Run a traversal to completion (batch kernels, tests).
 * Returns true if something was hit; for closest-hit `hit` holds t,u,v,tri,mesh.
 * In the CPU harness the walk goes through the image of the hot records (rt_top.h) when the scene has one, and through the
 * 32-B node records (rt_nodeq.h) when the tree has them -- the links and records wf_extend reads, checked there against the
 * brute-force scan; on the device this function serves
 * the batch twins and wf_finish, which keep nothing in LDS. */
template <bool COUNT, class Stack>
NORI_HD bool traverse(const DevScene &sc, const RayIn &ray, bool any, Stack &stack, Hit &hit, TraversalCounters &cnt) {
    Trav tv;
    trav_begin<kLayoutAny>(sc, ray, any, stack, tv);
#if defined(__HIP_DEVICE_COMPILE__)
    const TopNodesP top = nullptr;
    const bool use_q = false;
#else
    /* the harness walks the 32-B records when the tree has them (rt_nodeq.h): wf_extend's default node loop */
    const bool use_q = sc.nodes_q != nullptr && !sc.wide;
    static const bool leaf_select = [] { const char *e = std::getenv("NORI_EMU_LEAF_SELECT"); return e == nullptr || std::atoi(e) != 0; }();
    const TopNodesP top = use_q ? sc.top_image_q : sc.top_image;
    if (top != nullptr && trav_active(tv)) tv.node = (int) f2u(top[0].x);
#endif
    while (trav_active(tv)) {
        if (trav_at_inner(tv)) {
            if (sc.wide) trav_wide_step<COUNT>(sc, stack, tv, cnt, top);
            else if (use_q) trav_inner_step_q<COUNT>(sc, stack, tv, cnt, top);
            else trav_inner_step<COUNT>(sc, stack, tv, cnt, top);
        } else {
#if !defined(__HIP_DEVICE_COMPILE__)
            if (leaf_select) trav_leaf_step<COUNT, false>(sc, stack, tv, cnt, top);      /* wf_extend's form of the leaf step */
            else
#endif
            trav_leaf_step<COUNT>(sc, stack, tv, cnt, top);
        }
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    if (leaf_select && tv.hit.tri != kNoHit) tv.hit.mesh = sc.tri_mesh[tv.hit.tri];      /* (that form does not keep the mesh) */
#endif
    hit = tv.hit;
    return hit.tri != kNoHit;
}

/* The part of the intersection record the integrators consume. */
struct Surface {
    f3 p;        /* its.p, accel.cpp:71 */
    f3 ns;       /* its.shFrame.n, accel.cpp:82-95 */
};

/* src/accel.cpp:45-96 -- barycentric position and shading normal, from the triangle's
 * pre-gathered shading record (rt_types.h) */
NORI_HD void surface_fill(const DevScene &sc, const Hit &h, Surface &s, f3 *geo_n, f2 *uv_out) {
    const MeshRec &m = sc.meshes[h.mesh];
    const f4 *rec = sc.shade_tris + (size_t) h.tri * kShadeQuads;
    const f3 p0 = xyz(rec[0]), p1 = xyz(rec[1]), p2 = xyz(rec[2]);
    const float b0 = 1.0f - (h.u + h.v), b1 = h.u, b2 = h.v;
    s.p = (b0 * p0 + b1 * p1) + b2 * p2;
    const bool needGeo = geo_n != nullptr || !(m.flags & kMeshHasNormals);
    f3 ng = mk3(0.0f);
    if (needGeo) ng = normalized(cross(p1 - p0, p2 - p0));
    if (geo_n) *geo_n = ng;
    if (m.flags & kMeshHasNormals) {
        const f3 n0 = xyz(rec[3]), n1 = xyz(rec[4]), n2 = xyz(rec[5]);
        s.ns = normalized((b0 * n0 + b1 * n1) + b2 * n2);
    } else {
        s.ns = ng;
    }
    if (uv_out) {
        if (m.flags & kMeshHasUV) {
            const uint32_t *idx = sc.indices + 3 * (size_t) h.tri;
            const f2 a = sc.texcoords[idx[0]], b = sc.texcoords[idx[1]], c = sc.texcoords[idx[2]];
            *uv_out = mk2((b0 * a.x + b1 * b.x) + b2 * c.x, (b0 * a.y + b1 * b.y) + b2 * c.y);
        } else {
            *uv_out = mk2(h.u, h.v);
        }
    }
}

/* PerspectiveCamera::sampleRay, src/perspective.cpp:76-97 */
NORI_HD f3 mat_point(const float *m, f3 p) {
    float r[4];
    for (int i = 0; i < 4; ++i)
        r[i] = ((m[4 * i] * p.x + m[4 * i + 1] * p.y) + m[4 * i + 2] * p.z) + m[4 * i + 3] * 1.0f;
    return mk3(r[0], r[1], r[2]) / r[3];
}
NORI_HD f3 mat_vector(const float *m, f3 v) {
    return mk3(m[0] * v.x + (m[1] * v.y + m[2] * v.z),
               m[4] * v.x + (m[5] * v.y + m[6] * v.z),
               m[8] * v.x + (m[9] * v.y + m[10] * v.z));
}
NORI_HD void camera_sample_ray(const CameraRec &cam, f2 samplePosition, RayIn &ray) {
    f3 nearP = mat_point(cam.sample_to_camera, mk3(samplePosition.x * cam.inv_w, samplePosition.y * cam.inv_h, 0.0f));
    f3 d = normalized(nearP);
    float invZ = exact_rcp(d.z);
    ray.o = mat_point(cam.camera_to_world, mk3(0.0f, 0.0f, 0.0f));
    ray.d = mat_vector(cam.camera_to_world, d);
    ray.mint = cam.near_clip * invZ;
    ray.maxt = cam.far_clip * invZ;
}

} // namespace nrt
