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

namespace nrt {

struct RayIn {
    f3 o, d;
    float mint, maxt;
};

/* Reciprocal direction for the slab test.  A zero component maps to a huge
 * finite value so that (plane - o) * rcp is +-inf off the plane and exactly 0
 * on it -- the containment rule of bbox.h:331-333 (boundary inclusive) without
 * producing 0 * inf = NaN. */
NORI_HD float slab_rcp(float d) {
    float r = 1.0f / d;
    if (!(fabsf(r) <= 3.0e38f)) r = (f2u(d) >> 31) ? -3.0e38f : 3.0e38f;
    return r;
}

/* Moeller-Trumbore on a pre-gathered leaf record; src/mesh.cpp:39-76 */
NORI_HD bool tri_test(f3 p0, f3 edge1, f3 edge2, f3 o, f3 d, float &u, float &v, float &t) {
    f3 pvec = cross(d, edge2);
    float det = dot(edge1, pvec);
    if (det > -1e-8f && det < 1e-8f) return false;
    float inv_det = 1.0f / det;
    f3 tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross(tvec, edge1);
    v = dot(d, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(edge2, qvec) * inv_det;
    return true;
}

/* Slab test (include/nori/bbox.h:323-350) of BOTH child boxes of a node.  The x and y planes are
 * processed as 2-wide vectors against (o.x, o.y) / (rcp.x, rcp.y), the z planes of each child as
 * (min, max) pairs: 6 v_pk_add_f32 + 6 v_pk_mul_f32 (full rate on gfx950: two floats per lane per
 * issue) instead of 24 scalar VALU ops.  Element-wise IEEE -- same bits as the scalar form.  No NaN
 * can occur (slab_rcp is finite, boxes are finite), so the first axis initialises the interval. */
typedef float v2f __attribute__((vector_size(8)));

NORI_HD void slab_two(const f4 &q0, const f4 &q1, const f4 &q2, f3 o, f3 rcp, float &nl, float &fl, float &nr, float &fr) {
    const v2f oxy = {o.x, o.y}, rxy = {rcp.x, rcp.y}, ozz = {o.z, o.z}, rzz = {rcp.z, rcp.z};
    const v2f lmn = {q0.x, q0.y}, lmx = {q0.z, q0.w}, rmn = {q1.x, q1.y}, rmx = {q1.z, q1.w};
    const v2f lz = {q2.x, q2.y}, rz = {q2.z, q2.w};
    const v2f a = (lmn - oxy) * rxy, b = (lmx - oxy) * rxy;
    const v2f c = (rmn - oxy) * rxy, d = (rmx - oxy) * rxy;
    const v2f e = (lz - ozz) * rzz, f = (rz - ozz) * rzz;
    nl = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(e[0], e[1]));
    fl = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(e[0], e[1]));
    nr = fmaxf(fmaxf(fminf(c[0], d[0]), fminf(c[1], d[1])), fminf(f[0], f[1]));
    fr = fminf(fminf(fmaxf(c[0], d[0]), fmaxf(c[1], d[1])), fmaxf(f[0], f[1]));
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

template <class Stack>
NORI_HD void trav_begin(const DevScene &sc, const RayIn &ray, bool any, Stack &stack, Trav &tv) {
    tv.o = ray.o; tv.d = ray.d;
    tv.rcp = mk3(slab_rcp(ray.d.x), slab_rcp(ray.d.y), slab_rcp(ray.d.z));
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

/* one inner-node step: both child boxes from one 64-B record */
template <bool COUNT, class Stack>
NORI_HD void trav_inner_step(const DevScene &sc, Stack &stack, Trav &tv, TraversalCounters &cnt) {
    const f4 *nq = sc.nodes + (size_t) tv.node * kNodeQuads;
    const f4 q0 = nq[0], q1 = nq[1], q2 = nq[2], q3 = nq[3];
    if (COUNT) cnt.nodes++;
    float nl, fl, nr, fr;
    slab_two(q0, q1, q2, tv.o, tv.rcp, nl, fl, nr, fr);
    fl *= 1.0000004f; fr *= 1.0000004f;
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
    inv[0] = 1.0f / det[0]; inv[1] = 1.0f / det[1];
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

/* one leaf step: ONE PAIR of triangles, then advance within the leaf */
template <bool COUNT, class Stack>
NORI_HD void trav_leaf_step(const DevScene &sc, Stack &stack, Trav &tv, TraversalCounters &cnt) {
    const uint32_t cursor = ~(uint32_t) tv.node;
    const f4 *tq = sc.tris + (size_t) (cursor >> 3) * kPairQuads;
    const f4 q0 = tq[0], q1 = tq[1], q2 = tq[2], q3 = tq[3], q4 = tq[4];
    if (COUNT) cnt.tris += 2;
    TriPairHit r;
    tri_pair_test(q0, q1, q2, q3, q4, tv.o, tv.d, tv.mint, tv.hit.t, r);
    if (r.ok[0] || r.ok[1]) {
        const f4 q5 = tq[5];
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
 * Returns true if something was hit; for closest-hit `hit` holds t,u,v,tri,mesh. */
template <bool COUNT, class Stack>
NORI_HD bool traverse(const DevScene &sc, const RayIn &ray, bool any, Stack &stack, Hit &hit, TraversalCounters &cnt) {
    Trav tv;
    trav_begin(sc, ray, any, stack, tv);
    while (trav_active(tv)) {
        if (trav_at_inner(tv)) trav_inner_step<COUNT>(sc, stack, tv, cnt);
        else trav_leaf_step<COUNT>(sc, stack, tv, cnt);
    }
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
    return mk3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}
NORI_HD f3 mat_vector(const float *m, f3 v) {
    return mk3(m[0] * v.x + (m[1] * v.y + m[2] * v.z),
               m[4] * v.x + (m[5] * v.y + m[6] * v.z),
               m[8] * v.x + (m[9] * v.y + m[10] * v.z));
}
NORI_HD void camera_sample_ray(const CameraRec &cam, f2 samplePosition, RayIn &ray) {
    f3 nearP = mat_point(cam.sample_to_camera, mk3(samplePosition.x * cam.inv_w, samplePosition.y * cam.inv_h, 0.0f));
    f3 d = normalized(nearP);
    float invZ = 1.0f / d.z;
    ray.o = mat_point(cam.camera_to_world, mk3(0.0f, 0.0f, 0.0f));
    ray.d = mat_vector(cam.camera_to_world, d);
    ray.mint = cam.near_clip * invZ;
    ray.maxt = cam.far_clip * invZ;
}

} // namespace nrt
