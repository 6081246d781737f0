/*
 * wf_records.h -- the records the wavefront engine keeps per path in HBM (wavefront.hip): flags, the 16-B
 * hit record wf_extend writes, and PathState (rt_path.h) <-> state arrays.  Plain per-lane code, shared
 * with the CPU emulation harness, which walks paths through the same encodings (tests/emu/emu.cpp,
 * emu_li_records).
 */
#pragma once
#include "rt_path.h"

namespace nrt {

/* path flags: F_* | prev_measure << 4 | depth << 8;  0 = no path in this slot */
constexpr uint32_t F_HAS_A = 1u, F_HAS_B = 2u, F_END_AFTER_B = 4u;
/* hit word: low 31 bits = global triangle of the closest hit or kMissA, bit 31 = shadow ray occluded */
constexpr uint32_t kMissA = 0x7fffffffu, kOccludedB = 0x80000000u;

/* (t, u, v, hit word) of a path vertex: closest hit of the continuation ray (null: none traced) and the
   answer of its shadow ray */
NORI_HD f4 hit_pack(const Hit *closest, bool shadow_occluded) {
    f4 h; h.x = kInf; h.y = h.z = 0.0f;
    uint32_t w = kMissA;
    if (closest) { h.x = closest->t; h.y = closest->u; h.z = closest->v; if (closest->tri != kNoHit) w = closest->tri; }
    h.w = u2f(w | (shadow_occluded ? kOccludedB : 0u));
    return h;
}

NORI_HD void hit_unpack(const DevScene &sc, const f4 &h, Hit &hit, bool &found) {
    const uint32_t hw = f2u(h.w);
    hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.tri = hw & kMissA;
    found = hit.tri != kMissA;
    if (!found) hit.tri = kNoHit;
    hit.mesh = found ? f2u(sc.shade_tris[(size_t) hit.tri * kShadeQuads].w) : kNoHit;      /* same fetch as p0 */
}

/* state record -> PathState */
NORI_HD void vertex_unpack(PathState &st, uint32_t fl, const f4 &L4, const f4 &t4, uint64_t rng_state, uint64_t rng_inc) {
    st.L = mk3(L4.x, L4.y, L4.z);
    st.T = mk3(t4.x, t4.y, t4.z); st.eta = t4.w; st.pdf_mat = L4.w;
    st.Ld = mk3(0.0f); st.cont_d = mk3(0.0f);
    st.prev_measure = (int32_t) ((fl >> 4) & 3u); st.depth = (int32_t) (fl >> 8);
    st.phase = PH_CLOSEST; st.end_after_shadow = 0;
    st.ray.o = st.ray.d = mk3(0.0f); st.ray.mint = st.ray.maxt = 0.0f;
    st.rng.inc = rng_inc; st.rng.state = rng_state;
}

/* the surviving path's next record: origin, continuation ray (slot A), shadow ray (slot B) */
NORI_HD void vertex_pack(const PathState &st, f4 &o, f4 &dA, f4 &dB, f4 &T, f4 &L, f4 &Ld, uint32_t &fl) {
    o.x = st.ray.o.x; o.y = st.ray.o.y; o.z = st.ray.o.z;
    fl = ((uint32_t) st.prev_measure << 4) | ((uint32_t) st.depth << 8);
    if (st.phase == PH_SHADOW) {
        dB.x = st.ray.d.x; dB.y = st.ray.d.y; dB.z = st.ray.d.z; dB.w = st.ray.maxt;
        Ld.x = st.Ld.x; Ld.y = st.Ld.y; Ld.z = st.Ld.z; Ld.w = 0.0f;
        fl |= F_HAS_B;
        o.w = kEpsilon;
        if (st.end_after_shadow) fl |= F_END_AFTER_B;
        else {
            dA.x = st.cont_d.x; dA.y = st.cont_d.y; dA.z = st.cont_d.z; dA.w = kInf;
            fl |= F_HAS_A;
        }
    } else {
        dA.x = st.ray.d.x; dA.y = st.ray.d.y; dA.z = st.ray.d.z; dA.w = st.ray.maxt;
        fl |= F_HAS_A;
        o.w = st.ray.mint;
    }
    T.x = st.T.x; T.y = st.T.y; T.z = st.T.z; T.w = st.eta;
    L.x = st.L.x; L.y = st.L.y; L.z = st.L.z; L.w = st.pdf_mat;
}

/* ---- how a path's record lies in HBM (wavefront.hip, WfState).  Two facts of rt_path.h shrink it: every stored ray leaves a
   surface with mint = kEpsilon, and every stored continuation ray has maxt = inf (only the camera ray differs, and the first
   vertex is never stored) -- so neither is kept, the origin and the emitter sample are three floats, and the word freed next
   to the continuation direction carries the path flags, which both kernels that read the direction need:
       o    12 B  origin (both rays)                       dA   16 B  (continuation direction, bits: path flags) -- always written
       dB   16 B  (shadow direction, maxt) if F_HAS_B      Ld   12 B  emitter sample added if the shadow ray is unoccluded
       T_eta, L_pdf 16 B each, sidx 4 B, rng 8 B                      = 100 B per record and copy (round 3: 112)
   wf_extend reads o + dA + dB = 44 B per path (+ dA again after the shadow ray: 16 B -- the origin is still in its registers;
   round 3: 52 + 32), wf_shade reads 88 B and writes 100 (96 / 112). */
NORI_HD P3 p3_of(const f4 &v) { P3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }
NORI_HD f4 state_dA(const f4 &dA, uint32_t fl, bool has_a) {      /* what is stored for the continuation ray: direction (or zeros) + flags */
    f4 r; r.x = has_a ? dA.x : 0.0f; r.y = has_a ? dA.y : 0.0f; r.z = has_a ? dA.z : 0.0f; r.w = u2f(fl);
    return r;
}
NORI_HD uint32_t state_flags(const f4 &stored_dA) { return f2u(stored_dA.w); }
constexpr float kStoredMint = kEpsilon;      /* of every stored ray */

} // namespace nrt
