/* TEST / TOOLING ONLY -- wave-level model of wf_extend's scheduling (tools/wave_sim.py).
 *
 * wf_extend's time goes into VALU instructions issued for 64 lanes whether 64 or 20 of them have work, so what a scheduling
 * policy (refill / leaf / repeat thresholds, postponed leaves, the order paths arrive in) is worth is a COUNT: how many
 * node steps, leaf steps, refills and loop trips a wave executes for the same rays.  This model runs the real per-lane
 * traversal code (rt_trace.h) for 64 lanes under the kernel's own voting rules and counts them -- no GPU needed; the
 * GPU census (NORI_HIP_CENSUS) pins the model for the policy the kernel ships with.
 */
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

struct SimEntry {            /* one path of one pass: its shadow ray (B) and / or continuation ray (A), wf_records.h */
    bool hasA, hasB;
    RayIn A, B;
};

struct SimPolicy {
    int refill_threshold, leaf_threshold, inner_repeat;
    int postpone;            /* 1: a lane that reaches a leaf parks it and walks on until it holds a second one */
    int chunk;               /* paths per wave */
    int sort_octant;         /* 1: within blocks of 256 paths, order by the direction octant of the first ray */
    int pend_threshold;      /* > 0: lanes whose shadow ray is answered start their continuation ray as soon as this many wait (no new paths) */
    int pretest;             /* 1: every ray tests the image's cached leaf pairs (the walls) when it starts -- dense, at the lanes of the refill --
                                and the walk skips those leaves (experiment: DESIGN.md section 9) */
};

struct SimCounts {
    uint64_t rays, trips, node_steps, node_lanes, leaf_steps, leaf_lanes, refills, refill_lanes, lane_node_steps, lane_leaf_steps, node_idle_lanes, node_leaf_lanes, pend_restarts, pend_lanes;
    uint64_t pre_steps, pre_lanes;
};

struct SimLeafStack {        /* the leaf step's pop when it works on a parked leaf: nothing to pop, the leaf is done */
    static constexpr int kLeafEnd = 0x7ffffffe;
    void reset() {}
    void push(int) {}
    int pop_or(int) { return kLeafEnd; }
};

struct SimLane {
    Trav tv;
    ArrayStack stack;
    int leaf2;               /* parked leaf link or kTravDone */
    bool pendA;
    RayIn nextA;
};

static inline bool sim_wants_leaf(const SimLane &l, bool postpone) {
    if (!postpone) return trav_at_leaf(l.tv);
    return trav_at_leaf(l.tv) || (l.leaf2 != kTravDone && !trav_at_inner(l.tv));
}
static inline bool sim_active(const SimLane &l) { return trav_active(l.tv) || l.leaf2 != kTravDone; }

static inline bool sim_cached_leaf(const SimLane &l) { return trav_at_leaf(l.tv) && ((~(uint32_t) l.tv.node) & (uint32_t) kTopBit) != 0u; }
/* pretest: a lane never works on a cached leaf -- it was tested when the ray started */
static inline void sim_skip_cached(SimLane &l, bool pretest) { if (pretest) while (sim_cached_leaf(l)) trav_pop(l.stack, l.tv); }

static void sim_begin(const DevScene &sc, SimLane &l, const RayIn &ray, bool any, TopNodesP top, bool pretest = false) {
    trav_begin<kLayoutAny>(sc, ray, any, l.stack, l.tv);
    l.leaf2 = kTravDone;
    if (top == nullptr || !trav_active(l.tv)) return;
    if (pretest) {
        const uint32_t n_pairs = f2u(top[0].z), first = f2u(top[0].w) / kPairQuads - n_pairs;      /* the pair records close the image */
        TraversalCounters tc; tc.nodes = tc.tris = 0;
        for (uint32_t p = 0; p < n_pairs && trav_active(l.tv); ++p) {
            l.tv.node = (int) ~(((kTopPairBase + first + p) << 3) | 0u);
            SimLeafStack ls;
            trav_leaf_step<false>(sc, ls, l.tv, tc, top);
        }
        if (l.tv.node == kTravDone) return;      /* an any-hit ray stopped by a wall */
    }
    l.tv.node = (int) f2u(top[0].x);
}

static void sim_wave(const DevScene &sc, const SimEntry *e, size_t n, const SimPolicy &P, SimCounts &C) {
    std::vector<SimLane> L(64);
    for (auto &l : L) { trav_idle(l.tv); l.leaf2 = kTravDone; l.pendA = false; }
    const TopNodesP top = sc.top_image;
    TraversalCounters tc; tc.nodes = tc.tris = 0;
    size_t pos = 0;
    while (true) {
        int nIdle = 0; bool anyPend = false;
        for (auto &l : L) { if (!sim_active(l)) { ++nIdle; anyPend |= l.pendA; } }
        const bool exhausted = pos >= n;
        if ((!exhausted || anyPend) && (nIdle >= P.refill_threshold || nIdle == 64)) {
            C.refills++; C.refill_lanes += (uint64_t) nIdle;
            if (P.pretest && top != nullptr) { C.pre_steps += f2u(top[0].z); C.pre_lanes += (uint64_t) nIdle * f2u(top[0].z); }
            for (auto &l : L) {
                if (sim_active(l)) continue;
                if (l.pendA) { sim_begin(sc, l, l.nextA, false, top, P.pretest != 0); l.pendA = false; C.rays++; continue; }
                if (pos >= n) continue;
                const SimEntry &en = e[pos++];
                if (en.hasB) { sim_begin(sc, l, en.B, true, top, P.pretest != 0); C.rays++; l.pendA = en.hasA; l.nextA = en.A; }
                else if (en.hasA) { sim_begin(sc, l, en.A, false, top, P.pretest != 0); C.rays++; }
            }
        }
        else if (P.pend_threshold > 0) {
            int np = 0;
            for (auto &l : L) if (!sim_active(l) && l.pendA) ++np;
            if (np >= P.pend_threshold) {
                C.pend_restarts++; C.pend_lanes += (uint64_t) np;
                for (auto &l : L) if (!sim_active(l) && l.pendA) { sim_begin(sc, l, l.nextA, false, top, P.pretest != 0); l.pendA = false; C.rays++; }
            }
        }
        bool anyActive = false;
        for (auto &l : L) anyActive |= sim_active(l);
        if (!anyActive) {
            bool pend = false;
            for (auto &l : L) pend |= l.pendA;
            if (pos >= n && !pend) break;
            continue;
        }
        C.trips++;
        while (true) {
            int ni = 0;
            for (auto &l : L) if (trav_at_inner(l.tv)) ++ni;
            if (ni) {
                C.node_steps++; C.node_lanes += (uint64_t) ni; C.lane_node_steps += (uint64_t) ni;
                for (auto &l : L) { if (!sim_active(l)) C.node_idle_lanes++; else if (!trav_at_inner(l.tv)) C.node_leaf_lanes++; }      /* what the other lanes wait for */
            }
            for (auto &l : L) {
                if (!trav_at_inner(l.tv)) continue;
                if (sc.wide) trav_wide_step<false>(sc, l.stack, l.tv, tc, top);
                else trav_inner_step<false>(sc, l.stack, l.tv, tc, top);
                sim_skip_cached(l, P.pretest != 0);
                if (P.postpone && trav_at_leaf(l.tv) && l.leaf2 == kTravDone) { l.leaf2 = l.tv.node; trav_pop(l.stack, l.tv); }
            }
            int left = 0;
            for (auto &l : L) if (trav_at_inner(l.tv)) ++left;
            if (left < P.inner_repeat) break;
        }
        int nLeaf = 0; bool innerLeft = false;
        for (auto &l : L) { if (sim_wants_leaf(l, P.postpone != 0)) ++nLeaf; innerLeft |= trav_at_inner(l.tv); }
        if (nLeaf && (nLeaf >= P.leaf_threshold || !innerLeft)) {
            C.leaf_steps++; C.leaf_lanes += (uint64_t) nLeaf; C.lane_leaf_steps += (uint64_t) nLeaf;
            for (auto &l : L) {
                if (!sim_wants_leaf(l, P.postpone != 0)) continue;
                if (P.postpone && l.leaf2 != kTravDone) {      /* the parked leaf first: its pairs one by one; the walk (tv.node) rests */
                    const int keep = l.tv.node;
                    l.tv.node = l.leaf2;
                    SimLeafStack ls;
                    trav_leaf_step<false>(sc, ls, l.tv, tc, top);
                    if (l.tv.node == kTravDone) { l.leaf2 = kTravDone; l.stack.reset(); }      /* any-hit answered: the ray is done */
                    else {
                        l.leaf2 = l.tv.node == SimLeafStack::kLeafEnd ? kTravDone : l.tv.node;
                        l.tv.node = keep;
                        /* a lane whose walk stands at a leaf takes it over as the parked one */
                        if (l.leaf2 == kTravDone && trav_at_leaf(l.tv)) { l.leaf2 = l.tv.node; trav_pop(l.stack, l.tv); }
                    }
                } else {
                    trav_leaf_step<false>(sc, l.stack, l.tv, tc, top);
                    sim_skip_cached(l, P.pretest != 0);
                }
            }
        }
    }
}

/* the passes of a frame as wf_extend sees them: level k = the paths alive at their k-th vertex, in the engine's order
   (tile-major, sample-major, pixel of the tile), each with its shadow and / or continuation ray */
template <int INTEG>
static void sim_capture_path(const DevScene &sc, const RayIn &cam, uint64_t rng_state, uint64_t rng_inc, ArrayStack &stack,
                             std::vector<std::vector<SimEntry>> &levels, size_t max_levels) {
    TraversalCounters tc; tc.nodes = tc.tris = 0;
    f4 o, dA, dB, T, L, Ld;
    o.x = cam.o.x; o.y = cam.o.y; o.z = cam.o.z; o.w = cam.mint;
    dA.x = cam.d.x; dA.y = cam.d.y; dA.z = cam.d.z; dA.w = cam.maxt;
    dB = dA; Ld.x = Ld.y = Ld.z = Ld.w = 0.0f;
    T.x = T.y = T.z = T.w = 1.0f; L.x = L.y = L.z = L.w = 0.0f;
    uint32_t fl = F_HAS_A | (2u << 4);
    for (size_t level = 0; ; ++level) {
        SimEntry en; en.hasA = (fl & F_HAS_A) != 0; en.hasB = (fl & F_HAS_B) != 0;
        en.B.o = mk3(o.x, o.y, o.z); en.B.d = mk3(dB.x, dB.y, dB.z); en.B.mint = kEpsilon; en.B.maxt = dB.w;
        en.A.o = mk3(o.x, o.y, o.z); en.A.d = mk3(dA.x, dA.y, dA.z); en.A.mint = o.w; en.A.maxt = dA.w;
        if (level < max_levels) { if (levels.size() <= level) levels.resize(level + 1); levels[level].push_back(en); }
        bool occluded = false;
        if (en.hasB) { Hit sh; occluded = traverse<false>(sc, en.B, true, stack, sh, tc); }
        f4 h;
        if (en.hasA) { Hit hit; (void) traverse<false>(sc, en.A, false, stack, hit, tc); h = hit_pack(&hit, occluded); }
        else h = hit_pack(nullptr, occluded);
        if (en.hasB && !(f2u(h.w) & kOccludedB)) { L.x = L.x + Ld.x; L.y = L.y + Ld.y; L.z = L.z + Ld.z; }
        if (fl & F_END_AFTER_B) break;
        Hit hit; bool found;
        hit_unpack(sc, h, hit, found);
        PathState st;
        vertex_unpack(st, fl, L, T, rng_state, rng_inc);
        const bool done = path_on_closest<INTEG>(sc, st, hit, found, mk3(dA.x, dA.y, dA.z));
        L.x = st.L.x; L.y = st.L.y; L.z = st.L.z;
        if (done) break;
        vertex_pack(st, o, dA, dB, T, L, Ld, fl);
        rng_state = st.rng.state;
    }
}

static int sim_octant(const SimEntry &e) {
    const RayIn &r = e.hasB ? e.B : e.A;
    return (r.d.x < 0.0f ? 1 : 0) | (r.d.y < 0.0f ? 2 : 0) | (r.d.z < 0.0f ? 4 : 0);
}
