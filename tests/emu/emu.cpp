/*
 * emu.cpp -- TEST-ONLY host emulation of the device kernels' per-lane code.
 *
 * There is no GPU in the development container, so the per-lane device code
 * (nori_amd/csrc/device/rt_*.h: traversal, Moeller-Trumbore, path state
 * machine, BSDFs, warps, camera, tile splat) is also compiled here with g++ and
 * driven by a sequential loop that mirrors the thread/tile loop of
 * render_kernel in nori_hip.hip.  This lets `pytest -m "not gpu"` check the
 * kernel LOGIC against the oracle on the CPU before GPU minutes are spent.
 *
 * It is NOT a CPU fallback: it is not part of libnori_hip.so, the product never
 * loads it, and the `-m gpu` tests exercise the real HIP kernels through the
 * C ABI.  Built by tests/emu/Makefile into tests/emu/libnori_emu.so.
 */
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nori_hip.h"
#include "emu_film.h"
#include "emu_builder.h"
#include "../../nori_amd/csrc/device/wf_records.h"
#include "../../nori_amd/csrc/device/rt_path.h"
#include "../../nori_amd/csrc/device/scene_prep.h"

using namespace nrt;

struct ArrayStack {
    int data[128];
    int sp = 0;
    int high = 0;
    void reset() { sp = 0; }
    bool empty() const { return sp == 0; }
    int pop_or(int empty_value) { return sp == 0 ? empty_value : pop(); }
    void push(int v) { data[sp++] = v; if (sp > high) high = sp; }
    int pop() { return data[--sp]; }
};

struct emu_ctx {
    HostScene host;
    HostBvh bvh;
    DevScene dev;
    std::vector<f4> top_image;      /* rt_top.h: the records wf_extend keeps in LDS; the harness walks through them too */
    std::vector<f4> nodes_q, top_image_q;      /* rt_nodeq.h: the 32-B node records and the image that holds them */
    std::string error;
    uint32_t n_refs = 0;      /* references the device builders' steps built the tree over (emu_builder.h) */
};

static void bind(emu_ctx *c) {
    DevScene &d = c->dev;
    std::memset(&d, 0, sizeof(d));
    HostScene &h = c->host;
    d.nodes = c->bvh.nodes.data(); d.tris = c->bvh.tris.data();
    d.positions = h.positions.data(); d.normals = h.normals.data(); d.texcoords = h.texcoords.data();
    d.indices = h.indices.data(); d.meshes = h.meshes.data(); d.emitter_cdf = h.emitter_cdf.data();
    d.emitters = h.emitters.data(); d.tri_mesh = h.tri_mesh.data();
    d.shade_tris = h.shade_tris.data();
    d.n_emitters = (uint32_t) h.emitters.size(); d.n_meshes = (uint32_t) h.meshes.size();
    d.n_triangles = (uint32_t) h.tri_mesh.size();
    d.n_cdf = (uint32_t) h.emitter_cdf.size();
    d.root = c->bvh.root;
    d.wide = c->bvh.wide ? 1u : 0u;
    /* the nodes as 32-B records (rt_nodeq.h) when the tree qualifies -- the harness then walks those, as wf_extend does */
    const char *nqe = std::getenv("NORI_EMU_NODEQ");
    d.nodes_q = nullptr; d.top_image_q = nullptr; d.top_image_q_quads = 0u; d.top_image = nullptr; d.top_image_quads = 0u;
    if (d.n_triangles > 0 && !d.wide && d.root >= 0 && !(nqe && atoi(nqe) == 0) && nodeq_grid(d.nodes, d.root, d.grid)) {
        const size_t n_nodes = c->bvh.nodes.size() / kNodeQuads;
        c->nodes_q.assign(n_nodes * kNodeqQuads, f4());
        bool ok = true;
        for (size_t i = 0; i < n_nodes; ++i) ok &= nodeq_from_node(d.nodes + i * kNodeQuads, d.grid, c->nodes_q.data() + i * kNodeqQuads);
        if (ok) d.nodes_q = c->nodes_q.data();
        /* experiment: what the planes would be worth as fp16 -- every plane moved outward to the next value
           binary16 represents in coordinates normalised to [-1, 1] around the grid's centre (11 bits of mantissa: finest at the
           centre, 2^-11 of the half extent at the faces); NORI_EMU_NODEQ_FP16=1, tools/trav_histogram.py counts the tests */
        if (ok && std::getenv("NORI_EMU_NODEQ_FP16") && atoi(std::getenv("NORI_EMU_NODEQ_FP16")) != 0) {
            auto snap = [](uint32_t q, bool up) {
                const double u = (double) q / 65535.0 * 2.0 - 1.0, a = fabs(u);
                if (a == 0.0) return q;
                int e; (void) frexp(a, &e);                               /* a = m 2^e, m in [0.5, 1) */
                if (e < -13) e = -13;                                      /* binary16 denormal range: fixed spacing 2^-24 */
                const double ulp = ldexp(1.0, e - 11);
                const bool away = (u > 0.0) == up;                         /* away from zero? */
                const double r = away ? ceil(a / ulp) * ulp : floor(a / ulp) * ulp;
                const double v = ((u > 0.0 ? r : -r) + 1.0) * 0.5 * 65535.0;
                const double qq = up ? ceil(v - 1e-9) : floor(v + 1e-9);
                return (uint32_t) (qq < 0.0 ? 0.0 : qq > 65535.0 ? 65535.0 : qq);
            };
            for (size_t i = 0; i < n_nodes; ++i) {
                uint32_t *w = reinterpret_cast<uint32_t *>(c->nodes_q.data() + i * kNodeqQuads);
                for (int a = 0; a < 3; ++a) {
                    w[2 * a] = snap(w[2 * a] & 0xffffu, false) | (snap(w[2 * a] >> 16, false) << 16);
                    w[2 * a + 1] = snap(w[2 * a + 1] & 0xffffu, true) | (snap(w[2 * a + 1] >> 16, true) << 16);
                }
            }
        }
    }
    const char *ti = std::getenv("NORI_EMU_TOP_IMAGE");
    if (d.n_triangles > 0 && !(ti && atoi(ti) == 0)) {
        /* as many node records as the device keeps (wavefront.hip, wf_top_capacity) unless told otherwise */
        const char *tn = std::getenv("NORI_EMU_TOP_NODES");
        c->top_image.assign(kTopImageMaxQuads, f4());
        top_image_build(d.nodes, d.nodes, top_layout(false), d.tris, d.root, d.wide != 0u, d.n_triangles, tn ? atoi(tn) : (d.wide ? 113 : 143), c->top_image.data());
        d.top_image = c->top_image.data();
        d.top_image_quads = f2u(c->top_image[0].w);
        if (d.nodes_q) {
            c->top_image_q.assign(kTopImageMaxQuads, f4());
            top_image_build(d.nodes, d.nodes_q, top_layout(true), d.tris, d.root, false, d.n_triangles, tn ? atoi(tn) : 359, c->top_image_q.data());
            d.top_image_q = c->top_image_q.data();
            d.top_image_q_quads = f2u(c->top_image_q[0].w);
        }
    }
    d.camera = h.camera; d.filter = h.filter; d.integrator = h.integrator;
}

template <int INTEG>
static f3 run_path(const DevScene &sc, PathState &st, ArrayStack &stack, uint64_t &nClosest, uint64_t &nShadow, TraversalCounters &tc) {
    while (true) {
        const bool any = st.phase == PH_SHADOW;
        Hit hit;
        const f3 qo = st.ray.o, qd = st.ray.d;
        const bool found = traverse<true>(sc, st.ray, any, stack, hit, tc);
        bool done;
        if (any) { ++nShadow; done = path_on_shadow(st, found, qo); }
        else { ++nClosest; done = path_on_closest<INTEG>(sc, st, hit, found, qd); }
        if (done) break;
    }
    return st.L;
}

static f3 run_path_dyn(const DevScene &sc, PathState &st, ArrayStack &stack, uint64_t &nc, uint64_t &ns, TraversalCounters &tc) {
    switch (sc.integrator.type) {
    case 0: return run_path<0>(sc, st, stack, nc, ns, tc);
    case 1: return run_path<1>(sc, st, stack, nc, ns, tc);
    case 2: return run_path<2>(sc, st, stack, nc, ns, tc);
    case 3: return run_path<3>(sc, st, stack, nc, ns, tc);
    case 4: return run_path<4>(sc, st, stack, nc, ns, tc);
    case 5: return run_path<5>(sc, st, stack, nc, ns, tc);
    default: return run_path<6>(sc, st, stack, nc, ns, tc);
    }
}

struct PlainAdd { void operator()(float *p, float v) const { *p += v; } };

/* Integrator::Li the way the wavefront engine walks a path (wavefront.hip: wf_extend / wf_shade / wf_finish):
   per vertex the shadow ray, then the continuation ray, the answers squeezed through the 16-B hit record, the
   path through the state records of wf_records.h.  Must equal emu_li bit for bit. */
template <int INTEG>
static f3 run_path_records(const DevScene &sc, const RayIn &cam, uint64_t rng_state, uint64_t rng_inc, ArrayStack &stack, TraversalCounters &tc) {
    f4 o, dA, dB, T, L, Ld;
    o.x = cam.o.x; o.y = cam.o.y; o.z = cam.o.z; o.w = cam.mint;
    dA.x = cam.d.x; dA.y = cam.d.y; dA.z = cam.d.z; dA.w = cam.maxt;
    dB = dA; Ld.x = Ld.y = Ld.z = Ld.w = 0.0f;
    T.x = T.y = T.z = T.w = 1.0f; L.x = L.y = L.z = L.w = 0.0f;
    uint32_t fl = F_HAS_A | (2u << 4);
    while (true) {
        bool occluded = false;
        if (fl & F_HAS_B) {
            RayIn ray; ray.o = mk3(o.x, o.y, o.z); ray.d = mk3(dB.x, dB.y, dB.z); ray.mint = kEpsilon; ray.maxt = dB.w;
            Hit sh; occluded = traverse<false>(sc, ray, true, stack, sh, tc);
        }
        f4 h;
        if (fl & F_HAS_A) {
            RayIn ray; ray.o = mk3(o.x, o.y, o.z); ray.d = mk3(dA.x, dA.y, dA.z); ray.mint = o.w; ray.maxt = dA.w;
            Hit hit; (void) traverse<false>(sc, ray, false, stack, hit, tc);
            h = hit_pack(&hit, occluded);
        } else {
            h = hit_pack(nullptr, occluded);
        }
        /* wf_shade */
        if ((fl & F_HAS_B) && !(f2u(h.w) & kOccludedB)) { L.x = L.x + Ld.x; L.y = L.y + Ld.y; L.z = L.z + Ld.z; }
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
        /* ... and through HBM as wf_shade stores it and wf_extend / wf_shade load it (wf_records.h): the origin and the emitter
           sample as three floats, mint / maxt of the continuation ray not at all, the flags in the continuation direction's w */
        { const f4 stored = state_dA(dA, fl, (fl & F_HAS_A) != 0u); const P3 o3 = p3_of(o), l3 = p3_of(Ld);
          fl = state_flags(stored); dA = stored; dA.w = kInf;
          o.x = o3.x; o.y = o3.y; o.z = o3.z; o.w = kStoredMint; Ld.x = l3.x; Ld.y = l3.y; Ld.z = l3.z; Ld.w = 0.0f; }
    }
    return mk3(L.x, L.y, L.z);
}


#include "emu_wavesim.h"
#include "../../nori_amd/csrc/device/group_merge.h"
#include "../../nori_amd/csrc/device/film.h"      /* film_block_rows_first_tile: the tile range of a share of block rows */

extern "C" {

/* tools/wave_sim.py: counts[level][14] for the first `max_levels` passes of a spp-sample frame under `policy` (7 ints) */
/* policy[8 .. 11] (emu_wave_sim_ex): tile stride (only tiles whose index is a multiple are simulated -- a full-resolution frame
   sampled), sort window in paths (0: none), key kind, origin-cell bits per axis.  Key kinds: 1 octant of the first ray traced;
   2 (Morton cell of the origin, octant of the continuation ray); 3 (octant of the continuation ray, Morton cell); 4 as 2 with the
   "has a shadow ray" bit on top; 5 (cell, 6-bit direction code of the continuation ray: octant + order of the components) */
static uint32_t sim_morton3(uint32_t x, uint32_t y, uint32_t z, int bits) {
    uint32_t m = 0;
    for (int b = 0; b < bits; ++b) m |= (((x >> b) & 1u) << (3 * b)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b + 2));
    return m;
}
static uint32_t sim_sort_key(const DevScene &sc, const SimEntry &e, int kind, int bits) {
    const RayIn &first = e.hasB ? e.B : e.A;
    const RayIn &cont = e.hasA ? e.A : e.B;
    auto oct = [](const RayIn &r) { return (uint32_t) ((r.d.x < 0.0f ? 1 : 0) | (r.d.y < 0.0f ? 2 : 0) | (r.d.z < 0.0f ? 4 : 0)); };
    if (kind == 1) return oct(first);
    const float o[3] = {cont.o.x, cont.o.y, cont.o.z};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {      /* the 16-bit grid of the node records (rt_nodeq.h) spans the scene: its top bits */
        const float t = (o[a] - sc.grid.mn[a]) / (sc.grid.scale[a] * 65536.0f);
        q[a] = (uint32_t) std::min((float) ((1 << bits) - 1), std::max(0.0f, t * (float) (1 << bits)));
    }
    const uint32_t cell = sim_morton3(q[0], q[1], q[2], bits);
    if (kind == 2) return (cell << 3) | oct(cont);
    if (kind == 3) return (oct(cont) << (3 * bits)) | cell;
    if (kind == 4) return ((e.hasB ? 1u : 0u) << (3 * bits + 3)) | (cell << 3) | oct(cont);
    if (kind == 5) {
        const float ax = std::fabs(cont.d.x), ay = std::fabs(cont.d.y), az = std::fabs(cont.d.z);
        const uint32_t major = ax >= ay && ax >= az ? 0u : ay >= az ? 1u : 2u;
        return (cell << 5) | (oct(cont) << 2) | major;
    }
    if (kind == 6 || kind == 7) {      /* the continuation direction on the octahedral map, 16 x 16 cells */
        const float n1 = std::fabs(cont.d.x) + std::fabs(cont.d.y) + std::fabs(cont.d.z);
        float u = cont.d.x / n1, v = cont.d.y / n1;
        if (cont.d.z < 0.0f) { const float uu = (1.0f - std::fabs(v)) * (u < 0.0f ? -1.0f : 1.0f), vv = (1.0f - std::fabs(u)) * (v < 0.0f ? -1.0f : 1.0f); u = uu; v = vv; }
        const uint32_t du = (uint32_t) std::min(15.0f, std::max(0.0f, (u * 0.5f + 0.5f) * 16.0f)), dv = (uint32_t) std::min(15.0f, std::max(0.0f, (v * 0.5f + 0.5f) * 16.0f));
        const uint32_t dir = sim_morton3(du, dv, 0u, 4);
        return kind == 6 ? (dir << (3 * bits)) | cell : (cell << 12) | dir;
    }
    return 0u;
}

int emu_wave_sim(emu_ctx *c, uint32_t spp, uint32_t max_levels, const int *policy, uint64_t *counts) {
    const DevScene &sc = c->dev;
    if (sc.integrator.type != 6) return NORI_ERR_INVALID_ARGUMENT;      /* path_mis only */
    const int W = sc.camera.width, H = sc.camera.height;
    const uint32_t tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    std::vector<std::vector<SimEntry>> levels;
    ArrayStack stack;
    const uint32_t tile_stride = policy[8] > 0 ? (uint32_t) policy[8] : 1u;
    for (uint32_t tile = 0; tile < tiles_x * tiles_y; tile += tile_stride)
        for (uint32_t s = 0; s < spp; ++s)
            for (int pix = 0; pix < 256; ++pix) {
                const int px = (int) (tile % tiles_x) * kTile + (((pix >> 6) & 1) << 3) + (pix & 7), py = (int) (tile / tiles_x) * kTile + ((pix >> 7) << 3) + ((pix & 63) >> 3);      /* film.h, film_tile_pixel */
                if (px >= W || py >= H) continue;
                Rng rng; rng_seed(rng, (uint64_t) py * (uint64_t) W + (uint64_t) px, (uint64_t) s);
                const f2 j = rng_next_2d(rng);
                (void) rng_next_2d(rng);
                RayIn cam; camera_sample_ray(sc.camera, mk2((float) px + j.x, (float) py + j.y), cam);
                sim_capture_path<6>(sc, cam, rng.state, rng.inc, stack, levels, max_levels);
            }
    SimPolicy P; P.refill_threshold = policy[0]; P.leaf_threshold = policy[1]; P.inner_repeat = policy[2]; P.postpone = policy[3]; P.chunk = policy[4]; P.sort_octant = policy[5]; P.pend_threshold = policy[6]; P.pretest = policy[7];
    const size_t window = (size_t) std::max(0, policy[9]);
    const int kind = policy[10], bits = policy[11];
    for (size_t k = 0; k < levels.size() && k < max_levels; ++k) {
        std::vector<SimEntry> &e = levels[k];
        if (P.sort_octant)
            for (size_t b = 0; b < e.size(); b += 256)
                std::stable_sort(e.begin() + b, e.begin() + std::min(e.size(), b + 256), [](const SimEntry &x, const SimEntry &y) { return sim_octant(x) < sim_octant(y); });
        if (window && kind && k > 0) {
            std::vector<std::pair<uint32_t, uint32_t>> keys(e.size());
            for (size_t i = 0; i < e.size(); ++i) keys[i] = std::make_pair(sim_sort_key(sc, e[i], kind, bits), (uint32_t) i);
            for (size_t b = 0; b < e.size(); b += window) std::stable_sort(keys.begin() + b, keys.begin() + std::min(e.size(), b + window));
            std::vector<SimEntry> sorted(e.size());
            for (size_t i = 0; i < e.size(); ++i) sorted[i] = e[keys[i].second];
            e.swap(sorted);
        }
        SimCounts C; std::memset(&C, 0, sizeof(C));
        for (size_t b = 0; b < e.size(); b += (size_t) P.chunk) sim_wave(sc, e.data() + b, std::min((size_t) P.chunk, e.size() - b), P, C);
        std::memcpy(counts + k * 16, &C, sizeof(C));
    }
    return (int) levels.size();
}

int emu_create(const nori_scene_desc *scene, emu_ctx **out) {
    emu_ctx *c = new emu_ctx();
    std::string err = prepare_scene(*scene, c->host);
    /* node layout as the library picks it (nori_hip_build_accel): NORI_HIP_ACCEL_LAYOUT=bvh4q forces wide nodes */
    const char *lay = std::getenv("NORI_HIP_ACCEL_LAYOUT");
    const bool wide = lay ? std::string(lay) == "bvh4q" : c->host.tri_mesh.size() >= ((size_t) 1 << 20);
    /* NORI_EMU_BUILDER=lbvh | ploc: the DEVICE builders' steps run on the CPU (emu_builder.h) instead of the host SAH builder */
    const char *bld = std::getenv("NORI_EMU_BUILDER");
    const std::string builder = bld ? bld : "sah";
    if (err.empty() && builder != "sah") {
        uint32_t radius = builder == "ploc" ? 8u : 0u;
        if (const char *e = std::getenv("NORI_HIP_PLOC_RADIUS")) if (radius) radius = (uint32_t) std::max(1, atoi(e));
        EmuBuilderStats st;
        err = build_bvh_steps_host(c->host, wide, radius, c->bvh, &st);
        if (err.empty() && c->bvh.max_depth + 1 > 64 && wide) err = build_bvh_steps_host(c->host, false, radius, c->bvh, &st);
        if (err.empty() && c->bvh.max_depth + 1 > 64 && radius) {      /* as nori_hip_build_accel: the radix tree instead */
            err = build_bvh_steps_host(c->host, wide, 0u, c->bvh, &st);
            if (err.empty() && c->bvh.max_depth + 1 > 64 && wide) err = build_bvh_steps_host(c->host, false, 0u, c->bvh, &st);
        }
        if (err.empty() && c->bvh.max_depth + 1 > 64) err = "tree deeper than the traversal stack";
        c->n_refs = st.n_refs;
    } else {
        if (err.empty()) err = build_bvh_sah(c->host, 64, c->bvh, wide);
        if (!err.empty() && wide) err = build_bvh_sah(c->host, 64, c->bvh, false);
    }
    if (!err.empty()) { fprintf(stderr, "emu_create: %s\n", err.c_str()); delete c; return NORI_ERR_INVALID_ARGUMENT; }
    bind(c);
    *out = c;
    return NORI_OK;
}
void emu_destroy(emu_ctx *c) { delete c; }

int emu_accel_info(const emu_ctx *c, nori_accel_info *in) {
    std::memset(in, 0, sizeof(*in));
    in->n_triangles = c->dev.n_triangles; in->n_nodes = c->bvh.n_nodes; in->n_leaves = c->bvh.n_leaves;
    in->max_depth = c->bvh.max_depth; in->node_bytes = kNodeQuads * 16; in->tri_bytes = kPairQuads * 16 / 2;
    in->total_bytes = (uint64_t) (c->bvh.nodes.size() + c->bvh.tris.size()) * 16;
    in->build_ms = c->bvh.build_ms; in->sah_cost = c->bvh.sah_cost;
    in->node_children = c->bvh.wide ? 4u : 2u;
    in->node_records_32b = c->dev.nodes_q != nullptr ? 1u : 0u;
    in->built_on_device = 0u; in->n_references = c->n_refs;
    return NORI_OK;
}
int emu_border_size(const emu_ctx *c) { return c->host.filter.border; }

/* the references split_emit (lbvh_steps.h) makes of ONE triangle cut `cuts` times, in a scene box [smin, smax]: their padded boxes as
   (mn.xyz, mx.xyz) rows and Morton keys; returns how many (tests/test_device_logic_cpu.py: every point of the triangle lies in one of them) */
int emu_split_parts(const float *tri9, uint32_t cuts, float pad0, const float *smin3, const float *smax3, float *boxes6, unsigned long long *keys, int cap) {
    f4 pos[3]; uint32_t idx[3] = {0u, 1u, 2u};
    for (int k = 0; k < 3; ++k) { pos[k].x = tri9[3 * k]; pos[k].y = tri9[3 * k + 1]; pos[k].z = tri9[3 * k + 2]; pos[k].w = 0.0f; }
    const f3 smin = mk3(smin3[0], smin3[1], smin3[2]);
    const f3 ext = mk3(smax3[0] - smin3[0], smax3[1] - smin3[1], smax3[2] - smin3[2]);
    const f3 sinv = mk3(ext.x > 0 ? 1.0f / ext.x : 0.0f, ext.y > 0 ? 1.0f / ext.y : 0.0f, ext.z > 0 ? 1.0f / ext.z : 0.0f);
    const RefOut none{nullptr, nullptr, nullptr, nullptr};
    const uint32_t n = split_emit(pos, idx, 0u, cuts, pad0, smin, sinv, false, none, 0u);
    if ((int) n > cap) return -1;
    std::vector<uint32_t> tri(n); std::vector<f4> mn(n), mx(n); std::vector<unsigned long long> key(n);
    const RefOut out{tri.data(), mn.data(), mx.data(), key.data()};
    if (split_emit(pos, idx, 0u, cuts, pad0, smin, sinv, true, out, 0u) != n) return -2;
    for (uint32_t k = 0; k < n; ++k) {
        boxes6[6 * k] = mn[k].x; boxes6[6 * k + 1] = mn[k].y; boxes6[6 * k + 2] = mn[k].z;
        boxes6[6 * k + 3] = mx[k].x; boxes6[6 * k + 4] = mx[k].y; boxes6[6 * k + 5] = mx[k].z;
        keys[k] = key[k];
    }
    return (int) n;
}

/* raw records for an independent look at the 32-B form (tests/test_device_logic_cpu.py): copies up to `cap` nodes as 16 floats
   (64-B form) and 8 dwords (32-B form) each, the grid as (mn[3], scale[3]); returns the number of nodes, -1 without 32-B records */
long long emu_node_records(emu_ctx *c, float *nodes64, uint32_t *nodes32, float *grid, size_t cap) {
    const DevScene &sc = c->dev;
    if (sc.nodes_q == nullptr) return -1;
    const size_t n = c->nodes_q.size() / kNodeqQuads;
    for (size_t i = 0; i < n && i < cap; ++i) {
        std::memcpy(nodes64 + 16 * i, sc.nodes + i * kNodeQuads, 64);
        std::memcpy(nodes32 + 8 * i, sc.nodes_q + i * kNodeqQuads, 32);
    }
    for (int a = 0; a < 3; ++a) { grid[a] = sc.grid.mn[a]; grid[3 + a] = sc.grid.scale[a]; }
    return (long long) n;
}

/* rt_nodeq.h: every ray against every node of the tree, the 32-B record's verdict per child against the exact one.
   Exact = the slab test of the stored 64-B box [c - h, c + h] in binary64 with the ray's own (binary32) reciprocal
   direction, clipped to [0, maxt] (trav_inner_step's rule).  Returns the number of (ray, child) pairs the exact test
   accepts and the record rejects -- must be 0 -- and writes how many pairs each accepts into counts[0..1].
   -1: the tree has no 32-B records. */
long long emu_nodeq_check(emu_ctx *c, const nori_ray *rays, size_t n, unsigned long long *counts) {
    const DevScene &sc = c->dev;
    if (sc.nodes_q == nullptr) return -1;
    const size_t n_nodes = c->nodes_q.size() / kNodeqQuads;
    unsigned long long bad = 0, acc_exact = 0, acc_q = 0;
    for (size_t i = 0; i < n; ++i) {
        const nori_ray &r = rays[i];
        const f3 o = mk3(r.o[0], r.o[1], r.o[2]);
        const f3 rcp = mk3(slab_rcp(r.d[0]), slab_rcp(r.d[1]), slab_rcp(r.d[2]));
        NodeqRay R; nodeq_ray(sc.grid, o, rcp, R);
        const double oo[3] = {o.x, o.y, o.z}, rr[3] = {rcp.x, rcp.y, rcp.z};
        for (size_t k = 0; k < n_nodes; ++k) {
            float nl, fl, nr, fr;
            nodeq_slabs(sc.nodes_q[k * kNodeqQuads], sc.nodes_q[k * kNodeqQuads + 1], R, nl, fl, nr, fr);
            const bool hq[2] = {nl <= fl && fl >= 0.0f && nl <= r.maxt, nr <= fr && fr >= 0.0f && nr <= r.maxt};
            for (int ch = 0; ch < 2; ++ch) {
                double lo[3], hi[3];
                if (!node_child_box(sc.nodes + k * kNodeQuads, ch, lo, hi)) continue;
                double tn = -1e300, tf = 1e300;
                for (int a = 0; a < 3; ++a) {
                    const double t0 = (lo[a] - oo[a]) * rr[a], t1 = (hi[a] - oo[a]) * rr[a];
                    tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1));
                }
                const bool he = tn <= tf && tf >= 0.0 && tn <= (double) r.maxt;
                acc_exact += he; acc_q += hq[ch];
                if (he && !hq[ch]) ++bad;
            }
        }
    }
    if (counts) { counts[0] = acc_exact; counts[1] = acc_q; }
    return (long long) bad;
}

int emu_intersect(emu_ctx *c, const nori_ray *rays, nori_intersection *out, size_t n, int shadow) {
    const DevScene &sc = c->dev;
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
        ArrayStack stack; TraversalCounters tc; tc.nodes = tc.tris = 0;
        for (size_t i = t; i < n; i += nt) {
            const nori_ray &r = rays[i];
            RayIn ray; ray.o = mk3(r.o[0], r.o[1], r.o[2]); ray.d = mk3(r.d[0], r.d[1], r.d[2]); ray.mint = r.mint; ray.maxt = r.maxt;
            Hit hit;
            const bool found = traverse<false>(sc, ray, shadow != 0, stack, hit, tc);
            nori_intersection o; std::memset(&o, 0, sizeof(o));
            o.mesh = NORI_NO_HIT; o.tri = NORI_NO_HIT;
            if (found && shadow) o.mesh = 0;
            else if (found) {
                Surface sf; f3 ng; f2 uv;
                surface_fill(sc, hit, sf, &ng, &uv);
                const Frame sh = make_frame(sf.ns), geo = make_frame(ng);
                o.p[0] = sf.p.x; o.p[1] = sf.p.y; o.p[2] = sf.p.z; o.t = hit.t; o.uv[0] = uv.x; o.uv[1] = uv.y;
                o.sh_s[0] = sh.s.x; o.sh_s[1] = sh.s.y; o.sh_s[2] = sh.s.z; o.sh_t[0] = sh.t.x; o.sh_t[1] = sh.t.y; o.sh_t[2] = sh.t.z;
                o.sh_n[0] = sh.n.x; o.sh_n[1] = sh.n.y; o.sh_n[2] = sh.n.z;
                o.geo_s[0] = geo.s.x; o.geo_s[1] = geo.s.y; o.geo_s[2] = geo.s.z; o.geo_t[0] = geo.t.x; o.geo_t[1] = geo.t.y; o.geo_t[2] = geo.t.z;
                o.geo_n[0] = geo.n.x; o.geo_n[1] = geo.n.y; o.geo_n[2] = geo.n.z;
                o.mesh = hit.mesh; o.tri = hit.tri - sc.meshes[hit.mesh].tri_offset;
            }
            out[i] = o;
        }
    });
    for (auto &t : th) t.join();
    return NORI_OK;
}

int emu_sample_rays(emu_ctx *c, const float *ps, size_t n, nori_ray *rays) {
    for (size_t i = 0; i < n; ++i) {
        RayIn r; camera_sample_ray(c->dev.camera, mk2(ps[2 * i], ps[2 * i + 1]), r);
        rays[i].o[0] = r.o.x; rays[i].o[1] = r.o.y; rays[i].o[2] = r.o.z;
        rays[i].d[0] = r.d.x; rays[i].d[1] = r.d.y; rays[i].d[2] = r.d.z; rays[i].mint = r.mint; rays[i].maxt = r.maxt;
    }
    return NORI_OK;
}

int emu_li(emu_ctx *c, const nori_ray *rays, size_t n, const uint64_t *ss, const uint64_t *sq, float *rgb) {
    const DevScene &sc = c->dev;
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    if (n < 1024) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
        ArrayStack stack; TraversalCounters tc; tc.nodes = tc.tris = 0; uint64_t a = 0, b = 0;
        for (size_t i = t; i < n; i += nt) {
            const nori_ray &r = rays[i];
            RayIn ray; ray.o = mk3(r.o[0], r.o[1], r.o[2]); ray.d = mk3(r.d[0], r.d[1], r.d[2]); ray.mint = r.mint; ray.maxt = r.maxt;
            PathState st; rng_seed(st.rng, ss[i], sq[i]); path_begin(st, ray);
            f3 L = run_path_dyn(sc, st, stack, a, b, tc);
            rgb[3 * i] = L.x; rgb[3 * i + 1] = L.y; rgb[3 * i + 2] = L.z;
        }
    });
    for (auto &t : th) t.join();
    return NORI_OK;
}

int emu_li_records(emu_ctx *c, const nori_ray *rays, size_t n, const uint64_t *ss, const uint64_t *sq, float *rgb) {
    const DevScene &sc = c->dev;
    ArrayStack stack; TraversalCounters tc; tc.nodes = tc.tris = 0;
    for (size_t i = 0; i < n; ++i) {
        const nori_ray &r = rays[i];
        RayIn ray; ray.o = mk3(r.o[0], r.o[1], r.o[2]); ray.d = mk3(r.d[0], r.d[1], r.d[2]); ray.mint = r.mint; ray.maxt = r.maxt;
        Rng g; rng_seed(g, ss[i], sq[i]);
        f3 L;
        switch (sc.integrator.type) {
        case 0: L = run_path_records<0>(sc, ray, g.state, g.inc, stack, tc); break;
        case 1: L = run_path_records<1>(sc, ray, g.state, g.inc, stack, tc); break;
        case 2: L = run_path_records<2>(sc, ray, g.state, g.inc, stack, tc); break;
        case 3: L = run_path_records<3>(sc, ray, g.state, g.inc, stack, tc); break;
        case 4: L = run_path_records<4>(sc, ray, g.state, g.inc, stack, tc); break;
        case 5: L = run_path_records<5>(sc, ray, g.state, g.inc, stack, tc); break;
        default: L = run_path_records<6>(sc, ray, g.state, g.inc, stack, tc); break;
        }
        rgb[3 * i] = L.x; rgb[3 * i + 1] = L.y; rgb[3 * i + 2] = L.z;
    }
    return NORI_OK;
}

static Bsdf from_desc(const nori_bsdf_desc &d) {
    Bsdf b; b.type = d.type; b.albedo = mk3(d.albedo[0], d.albedo[1], d.albedo[2]);
    b.alpha = d.alpha; b.int_ior = d.int_ior; b.ext_ior = d.ext_ior; b.ks = d.ks; return b;
}
int emu_bsdf_sample(const nori_bsdf_desc *bsdf, const float *wi, const float *sample, size_t n, float *wo, float *weight, float *eta, int32_t *measure) {
    Bsdf b = from_desc(*bsdf);
    for (size_t i = 0; i < n; ++i) {
        f3 o; float e; int m;
        f3 w = bsdf_sample(b, mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), mk2(sample[2 * i], sample[2 * i + 1]), o, e, m);
        wo[3 * i] = o.x; wo[3 * i + 1] = o.y; wo[3 * i + 2] = o.z; weight[3 * i] = w.x; weight[3 * i + 1] = w.y; weight[3 * i + 2] = w.z;
        if (eta) eta[i] = e; if (measure) measure[i] = m;
    }
    return NORI_OK;
}
int emu_bsdf_eval(const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *value) {
    Bsdf b = from_desc(*bsdf);
    for (size_t i = 0; i < n; ++i) {
        f3 v = bsdf_eval(b, mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]));
        value[3 * i] = v.x; value[3 * i + 1] = v.y; value[3 * i + 2] = v.z;
    }
    return NORI_OK;
}
int emu_bsdf_pdf(const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *pdf) {
    Bsdf b = from_desc(*bsdf);
    for (size_t i = 0; i < n; ++i)
        pdf[i] = bsdf_pdf(b, mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]));
    return NORI_OK;
}
int emu_warp(int warp, float param, const float *s, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) { f3 r = warp_dispatch(warp, param, mk2(s[2 * i], s[2 * i + 1])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
    return NORI_OK;
}
int emu_warp_pdf(int warp, float param, const float *p, size_t n, float *pdf) {
    for (size_t i = 0; i < n; ++i) pdf[i] = warp_pdf_dispatch(warp, param, mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
    return NORI_OK;
}
int emu_pcg32_floats(const uint64_t *ss, const uint64_t *sq, size_t n, uint32_t count, float *out) {
    for (size_t i = 0; i < n; ++i) { Rng r; rng_seed(r, ss[i], sq[i]); for (uint32_t j = 0; j < count; ++j) out[i * count + j] = rng_next_float(r); }
    return NORI_OK;
}

/* Mirrors render_kernel: tiles, spp chunks, thread<->pixel, splat to a tile,
 * merge tile into the frame. */
/* Unit check of the packed kernels of rt_trace.h against their scalar statements: n random (triangle pair,
   ray) and (node, ray) cases from `seed`; returns the number of mismatching results (bit level). */
/* sin (0) / cos (1) / log (2) / exp (3) of the device headers (rt_math.h), evaluated on the CPU */
int emu_libm_eval(int op, const float *x, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) {
        float s, c;
        switch (op) {
        case 0: det_sincosf(x[i], &s, &c); out[i] = s; break;
        case 1: det_sincosf(x[i], &s, &c); out[i] = c; break;
        case 2: out[i] = det_logf(x[i]); break;
        case 3: out[i] = det_expf(x[i]); break;
        default: return -1;
        }
    }
    return 0;
}

size_t emu_packed_vs_scalar(size_t n, uint64_t seed) {
    Rng r; rng_seed(r, seed, 7);
    auto rnd = [&](float lo, float hi) { return lo + (hi - lo) * rng_next_float(r); };
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const float scale = i % 3 == 0 ? 100.0f : (i % 3 == 1 ? 1.0f : 1e-2f);
        float p0[2][3], e1[2][3], e2[2][3];
        f4 q[6];
        for (int k = 0; k < 2; ++k) {
            for (int a = 0; a < 3; ++a) { p0[k][a] = rnd(-scale, scale); e1[k][a] = rnd(-scale, scale) * 0.3f; e2[k][a] = rnd(-scale, scale) * 0.3f; }
            if (i % 11 == 5 && k == 1) for (int a = 0; a < 3; ++a) e2[k][a] = 0.5f * e1[k][a];      /* collinear */
            if (i % 13 == 7 && k == 0) for (int a = 0; a < 3; ++a) e1[k][a] = e2[k][a] = p0[k][a] = 0.0f;      /* padding triangle */
            pair_pack(q, k, p0[k], e1[k], e2[k], (uint32_t) (2 * i + k), 0u);
        }
        const f3 o = mk3(rnd(-scale, scale), rnd(-scale, scale), rnd(-scale, scale));
        f3 d = mk3(rnd(-1, 1), rnd(-1, 1), rnd(-1, 1));
        if (i % 7 == 3) d.y = 0.0f;
        d = normalized(d);
        const float mint = 1e-4f * scale, maxt = i % 5 == 0 ? scale : kInf;
        TriPairHit h;
        tri_pair_test(q[0], q[1], q[2], q[3], q[4], o, d, mint, maxt, h);
        for (int k = 0; k < 2; ++k) {
            float u, v, t;
            const bool ok = tri_test(mk3(p0[k][0], p0[k][1], p0[k][2]), mk3(e1[k][0], e1[k][1], e1[k][2]), mk3(e2[k][0], e2[k][1], e2[k][2]), o, d, u, v, t) &&
                            t >= mint && t <= maxt;
            if (ok != h.ok[k]) ++bad;
            else if (ok && (f2u(u) != f2u(h.u[k]) || f2u(v) != f2u(h.v[k]) || f2u(t) != f2u(h.t[k]))) ++bad;
        }
        /* node test: slab_two (centre / half-extent boxes, fused multiply-adds) must be CONSERVATIVE against the exact slab test
           of bbox.h:323-350 evaluated in binary64: whenever the ray meets the box shrunk by a seventh of the padding every leaf
           box carries (kBoxPadRel x scene diagonal ~ 7e-6 x scale here), the device's test must report the box */
        f4 n[4]; float lmn[3], lmx[3], rmn[3], rmx[3];
        for (int a = 0; a < 3; ++a) {
            float x0 = rnd(-scale, scale), x1 = rnd(-scale, scale), y0 = rnd(-scale, scale), y1 = rnd(-scale, scale);
            if (i % 4 == 1) { x1 = x0 + rnd(0.0f, 1e-3f) * scale; y1 = y0 + rnd(0.0f, 2e-5f) * scale; }      /* thin boxes */
            lmn[a] = fminf(x0, x1); lmx[a] = i % 9 == 4 ? lmn[a] : fmaxf(x0, x1); rmn[a] = fminf(y0, y1); rmx[a] = fmaxf(y0, y1);
        }
        node_pack(lmn, lmx, rmn, rmx, 1, 2, n);
        const f3 rcp = mk3(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
        float nl, fl, nr, fr;
        slab_two(n[0], n[1], n[2], o, rcp, nl, fl, nr, fr);
        fl *= 1.0000006f; fr *= 1.0000006f;                          /* as trav_inner_step */
        const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
        const double shrink = 1e-6 * scale;
        for (int side = 0; side < 2; ++side) {
            const float *mn = side ? rmn : lmn, *mx = side ? rmx : lmx;
            double tn = -1e300, tf = 1e300;
            bool miss = false;
            for (int a = 0; a < 3; ++a) {
                const double lo = (double) mn[a] + shrink, hi = (double) mx[a] - shrink;
                if (lo > hi) { miss = true; break; }                 /* thinner than the margin: nothing to demand */
                if (dd[a] == 0.0) { if (oo[a] < lo || oo[a] > hi) miss = true; continue; }
                const double t1 = (lo - oo[a]) / dd[a], t2 = (hi - oo[a]) / dd[a];
                tn = std::max(tn, std::min(t1, t2)); tf = std::min(tf, std::max(t1, t2));
            }
            const bool exact_hit = !miss && tn <= tf && tf >= 0.0 && tn <= (double) maxt;
            const float gn = side ? nr : nl, gf = side ? fr : fl;
            const bool dev_hit = (gn <= gf) && (gf >= 0.0f) && (gn <= maxt);
            if (exact_hit && !dev_hit) ++bad;
        }
    }
    return bad;
}

int emu_render(emu_ctx *c, const nori_render_params *p, float *rgbw, nori_render_stats *stats) {
    const DevScene &sc = c->dev;
    const int W = sc.camera.width, H = sc.camera.height, border = sc.filter.border;
    const int tile_w = kTile + 2 * border, cols = W + 2 * border, rows = H + 2 * border;
    const uint32_t tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile, n_tiles = tiles_x * tiles_y;
    std::vector<uint32_t> sel;
    for (uint32_t t = p->tile_rem; t < n_tiles; t += p->tile_mod) sel.push_back(t);
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::vector<float>> frames(nt, std::vector<float>((size_t) cols * rows * 4, 0.0f));
    std::vector<nori_render_stats> st(nt);
    std::vector<std::thread> th;
    int maxHigh = 0;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
        std::memset(&st[t], 0, sizeof(st[t]));
        std::vector<float> tile((size_t) tile_w * tile_w * 4);
        ArrayStack stack; TraversalCounters tc; tc.nodes = tc.tris = 0;
        uint64_t nClosest = 0, nShadow = 0, nCam = 0, nInvalid = 0;
        for (size_t k = t; k < sel.size(); k += nt) {
            const uint32_t tile_id = sel[k];
            const int x0 = (int) (tile_id % tiles_x) * kTile, y0 = (int) (tile_id / tiles_x) * kTile;
            std::fill(tile.begin(), tile.end(), 0.0f);
            for (int tid = 0; tid < 256; ++tid) {
                const int wave = tid >> 6, lane = tid & 63;
                const int px = x0 + ((wave & 1) << 3) + (lane & 7), py = y0 + ((wave >> 1) << 3) + (lane >> 3);
                if (!(px < W && py < H)) continue;
                for (uint32_t s = p->spp_begin; s < p->spp_begin + p->spp_count; ++s) {
                    PathState ps;
                    rng_seed(ps.rng, (uint64_t) py * (uint64_t) W + (uint64_t) px, (uint64_t) s);
                    const f2 j = rng_next_2d(ps.rng);
                    const f2 pixelSample = mk2((float) px + j.x, (float) py + j.y);
                    (void) rng_next_2d(ps.rng);
                    RayIn cam; camera_sample_ray(sc.camera, pixelSample, cam);
                    path_begin(ps, cam); ++nCam;
                    const f3 L = run_path_dyn(sc, ps, stack, nClosest, nShadow, tc);
                    if (color_valid(L)) splat_tile(tile.data(), tile_w, x0, y0, sc.filter.table, sc.filter.radius, sc.filter.lookup_factor, border, pixelSample, L, PlainAdd());
                    else ++nInvalid;
                }
            }
            float *frame = frames[t].data();
            for (int i = 0; i < tile_w * tile_w; ++i) {
                const int ty = i / tile_w, tx = i - ty * tile_w, gx = x0 + tx, gy = y0 + ty;
                if (gx >= cols || gy >= rows) continue;
                for (int ch = 0; ch < 4; ++ch) frame[((size_t) gy * cols + gx) * 4 + ch] += tile[(size_t) i * 4 + ch];
            }
        }
        st[t].n_camera_samples = nCam; st[t].n_closest_rays = nClosest; st[t].n_shadow_rays = nShadow;
        st[t].n_node_tests = tc.nodes; st[t].n_tri_tests = tc.tris; st[t].n_invalid = nInvalid;
        if (stack.high > maxHigh) maxHigh = stack.high;
    });
    for (auto &t : th) t.join();
    for (unsigned t = 0; t < nt; ++t) for (size_t i = 0; i < frames[t].size(); ++i) rgbw[i] += frames[t][i];
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (unsigned t = 0; t < nt; ++t) {
            stats->n_camera_samples += st[t].n_camera_samples; stats->n_closest_rays += st[t].n_closest_rays;
            stats->n_shadow_rays += st[t].n_shadow_rays; stats->n_node_tests += st[t].n_node_tests;
            stats->n_tri_tests += st[t].n_tri_tests; stats->n_invalid += st[t].n_invalid;
        }
        stats->kernel_ms = (float) maxHigh;   /* emu: max stack depth seen */
    }
    return NORI_OK;
}

} // extern "C"


/* The device group of libnori_hip (group.hip) with CPU ranks: the same shares, the same threads-then-merge driver and the
   same strip lists (group_merge.h), the emulated device code as every rank's renderer.  Checked against ONE render of the
   whole frame (tests/test_distributed_cpu.py). */
extern "C" int emu_group_render(emu_ctx *c, int n_ranks, const nori_render_params *params, int split, int merge, float *rgbw, nori_render_stats *stats) {
    if (!c || !params || !rgbw || n_ranks < 1) return NORI_ERR_INVALID_ARGUMENT;
    const int W = c->dev.camera.width, H = c->dev.camera.height, border = c->dev.filter.border;
    const int rows = H + 2 * border, cols = W + 2 * border;
    const uint32_t tiles_x = (uint32_t) ((W + kTile - 1) / kTile);
    if (merge == kMergeGather && (split != kSplitTile || tiles_x % (uint32_t) n_ranks != 0u)) return NORI_ERR_INVALID_ARGUMENT;
    const size_t frame_floats = (size_t) rows * cols * 4;
    std::vector<std::vector<float>> frames((size_t) n_ranks, std::vector<float>(frame_floats, 0.0f));
    std::vector<nori_render_stats> st((size_t) n_ranks);
    std::vector<int> rc((size_t) n_ranks, NORI_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < n_ranks; ++k) th.emplace_back([&, k] {
        const GroupShare sh = group_share(split, k, n_ranks, params->spp_begin, params->spp_count);
        nori_render_params p = *params;
        p.spp_begin = sh.spp_begin; p.spp_count = sh.spp_count; p.tile_mod = sh.tile_mod; p.tile_rem = sh.tile_rem;
        rc[(size_t) k] = emu_render(c, &p, frames[(size_t) k].data(), &st[(size_t) k]);
    });
    for (auto &t : th) t.join();
    for (int k = 0; k < n_ranks; ++k) if (rc[(size_t) k] != NORI_OK) return rc[(size_t) k];
    std::vector<float> &root = frames[0];
    for (int k = 1; k < n_ranks; ++k) {
        if (merge == kMergeGather) {
            const std::vector<int32_t> x = group_strip_columns(k, n_ranks, tiles_x, border, cols);
            std::vector<float> pack(x.size() * (size_t) rows * 4);
            group_pack_strips(frames[(size_t) k].data(), rows, cols, x, pack.data());      /* "device k" */
            group_add_strips(root.data(), rows, cols, x, pack.data());                     /* "device 0" */
        } else {
            for (size_t i = 0; i < frame_floats; ++i) root[i] += frames[(size_t) k][i];
        }
    }
    std::memcpy(rgbw, root.data(), frame_floats * sizeof(float));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (int k = 0; k < n_ranks; ++k) {
            stats->n_camera_samples += st[(size_t) k].n_camera_samples; stats->n_closest_rays += st[(size_t) k].n_closest_rays;
            stats->n_shadow_rays += st[(size_t) k].n_shadow_rays; stats->n_invalid += st[(size_t) k].n_invalid;
        }
    }
    return NORI_OK;
}


/* film_order = reference over a group: the rows of 32x32 blocks of rank `rank` (group_merge.h) and the contiguous range of 16x16
   tiles they are rendered as (film.h; nori_hip.hip render_impl clips it to the frame the same way): out = row_begin, row_count,
   tile_begin, tile_count */
extern "C" int emu_group_block_rows(int rank, int world, int width, int height, uint32_t out[4]) {
    if (world < 1 || rank < 0 || rank >= world || !out) return NORI_ERR_INVALID_ARGUMENT;
    const uint32_t tiles_x = (uint32_t) ((width + kTile - 1) / kTile), tiles_y = (uint32_t) ((height + kTile - 1) / kTile), n_tiles = tiles_x * tiles_y;
    const uint32_t byn = (uint32_t) ((height + 31) / 32);
    const GroupRows r = group_block_rows(rank, world, byn);
    const uint32_t t0 = std::min(film_block_rows_first_tile(r.row_begin, tiles_x), n_tiles), t1 = std::min(film_block_rows_first_tile(r.row_begin + r.row_count, tiles_x), n_tiles);
    out[0] = r.row_begin; out[1] = r.row_count; out[2] = t0; out[3] = t1 - t0;
    return NORI_OK;
}
