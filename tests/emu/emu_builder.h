/*
 * emu_builder.h -- TEST-ONLY: the device BVH builders (nori_amd/csrc/device/lbvh.hip) run on the CPU.
 *
 * lbvh.hip's kernels are one thread per element over the functions of lbvh_steps.h; here the same functions run in
 * loops, in the order of build_bvh_lbvh_device, with std::sort / serial prefix sums where the device uses hipcub.
 * The result is a HostBvh the emulation harness traverses like any other, so `pytest -m "not gpu"` checks the
 * builders' LOGIC (radix tree, PLOC, leaf collapse, pair / node / wide-node emission) against the oracle's
 * brute force and reports what the trees cost per ray before GPU minutes are spent.
 */
#pragma once
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>

#include <cstdio>
#include "../../nori_amd/csrc/device/lbvh.h"
#include "../../nori_amd/csrc/device/lbvh_steps.h"
#include "../../nori_amd/csrc/device/scene_prep.h"

namespace nrt {

struct EmuBuilderStats { uint32_t ploc_iterations = 0, reinserted = 0, n_refs = 0; float sah_cost = 0.0f; };

inline std::string build_bvh_steps_host(const HostScene &scene, bool wide, uint32_t ploc_radius, HostBvh &out, EmuBuilderStats *stats = nullptr) {
    out = HostBvh();
    uint32_t n = (uint32_t) scene.tri_mesh.size();
    if (n == 0) return "empty scene";
    if (n <= 4) wide = false;
    const uint32_t pair_base = wide ? 1u : 0u;
    const f4 *pos = scene.positions.data();
    const uint32_t *idx = scene.indices.data();

    /* 1. scene bounds */
    f3 smin = mk3(kInf), smax = mk3(-kInf);
    for (uint32_t t = 0; t < n; ++t) {
        f3 mn, mx; tri_box(pos, idx, t, mn, mx);
        smin = mk3(fminf(smin.x, mn.x), fminf(smin.y, mn.y), fminf(smin.z, mn.z));
        smax = mk3(fmaxf(smax.x, mx.x), fmaxf(smax.y, mx.y), fmaxf(smax.z, mx.z));
    }
    const float ex = smax.x - smin.x, ey = smax.y - smin.y, ez = smax.z - smin.z;
    const float pad = box_pad_rel() * sqrtf(ex * ex + ey * ey + ez * ez) + 1e-30f;
    const f3 sinv = mk3(ex > 0 ? 1.0f / ex : 0.0f, ey > 0 ? 1.0f / ey : 0.0f, ez > 0 ? 1.0f / ez : 0.0f);

    /* 2. references (lbvh_steps.h): every triangle once, or -- with a budget -- the triangles with the emptiest boxes as several parts */
    const uint32_t n_tris = n;
    const SplitTuning stn = split_tuning(n_tris);
    std::vector<uint32_t> cuts(n_tris, 0u), ref_first(n_tris, 0u);
    if (stn.budget > 0.0f && ploc_radius != 0u) {      /* (as build_bvh_lbvh_device: not in front of the radix tree) */
        std::vector<float> prio(n_tris);
        for (uint32_t t = 0; t < n_tris; ++t) prio[t] = split_priority(pos, idx, t, smin, sinv);
        /* what else lies in every triangle's box (lbvh_steps.h): centres per grid cell, summed-volume table, a cap on the triangle's cuts */
        std::vector<uint32_t> limit(n_tris, stn.cap);
        if (stn.inside > 0u) {
            const int G = kSplitGrid, G1 = kSplitGrid + 1;
            std::vector<uint32_t> cells((size_t) G * G * G, 0u), sat((size_t) G1 * G1 * G1, 0u);
            for (uint32_t t = 0; t < n_tris; ++t) { f3 mn, mx; tri_box(pos, idx, t, mn, mx); if (!tri_unbounded(pos, idx, t)) cells[split_grid_cell(mn, mx, smin, sinv)]++; }
            split_sat(cells.data(), sat.data());
            for (uint32_t t = 0; t < n_tris; ++t) {
                if (!(prio[t] > 0.0f)) continue;
                f3 mn, mx; tri_box(pos, idx, t, mn, mx);
                limit[t] = std::min(stn.cap, split_inside(sat.data(), mn, mx, smin, sinv) / stn.inside);
            }
        }
        const unsigned long long want = (unsigned long long) ((double) stn.budget * (double) n_tris);
        auto total = [&](float D) { unsigned long long sum = 0; for (uint32_t t = 0; t < n_tris; ++t) sum += split_count(prio[t], D, limit[t]); return sum; };
        unsigned long long sum_bits = 0, have = 0;
        for (uint32_t t = 0; t < n_tris; ++t) if (prio[t] > 0.0f) { sum_bits += f2u(prio[t]); ++have; }
        const float D = have ? split_choose_D(total, want, split_scale_D(sum_bits, have, stn.scale)) : 0.0f;
        for (uint32_t t = 0; t < n_tris; ++t) cuts[t] = split_count(prio[t], D, limit[t]);
    }
    uint32_t m = 0;
    const RefOut none{nullptr, nullptr, nullptr, nullptr};
    for (uint32_t t = 0; t < n_tris; ++t) { ref_first[t] = m; m += split_emit(pos, idx, t, cuts[t], pad, smin, sinv, false, none, 0u); }
    std::vector<uint32_t> ref_tri(m); std::vector<f4> ref_mn(m), ref_mx(m); std::vector<unsigned long long> key(m);
    const RefOut refs{ref_tri.data(), ref_mn.data(), ref_mx.data(), key.data()};
    for (uint32_t t = 0; t < n_tris; ++t) (void) split_emit(pos, idx, t, cuts[t], pad, smin, sinv, true, refs, ref_first[t]);
    if (stats) stats->n_refs = m;
    n = m;      /* from here on: references */

    /* 3. sorted by Morton key (stable in the reference index, like the radix sort); order[k] = triangle of the reference at position k,
       lmn / lmx[k] = its padded box */
    std::vector<unsigned long long> keys(n);
    std::vector<uint32_t> order(n);
    std::vector<f4> lmn(n), lmx(n);
    {
        std::vector<uint32_t> by_key(n);
        std::iota(by_key.begin(), by_key.end(), 0u);
        std::stable_sort(by_key.begin(), by_key.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        for (uint32_t k = 0; k < n; ++k) { keys[k] = key[by_key[k]]; order[k] = ref_tri[by_key[k]]; lmn[k] = ref_mn[by_key[k]]; lmx[k] = ref_mx[by_key[k]]; }
    }

    std::vector<uint32_t> leaf_cnt(n, 0u), leaf_pairs(n, 0u), pair_start(n, 0u);
    std::vector<RadixNode> rnodes;
    std::vector<uint32_t> pin(n, 0u), plf(n, 0u), keep(n, 0u), node_index(n, 0u), collapse(n, 0u);
    std::vector<f4> tmin, tmax;
    uint32_t N = 1;
    if (n <= 4) {
        leaf_cnt[0] = n; leaf_pairs[0] = (n + 1) / 2;
    } else {
        rnodes.resize(n - 1);
        if (ploc_radius == 0u) {
            for (uint32_t i = 0; i + 1 < n; ++i) rnodes[i] = radix_node(keys.data(), (int) n, (int) i, pin.data(), plf.data());
        } else {
            std::vector<f4> amn(n), amx(n), bmn(n), bmx(n);
            std::vector<uint32_t> nearest(n), nl(n), nr(n), nc(n), npn(n), npp(n), leaf_pos(n), order2(n);
            PlocClusters ca{amn.data(), amx.data()}, cb{bmn.data(), bmx.data()};
            PlocNodes pn{nl.data(), nr.data(), nc.data(), npn.data(), npp.data()};
            for (uint32_t k = 0; k < n; ++k) {
                f4 mn = lmn[k], mx = lmx[k];
                mn.w = u2f(kLeafBit | k); mx.w = u2f(1u);
                ca.mn[k] = mn; ca.mx[k] = mx;
            }
            uint32_t m = n, node_base = 0u, iterations = 0u;
            std::vector<uint32_t> lead(n), stays(n), lead_rank(n), stays_rank(n);
            while (m > 1u) {
                for (uint32_t i = 0; i < m; ++i) nearest[i] = ploc_nearest(ca, m, i, ploc_radius);
                uint32_t nl_ = 0u, ns_ = 0u;
                for (uint32_t i = 0; i < m; ++i) {
                    ploc_decide(nearest.data(), i, lead[i], stays[i]);
                    lead_rank[i] = nl_; stays_rank[i] = ns_; nl_ += lead[i]; ns_ += stays[i];
                }
                for (uint32_t i = 0; i < m; ++i) ploc_apply(ca, cb, pn, nearest.data(), i, lead[i], stays[i], lead_rank[i], stays_rank[i], node_base);
                if (nl_ == 0u || ns_ + nl_ != m) return "a PLOC iteration merged nothing";
                node_base += nl_; m = ns_;
                std::swap(ca, cb);
                ++iterations;
            }
            if (node_base != n - 1u) return "PLOC node count";
            if (stats) stats->ploc_iterations = iterations;
            {   /* treelet restructuring (lbvh_steps.h), as build_bvh_lbvh_device runs it */
                const BuildTuning rp = build_tuning(n);
                const int sweeps = rp.sweeps;
                std::vector<f4> nmn(n), nmx(n); std::vector<float> ncost(n);
                TreeletData td{nmn.data(), nmx.data(), ncost.data(), lmn.data(), lmx.data()};
                TreeletParams tp; tp.c_node = 1.0f; tp.c_tri = 1.0f;
                std::vector<uint32_t> visits(n);
                auto sweep = [&] {
                    std::fill(visits.begin(), visits.end(), 0u);
                    for (uint32_t k = 0; k < n; ++k) treelet_climb(pn, td, tp, visits.data(), n - 2u, k);
                };
                for (int sw = 0; sw < sweeps; ++sw) sweep();
                /* parallel re-insertion (lbvh_steps.h), as build_bvh_lbvh_device runs it: every iteration the candidates of one
                   residue class search the same tree, mark, check, the winners move, the tree is refitted */
                if (rp.iterations > 0) {
                    const uint32_t n_inner = n - 1u, n_slots = 2u * n - 1u;
                    std::vector<unsigned long long> lock(n_slots), rkey(n_slots);
                    std::vector<uint32_t> target(n_slots), pivot(n_slots), win(n_slots);
                    ReinsData rd{lock.data(), rkey.data(), target.data(), pivot.data(), win.data()};
                    auto refit = [&] {
                        std::fill(visits.begin(), visits.end(), 0u);
                        for (uint32_t k = 0; k < n; ++k) reins_refit_climb(pn, td, rd, tp, visits.data(), n - 2u, k);
                    };
                    refit();
                    uint32_t moved_total = 0;
                    for (int it = 0; it < rp.iterations; ++it) {
                        const uint32_t phase = (uint32_t) it % rp.stride;
                        std::fill(lock.begin(), lock.end(), 0ull);
                        std::fill(rkey.begin(), rkey.end(), 0ull);
                        for (uint32_t sl = phase; sl < n_slots; sl += rp.stride) reins_search(pn, td, rd, n_inner, sl);
                        for (uint32_t sl = phase; sl < n_slots; sl += rp.stride) (void) reins_locks(pn, rd, n_inner, sl, true);
                        for (uint32_t sl = phase; sl < n_slots; sl += rp.stride) win[sl] = reins_locks(pn, rd, n_inner, sl, false) ? 1u : 0u;
                        uint32_t moved = 0;
                        for (uint32_t sl = phase; sl < n_slots; sl += rp.stride) if (win[sl]) { reins_apply(pn, rd, n_inner, sl); ++moved; }
                        refit();
                        moved_total += moved;
                        if (std::getenv("NORI_HIP_BUILD_TIMING")) fprintf(stderr, "[reinsert] iteration %d: %u moved, root cost %.6g\n", it, moved, (double) ncost[n - 2u]);
                    }
                    if (stats) stats->reinserted = moved_total;
                    for (int sw = 0; sw < rp.sweeps_after; ++sw) sweep();
                }
            }
            std::vector<f4> lmn2(n), lmx2(n);
            for (uint32_t k = 0; k < n; ++k) { leaf_pos[k] = ploc_first_position(pn, n - 1u, kLeafBit | k); order2[leaf_pos[k]] = order[k]; lmn2[leaf_pos[k]] = lmn[k]; lmx2[leaf_pos[k]] = lmx[k]; }
            for (uint32_t id = 0; id + 1 < n; ++id) rnodes[n - 2u - id] = ploc_finish(pn, n - 1u, id, leaf_pos.data(), pin.data(), plf.data());
            order.swap(order2); lmn.swap(lmn2); lmx.swap(lmx2);
        }
        /* 5. segment tree of boxes */
        while (N < n) N <<= 1;
        tmin.resize((size_t) 2 * N); tmax.resize((size_t) 2 * N);
        for (uint32_t k = 0; k < N; ++k) {
            f4 mn4, mx4;
            if (k < n) { mn4 = lmn[k]; mx4 = lmx[k]; mn4.w = mx4.w = 0.0f; }
            else { mn4.x = mn4.y = mn4.z = kInf; mx4.x = mx4.y = mx4.z = -kInf; mn4.w = mx4.w = 0.0f; }
            tmin[N + k] = mn4; tmax[N + k] = mx4;
        }
        for (uint32_t i = N - 1; i >= 1; --i) seg_tree_combine(i, tmin.data(), tmax.data());
        /* 6. leaves */
        CollapseParams cp; cp.c_pair = 1.5f; cp.c_node = 1.0f;
        for (uint32_t i = 0; i + 1 < n; ++i) collapse[i] = collapse_decide(rnodes.data(), i, tmin.data(), tmax.data(), N, cp);
        for (uint32_t i = 0; i + 1 < n; ++i) keep[i] = mark_leaves(rnodes.data(), i, collapse.data(), pin.data(), leaf_cnt.data(), leaf_pairs.data());
        if (stats) {      /* cost of the tree as built: node areas + leaf areas x pairs, relative to the root's area */
            f3 a, b; range_box(tmin.data(), tmax.data(), N, 0, n - 1, a, b);
            double root_area = 0.0, cost = 0.0;
            /* unbounded triangles (infinite boxes) would make every number infinite: measure the bounded part */
            for (uint32_t i = 0; i + 1 < n; ++i) {
                if (!keep[i]) continue;
                range_box(tmin.data(), tmax.data(), N, rnodes[i].lo, rnodes[i].hi, a, b);
                const double area = box_area(a, b);
                if (!(area < 1e30)) continue;
                if (area > root_area) root_area = area;
                cost += area * cp.c_node;
                for (uint32_t child : {rnodes[i].left, rnodes[i].right}) {
                    uint32_t lo, hi;
                    if (!child_range(rnodes.data(), collapse.data(), child, lo, hi)) continue;
                    range_box(tmin.data(), tmax.data(), N, lo, hi, a, b);
                    cost += (double) box_area(a, b) * ((hi - lo + 2) / 2) * cp.c_pair;
                }
            }
            stats->sah_cost = root_area > 0.0 ? (float) (cost / root_area) : 0.0f;
        }
    }
    for (uint32_t k = 1; k < n; ++k) pair_start[k] = pair_start[k - 1] + leaf_pairs[k - 1];
    const uint32_t n_pairs = pair_start[n - 1] + leaf_pairs[n - 1] + pair_base;
    out.n_pairs = n_pairs;
    out.tris.assign((size_t) std::max<uint32_t>(n_pairs, 1) * kPairQuads, f4{0.0f, 0.0f, 0.0f, 0.0f});
    for (uint32_t k = 0; k < n; ++k)
        if (leaf_cnt[k]) emit_leaf_pairs(pos, idx, scene.tri_mesh.data(), order.data(), k, leaf_cnt[k], out.tris.data() + (size_t) (pair_start[k] + pair_base) * kPairQuads);

    if (n <= 4) {
        out.nodes.assign(kNodeQuads, f4{0.0f, 0.0f, 0.0f, 0.0f});
        out.root = (int32_t) ~((0u << 3) | ((n + 1u) / 2u - 1u));
        out.n_nodes = 0; out.n_leaves = 1; out.max_depth = 0;
    } else if (wide) {
        std::vector<uint32_t> kids((size_t) n * 4, 0u), n_kids(n, 0u), is_wide(n, 0u), wide_index(n, 0u), frontier{0u}, next;
        is_wide[0] = 1u;
        uint32_t levels = 0;
        while (!frontier.empty()) {
            next.clear();
            for (uint32_t i : frontier) {
                uint32_t kid[4];
                const int nk = wide_children(rnodes.data(), collapse.data(), tmin.data(), tmax.data(), N, i, kid);
                n_kids[i] = (uint32_t) nk;
                for (int k = 0; k < nk; ++k) {
                    kids[4 * (size_t) i + k] = kid[k];
                    uint32_t lo, hi;
                    if (!child_range(rnodes.data(), collapse.data(), kid[k], lo, hi)) { is_wide[kid[k]] = 1u; next.push_back(kid[k]); }
                }
            }
            frontier.swap(next);
            if (++levels > 4096) return "wide levels did not terminate";
        }
        uint32_t n_wide = 0;
        for (uint32_t i = 0; i + 1 < n; ++i) { wide_index[i] = n_wide; n_wide += is_wide[i]; }
        out.nodes.assign((size_t) std::max<uint32_t>(n_wide, 1) * kNodeQuads, f4{0.0f, 0.0f, 0.0f, 0.0f});
        for (uint32_t i = 0; i + 1 < n; ++i)
            if (is_wide[i]) emit_wide_node(rnodes.data(), tmin.data(), tmax.data(), N, collapse.data(), pair_start.data(), wide_index.data(),
                                           kids.data() + 4 * (size_t) i, (int) n_kids[i], out.nodes.data() + (size_t) wide_index[i] * kNodeQuads);
        out.root = 0; out.n_nodes = n_wide; out.n_leaves = 0; out.max_depth = 3 * levels; out.wide = true;
    } else {
        uint32_t n_nodes = 0;
        for (uint32_t i = 0; i + 1 < n; ++i) { node_index[i] = n_nodes; n_nodes += keep[i]; }
        out.nodes.assign((size_t) std::max<uint32_t>(n_nodes, 1) * kNodeQuads, f4{0.0f, 0.0f, 0.0f, 0.0f});
        for (uint32_t i = 0; i + 1 < n; ++i)
            if (keep[i]) emit_node(rnodes.data(), i, tmin.data(), tmax.data(), N, collapse.data(), pair_start.data(), node_index.data(),
                                   out.nodes.data() + (size_t) node_index[i] * kNodeQuads);
        uint32_t depth = 0;
        for (uint32_t k = 0; k < n; ++k) depth = std::max(depth, leaf_depth(pin.data(), plf.data(), keep.data(), k));
        out.root = 0; out.n_nodes = n_nodes; out.n_leaves = 0; out.max_depth = depth;
    }
    if (stats) out.sah_cost = stats->sah_cost;
    return std::string();
}

} // namespace nrt
