/*
 * rt_top.h -- the part of the acceleration structure that wf_extend keeps in LDS.
 *
 * Every ray starts at the root, and in the scenes of this renderer a handful of records carry most of the traffic:
 * in the pa4 Cornell box the 8 most visited nodes take 63 % of all node visits and SIX leaf pair records -- the walls,
 * whose leaves hang off the first levels of the tree -- take 76 % of all triangle steps (tools/trav_histogram.py).
 * A per-lane 64-B node fetch is four 16-B accesses of the CU's vector L1 (one access per clock), a leaf step five:
 * served from LDS they cost neither an L1 access nor an L2 round trip.
 *
 * The image is built ONCE per acceleration structure (top_image_build: one thread, on the device for any builder and
 * either node layout; the CPU harness runs the same function) and every wf_extend workgroup copies it into its LDS:
 *
 *   quad 0                       header: (bits root link, bits cached nodes, bits cached pairs, 0)
 *   kTopNodes x kTopStrideQuads  node records as in memory (rt_types.h), 80 B apart (consecutive slots start in
 *                                different banks); links to cached nodes rewritten to  kTopBit | slot,  links to cached
 *                                leaves to the leaf code of  first_pair = kTopPairBase + slot  (rt_trace.h)
 *   kTopPairs x kPairQuads       leaf pair records as in memory
 *
 * Which records: greedy by the surface area of the (child) box -- the probability that a ray crossing the parent's box
 * visits the child -- starting from the root: the inner node of largest area among the children of the nodes chosen so
 * far, kTopNodes times; then the leaves hanging off the chosen nodes, largest area first, while their pairs fit.
 */
#pragma once
#include "rt_types.h"

namespace nrt {

constexpr int kTopNodes = 24;
constexpr int kTopStrideQuads = 5;
constexpr int kTopPairs = 8;
constexpr int kTopBit = 0x40000000;          /* node indices stay below 2^30 */
/* a cached leaf keeps the form ~link = (first_pair << 3) | (n_pairs - 1) with first_pair = kTopPairBase + slot: bit 30 of
   the cursor tells the leaf step where the record lives; pair indices of real records stay below 2^27 (checked) */
constexpr uint32_t kTopPairBase = (uint32_t) kTopBit >> 3;
constexpr int kTopImageQuads = 1 + kTopNodes * kTopStrideQuads + kTopPairs * kPairQuads;

/* surface area (up to a constant factor) of child k's box of a node record, either layout; unbounded boxes: infinity */
NORI_HD float top_child_area(const f4 q[4], int k, bool wide) {
    float ex, ey, ez;
    if (!wide) {
        ex = k == 0 ? q[2].x : q[2].z; ey = k == 0 ? q[2].y : q[2].w; ez = k == 0 ? q[1].z : q[1].w;      /* half extents */
    } else {
        const uint32_t meta = f2u(q[0].w);
        if (meta & kWideAllHit) return kInf;
        const uint32_t lo[3] = {f2u(q[1].x), f2u(q[1].y), f2u(q[1].z)}, hi[3] = {f2u(q[1].w), f2u(q[2].x), f2u(q[2].y)};
        float e[3];
        for (int a = 0; a < 3; ++a) {
            const int l = (int) ((lo[a] >> (8 * k)) & 255u), h = (int) ((hi[a] >> (8 * k)) & 255u);
            e[a] = h >= l ? ldexpf((float) (h - l), (int) ((meta >> (8 * a)) & 255u) - 128) : 0.0f;
        }
        ex = e[0]; ey = e[1]; ez = e[2];
    }
    if (!(ex < 1e30f) || !(ey < 1e30f) || !(ez < 1e30f)) return kInf;
    return ex * ey + (ey * ez + ez * ex);
}

/* builds the image (kTopImageQuads quads) of a tree; `nodes` / `tris` as in DevScene */
NORI_HD void top_image_build(const f4 *nodes, const f4 *tris, int32_t root, bool wide, uint32_t n_triangles, f4 *image) {
    const int n_links = wide ? 4 : 2;
    for (int i = 0; i < kTopImageQuads; ++i) { image[i].x = image[i].y = image[i].z = image[i].w = 0.0f; }
    image[0].x = u2f((uint32_t) root);
    if (n_triangles == 0u || root < 0) return;                /* empty scene, or the root is a leaf: nothing cached */
    const bool pairs_ok = n_triangles < (1u << 27);           /* pair indices < 2^27: bit 30 of a leaf cursor is free */
    f4 *inodes = image + 1, *ipairs = image + 1 + kTopNodes * kTopStrideQuads;
    /* candidates: children of chosen nodes -- (link, area, slot of the parent, which link of the parent) */
    constexpr int kMaxCand = kTopNodes * 4 + 1;
    int32_t c_link[kMaxCand]; float c_area[kMaxCand]; int c_parent[kMaxCand], c_which[kMaxCand]; bool c_used[kMaxCand];
    int n_cand = 0, n_nodes = 0, n_pairs = 0;
    c_link[0] = root; c_area[0] = kInf; c_parent[0] = -1; c_which[0] = 0; c_used[0] = false; n_cand = 1;
    float *link_of[4];
    while (n_nodes < kTopNodes) {
        int best = -1;
        for (int i = 0; i < n_cand; ++i)
            if (!c_used[i] && c_link[i] >= 0 && (best < 0 || c_area[i] > c_area[best])) best = i;
        if (best < 0) break;
        c_used[best] = true;
        const int slot = n_nodes++;
        f4 *dst = inodes + slot * kTopStrideQuads;
        const f4 *src = nodes + (size_t) c_link[best] * kNodeQuads;
        for (int q = 0; q < kNodeQuads; ++q) dst[q] = src[q];
        if (c_parent[best] >= 0) {
            f4 &pl = inodes[c_parent[best] * kTopStrideQuads + 3];
            link_of[0] = &pl.x; link_of[1] = &pl.y; link_of[2] = &pl.z; link_of[3] = &pl.w;
            *link_of[c_which[best]] = u2f((uint32_t) (kTopBit | slot));
        }
        const float *links = &dst[3].x;
        for (int k = 0; k < n_links; ++k) {
            const int32_t lk = (int32_t) f2u(links[k]);
            if (wide && lk == kWideEmpty) continue;
            c_link[n_cand] = lk; c_area[n_cand] = top_child_area(dst, k, wide); c_parent[n_cand] = slot; c_which[n_cand] = k; c_used[n_cand] = false;
            ++n_cand;
        }
    }
    image[0].x = u2f((uint32_t) (kTopBit | 0));
    while (pairs_ok) {
        int best = -1;
        for (int i = 0; i < n_cand; ++i) {
            if (c_used[i] || c_link[i] >= 0) continue;
            const uint32_t cur = ~(uint32_t) c_link[i];
            if ((int) (cur & 7u) + 1 > kTopPairs - n_pairs) continue;          /* does not fit any more */
            if (best < 0 || c_area[i] > c_area[best]) best = i;
        }
        if (best < 0) break;
        c_used[best] = true;
        const uint32_t cur = ~(uint32_t) c_link[best], first = cur >> 3, cnt = (cur & 7u) + 1u;
        for (uint32_t q = 0; q < cnt * kPairQuads; ++q) ipairs[(size_t) n_pairs * kPairQuads + q] = tris[(size_t) first * kPairQuads + q];
        f4 &pl = inodes[c_parent[best] * kTopStrideQuads + 3];
        link_of[0] = &pl.x; link_of[1] = &pl.y; link_of[2] = &pl.z; link_of[3] = &pl.w;
        *link_of[c_which[best]] = u2f(~(((kTopPairBase + (uint32_t) n_pairs) << 3) | (cnt - 1u)));
        n_pairs += (int) cnt;
    }
    image[0].y = u2f((uint32_t) n_nodes); image[0].z = u2f((uint32_t) n_pairs);
}

} // namespace nrt
