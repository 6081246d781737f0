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
 *   quad 0                       header: (bits root link, bits cached nodes, bits cached pairs, bits quads of the image)
 *   n_nodes x kTopStrideQuads    node records as in memory (rt_types.h), 80 B apart (consecutive slots start in
 *                                different banks); links to cached nodes rewritten to  kTopBit | quad offset of the record
 *                                in the image (the fetch needs no multiply),  links to cached leaves to the leaf code of
 *                                first_pair = kTopPairBase + index, the record at quad  index x kPairQuads  of the image
 *   (<= 5 quads of padding)      so that the pair records start at a multiple of kPairQuads
 *   n_pairs x kPairQuads         leaf pair records as in memory
 *
 * Two forms: with the 64-B node records (n_nodes x kTopStrideQuads, as drawn) for the kernels that walk those, and with the
 * 32-B records of rt_nodeq.h (n_nodes x kTopStrideQuadsQ, links in the record's last two dwords) for wf_extend's hand-written
 * node loop.  How many nodes: what the LDS of a CU leaves to a workgroup of wf_extend next to its traversal stacks
 * (wavefront.hip, wf_top_capacity) -- BVH2 trees (workgroups of 1024 threads, two per CU): 359 records of 32 B (230 if the tree
 * is deeper than the LDS stack) or 143 of 64 B; wide nodes (workgroups of 256, six per CU): 113.
 *
 * Which records: greedy by the surface area of the (child) box -- the probability that a ray crossing the parent's box
 * visits the child -- starting from the root: the inner node of largest area among the children of the nodes chosen so
 * far, as many times as nodes fit; then the leaves hanging off the chosen nodes, largest area first, while their pairs fit.
 */
#pragma once
#include "rt_types.h"

namespace nrt {

constexpr int kTopMaxNodes = 384;           /* capacity of the buffers; how many are used is the caller's LDS budget */
constexpr int kTopStrideQuads = 5;          /* 64-B node records */
constexpr int kTopStrideQuadsQ = 2;         /* 32-B node records (rt_nodeq.h) */
/* how the node records of an image look: quads per record, quads from one slot to the next, where the links are */
struct TopLayout { int rec_quads, stride_quads, link_quad, link_first; };
NORI_HD constexpr TopLayout top_layout(bool q) { return q ? TopLayout{2, kTopStrideQuadsQ, 1, 2} : TopLayout{4, kTopStrideQuads, 3, 0}; }
constexpr int kTopPairs = 8;
constexpr int kTopBit = 0x40000000;          /* node indices stay below 2^30 */
/* a cached leaf keeps the form ~link = (first_pair << 3) | (n_pairs - 1) with first_pair = kTopPairBase + slot: bit 30 of
   the cursor tells the leaf step where the record lives; pair indices of real records stay below 2^27 (checked) */
constexpr uint32_t kTopPairBase = (uint32_t) kTopBit >> 3;
constexpr int kTopQuadMask = 0xffff;         /* a cached node's link carries the quad offset of its record in these bits */
constexpr int kTopPairMask = 0xfff;          /* a cached leaf's cursor: index of its first pair record, in units of kPairQuads */
/* quads of an image of `n` nodes and all kTopPairs pair records */
NORI_HD constexpr int top_image_quads(int n, int stride_quads = kTopStrideQuads) { return (1 + n * stride_quads + kPairQuads - 1) / kPairQuads * kPairQuads + kTopPairs * kPairQuads; }
constexpr int kTopImageMaxQuads = top_image_quads(kTopMaxNodes);
static_assert(kTopImageMaxQuads <= kTopQuadMask + 1 && kTopImageMaxQuads / kPairQuads <= kTopPairMask + 1, "offsets must fit the links' low bits");

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

/* argmax over candidates 0 .. n-1 of a score (negative: not eligible), lowest index among equals; -1 if none.  The device
   kernel passes a wave-wide version (nori_hip.hip): the selection below runs in all 64 lanes at once, identically. */
struct TopSerialArgMax {
    template <class F> NORI_HD int operator()(int n, F score) const {
        int best = -1; float bs = -1.0f;
        for (int i = 0; i < n; ++i) { const float sc = score(i); if (sc > bs) { bs = sc; best = i; } }
        return best;
    }
};
constexpr int kTopMaxCand = kTopMaxNodes * 4 + 1;
/* work arrays of the selection: kTopMaxCand entries each (LDS on the device) */
struct TopWork { int32_t *link; float *area; int16_t *parent; int8_t *which; };

/* builds the image of a tree with at most `max_nodes` (<= kTopMaxNodes) node records into `image` (kTopImageMaxQuads quads);
   `nodes` / `tris` as in DevScene.  Candidates: the children of the nodes chosen so far -- (link, area of the box, slot of the
   parent, which link of the parent); a taken candidate's area becomes -1. */
/* `records` / `layout`: what is copied into the image -- the 64-B nodes themselves (records = nodes, top_layout(false)) or their
   32-B form (rt_nodeq.h: records = nodes_q, top_layout(true)); the selection always reads the 64-B nodes. */
template <class ArgMax>
NORI_HD void top_image_build_with(const f4 *nodes, const f4 *records, TopLayout layout, const f4 *tris, int32_t root, bool wide, uint32_t n_triangles,
                                  int max_nodes, f4 *image, TopWork w, ArgMax argmax) {
    const int n_links = wide ? 4 : 2;
    const int kStride = layout.stride_quads;
    if (max_nodes > kTopMaxNodes) max_nodes = kTopMaxNodes;
    image[0].x = u2f((uint32_t) root); image[0].y = image[0].z = u2f(0u); image[0].w = u2f(1u);
    if (n_triangles == 0u || root < 0 || max_nodes < 1) return;      /* empty scene, or the root is a leaf: nothing cached */
    const bool pairs_ok = n_triangles < (1u << 27);           /* pair indices < 2^27: bit 30 of a leaf cursor is free */
    f4 *inodes = image + 1;
    int n_cand = 0, n_nodes = 0, n_pairs = 0;
    w.link[0] = root; w.area[0] = kInf; w.parent[0] = -1; w.which[0] = 0; n_cand = 1;
    float *link_of[4];
    while (n_nodes < max_nodes) {
        const int best = argmax(n_cand, [&](int i) { return w.link[i] >= 0 ? w.area[i] : -1.0f; });
        if (best < 0) break;
        w.area[best] = -1.0f;
        const int slot = n_nodes++;
        f4 *dst = inodes + slot * kStride;
        const f4 *node = nodes + (size_t) w.link[best] * kNodeQuads;
        const f4 *src = records + (size_t) w.link[best] * layout.rec_quads;
        for (int q = 0; q < layout.rec_quads; ++q) dst[q] = src[q];
        for (int q = layout.rec_quads; q < kStride; ++q) dst[q].x = dst[q].y = dst[q].z = dst[q].w = 0.0f;      /* the slot's padding */
        if (w.parent[best] >= 0) {
            f4 &pl = inodes[w.parent[best] * kStride + layout.link_quad];
            link_of[0] = &pl.x; link_of[1] = &pl.y; link_of[2] = &pl.z; link_of[3] = &pl.w;
            *link_of[layout.link_first + w.which[best]] = u2f((uint32_t) (kTopBit | (1 + slot * kStride)));
        }
        const float *links = &node[3].x;
        for (int k = 0; k < n_links; ++k) {
            const int32_t lk = (int32_t) f2u(links[k]);
            if (wide && lk == kWideEmpty) continue;
            float ar = top_child_area(node, k, wide);
            if (!(ar >= 0.0f)) ar = 0.0f;
            w.link[n_cand] = lk; w.area[n_cand] = ar; w.parent[n_cand] = (int16_t) slot; w.which[n_cand] = (int8_t) k;
            ++n_cand;
        }
    }
    image[0].x = u2f((uint32_t) (kTopBit | 1));
    const int pair_base = (1 + n_nodes * kStride + kPairQuads - 1) / kPairQuads;      /* in pair records */
    for (int q = 1 + n_nodes * kStride; q < pair_base * kPairQuads; ++q) image[q].x = image[q].y = image[q].z = image[q].w = 0.0f;
    f4 *ipairs = image + (size_t) pair_base * kPairQuads;
    while (pairs_ok) {
        const int room = kTopPairs - n_pairs;
        const int best = argmax(n_cand, [&](int i) {
            if (w.link[i] >= 0) return -1.0f;
            return (int) (~(uint32_t) w.link[i] & 7u) + 1 > room ? -1.0f : w.area[i];          /* a leaf that does not fit any more */
        });
        if (best < 0) break;
        w.area[best] = -1.0f;
        const uint32_t cur = ~(uint32_t) w.link[best], first = cur >> 3, cnt = (cur & 7u) + 1u;
        for (uint32_t q = 0; q < cnt * kPairQuads; ++q) ipairs[(size_t) n_pairs * kPairQuads + q] = tris[(size_t) first * kPairQuads + q];
        f4 &pl = inodes[w.parent[best] * kStride + layout.link_quad];
        link_of[0] = &pl.x; link_of[1] = &pl.y; link_of[2] = &pl.z; link_of[3] = &pl.w;
        *link_of[layout.link_first + w.which[best]] = u2f(~(((kTopPairBase + (uint32_t) (pair_base + n_pairs)) << 3) | (cnt - 1u)));
        n_pairs += (int) cnt;
    }
    image[0].y = u2f((uint32_t) n_nodes); image[0].z = u2f((uint32_t) n_pairs);
    image[0].w = u2f((uint32_t) ((pair_base + n_pairs) * kPairQuads));      /* quads in use */
}

#if !defined(__HIP_DEVICE_COMPILE__)
/* the serial form (CPU harness) */
inline void top_image_build(const f4 *nodes, const f4 *records, TopLayout layout, const f4 *tris, int32_t root, bool wide, uint32_t n_triangles,
                            int max_nodes, f4 *image) {
    static thread_local int32_t link[kTopMaxCand]; static thread_local float area[kTopMaxCand];
    static thread_local int16_t parent[kTopMaxCand]; static thread_local int8_t which[kTopMaxCand];
    TopWork w; w.link = link; w.area = area; w.parent = parent; w.which = which;
    top_image_build_with(nodes, records, layout, tris, root, wide, n_triangles, max_nodes, image, w, TopSerialArgMax());
}
#endif

} // namespace nrt
