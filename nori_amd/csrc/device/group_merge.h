/*
 * group_merge.h -- how a frame is shared out over the GPUs of one node and put back together.
 *
 * Replaces, across devices, what the reference does across TBB workers: the parallel_for over image blocks
 * (src/main.cpp:85-113) and the additive merge of every worker's block into the frame under a mutex,
 * ImageBlock::put(ImageBlock&) (src/block.cpp:93-102).  The path shards with no data-path exchange -- every device
 * renders its share into its own zero-initialised RGBW frame -- and ends with ONE merge on device 0.
 *
 *   split tile    16x16 tiles round-robin over the devices (tile_mod = N, tile_rem = rank)
 *   split sample  every device renders the whole frame with a contiguous share of the sample indices per pixel
 *                 (disjoint pcg32 streams)
 *   merge reduce  sum of the N whole frames into device 0's (RCCL ncclReduce; either split)
 *   (film_order = reference: rows of 32x32 blocks per device, the blocks' accumulators merged, then added into the frame in the
 *   reference's block order on device 0 -- group_block_rows below, nori_hip.h)
 *   merge gather  tile split with tiles_x divisible by N only: a device's tiles are whole tile COLUMNS, so what it touched
 *                 is a set of (16 + 2 border)-pixel-wide column strips; each device packs its strips, device 0 receives
 *                 them (RCCL send / receive) and adds them, overlapping halos included -- 1/N of the frame per device
 *                 over the links instead of a reduction of all of it
 *
 * This header is plain host C++ (no HIP): the index arithmetic below is shared by the device group of libnori_hip
 * (group.hip: kernels pack and add by these index lists) and by the CPU harness (tests/emu: the same lists drive plain
 * loops), where the threaded driver and the merges are tested against a single render.
 */
#pragma once
#include <stdint.h>
#include <vector>

namespace nrt {

enum { kSplitTile = 0, kSplitSample = 1 };
enum { kMergeReduce = 0, kMergeGather = 1 };

struct GroupShare { uint32_t spp_begin, spp_count, tile_mod, tile_rem; };

/* share of device `rank` of a job of samples [spp_begin, spp_begin + spp_count) per pixel */
inline GroupShare group_share(int split, int rank, int world, uint32_t spp_begin, uint32_t spp_count) {
    GroupShare s;
    if (split == kSplitSample) {
        const uint32_t base = spp_count / (uint32_t) world, extra = spp_count % (uint32_t) world;
        const uint32_t r = (uint32_t) rank;
        s.spp_begin = spp_begin + r * base + (r < extra ? r : extra);
        s.spp_count = base + (r < extra ? 1u : 0u);
        s.tile_mod = 1u; s.tile_rem = 0u;
    } else {
        s.spp_begin = spp_begin; s.spp_count = spp_count; s.tile_mod = (uint32_t) world; s.tile_rem = (uint32_t) rank;
    }
    return s;
}

/* film_order = reference: the share is whole rows of 32x32 blocks (a block's samples are added consecutively, film.h) --
   contiguous, the first (block_rows % world) devices one row more; devices beyond the rows get none */
struct GroupRows { uint32_t row_begin, row_count; };
inline GroupRows group_block_rows(int rank, int world, uint32_t block_rows) {
    const uint32_t base = block_rows / (uint32_t) world, extra = block_rows % (uint32_t) world, r = (uint32_t) rank;
    GroupRows s;
    s.row_begin = r * base + (r < extra ? r : extra);
    s.row_count = base + (r < extra ? 1u : 0u);
    return s;
}

/* Column strips of device `rank` under the tile split (tiles_x % world == 0): tile column c = rank, rank + world, ...
 * covers the bordered-frame columns [16 c, 16 c + 16 + 2 border).  Returns, strip after strip, the frame column of every
 * packed column, or -1 for a column that is NOT to be packed: outside the frame (the last tile column of an image whose
 * width is not a multiple of 16 is clipped) or already packed by an earlier strip of the same device -- a device's own
 * strips overlap when 16 world < 16 + 2 border (always for one device), and a column packed twice would be added twice. */
inline std::vector<int32_t> group_strip_columns(int rank, int world, uint32_t tiles_x, int border, int frame_cols) {
    std::vector<int32_t> x;
    int32_t seen_up_to = -1;                      /* strips ascend: everything <= this is packed already */
    for (uint32_t c = (uint32_t) rank; c < tiles_x; c += (uint32_t) world)
        for (int k = 0; k < 16 + 2 * border; ++k) {
            const int32_t col = (int32_t) (16u * c) + k;
            if (col >= frame_cols || col <= seen_up_to) { x.push_back(-1); continue; }
            x.push_back(col); seen_up_to = col;
        }
    return x;
}

/* host twins of the device kernels (the CPU harness): frame = rows x cols x 4 floats */
inline void group_pack_strips(const float *frame, int rows, int cols, const std::vector<int32_t> &x, float *pack) {
    const size_t w = x.size();
    for (int y = 0; y < rows; ++y)
        for (size_t i = 0; i < w; ++i)
            for (int ch = 0; ch < 4; ++ch)
                pack[((size_t) y * w + i) * 4 + ch] = x[i] >= 0 ? frame[((size_t) y * cols + x[i]) * 4 + ch] : 0.0f;
}
inline void group_add_strips(float *frame, int rows, int cols, const std::vector<int32_t> &x, const float *pack) {
    const size_t w = x.size();
    for (int y = 0; y < rows; ++y)
        for (size_t i = 0; i < w; ++i)
            if (x[i] >= 0)
                for (int ch = 0; ch < 4; ++ch) frame[((size_t) y * cols + x[i]) * 4 + ch] += pack[((size_t) y * w + i) * 4 + ch];
}

} // namespace nrt
