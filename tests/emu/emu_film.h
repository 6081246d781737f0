/*
 * emu_film.h -- ImageBlock::put(pos, value), src/block.cpp:62-91, into a pixel tile of
 * (kTile + 2*border)^2 RGBW accumulators: the CPU emulation harness's film (a plain scatter; the
 * product's film is film.hip: sample store + gather, checked against the oracle on the GPU).
 *
 * Weights are computed exactly as the reference does for the 32x32 block (NORI_BLOCK_SIZE,
 * include/nori/block.h:17) that contains the tile -- same block-relative float arithmetic, so the
 * filter-table index of every (sample, pixel) pair is the one Nori computes -- and then shifted to
 * tile coordinates with integer offsets only.
 */
#pragma once
#include "../../nori_amd/csrc/device/rt_types.h"

namespace nrt {

template <class Add>
NORI_HD void splat_tile(float *tile, int tile_w, int x0, int y0, const float *ftab, float radius, float lookup,
                        int border, f2 pos, f3 value, Add add) {
    const int bx0 = x0 & ~31, by0 = y0 & ~31;
    const float px = pos.x - 0.5f - (float) (bx0 - border);
    const float py = pos.y - 0.5f - (float) (by0 - border);
    const int offx = x0 - bx0, offy = y0 - by0;            /* block -> tile */
    int minX = (int) ceilf(px - radius) - offx, maxX = (int) floorf(px + radius) - offx;
    int minY = (int) ceilf(py - radius) - offy, maxY = (int) floorf(py + radius) - offy;
    minX = minX < 0 ? 0 : minX; minY = minY < 0 ? 0 : minY;
    maxX = maxX > tile_w - 1 ? tile_w - 1 : maxX; maxY = maxY > tile_w - 1 ? tile_w - 1 : maxY;
    for (int y = minY; y <= maxY; ++y) {
        const float wy = ftab[(int) (fabsf((float) (y + offy) - py) * lookup)];
        for (int x = minX; x <= maxX; ++x) {
            const float wx = ftab[(int) (fabsf((float) (x + offx) - px) * lookup)];
            float *p = tile + ((y * tile_w + x) << 2);
            add(p + 0, value.x * wx * wy);
            add(p + 1, value.y * wx * wy);
            add(p + 2, value.z * wx * wy);
            add(p + 3, 1.0f * wx * wy);
        }
    }
}

} // namespace nrt
