/*
 * film.h -- ImageBlock on the device: sample store, filtered splat, block merge.
 *
 * Both render engines finish a camera sample by storing (pixelSample, radiance)
 * -- 20 B -- into a sample store in HBM laid out tile-major:
 *     index = (tile_ordinal * n_spp + sample_in_launch) * 256 + pixel_in_tile
 * (pixel_in_tile uses the 8x8-quad numbering of film_tile_pixel).  Then
 *   film_gather   one workgroup per tile: ImageBlock::put(pos, value)
 *                 (src/block.cpp:62-91) turned from a scatter into a gather --
 *                 every pixel of the tile's bordered block sums, in a fixed
 *                 order, the samples of the tile that reach it, with the
 *                 reference's own block-relative weights.  No atomics (measured:
 *                 ds_add_f32 splatting cost 22 % of the render kernel).
 *                 The block is added to the tile's private accumulator in HBM.
 *   film_resolve  ImageBlock::put(ImageBlock&) (src/block.cpp:93-102): every
 *                 frame pixel gathers the <= 4 tile accumulators that cover it.
 * Deterministic: the same inputs give the same bits, independent of scheduling.
 *
 * REFERENCE ORDER (option film_order = "reference").  The fast path above adds a pixel's samples round by round and
 * tap by tap; the reference adds them in the order of renderBlock / ImageBlock::put (src/main.cpp:33-53,
 * src/block.cpp:62-102): per 32x32 block, source pixel after source pixel in raster order, sample after sample, each
 * (value * wx) * wy into the block's own accumulator -- and the blocks into the frame in BlockGenerator's spiral order
 * (src/block.cpp:109-152).  Float addition is not associative, so the two differ in the last bits.
 * film_reference_order reproduces the reference's order: one workgroup per 32x32 block, one thread per pixel of the
 * block's bordered accumulator walking the source pixels that reach it in raster order and their samples in index
 * order; then every frame pixel adds the (at most four) blocks that cover it by ascending spiral rank.  With
 * bit-identical radiance per camera sample the FRAME is then bit-identical to a single-threaded
 * render of the CPU oracle.  Slower (every sample is read by up to 25 threads straight from the store): a mode for
 * comparisons, not the default.
 */
#pragma once
#include <string>

#include "rt_types.h"

namespace nrt {

struct FilmStore {
    f2 *pos = nullptr;          /* pixelSample */
    P3 *L = nullptr;            /* radiance rgb, 12 B apart */
    size_t capacity = 0;        /* samples */
    float *tile_acc = nullptr;  /* n_tiles x n_parts x tile_w^2 x 4 */
    uint32_t n_parts = 1;       /* workgroups per tile in film_gather: each sums its share of the samples per pixel
                                   into its own accumulator (few tiles per GPU: keeps all CUs busy); fixed per render */
    size_t acc_floats = 0;
    unsigned long long *d_invalid = nullptr;
    float *block_acc = nullptr;          /* reference order: n_blocks x (32 + 2 border)^2 x 4 */
    size_t block_floats = 0;
    uint32_t *spiral_rank = nullptr;     /* reference order: position of every 32x32 block in BlockGenerator's sequence */
    size_t n_rank = 0;
    uint32_t rank_bx = 0, rank_by = 0;   /* the block grid the ranks on the device were computed for (uploaded once per frame geometry) */
};

struct FilmLaunch {
    uint32_t tile_first, n_tiles;   /* ordinals (within the selected tiles) handled by this gather */
    uint32_t store_tile_first;      /* ordinal of the tile whose samples start at index 0 of the store */
    uint32_t n_spp;                 /* samples per pixel present in the store */
    uint32_t tile_mod, tile_rem, tiles_x, tiles_y;
    int32_t tile_w;
};

/* pixel of index `pix` (0..255) inside the tile at (x0, y0): wave w covers the 8x8 quad (w&1, w>>1) */
NORI_HD void film_tile_pixel(int pix, int x0, int y0, int &px, int &py) {
    const int wave = pix >> 6, lane = pix & 63;
    px = x0 + ((wave & 1) << 3) + (lane & 7);
    py = y0 + ((wave >> 1) << 3) + (lane >> 3);
}

/* (Re)allocate the context's store `store` (owned by nori_hip_ctx, grown on demand, freed by film_release);
   zeroes the tile accumulators; `out` = the view to launch with.  "" or an error. */
std::string film_prepare(FilmStore &store, size_t n_samples, size_t n_sel_tiles, int tile_w, void *stream, FilmStore &out);
/* splat the store's samples of tiles [tile_first, tile_first + n_tiles) into their accumulators */
void film_gather(const DevScene &sc, const float *d_filter_table, const FilmStore &st, const FilmLaunch &fl, void *stream);
/* add all accumulators into the caller's RGBW frame */
void film_resolve(const DevScene &sc, const FilmStore &st, const FilmLaunch &fl, float *d_rgbw, void *stream);
/* A share of the reference-order film for one of several devices: whole ROWS of 32x32 blocks (a block's samples are added
   consecutively, so a block is the smallest share), i.e. the contiguous range of 16x16 tiles from
   film_block_rows_first_tile(row_begin) on.  With block_acc set, the call stops after the blocks: it writes the accumulators of
   its own blocks into the caller's array for ALL blocks of the frame (film_block_acc_floats; zeroed by the caller).  The shares'
   arrays are disjoint, so their element-wise sum is exact whatever its order (x + 0 = x; an accumulator is never -0), and
   film_resolve_blocks on the sum gives the bits of the one-device frame. */
struct FilmBlockRows {
    uint32_t row_begin = 0, row_count = 0xffffffffu;      /* clipped to the frame's block rows */
    float *block_acc = nullptr;
};
NORI_HD uint32_t film_block_rows_first_tile(uint32_t block_row, uint32_t tiles_x) { return block_row * 2u * tiles_x; }      /* NORI_BLOCK_SIZE = 2 tiles */
size_t film_block_acc_floats(const DevScene &sc);
/* Reference-order film (see the header comment): the store must hold EVERY sample of the frame (of the share's block rows) --
   all tiles, samples [0, n_spp) of every pixel, index (tile * n_spp + s) * 256 + pixel -- because a pixel's samples are added
   consecutively.  Accumulates into d_rgbw (unless share->block_acc is set).  "" or an error. */
std::string film_reference_order(FilmStore &store, const FilmStore &view, const DevScene &sc, const float *d_filter_table,
                                 uint32_t n_spp, uint32_t tiles_x, const FilmBlockRows *share, float *d_rgbw, void *stream);
/* ImageBlock::put(ImageBlock&) of every block in BlockGenerator's order (src/block.cpp:93-152) into d_rgbw */
std::string film_resolve_blocks(FilmStore &store, const DevScene &sc, const float *d_block_acc, float *d_rgbw, void *stream);
/* samples dropped by the isValid() guard (src/block.cpp:63-67) since film_prepare; synchronises `stream` */
unsigned long long film_invalid_count(const FilmStore &st, void *stream);
void film_release(FilmStore &store);

} // namespace nrt
