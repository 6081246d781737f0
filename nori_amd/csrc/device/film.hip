/* film.hip -- sample store, gather splat and block merge (see film.h). */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "film.h"

using namespace nrt;

namespace {

constexpr int kB = 256;

constexpr int kMaxBorder = 8;                  /* (16 + 2 * 8)^2 / 256 = 4 output pixels per thread */

/* a workgroup barrier that waits for LDS only (what film_gather's rounds exchange lives there) */
__device__ __forceinline__ void film_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* ImageBlock::put(pos, value) as a gather.  Sample round by sample round, the tile's 256 current
 * samples are staged in LDS together with their 1-D filter weights; then every pixel of the tile's
 * bordered block (tile_w^2 pixels, <= 4 per thread) adds the samples that can reach it.  Weights follow
 * src/block.cpp:70-90 in the coordinates of the reference's 32x32 block containing the tile, so every
 * filter-table index is the one Nori computes:
 *     pos   = p - 0.5 - (block_offset - border)
 *     pixel x is touched iff ceil(pos.x - r) <= x <= floor(pos.x + r)   <=>  pos.x - r <= x <= pos.x + r
 *     wx[x] = filter[(int)(|x - pos.x| * lookupFactor)],  wy[y] likewise       (block.cpp:79-84)
 *     px   += ((r, g, b, 1) * wx[x]) * wy[y]                                    (block.cpp:88-90)
 * A sample of tile pixel (sx, sy) reaches the output pixels (sx + k, sy + m), k, m in [0, 2 border]
 * of the bordered tile frame: 2 border + 1 taps per axis, zero where out of the filter's reach or
 * for samples the isValid() guard (block.cpp:63-67) rejects. */
__global__ __launch_bounds__(kB) void film_gather_kernel(int width, int height, FilterRec fr, const float *__restrict__ filter_table,
                                                         FilmStore st, FilmLaunch fl) {
    extern __shared__ float s_dyn[];                          /* [taps][256] wx, [taps][256] wy, r, g, b */
    __shared__ float ftab[kFilterRes + 1];
    __shared__ unsigned int s_invalid;
    const int tid = threadIdx.x;
    if (tid <= kFilterRes) ftab[tid] = filter_table[tid];
    if (tid == 0) s_invalid = 0u;

    const uint32_t part = blockIdx.x % st.n_parts;
    const uint32_t ord = fl.tile_first + blockIdx.x / st.n_parts;
    const uint32_t tile_id = fl.tile_rem + ord * fl.tile_mod;
    const int x0 = (int) (tile_id % fl.tiles_x) * kTile, y0 = (int) (tile_id / fl.tiles_x) * kTile;
    const int border = fr.border, tile_w = fl.tile_w, taps = 2 * border + 1;
    float (*s_wx)[256] = reinterpret_cast<float (*)[256]>(s_dyn), (*s_wy)[256] = s_wx + taps;
    float *s_Lr = s_dyn + 2 * taps * 256, *s_Lg = s_Lr + 256, *s_Lb = s_Lg + 256;      /* (a float4 array: ds_read_b128 bank conflicts, 2.6x slower) */
    const float radius = fr.radius, lookup = fr.lookup_factor;
    const int bx0 = x0 & ~31, by0 = y0 & ~31;                 /* NORI_BLOCK_SIZE = 32 */
    const int offx = x0 - bx0, offy = y0 - by0;               /* tile frame -> block frame */

    /* this thread's sample slot (pixel of the tile) */
    int px, py; film_tile_pixel(tid, x0, y0, px, py);
    const bool live = px < width && py < height;
    const int sxl = px - x0, syl = py - y0, raster = syl * kTile + sxl;

    /* this thread's output pixels in the bordered tile frame, and -- fixed for all sample rounds -- the taps that
       reach a source pixel of the tile from each: source (ox - k, oy - m) lies in [0, 16)^2 for k in [k0, k1],
       m in [m0, m1].  The tap loops run over exactly those: no per-tap bounds test. */
    const int n_out = tile_w * tile_w;
    constexpr int kMaxOut = 4;
    int out_x[kMaxOut], out_y[kMaxOut], k_lo[kMaxOut], k_hi[kMaxOut], m_lo[kMaxOut], m_hi[kMaxOut];
    f4 acc[kMaxOut];
    for (int o = 0; o < kMaxOut; ++o) {
        const int i = tid + o * kB;
        out_y[o] = i < n_out ? i / tile_w : -1000; out_x[o] = i - (i / tile_w) * tile_w;
        k_lo[o] = max(0, out_x[o] - (kTile - 1)); k_hi[o] = min(taps - 1, out_x[o]);
        m_lo[o] = max(0, out_y[o] - (kTile - 1)); m_hi[o] = min(taps - 1, out_y[o]);
        acc[o].x = acc[o].y = acc[o].z = acc[o].w = 0.0f;
    }

    const size_t first = (size_t) (ord - fl.store_tile_first) * fl.n_spp * 256u;
    unsigned int invalid = 0;
    __syncthreads();
    /* this workgroup's share of the samples per pixel */
    const uint32_t s_lo = (uint32_t) ((uint64_t) fl.n_spp * part / st.n_parts), s_hi = (uint32_t) ((uint64_t) fl.n_spp * (part + 1) / st.n_parts);
    /* the NEXT round's sample is requested while this round's taps run (its trip to HBM used to open every round; the barriers of a
       round wait for LDS only -- __syncthreads() would wait for the request as well) */
    f2 p_next = mk2(0.0f, 0.0f); P3 l_next; l_next.x = l_next.y = l_next.z = 0.0f;
    if (live && s_lo < s_hi) { const size_t i0 = first + (size_t) s_lo * 256u + (size_t) tid; p_next = st.pos[i0]; l_next = st.L[i0]; }
    for (uint32_t s = s_lo; s < s_hi; ++s) {
        {
            f4 L; L.x = L.y = L.z = 0.0f;
            bool ok = false;
            float bpx = 0.0f, bpy = 0.0f;
            const f2 p = p_next; const P3 l3 = l_next;
            if (live && s + 1u < s_hi) { const size_t i1 = first + (size_t) (s + 1u) * 256u + (size_t) tid; p_next = st.pos[i1]; l_next = st.L[i1]; }
            if (live) {
                L.x = l3.x; L.y = l3.y; L.z = l3.z;
                ok = color_valid(mk3(L.x, L.y, L.z));
                if (!ok) { ++invalid; L.x = L.y = L.z = 0.0f; }
                bpx = p.x - 0.5f - (float) (bx0 - border);
                bpy = p.y - 0.5f - (float) (by0 - border);
            }
            s_Lr[raster] = L.x; s_Lg[raster] = L.y; s_Lb[raster] = L.z;
            for (int k = 0; k < taps; ++k) {
                const float xb = (float) (sxl + k + offx), yb = (float) (syl + k + offy);
                const bool inx = ok && xb >= bpx - radius && xb <= bpx + radius;
                const bool iny = ok && yb >= bpy - radius && yb <= bpy + radius;
                s_wx[k][raster] = inx ? ftab[(int) (fabsf(xb - bpx) * lookup)] : 0.0f;
                s_wy[k][raster] = iny ? ftab[(int) (fabsf(yb - bpy) * lookup)] : 0.0f;
            }
        }
        film_barrier();
        for (int o = 0; o < kMaxOut; ++o) {
            const int oy = out_y[o], ox = out_x[o];
            if (oy < 0) break;
            for (int m = m_lo[o]; m <= m_hi[o]; ++m) {
                const int row = (oy - m) * kTile + ox;
                const float *wy_row = s_wy[m];
                for (int k = k_lo[o]; k <= k_hi[o]; ++k) {
                    const int r = row - k;
                    /* Color4f(value) * wx * wy (block.cpp:88-90) as value * (wx * wy), multiply-add fused: 5 instructions per tap where
                       the reference's own left-to-right product ((L * wx) * wy per channel, then the add) takes 11 -- this kernel
                       is bound by the vector ALU.  The fast film does not reproduce the reference's summation ORDER anyway
                       (film.h): its frames differ from a render in reference order by rounding, ~1e-7 relative either way;
                       film_order = reference keeps order and products exactly. */
                    const float w = s_wx[k][r] * wy_row[r];
                    acc[o].x = __builtin_fmaf(s_Lr[r], w, acc[o].x); acc[o].y = __builtin_fmaf(s_Lg[r], w, acc[o].y);
                    acc[o].z = __builtin_fmaf(s_Lb[r], w, acc[o].z); acc[o].w += w;
                }
            }
        }
        film_barrier();                                        /* round consumed */
    }
    f4 *dst = reinterpret_cast<f4 *>(st.tile_acc) + ((size_t) ord * st.n_parts + part) * n_out;
    for (int o = 0; o < kMaxOut; ++o) {
        const int i = tid + o * kB;
        if (i >= n_out) break;
        f4 v = dst[i];
        v.x += acc[o].x; v.y += acc[o].y; v.z += acc[o].z; v.w += acc[o].w;
        dst[i] = v;
    }
    if (invalid) atomicAdd(&s_invalid, invalid);
    __syncthreads();
    if (tid == 0 && s_invalid) atomicAdd(st.d_invalid, (unsigned long long) s_invalid);
}

/* ImageBlock::put(ImageBlock&): every frame pixel gathers the (at most four) tile accumulators
   whose bordered area covers it, in a fixed order */
__global__ void film_resolve_kernel(int width, int height, int border, int tile_w, uint32_t tiles_x, uint32_t tiles_y,
                                    uint32_t tile_mod, uint32_t tile_rem, uint32_t n_parts, const float *tile_acc, float *rgbw) {
    const int cols = width + 2 * border, rows = height + 2 * border;
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y;
    if (gx >= cols || gy >= rows) return;
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int tx1 = min(gx / kTile, (int) tiles_x - 1), ty1 = min(gy / kTile, (int) tiles_y - 1);
    const int tx0 = max(0, (gx - tile_w + kTile) / kTile), ty0 = max(0, (gy - tile_w + kTile) / kTile);
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int lx = gx - tx * kTile, ly = gy - ty * kTile;
            if (lx < 0 || ly < 0 || lx >= tile_w || ly >= tile_w) continue;
            const uint32_t tile_id = (uint32_t) ty * tiles_x + (uint32_t) tx;
            if (tile_id < tile_rem || (tile_id - tile_rem) % tile_mod != 0u) continue;
            const uint32_t ord = (tile_id - tile_rem) / tile_mod;
            for (uint32_t part = 0; part < n_parts; ++part) {      /* fixed order: deterministic */
                const float4 v = *reinterpret_cast<const float4 *>(tile_acc + ((((size_t) ord * n_parts + part) * tile_w + (size_t) ly) * tile_w + lx) * 4);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
    float4 *dst = reinterpret_cast<float4 *>(rgbw) + (size_t) gy * cols + gx;
    float4 cur = *dst;
    cur.x += sum.x; cur.y += sum.y; cur.z += sum.z; cur.w += sum.w;
    *dst = cur;
}

/* ---- reference order ---- */
constexpr int kBlock32 = 32;          /* NORI_BLOCK_SIZE, include/nori/block.h:17 */

/* one workgroup per 32x32 block; a thread owns pixels of the block's bordered accumulator and adds, for each, the
   samples that reach it in the reference's order: source pixels in raster order (renderBlock's y, x loops,
   src/main.cpp:33-34), samples in index order (:35), each as ImageBlock::put does (src/block.cpp:62-91) */
__global__ __launch_bounds__(kB) void film_block_reference_kernel(int width, int height, FilterRec fr, const float *__restrict__ filter_table,
                                                                  FilmStore st, uint32_t n_spp, uint32_t tiles_x, uint32_t blocks_x,
                                                                  uint32_t block_first, uint32_t tile_first) {
    __shared__ float ftab[kFilterRes + 1];
    if (threadIdx.x <= kFilterRes) ftab[threadIdx.x] = filter_table[threadIdx.x];
    __syncthreads();
    const uint32_t blk = block_first + blockIdx.x;      /* a share of the frame: whole block rows from block_first on, their tiles from tile_first on in the store */
    const int bx = (int) (blk % blocks_x), by = (int) (blk / blocks_x);
    const int offx = bx * kBlock32, offy = by * kBlock32;
    const int bw = min(kBlock32, width - offx), bh = min(kBlock32, height - offy);
    const int border = fr.border, cols = bw + 2 * border, rows = bh + 2 * border;
    const float radius = fr.radius, lookup = fr.lookup_factor;
    float4 *dst = reinterpret_cast<float4 *>(st.block_acc) + (size_t) blk * (kBlock32 + 2 * border) * (kBlock32 + 2 * border);
    unsigned long long invalid = 0;
    for (int i = (int) threadIdx.x; i < cols * rows; i += kB) {
        const int oy = i / cols, ox = i - oy * cols;
        const float fx = (float) ox, fy = (float) oy;
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int sy = max(0, oy - 2 * border); sy <= min(bh - 1, oy); ++sy)
            for (int sx = max(0, ox - 2 * border); sx <= min(bw - 1, ox); ++sx) {
                const int px = offx + sx, py = offy + sy;
                const uint32_t tile = (uint32_t) (py / kTile) * tiles_x + (uint32_t) (px / kTile) - tile_first;
                const int lx = px % kTile, ly = py % kTile;
                const uint32_t pix = (uint32_t) ((((lx >> 3) | ((ly >> 3) << 1)) << 6) | ((lx & 7) | ((ly & 7) << 3)));      /* inverse of film_tile_pixel */
                const bool centre = ox == sx + border && oy == sy + border;      /* one thread per source pixel counts its invalid samples */
                const size_t idx0 = (size_t) tile * n_spp * 256u + pix;
                /* branch-free body, four samples in flight: a sample outside its bounding box (block.cpp:76-80) or rejected
                   by the isValid() guard (:63-67) is given weight 0 and radiance 0 -- adding (0 * 0) * wy = +0 leaves the
                   accumulator's bits alone (it can never be -0), exactly as skipping it does */
                auto term = [&](size_t idx, float4 &a) {
                    f4 L; { const P3 l3 = st.L[idx]; L.x = l3.x; L.y = l3.y; L.z = l3.z; L.w = 0.0f; }
                    const f2 p = st.pos[idx];
                    const bool ok = color_valid(mk3(L.x, L.y, L.z));
                    if (centre && !ok) ++invalid;
                    const float bpx = p.x - 0.5f - (float) (offx - border), bpy = p.y - 0.5f - (float) (offy - border);
                    const bool in = ok && fx >= bpx - radius && fx <= bpx + radius && fy >= bpy - radius && fy <= bpy + radius;
                    const float wx = in ? ftab[(int) (fabsf(fx - bpx) * lookup)] : 0.0f, wy = in ? ftab[(int) (fabsf(fy - bpy) * lookup)] : 0.0f;
                    if (!in) L.x = L.y = L.z = 0.0f;
                    a.x += (L.x * wx) * wy; a.y += (L.y * wx) * wy; a.z += (L.z * wx) * wy; a.w += (1.0f * wx) * wy;
                };
                uint32_t s = 0;
                for (; s + 4u <= n_spp; s += 4u) {
                    term(idx0 + (size_t) s * 256u, acc); term(idx0 + (size_t) (s + 1u) * 256u, acc);
                    term(idx0 + (size_t) (s + 2u) * 256u, acc); term(idx0 + (size_t) (s + 3u) * 256u, acc);
                }
                for (; s < n_spp; ++s) term(idx0 + (size_t) s * 256u, acc);
            }
        dst[oy * (kBlock32 + 2 * border) + ox] = acc;
    }
    for (int off = 32; off > 0; off >>= 1) invalid += __shfl_down(invalid, off);
    if ((threadIdx.x & 63u) == 0u && invalid) atomicAdd(st.d_invalid, invalid);
}

/* The same sums in the same order, with every sample fetched 2 border + 1 times instead of (2 border + 1)^2 times.
 *
 * For ONE output pixel the reference's order is a chain: source rows ascending, within a row the sources left to right,
 * within a source its samples by index.  Chains of different output pixels are independent, and at any moment the
 * outputs that take from source row sy are the 2 border + 1 accumulator rows [sy, sy + 2 border]; each of them meets the
 * row's sources in 2 border + 1 PHASES: in phase j the output at column ox adds source sx = ox - (2 border - j) -- every
 * output its own source, all with the same horizontal tap k = 2 border - j.  So, per source row and per phase, the row's
 * samples are staged through LDS chunk by chunk (premultiplied: L wx[k] -- (L wx) wy keeps the reference's
 * association, src/block.cpp:88-90 -- the weight wx[k] itself for the W channel, and wy for all 2 border + 1 rows), and
 * every thread -- one output pixel of a ring of 2 border + 1 accumulator rows -- adds its source's samples of the chunk in
 * order.  When a source row is done, the accumulator row that just received its last source is written out and its ring
 * slot starts the row that enters below. */
constexpr int kRefChunk = 32;          /* samples of one source pixel staged per round (16 for filters of more than 11 taps: LDS) */
#define NORI_REF_UNROLL 4      /* terms whose LDS reads are in flight together (2: 8.5 ms, 4: 8.4, 8: 12.9 -- registers) */
constexpr int kRefMaxSlots = 4;        /* (2 * 8 + 1) * (32 + 16) / 256 rounded up */

__global__ __launch_bounds__(kB) void film_block_reference_staged_kernel(int width, int height, FilterRec fr, const float *__restrict__ filter_table,
                                                                         FilmStore st, uint32_t n_spp, uint32_t tiles_x, uint32_t blocks_x, int chunk,
                                                                         uint32_t block_first, uint32_t tile_first) {
    extern __shared__ __attribute__((aligned(16))) float s_ref[];      /* [chunk][32] float4 (r wx, g wx, b wx, wx), then wy[taps][chunk][32] */
    __shared__ float ftab[kFilterRes + 1];
    if (threadIdx.x <= kFilterRes) ftab[threadIdx.x] = filter_table[threadIdx.x];
    const int tid = (int) threadIdx.x;
    const uint32_t blk = block_first + blockIdx.x;
    const int bx = (int) (blk % blocks_x), by = (int) (blk / blocks_x);
    const int offx = bx * kBlock32, offy = by * kBlock32;
    const int bw = min(kBlock32, width - offx), bh = min(kBlock32, height - offy);
    const int border = fr.border, taps = 2 * border + 1, cols = bw + 2 * border;
    const int stride = kBlock32 + 2 * border;
    const float radius = fr.radius, lookup = fr.lookup_factor;
    float4 *dst = reinterpret_cast<float4 *>(st.block_acc) + (size_t) blk * stride * stride;
    const int plane = chunk * kBlock32;
    float4 *s_lw = reinterpret_cast<float4 *>(s_ref);      /* one ds_read_b128 per term: consecutive lanes read consecutive float4s (conflict free) */
    float *s_wy = s_ref + 4 * plane;                       /* s_wy[m][s][sx] */
    const int n_slots = taps * cols;
    float4 acc[kRefMaxSlots];
    for (int a = 0; a < kRefMaxSlots; ++a) acc[a] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned long long invalid = 0;
    __syncthreads();
    for (int sy = 0; sy < bh; ++sy) {
        const int py = offy + sy;
        for (int j = 0; j < taps; ++j) {
            const int k = taps - 1 - j;                       /* output column = source column + k */
            for (uint32_t c0 = 0; c0 < n_spp; c0 += (uint32_t) chunk) {
                const int cs = (int) min((uint32_t) chunk, n_spp - c0);
                /* stage samples [c0, c0 + cs) of the row's bw source pixels (requesting a thread's four samples ahead of their
                   weights measured slower: 10.7 against 8.4 ms) */
                for (int e = tid; e < cs * bw; e += kB) {
                    const int sl = e / bw, sx = e - sl * bw;
                    const int px = offx + sx;
                    const uint32_t tile = (uint32_t) (py / kTile) * tiles_x + (uint32_t) (px / kTile) - tile_first;
                    const int lx = px % kTile, ly = py % kTile;
                    const uint32_t pix = (uint32_t) ((((lx >> 3) | ((ly >> 3) << 1)) << 6) | ((lx & 7) | ((ly & 7) << 3)));      /* inverse of film_tile_pixel */
                    const size_t idx = (size_t) tile * n_spp * 256u + pix + (size_t) (c0 + (uint32_t) sl) * 256u;
                    f4 L; { const P3 l3 = st.L[idx]; L.x = l3.x; L.y = l3.y; L.z = l3.z; L.w = 0.0f; }
                    const f2 p = st.pos[idx];
                    const bool ok = color_valid(mk3(L.x, L.y, L.z));
                    if (j == 0 && !ok) ++invalid;              /* every sample is staged once with j == 0 */
                    const float bpx = p.x - 0.5f - (float) (offx - border), bpy = p.y - 0.5f - (float) (offy - border);
                    const float fx = (float) (sx + k);
                    const bool inx = ok && fx >= bpx - radius && fx <= bpx + radius;
                    const float wx = inx ? ftab[(int) (fabsf(fx - bpx) * lookup)] : 0.0f;
                    const int o = sl * kBlock32 + sx;
                    /* a sample outside the pixel's bounding box (block.cpp:76-80) or rejected by isValid() (:63-67) gets weight 0 and
                       radiance 0: it adds (0 * 0) * wy = +0, which leaves an accumulator's bits alone (it is never -0) */
                    s_lw[o] = make_float4((inx ? L.x : 0.0f) * wx, (inx ? L.y : 0.0f) * wx, (inx ? L.z : 0.0f) * wx, 1.0f * wx);
                    for (int m = 0; m < taps; ++m) {
                        const float fy = (float) (sy + m);
                        const bool iny = ok && fy >= bpy - radius && fy <= bpy + radius;
                        s_wy[m * plane + o] = iny ? ftab[(int) (fabsf(fy - bpy) * lookup)] : 0.0f;
                    }
                }
                __syncthreads();
                for (int a = 0; a < kRefMaxSlots; ++a) {
                    const int slot = tid + a * kB;
                    if (slot >= n_slots) break;
                    const int q = slot / cols, ox = slot - q * cols;
                    const int sx = ox - k;
                    if (sx < 0 || sx >= bw) continue;
                    /* the accumulator row of ring slot q that is active at source row sy: oy in [sy, sy + taps) with oy % taps == q */
                    const int oy = sy + ((q - sy % taps) + taps) % taps;
                    const float *wy = s_wy + (oy - sy) * plane + sx;
                    float4 v = acc[a];
                    int sl = 0;
                    for (; sl + NORI_REF_UNROLL <= cs; sl += NORI_REF_UNROLL) {      /* several terms' LDS reads in flight, then their adds in order */
                        float4 lw[NORI_REF_UNROLL]; float w[NORI_REF_UNROLL];
#pragma unroll
                        for (int u = 0; u < NORI_REF_UNROLL; ++u) { lw[u] = s_lw[(sl + u) * kBlock32 + sx]; w[u] = wy[(sl + u) * kBlock32]; }
#pragma unroll
                        for (int u = 0; u < NORI_REF_UNROLL; ++u) { v.x += lw[u].x * w[u]; v.y += lw[u].y * w[u]; v.z += lw[u].z * w[u]; v.w += lw[u].w * w[u]; }
                    }
                    for (; sl < cs; ++sl) {
                        const float4 lw = s_lw[sl * kBlock32 + sx];
                        const float w = wy[sl * kBlock32];
                        v.x += lw.x * w; v.y += lw.y * w; v.z += lw.z * w; v.w += lw.w * w;
                    }
                    acc[a] = v;
                }
                __syncthreads();
            }
        }
        /* accumulator row sy has received its last source row (sy); at the block's last source row all open rows have */
        for (int a = 0; a < kRefMaxSlots; ++a) {
            const int slot = tid + a * kB;
            if (slot >= n_slots) break;
            const int q = slot / cols, ox = slot - q * cols;
            const int oy = sy + ((q - sy % taps) + taps) % taps;
            if (oy == sy || sy == bh - 1) { dst[oy * stride + ox] = acc[a]; acc[a] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
        }
    }
    for (int off = 32; off > 0; off >>= 1) invalid += __shfl_down(invalid, off);
    if ((threadIdx.x & 63u) == 0u && invalid) atomicAdd(st.d_invalid, invalid);
}

/* ImageBlock::put(ImageBlock&) in the order the blocks arrive from BlockGenerator: every frame pixel adds the blocks
   covering it by ascending spiral rank */
__global__ void film_resolve_reference_kernel(int width, int height, int border, uint32_t blocks_x, uint32_t blocks_y,
                                              const float *block_acc, const uint32_t *spiral_rank, float *rgbw) {
    const int cols = width + 2 * border, rows = height + 2 * border;
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y;
    if (gx >= cols || gy >= rows) return;
    const int stride = kBlock32 + 2 * border;
    float4 val[4]; uint32_t rank[4]; int n = 0;
    const int bx1 = min(gx / kBlock32, (int) blocks_x - 1), by1 = min(gy / kBlock32, (int) blocks_y - 1);
    const int bx0 = max(0, (gx - 2 * border) / kBlock32), by0 = max(0, (gy - 2 * border) / kBlock32);
    for (int by = by0; by <= by1; ++by)
        for (int bx = bx0; bx <= bx1; ++bx) {
            const int bw = min(kBlock32, width - bx * kBlock32), bh = min(kBlock32, height - by * kBlock32);
            const int lx = gx - bx * kBlock32, ly = gy - by * kBlock32;
            if (lx < 0 || ly < 0 || lx >= bw + 2 * border || ly >= bh + 2 * border || n >= 4) continue;
            const uint32_t b = (uint32_t) by * blocks_x + (uint32_t) bx;
            val[n] = *reinterpret_cast<const float4 *>(block_acc + (((size_t) b * stride + (size_t) ly) * stride + lx) * 4);
            rank[n] = spiral_rank[b]; ++n;
        }
    for (int i = 1; i < n; ++i)                                   /* insertion sort by spiral rank */
        for (int j = i; j > 0 && rank[j] < rank[j - 1]; --j) {
            const uint32_t tr = rank[j]; rank[j] = rank[j - 1]; rank[j - 1] = tr;
            const float4 tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
        }
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);              /* the frame starts cleared (main.cpp:68) */
    for (int i = 0; i < n; ++i) { sum.x += val[i].x; sum.y += val[i].y; sum.z += val[i].z; sum.w += val[i].w; }
    float4 *dst = reinterpret_cast<float4 *>(rgbw) + (size_t) gy * cols + gx;
    float4 cur = *dst;
    cur.x += sum.x; cur.y += sum.y; cur.z += sum.z; cur.w += sum.w;
    *dst = cur;
}

/* BlockGenerator (src/block.cpp:109-152): from the centre block (nbx / 2, nby / 2) a square spiral -- right 1, down 1,
   left 2, up 2, right 3, ... -- skipping positions outside the grid; rank = position in that sequence */
std::vector<uint32_t> spiral_ranks(int nbx, int nby) {
    std::vector<uint32_t> rank((size_t) nbx * nby, 0u);
    int x = nbx / 2, y = nby / 2, placed = 0, run = 1, dir = 0;      /* dir: 0 right, 1 down, 2 left, 3 up */
    const int dx[4] = {1, 0, -1, 0}, dy[4] = {0, 1, 0, -1};
    if (x >= 0 && y >= 0 && x < nbx && y < nby) rank[(size_t) y * nbx + x] = (uint32_t) placed++;
    while (placed < nbx * nby) {
        for (int leg = 0; leg < 2 && placed < nbx * nby; ++leg) {    /* two legs share a run length */
            for (int k = 0; k < run && placed < nbx * nby; ++k) {
                x += dx[dir]; y += dy[dir];
                if (x >= 0 && y >= 0 && x < nbx && y < nby) rank[(size_t) y * nbx + x] = (uint32_t) placed++;
            }
            dir = (dir + 1) & 3;
        }
        ++run;
    }
    return rank;
}

} // namespace

namespace nrt {

#define FILM_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); } while (0)

void film_release(FilmStore &g_film) {
    if (g_film.block_acc) (void) hipFree(g_film.block_acc);
    if (g_film.spiral_rank) (void) hipFree(g_film.spiral_rank);
    if (g_film.pos) (void) hipFree(g_film.pos);
    if (g_film.L) (void) hipFree(g_film.L);
    if (g_film.tile_acc) (void) hipFree(g_film.tile_acc);
    if (g_film.d_invalid) (void) hipFree(g_film.d_invalid);
    g_film = FilmStore();
}

std::string film_prepare(FilmStore &g_film, size_t n_samples, size_t n_sel_tiles, int tile_w, void *stream, FilmStore &out) {
    if (g_film.capacity < n_samples) {
        if (g_film.pos) (void) hipFree(g_film.pos);
        if (g_film.L) (void) hipFree(g_film.L);
        g_film.pos = nullptr; g_film.L = nullptr; g_film.capacity = 0;
        FILM_TRY(hipMalloc((void **) &g_film.pos, std::max<size_t>(n_samples, 1) * sizeof(f2)));
        FILM_TRY(hipMalloc((void **) &g_film.L, std::max<size_t>(n_samples, 1) * sizeof(P3)));
        g_film.capacity = n_samples;
    }
    /* enough workgroups to fill the chip even when this GPU owns few tiles */
    g_film.n_parts = (uint32_t) std::min<size_t>(8, std::max<size_t>(1, 2048 / std::max<size_t>(n_sel_tiles, 1)));
    const size_t acc = n_sel_tiles * g_film.n_parts * (size_t) tile_w * tile_w * 4;
    if (g_film.acc_floats < acc) {
        if (g_film.tile_acc) (void) hipFree(g_film.tile_acc);
        g_film.tile_acc = nullptr; g_film.acc_floats = 0;
        FILM_TRY(hipMalloc((void **) &g_film.tile_acc, std::max<size_t>(acc, 4) * sizeof(float)));
        g_film.acc_floats = acc;
    }
    if (!g_film.d_invalid) FILM_TRY(hipMalloc((void **) &g_film.d_invalid, sizeof(unsigned long long)));
    FILM_TRY(hipMemsetAsync(g_film.tile_acc, 0, std::max<size_t>(acc, 4) * sizeof(float), (hipStream_t) stream));
    FILM_TRY(hipMemsetAsync(g_film.d_invalid, 0, sizeof(unsigned long long), (hipStream_t) stream));
    out = g_film;
    return std::string();
}

void film_gather(const DevScene &sc, const float *d_filter_table, const FilmStore &st, const FilmLaunch &fl, void *stream) {
    if (fl.n_tiles == 0 || fl.n_spp == 0) return;
    const size_t lds = (size_t) (2 * (2 * sc.filter.border + 1) + 3) * 256 * sizeof(float);
    hipLaunchKernelGGL(film_gather_kernel, dim3(fl.n_tiles * st.n_parts), dim3(kB), lds, (hipStream_t) stream, sc.camera.width, sc.camera.height,
                       sc.filter, d_filter_table, st, fl);
}

void film_resolve(const DevScene &sc, const FilmStore &st, const FilmLaunch &fl, float *d_rgbw, void *stream) {
    const int border = sc.filter.border, cols = sc.camera.width + 2 * border, rows = sc.camera.height + 2 * border;
    hipLaunchKernelGGL(film_resolve_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, (hipStream_t) stream, sc.camera.width,
                       sc.camera.height, border, fl.tile_w, fl.tiles_x, fl.tiles_y, fl.tile_mod, fl.tile_rem, st.n_parts,
                       (const float *) st.tile_acc, d_rgbw);
}

size_t film_block_acc_floats(const DevScene &sc) {
    const int border = sc.filter.border;
    const size_t nb = (size_t) ((sc.camera.width + kBlock32 - 1) / kBlock32) * (size_t) ((sc.camera.height + kBlock32 - 1) / kBlock32);
    return nb * (kBlock32 + 2 * border) * (kBlock32 + 2 * border) * 4;
}

std::string film_resolve_blocks(FilmStore &store, const DevScene &sc, const float *d_block_acc, float *d_rgbw, void *stream) {
    const int w = sc.camera.width, h = sc.camera.height, border = sc.filter.border;
    const uint32_t bxn = (uint32_t) ((w + kBlock32 - 1) / kBlock32), byn = (uint32_t) ((h + kBlock32 - 1) / kBlock32), nb = bxn * byn;
    if (store.n_rank != nb) {
        if (store.spiral_rank) (void) hipFree(store.spiral_rank);
        store.spiral_rank = nullptr; store.n_rank = 0; store.rank_bx = store.rank_by = 0;
        FILM_TRY(hipMalloc((void **) &store.spiral_rank, (size_t) nb * sizeof(uint32_t)));
        store.n_rank = nb;
    }
    if (store.rank_bx != bxn || store.rank_by != byn) {      /* once per frame geometry: a merge that is timed (group, nori_amd.dist) then holds no upload and no wait */
        const std::vector<uint32_t> rank = spiral_ranks((int) bxn, (int) byn);
        FILM_TRY(hipMemcpyAsync(store.spiral_rank, rank.data(), (size_t) nb * sizeof(uint32_t), hipMemcpyHostToDevice, (hipStream_t) stream));
        FILM_TRY(hipStreamSynchronize((hipStream_t) stream));                       /* `rank` is a host temporary */
        store.rank_bx = bxn; store.rank_by = byn;
    }
    const int cols = w + 2 * border, rows = h + 2 * border;
    hipLaunchKernelGGL(film_resolve_reference_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, (hipStream_t) stream, w, h, border, bxn, byn,
                       d_block_acc, (const uint32_t *) store.spiral_rank, d_rgbw);
    FILM_TRY(hipGetLastError());
    return std::string();
}

std::string film_reference_order(FilmStore &store, const FilmStore &view, const DevScene &sc, const float *d_filter_table,
                                 uint32_t n_spp, uint32_t tiles_x, const FilmBlockRows *share, float *d_rgbw, void *stream) {
    const int w = sc.camera.width, h = sc.camera.height, border = sc.filter.border;
    const uint32_t bxn = (uint32_t) ((w + kBlock32 - 1) / kBlock32), byn = (uint32_t) ((h + kBlock32 - 1) / kBlock32);
    const size_t floats = film_block_acc_floats(sc);
    float *acc = share ? share->block_acc : nullptr;
    if (!acc) {
        if (store.block_floats < floats) {
            if (store.block_acc) (void) hipFree(store.block_acc);
            store.block_acc = nullptr; store.block_floats = 0;
            FILM_TRY(hipMalloc((void **) &store.block_acc, floats * sizeof(float)));
            store.block_floats = floats;
        }
        acc = store.block_acc;
    }
    /* the blocks of this call: all of them, or the rows of a share -- whose tiles are the store's tiles 0, 1, ... */
    const uint32_t row0 = share ? std::min(share->row_begin, byn) : 0u, rown = share ? std::min(share->row_count, byn - row0) : byn;
    const uint32_t nb = rown * bxn, block_first = row0 * bxn, tile_first = film_block_rows_first_tile(row0, tiles_x);
    FilmStore v = view; v.block_acc = acc;
    static const bool unstaged = getenv("NORI_HIP_FILM_REF_UNSTAGED") != nullptr;      /* the first implementation, for A / B: one thread per output pixel reading its sources from memory */
    if (nb == 0) { /* an empty share */ }
    else if (unstaged) hipLaunchKernelGGL(film_block_reference_kernel, dim3(nb), dim3(kB), 0, (hipStream_t) stream, w, h, sc.filter, d_filter_table, v, n_spp, tiles_x, bxn, block_first, tile_first);
    else {
        int chunk = 2 * border + 1 > 11 ? kRefChunk / 2 : kRefChunk;
        if (const char *e = getenv("NORI_HIP_FILM_REF_CHUNK")) chunk = std::min(kRefChunk, std::max(4, atoi(e)));
        hipLaunchKernelGGL(film_block_reference_staged_kernel, dim3(nb), dim3(kB), (size_t) (4 + 2 * border + 1) * chunk * kBlock32 * sizeof(float), (hipStream_t) stream,
                           w, h, sc.filter, d_filter_table, v, n_spp, tiles_x, bxn, chunk, block_first, tile_first);
    }
    FILM_TRY(hipGetLastError());
    if (share && share->block_acc) return std::string();      /* the caller merges the shares' accumulators, then film_resolve_blocks */
    return film_resolve_blocks(store, sc, acc, d_rgbw, stream);
}

unsigned long long film_invalid_count(const FilmStore &st, void *stream) {
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, st.d_invalid, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return 0;
    (void) hipStreamSynchronize((hipStream_t) stream);
    return v;
}

} // namespace nrt
