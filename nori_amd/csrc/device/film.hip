/* film.hip -- sample store, gather splat and block merge (see film.h). */
#include <hip/hip_runtime.h>

#include <algorithm>

#include "film.h"

using namespace nrt;

namespace {

constexpr int kB = 256;

/* ImageBlock::put(pos, value) as a gather.  For the tile's bordered block (tile_w^2 pixels) each
 * thread owns <= 2 output pixels and, sample round by sample round, adds the contributions of the
 * tile's 256 current samples (staged in LDS) that reach them.  Weights follow src/block.cpp:70-90
 * in the coordinates of the reference's 32x32 block containing the tile, so every filter-table
 * index is the one Nori computes:
 *     pos   = p - 0.5 - (block_offset - border)
 *     pixel x is touched iff ceil(pos.x - r) <= x <= floor(pos.x + r)   <=>  pos.x - r <= x <= pos.x + r
 *     w     = filter[(int)(|x - pos.x| * lookupFactor)] * filter[(int)(|y - pos.y| * lookupFactor)]
 *     px   += (r, g, b, 1) * wx * wy */
__global__ __launch_bounds__(kB) void film_gather_kernel(int width, int height, FilterRec fr, const float *__restrict__ filter_table,
                                                         FilmStore st, FilmLaunch fl) {
    __shared__ float s_px[256], s_py[256];
    __shared__ f4 s_L[256];                 /* w = 1: valid sample of a pixel inside the image */
    __shared__ float ftab[kFilterRes + 1];
    __shared__ unsigned int s_invalid;
    const int tid = threadIdx.x;
    if (tid <= kFilterRes) ftab[tid] = filter_table[tid];
    if (tid == 0) s_invalid = 0u;

    const uint32_t ord = fl.tile_first + blockIdx.x;
    const uint32_t tile_id = fl.tile_rem + ord * fl.tile_mod;
    const int x0 = (int) (tile_id % fl.tiles_x) * kTile, y0 = (int) (tile_id / fl.tiles_x) * kTile;
    const int border = fr.border, tile_w = fl.tile_w;
    const float radius = fr.radius, lookup = fr.lookup_factor;
    const int bx0 = x0 & ~31, by0 = y0 & ~31;                 /* NORI_BLOCK_SIZE = 32 */
    const int offx = x0 - bx0, offy = y0 - by0;               /* tile frame -> block frame */

    /* this thread's sample slot (pixel of the tile) */
    int px, py; film_tile_pixel(tid, x0, y0, px, py);
    const bool live = px < width && py < height;
    const int raster = (py - y0) * kTile + (px - x0);

    /* this thread's output pixels in the bordered tile frame */
    const int n_out = tile_w * tile_w;
    constexpr int kMaxOut = 4;                                /* (16 + 2 * 8)^2 / 256: borders up to 8 */
    int out_i[kMaxOut];
    f4 acc[kMaxOut];
    for (int o = 0; o < kMaxOut; ++o) { out_i[o] = tid + o * kB; acc[o].x = acc[o].y = acc[o].z = acc[o].w = 0.0f; }

    const size_t first = (size_t) (ord - fl.store_tile_first) * fl.n_spp * 256u;
    unsigned int invalid = 0;
    for (uint32_t s = 0; s < fl.n_spp; ++s) {
        __syncthreads();                                       /* previous round consumed */
        {
            const size_t idx = first + (size_t) s * 256u + (size_t) tid;
            f4 L; L.x = L.y = L.z = L.w = 0.0f;
            float bpx = 0.0f, bpy = 0.0f;
            if (live) {
                const f2 p = st.pos[idx];
                L = st.L[idx];
                const bool ok = color_valid(mk3(L.x, L.y, L.z));
                if (!ok) ++invalid;
                L.w = ok ? 1.0f : 0.0f;
                bpx = p.x - 0.5f - (float) (bx0 - border);
                bpy = p.y - 0.5f - (float) (by0 - border);
            }
            s_px[raster] = bpx; s_py[raster] = bpy; s_L[raster] = L;
        }
        __syncthreads();
        for (int o = 0; o < kMaxOut; ++o) {
            const int i = out_i[o];
            if (i >= n_out) break;
            const int oy = i / tile_w, ox = i - oy * tile_w;
            const float xb = (float) (ox + offx), yb = (float) (oy + offy);
            const int sx0 = max(0, ox - 2 * border), sx1 = min(kTile - 1, ox);       /* source pixels within reach */
            const int sy0 = max(0, oy - 2 * border), sy1 = min(kTile - 1, oy);
            for (int sy = sy0; sy <= sy1; ++sy) {
                for (int sx = sx0; sx <= sx1; ++sx) {
                    const int r = sy * kTile + sx;
                    const f4 L = s_L[r];
                    if (L.w == 0.0f) continue;
                    const float bx = s_px[r], by = s_py[r];
                    if (!(xb >= bx - radius && xb <= bx + radius && yb >= by - radius && yb <= by + radius)) continue;
                    const float wx = ftab[(int) (fabsf(xb - bx) * lookup)];
                    const float wy = ftab[(int) (fabsf(yb - by) * lookup)];
                    acc[o].x += L.x * wx * wy; acc[o].y += L.y * wx * wy; acc[o].z += L.z * wx * wy; acc[o].w += 1.0f * wx * wy;
                }
            }
        }
    }
    f4 *dst = reinterpret_cast<f4 *>(st.tile_acc) + (size_t) ord * n_out;
    for (int o = 0; o < kMaxOut; ++o) {
        const int i = out_i[o];
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
                                    uint32_t tile_mod, uint32_t tile_rem, const float *tile_acc, float *rgbw) {
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
            const float4 v = *reinterpret_cast<const float4 *>(tile_acc + ((size_t) ord * tile_w * tile_w + (size_t) ly * tile_w + lx) * 4);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
    float4 *dst = reinterpret_cast<float4 *>(rgbw) + (size_t) gy * cols + gx;
    float4 cur = *dst;
    cur.x += sum.x; cur.y += sum.y; cur.z += sum.z; cur.w += sum.w;
    *dst = cur;
}

FilmStore g_film;
int g_film_device = -1;

} // namespace

namespace nrt {

#define FILM_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); } while (0)

void film_release() {
    if (g_film.pos) (void) hipFree(g_film.pos);
    if (g_film.L) (void) hipFree(g_film.L);
    if (g_film.tile_acc) (void) hipFree(g_film.tile_acc);
    if (g_film.d_invalid) (void) hipFree(g_film.d_invalid);
    g_film = FilmStore();
}

std::string film_prepare(size_t n_samples, size_t n_sel_tiles, int tile_w, void *stream, FilmStore &out) {
    int dev = 0; (void) hipGetDevice(&dev);
    if (dev != g_film_device) { film_release(); g_film_device = dev; }
    if (g_film.capacity < n_samples) {
        if (g_film.pos) (void) hipFree(g_film.pos);
        if (g_film.L) (void) hipFree(g_film.L);
        g_film.pos = nullptr; g_film.L = nullptr; g_film.capacity = 0;
        FILM_TRY(hipMalloc((void **) &g_film.pos, std::max<size_t>(n_samples, 1) * sizeof(f2)));
        FILM_TRY(hipMalloc((void **) &g_film.L, std::max<size_t>(n_samples, 1) * sizeof(f4)));
        g_film.capacity = n_samples;
    }
    const size_t acc = n_sel_tiles * (size_t) tile_w * tile_w * 4;
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
    hipLaunchKernelGGL(film_gather_kernel, dim3(fl.n_tiles), dim3(kB), 0, (hipStream_t) stream, sc.camera.width, sc.camera.height,
                       sc.filter, d_filter_table, st, fl);
}

void film_resolve(const DevScene &sc, const FilmStore &st, const FilmLaunch &fl, float *d_rgbw, void *stream) {
    const int border = sc.filter.border, cols = sc.camera.width + 2 * border, rows = sc.camera.height + 2 * border;
    hipLaunchKernelGGL(film_resolve_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, (hipStream_t) stream, sc.camera.width,
                       sc.camera.height, border, fl.tile_w, fl.tiles_x, fl.tiles_y, fl.tile_mod, fl.tile_rem,
                       (const float *) st.tile_acc, d_rgbw);
}

unsigned long long film_invalid_count(const FilmStore &st, void *stream) {
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, st.d_invalid, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return 0;
    (void) hipStreamSynchronize((hipStream_t) stream);
    return v;
}

} // namespace nrt
