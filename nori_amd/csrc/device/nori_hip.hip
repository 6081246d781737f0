/*
 * nori_hip.hip -- gfx950 kernels and the C ABI of include/nori_hip.h.
 *
 * Kernel shapes (wave64, 256-thread workgroups = 4 waves):
 *
 *  render_kernel   the megakernel engine (default for small jobs): one workgroup per
 *                  (16x16 pixel tile, spp chunk); thread <-> pixel, each wave an 8x8
 *                  quad of the tile.  Every lane runs the path state machine of
 *                  rt_path.h persistently; each loop trip the wave votes (__ballot)
 *                  over what its lanes want -- shade / node test / triangle test --
 *                  and runs a kind only when enough lanes want it.  A finished
 *                  sample goes to the film's sample store (film.h) and the lane
 *                  regenerates the next camera sample of its pixel in place.  LDS
 *                  holds the per-lane traversal stacks, [depth][thread] so that
 *                  bank = lane (conflict free).
 *  wavefront.hip   the wavefront engine (default for >= 2^24 samples per call):
 *                  path state in HBM, wf_extend / wf_shade / wf_finish.
 *  film.hip        ImageBlock::put (src/block.cpp:62-102) for both engines: sample
 *                  store -> gather splat per tile -> block merge, no atomics.
 *  lbvh.hip        Accel::build on the GPU (NORI_ACCEL_GPU_LBVH).
 *  intersect_kernel / li_kernel / bsdf_* / warp_* / camera / pcg32 / splat
 *                  batch twins of the reference's virtual calls for parity tests
 *                  and for the host-side plugin classes.
 *
 * No MFMA anywhere: there is no dense contraction on this path.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/nori_hip.h"
#include "film.h"
#include "ktimer.h"
#include "shade_tables.h"
#include "rt_path.h"
#include "scene_prep.h"
#include "lbvh.h"
#include "wavefront.h"

using namespace nrt;

/* ------------------------------------------------------------ LDS stack */
template <int DEPTH, int STRIDE>
struct LdsStack {
    int *base;      /* &lds[threadIdx.x] */
    int sp;
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ void push(int v) {
        if (sp < DEPTH) base[sp * STRIDE] = v;
        sp++;
    }
    __device__ __forceinline__ int pop() {
        sp--;
        return sp < DEPTH ? base[sp * STRIDE] : 0;
    }
    __device__ __forceinline__ int pop_or(int empty_value) { return sp == 0 ? empty_value : pop(); }
};

constexpr int kBlock = 256;
#ifndef NORI_RENDER_MIN_WAVES
#define NORI_RENDER_MIN_WAVES 4   /* waves per SIMD the render kernel is register-budgeted for */
#endif

/* ----------------------------------------------------------- render kernel */
struct RenderArgs {
    uint32_t spp_begin, spp_count;
    uint32_t tile_mod, tile_rem;
    uint32_t tiles_x, tiles_y;
    uint32_t n_sel_tiles;       /* tiles handled by this launch            */
    uint32_t n_chunks;          /* spp chunks per tile                     */
    uint32_t chunk_spp;
    int32_t tile_w;             /* kTile + 2 * border                      */
    uint32_t debug_flags;       /* bit0: skip the filtered splat (experiments only) */
    uint32_t th_shade, th_inner, th_leaf;   /* lanes of a wave that must want a kind of work for it to run */
};

template <int INTEG, int STACK, bool COUNT>
__global__ __launch_bounds__(kBlock, NORI_RENDER_MIN_WAVES) void render_kernel(DevScene sc, RenderArgs args, FilmStore film,
                                                        unsigned long long *stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *lds_stack = reinterpret_cast<int *>(smem);
    unsigned int *cnt = reinterpret_cast<unsigned int *>(smem + sizeof(int) * STACK * kBlock);

    const int tid = threadIdx.x;
    if (tid < 16) cnt[tid] = 0u;
    __syncthreads();
    __shared__ uint4 s_tab[kShadeTabWords / 4];
    shade_tables_to_lds(sc, s_tab);      /* mesh / emitter tables: LDS instead of L2 round trips (shade_tables.h) */

    /* which tile / which samples */
    const uint32_t sel = blockIdx.x / args.n_chunks, chunk = blockIdx.x % args.n_chunks;
    const uint32_t tile_id = args.tile_rem + sel * args.tile_mod;
    const int x0 = (int) (tile_id % args.tiles_x) * kTile, y0 = (int) (tile_id / args.tiles_x) * kTile;
    const uint32_t s0 = args.spp_begin + chunk * args.chunk_spp;
    const uint32_t s1 = min(s0 + args.chunk_spp, args.spp_begin + args.spp_count);

    /* wave w covers the 8x8 quad (w&1, w>>1) of the tile; lane l the pixel (l&7, l>>3) */
    const int wave = tid >> 6, lane = tid & 63;
    const int px = x0 + ((wave & 1) << 3) + (lane & 7);
    const int py = y0 + ((wave >> 1) << 3) + (lane >> 3);
    const bool live = px < sc.camera.width && py < sc.camera.height;

    LdsStack<STACK, kBlock> stack;
    stack.base = lds_stack + tid; stack.sp = 0;

    PathState st;
    st.phase = PH_NEW;
    st.ray.o = st.ray.d = mk3(0.0f); st.ray.mint = st.ray.maxt = 0.0f;
    f2 pixelSample = mk2(0.0f, 0.0f);
    uint32_t s = s0;
    uint32_t nCam = 0, nClosest = 0, nShadow = 0;
    TraversalCounters tc; tc.nodes = 0; tc.tris = 0;

    /* Persistent loop.  At any time a lane wants exactly one of three kinds of
       work: SHADE (its query finished: consume the result, splat, regenerate a
       camera sample, start the next query), INNER (test a BVH node) or LEAF (test
       one triangle).  Each trip the wave votes (__ballot) and runs a kind only
       if enough lanes want it -- thresholds from RenderArgs -- so that no block
       of code executes for a handful of the 64 lanes; if no kind reaches its
       threshold the most wanted one runs, which guarantees progress. */
    Trav tv;
    trav_idle(tv);
    bool finished = !live;
    const int thS = (int) args.th_shade, thI = (int) args.th_inner, thL = (int) args.th_leaf;
    while (true) {
        const bool wantS = !trav_active(tv) && !finished;
        const bool wantI = trav_at_inner(tv);
        const bool wantL = trav_at_leaf(tv);
        const int cS = __popcll(__ballot(wantS)), cI = __popcll(__ballot(wantI)), cL = __popcll(__ballot(wantL));
        if ((cS | cI | cL) == 0) break;
        bool runS = cS >= thS, runI = cI >= thI, runL = cL >= thL;
        if (!(runS | runI | runL)) {
            runS = cS >= cI && cS >= cL;
            runI = !runS && cI >= cL;
            runL = !runS && !runI;
        }
        if (COUNT && lane == 0) {       /* scheduler census: wave-level runs and participating lanes per kind */
            if (runS) { atomicAdd(&cnt[8], 1u); atomicAdd(&cnt[9], (unsigned) cS); }
            if (runI) { atomicAdd(&cnt[10], 1u); atomicAdd(&cnt[11], (unsigned) cI); }
            if (runL) { atomicAdd(&cnt[12], 1u); atomicAdd(&cnt[13], (unsigned) cL); }
        }
        if (runS && wantS) {
            bool gen = st.phase == PH_NEW;
            if (!gen) {
                bool done;
                if (st.phase == PH_SHADOW) done = path_on_shadow(st, tv.hit.tri != kNoHit, tv.o);
                else done = path_on_closest<INTEG>(sc, st, tv.hit, tv.hit.tri != kNoHit, tv.d);
                if (done) {
                    /* block.put(pixelSample, value), src/main.cpp:52: the sample goes to the film's
                       store (20 B); film_gather applies the reconstruction filter afterwards */
                    const size_t idx = ((size_t) sel * args.spp_count + (size_t) (s - 1u - args.spp_begin)) * 256u + (size_t) tid;
                    film.pos[idx] = pixelSample;
                    P3 out; out.x = st.L.x; out.y = st.L.y; out.z = st.L.z;
                    film.L[idx] = out;
                    gen = true;
                }
            }
            if (gen) {
                if (s >= s1) {
                    finished = true;
                } else {
                    /* renderBlock, src/main.cpp:41-46: jitter, aperture draw, sampleRay */
                    rng_seed(st.rng, (uint64_t) py * (uint64_t) sc.camera.width + (uint64_t) px, (uint64_t) s);
                    const f2 j = rng_next_2d(st.rng);
                    pixelSample = mk2((float) px + j.x, (float) py + j.y);
                    (void) rng_next_2d(st.rng);        /* apertureSample: drawn, unused (perspective.cpp:78) */
                    RayIn cam;
                    camera_sample_ray(sc.camera, pixelSample, cam);
                    path_begin(st, cam);
                    ++s; ++nCam;
                }
            }
            if (!finished) {
                const bool any = st.phase == PH_SHADOW;
                if (any) ++nShadow; else ++nClosest;
                trav_begin(sc, st.ray, any, stack, tv);
            }
        }
        if (runI && trav_at_inner(tv)) trav_inner_step<COUNT>(sc, stack, tv, tc);
        if (runL && trav_at_leaf(tv)) trav_leaf_step<COUNT>(sc, stack, tv, tc);
    }

    atomicAdd(&cnt[0], nCam); atomicAdd(&cnt[1], nClosest); atomicAdd(&cnt[2], nShadow);
    if (COUNT) { atomicAdd(&cnt[3], tc.nodes); atomicAdd(&cnt[4], tc.tris); }
    __syncthreads();
    if (tid < 16 && cnt[tid] != 0u) atomicAdd(&stats[tid], (unsigned long long) cnt[tid]);
}

/* ------------------------------------------------------- batch operators */
template <int STACK>
__global__ __launch_bounds__(kBlock) void intersect_kernel(DevScene sc, const nori_ray *rays, nori_intersection *out,
                                                           size_t n, int shadow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsStack<STACK, kBlock> stack;
    stack.base = reinterpret_cast<int *>(smem) + threadIdx.x; stack.sp = 0;
    TraversalCounters tc; tc.nodes = tc.tris = 0;
    for (size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t) gridDim.x * kBlock) {
        const nori_ray r = rays[i];
        RayIn ray; ray.o = mk3(r.o[0], r.o[1], r.o[2]); ray.d = mk3(r.d[0], r.d[1], r.d[2]);
        ray.mint = r.mint; ray.maxt = r.maxt;
        Hit hit;
        const bool found = traverse<false>(sc, ray, shadow != 0, stack, hit, tc);
        nori_intersection o;
        memset(&o, 0, sizeof(o));
        o.mesh = NORI_NO_HIT; o.tri = NORI_NO_HIT;
        if (found && shadow) {
            o.mesh = 0;
        } else if (found) {
            Surface sf; f3 ng; f2 uv;
            surface_fill(sc, hit, sf, &ng, &uv);
            const Frame sh = make_frame(sf.ns), geo = make_frame(ng);
            o.p[0] = sf.p.x; o.p[1] = sf.p.y; o.p[2] = sf.p.z; o.t = hit.t; o.uv[0] = uv.x; o.uv[1] = uv.y;
            o.sh_s[0] = sh.s.x; o.sh_s[1] = sh.s.y; o.sh_s[2] = sh.s.z;
            o.sh_t[0] = sh.t.x; o.sh_t[1] = sh.t.y; o.sh_t[2] = sh.t.z;
            o.sh_n[0] = sh.n.x; o.sh_n[1] = sh.n.y; o.sh_n[2] = sh.n.z;
            o.geo_s[0] = geo.s.x; o.geo_s[1] = geo.s.y; o.geo_s[2] = geo.s.z;
            o.geo_t[0] = geo.t.x; o.geo_t[1] = geo.t.y; o.geo_t[2] = geo.t.z;
            o.geo_n[0] = geo.n.x; o.geo_n[1] = geo.n.y; o.geo_n[2] = geo.n.z;
            o.mesh = hit.mesh; o.tri = hit.tri - sc.meshes[hit.mesh].tri_offset;
        }
        out[i] = o;
    }
}

template <int INTEG, int STACK>
__global__ __launch_bounds__(kBlock) void li_kernel(DevScene sc, const nori_ray *rays, size_t n,
                                                    const uint64_t *seed_state, const uint64_t *seed_seq, float *rgb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsStack<STACK, kBlock> stack;
    stack.base = reinterpret_cast<int *>(smem) + threadIdx.x; stack.sp = 0;
    TraversalCounters tc; tc.nodes = tc.tris = 0;
    for (size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t) gridDim.x * kBlock) {
        const nori_ray r = rays[i];
        RayIn ray; ray.o = mk3(r.o[0], r.o[1], r.o[2]); ray.d = mk3(r.d[0], r.d[1], r.d[2]);
        ray.mint = r.mint; ray.maxt = r.maxt;
        PathState st;
        rng_seed(st.rng, seed_state[i], seed_seq[i]);
        path_begin(st, ray);
        while (true) {
            const bool any = st.phase == PH_SHADOW;
            Hit hit;
            const f3 qo = st.ray.o, qd = st.ray.d;
            const bool found = traverse<false>(sc, st.ray, any, stack, hit, tc);
            const bool done = any ? path_on_shadow(st, found, qo) : path_on_closest<INTEG>(sc, st, hit, found, qd);
            if (done) break;
        }
        rgb[3 * i] = st.L.x; rgb[3 * i + 1] = st.L.y; rgb[3 * i + 2] = st.L.z;
    }
}

/* NORI_SEED_NORI_BLOCK -- the reference's own sampler streams.  Independent::prepare(block) seeds ONE pcg32 stream per
   32x32 block with (offset.x, offset.y) (src/independent.cpp:36-41) and renderBlock (src/main.cpp:27-56) consumes it
   serially: for y, for x, for sample: next2D (pixel jitter), next2D (aperture, unused), then whatever Li draws.  The
   number of draws per sample depends on the path, so the stream cannot be split: this mode runs ONE LANE PER BLOCK,
   each walking its block's pixels and samples in the reference's order with the block's stream.  Slow by design (a
   1024^2 frame is 1024 lanes of work) -- it exists for bit-level comparison with a render of the reference / the
   oracle in its native seeding, not for throughput.  Samples go to the film's store like everywhere else. */
constexpr int kNoriBlock = 32;        /* NORI_BLOCK_SIZE, include/nori/block.h:17 */

template <int INTEG, int STACK>
__global__ __launch_bounds__(64) void render_block_serial_kernel(DevScene sc, uint32_t blocks_x, uint32_t n_blocks, uint32_t spp,
                                                                 uint32_t tiles_x, FilmStore film, unsigned long long *stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsStack<STACK, 64> stack;
    stack.base = reinterpret_cast<int *>(smem) + threadIdx.x; stack.sp = 0;
    TraversalCounters tc; tc.nodes = tc.tris = 0;
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    unsigned long long nCam = 0, nClosest = 0, nShadow = 0;
    if (b < n_blocks) {
        const int ox = (int) (b % blocks_x) * kNoriBlock, oy = (int) (b / blocks_x) * kNoriBlock;
        const int sx = min(kNoriBlock, sc.camera.width - ox), sy = min(kNoriBlock, sc.camera.height - oy);
        Rng rng; rng_seed(rng, (uint64_t) ox, (uint64_t) oy);
        for (int y = 0; y < sy; ++y)
            for (int x = 0; x < sx; ++x) {
                const int px = x + ox, py = y + oy;
                const uint32_t tile = (uint32_t) (py / kTile) * tiles_x + (uint32_t) (px / kTile);
                const int lx = px % kTile, ly = py % kTile;
                const uint32_t pix = (uint32_t) ((((lx >> 3) | ((ly >> 3) << 1)) << 6) | ((lx & 7) | ((ly & 7) << 3)));      /* inverse of film_tile_pixel */
                for (uint32_t i = 0; i < spp; ++i) {
                    const f2 j = rng_next_2d(rng);
                    const f2 ps = mk2((float) px + j.x, (float) py + j.y);
                    (void) rng_next_2d(rng);                       /* apertureSample */
                    RayIn cam; camera_sample_ray(sc.camera, ps, cam);
                    PathState st; st.rng = rng;
                    path_begin(st, cam);
                    while (true) {
                        const bool any = st.phase == PH_SHADOW;
                        Hit hit;
                        const f3 qo = st.ray.o, qd = st.ray.d;
                        if (any) ++nShadow; else ++nClosest;
                        const bool found = traverse<false>(sc, st.ray, any, stack, hit, tc);
                        const bool done = any ? path_on_shadow(st, found, qo) : path_on_closest<INTEG>(sc, st, hit, found, qd);
                        if (done) break;
                    }
                    rng = st.rng;                                  /* the stream goes on where Li left it */
                    const size_t idx = ((size_t) tile * spp + i) * 256u + pix;
                    film.pos[idx] = ps;
                    P3 L; L.x = st.L.x; L.y = st.L.y; L.z = st.L.z;
                    film.L[idx] = L;
                    ++nCam;
                }
            }
    }
    if (nCam) { atomicAdd(&stats[0], nCam); atomicAdd(&stats[1], nClosest); atomicAdd(&stats[2], nShadow); }
}

__global__ void sample_rays_kernel(CameraRec cam, const float *ps, size_t n, nori_ray *rays) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RayIn r;
    camera_sample_ray(cam, mk2(ps[2 * i], ps[2 * i + 1]), r);
    nori_ray o;
    o.o[0] = r.o.x; o.o[1] = r.o.y; o.o[2] = r.o.z; o.d[0] = r.d.x; o.d[1] = r.d.y; o.d[2] = r.d.z;
    o.mint = r.mint; o.maxt = r.maxt;
    rays[i] = o;
}

__global__ void bsdf_kernel(Bsdf b, int op, const float *wi, const float *wo_in, const float *sample, size_t n,
                            float *wo_out, float *value, float *eta, int32_t *measure) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 a = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
    if (op == 0) {
        f3 wo; float e; int m;
        const f3 w = bsdf_sample(b, a, mk2(sample[2 * i], sample[2 * i + 1]), wo, e, m);
        wo_out[3 * i] = wo.x; wo_out[3 * i + 1] = wo.y; wo_out[3 * i + 2] = wo.z;
        value[3 * i] = w.x; value[3 * i + 1] = w.y; value[3 * i + 2] = w.z;
        if (eta) eta[i] = e;
        if (measure) measure[i] = m;
    } else {
        const f3 o = mk3(wo_in[3 * i], wo_in[3 * i + 1], wo_in[3 * i + 2]);
        if (op == 1) { const f3 v = bsdf_eval(b, a, o); value[3 * i] = v.x; value[3 * i + 1] = v.y; value[3 * i + 2] = v.z; }
        else value[i] = bsdf_pdf(b, a, o);
    }
}

__global__ void warp_kernel(int warp, float param, int pdf, const float *in, size_t n, float *out) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!pdf) {
        const f3 r = warp_dispatch(warp, param, mk2(in[2 * i], in[2 * i + 1]));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    } else {
        out[i] = warp_pdf_dispatch(warp, param, mk3(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
    }
}

__global__ void pcg32_kernel(const uint64_t *state, const uint64_t *seq, uint64_t skip, size_t n, uint32_t count, float *out) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r; rng_seed(r, state[i], seq[i]);
    if (skip) rng_advance(r, skip);
    for (uint32_t j = 0; j < count; ++j) out[i * count + j] = rng_next_float(r);
}

/* ImageBlock::put(pos, value) straight into the global frame: the batch twin of the film (film.hip
 * turns the same weights into a gather over a sample store). */
__global__ void splat_kernel(FilterRec fl, const float *ftab, int width, int height, const float *pos, const float *val,
                             size_t n, float *rgbw) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 value = mk3(val[3 * i], val[3 * i + 1], val[3 * i + 2]);
    if (!color_valid(value)) return;
    const int border = fl.border, cols = width + 2 * border, rows = height + 2 * border;
    const float px = pos[2 * i] - 0.5f - (float) (0 - border), py = pos[2 * i + 1] - 0.5f - (float) (0 - border);
    int minX = max((int) ceilf(px - fl.radius), 0), maxX = min((int) floorf(px + fl.radius), cols - 1);
    int minY = max((int) ceilf(py - fl.radius), 0), maxY = min((int) floorf(py + fl.radius), rows - 1);
    for (int y = minY; y <= maxY; ++y) {
        const float wy = ftab[(int) (fabsf((float) y - py) * fl.lookup_factor)];
        for (int x = minX; x <= maxX; ++x) {
            const float wx = ftab[(int) (fabsf((float) x - px) * fl.lookup_factor)];
            float *p = rgbw + (((size_t) y * cols + x) << 2);
            unsafeAtomicAdd(p + 0, value.x * wx * wy); unsafeAtomicAdd(p + 1, value.y * wx * wy);
            unsafeAtomicAdd(p + 2, value.z * wx * wy); unsafeAtomicAdd(p + 3, 1.0f * wx * wy);
        }
    }
}

/* ImageBlock::toBitmap, src/block.cpp:45-51 */
__global__ void develop_kernel(const float4 *rgbw, float *rgb, int width, int height, int border) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    const float4 p = rgbw[(size_t) (y + border) * (width + 2 * border) + (x + border)];
    float *o = rgb + ((size_t) y * width + x) * 3;
    if (p.w != 0.0f) { o[0] = p.x / p.w; o[1] = p.y / p.w; o[2] = p.z / p.w; }
    else { o[0] = o[1] = o[2] = 0.0f; }
}

/* ---------------------------------------------------------------- context */
struct nori_hip_ctx {
    int device = 0;
    std::string error;
    bool have_scene = false, have_accel = false;
    HostScene host;
    HostBvh bvh;
    DevScene dev;
    std::vector<void *> allocs_scene, allocs_accel;
    float *d_filter = nullptr;
    const uint32_t *d_tri_mesh = nullptr;
    unsigned long long *d_stats = nullptr;
    nori_accel_info info;
    int stack_depth = 32;
    uint64_t lbvh_bytes = 0;
    uint32_t lbvh_refs = 0;      /* references the device builder built the tree over (0: the host built it) */
    int engine = -1;                /* -1 auto, 0 megakernel, 1 wavefront */
    int accel_layout = -1;          /* -1 auto (wide from 2^20 triangles), 0 bvh2, 1 bvh4q (wide nodes) */
    bool film_reference = false;    /* film_order = reference: samples added in the reference's own order (film.h) */
    /* 216 B of state each (two copies of the 100-B record + the hit record) + 20 B of film, twice when a call has several batches (the
       sample store's two halves, wavefront.hip "tail overlap"): 2^29 paths = 137 GB of the 288 GB.
       A batch ends with a tail that lasts as long as its longest path (wf_finish: ~19 ms on the pa5 table scene), so fewer,
       bigger batches are cheaper: the table scene at 2048^2 x 1024 spp is 8 batches instead of the 16 of 2^28 */
    size_t wavefront_paths = (size_t) 1 << 29;
    /* camera samples per batch (0: as many as the pool holds paths -- every batch starts all its samples in its first pass, the
       fastest schedule: profiles/r6_12_pool_sweep_two_launches.txt): 20 B each in the film's sample store, twice when a call has several
       batches.  A batch bigger than the pool starts its samples pass by pass in the slots finished paths leave (regeneration,
       wavefront.hip): the pool bounds the STATE, this bounds the film store.  What it is for: a context short of memory keeps its
       batches -- hence the bits of its frame -- on a smaller pool (the out-of-memory retry below), and film_order = reference,
       which needs the frame in ONE batch, runs on any pool */
    size_t wavefront_samples = 0;
    /* render-time resources of THIS context (never shared, freed in nori_hip_destroy): the wavefront
       engine's state pool / streams / events and the film's sample store + tile accumulators */
    WfEngine *wf = nullptr;
    FilmStore film;
};

static std::string g_create_error;

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e__);                     \
            return e__ == hipErrorOutOfMemory ? NORI_ERR_OUT_OF_MEMORY :                          \
                   (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice) ? NORI_ERR_NO_DEVICE : NORI_ERR_INTERNAL; \
        }                                                                                          \
    } while (0)

template <class T>
static int upload(nori_hip_ctx *ctx, std::vector<void *> &pool, const std::vector<T> &v, const T **out) {
    void *d = nullptr;
    const size_t bytes = (std::max<size_t>(v.size() * sizeof(T), 16) + 15) & ~(size_t) 15;      /* whole 16-B words: tables are copied as uint4 */
    HIP_TRY(ctx, hipMalloc(&d, bytes));
    pool.push_back(d);
    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(d);
    return NORI_OK;
}

/* rt_top.h: the image of the tree's hottest records (any builder, either layout), once per acceleration structure.  One wave:
   every lane runs the selection (identically -- they all write the same values), the scan for the next best candidate is
   shared out over the lanes; the candidate lists live in LDS. */
struct TopWaveArgMax {
    template <class F> __device__ int operator()(int n, F score) const {
        int best = -1; float bs = -1.0f;
        for (int i = (int) (threadIdx.x & 63u); i < n; i += 64) { const float sc = score(i); if (sc > bs) { bs = sc; best = i; } }
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off); const int oi = __shfl_xor(best, off);
            if (os > bs || (os == bs && oi >= 0 && (best < 0 || oi < best))) { bs = os; best = oi; }
        }
        return best;
    }
};
__global__ __launch_bounds__(64) void top_image_kernel(DevScene sc, const f4 *records, TopLayout layout, int max_nodes, f4 *image) {
    __shared__ int32_t s_link[kTopMaxCand]; __shared__ float s_area[kTopMaxCand];
    __shared__ int16_t s_parent[kTopMaxCand]; __shared__ int8_t s_which[kTopMaxCand];
    TopWork w; w.link = s_link; w.area = s_area; w.parent = s_parent; w.which = s_which;
    top_image_build_with(sc.nodes, records, layout, sc.tris, sc.root, sc.wide != 0u, sc.n_triangles, max_nodes, image, w, TopWaveArgMax());
}

/* rt_nodeq.h: the 32-B records of a BVH2 tree -- the grid from the root's children, then one thread per node.
   status[0] = 1: the grid is valid; status[1] != 0: some node has an unbounded box (the tree does not qualify) */
__global__ void nodeq_grid_kernel(DevScene sc, NodeqGrid *grid, uint32_t *status) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { NodeqGrid g; const bool ok = nodeq_grid(sc.nodes, sc.root, g); if (ok) *grid = g; status[0] = ok ? 1u : 0u; status[1] = 0u; }
}
__global__ void nodeq_convert_kernel(const f4 *nodes, uint32_t n_nodes, const NodeqGrid *grid, const uint32_t *status_in, f4 *out, uint32_t *status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || status_in[0] == 0u) return;
    const NodeqGrid g = *grid;
    f4 q[4] = {nodes[(size_t) i * kNodeQuads], nodes[(size_t) i * kNodeQuads + 1], nodes[(size_t) i * kNodeQuads + 2], nodes[(size_t) i * kNodeQuads + 3]};
    f4 r[2];
    if (!nodeq_from_node(q, g, r)) { atomicAdd(&status[1], 1u); r[0].x = r[0].y = r[0].z = r[0].w = 0.0f; r[1] = r[0]; }
    out[(size_t) i * kNodeqQuads] = r[0]; out[(size_t) i * kNodeqQuads + 1] = r[1];
}

static void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) (void) hipFree(p);
    pool.clear();
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void) hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

extern "C" {

int nori_hip_abi_version(void) { return NORI_HIP_ABI_VERSION; }

int nori_hip_create(int device, nori_hip_ctx **out) {
    if (!out) return NORI_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return NORI_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { g_create_error = "device index out of range"; return NORI_ERR_INVALID_ARGUMENT; }
    nori_hip_ctx *ctx = new nori_hip_ctx();
    ctx->device = device;
    memset(&ctx->dev, 0, sizeof(ctx->dev));
    memset(&ctx->info, 0, sizeof(ctx->info));
    DeviceGuard g(device);
    e = hipMalloc((void **) &ctx->d_stats, 16 * sizeof(unsigned long long));
    if (e != hipSuccess) { g_create_error = std::string("hipMalloc: ") + hipGetErrorString(e); delete ctx; return NORI_ERR_NO_DEVICE; }
    ctx->wf = wavefront_create();
    *out = ctx;
    return NORI_OK;
}

void nori_hip_destroy(nori_hip_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    free_pool(ctx->allocs_scene); free_pool(ctx->allocs_accel);
    wavefront_destroy(ctx->wf);
    film_release(ctx->film);
    if (ctx->d_stats) (void) hipFree(ctx->d_stats);
    delete ctx;
}

const char *nori_hip_last_error(const nori_hip_ctx *ctx) {
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int nori_hip_upload_scene(nori_hip_ctx *ctx, const nori_scene_desc *scene) {
    if (!ctx || !scene) return NORI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    free_pool(ctx->allocs_scene); free_pool(ctx->allocs_accel);
    ctx->have_scene = ctx->have_accel = false;
    std::string err = prepare_scene(*scene, ctx->host);
    if (!err.empty()) { ctx->error = err; return NORI_ERR_INVALID_ARGUMENT; }
    HostScene &h = ctx->host;
    DevScene &d = ctx->dev;
    memset(&d, 0, sizeof(d));
    int rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.positions, &d.positions))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.normals, &d.normals))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.texcoords, &d.texcoords))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.indices, &d.indices))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.meshes, &d.meshes))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.emitter_cdf, &d.emitter_cdf))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.emitters, &d.emitters))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.tri_mesh, &ctx->d_tri_mesh))) return rc;
    if ((rc = upload(ctx, ctx->allocs_scene, h.shade_tris, &d.shade_tris))) return rc;
    std::vector<float> ft(h.filter.table, h.filter.table + kFilterRes + 1);
    const float *dft = nullptr;
    if ((rc = upload(ctx, ctx->allocs_scene, ft, &dft))) return rc;
    ctx->d_filter = const_cast<float *>(dft);
    d.tri_mesh = ctx->d_tri_mesh;
    d.n_emitters = (uint32_t) h.emitters.size();
    d.n_meshes = (uint32_t) h.meshes.size();
    d.bsdf_mask = 0u; for (const MeshRec &m : h.meshes) d.bsdf_mask |= 1u << (uint32_t) m.bsdf_type;
    d.n_triangles = (uint32_t) h.tri_mesh.size();
    d.n_cdf = (uint32_t) h.emitter_cdf.size();
    d.camera = h.camera; d.filter = h.filter; d.integrator = h.integrator;
    ctx->have_scene = true;
    return NORI_OK;
}

int nori_hip_build_accel(nori_hip_ctx *ctx, int builder) {
    if (!ctx) return NORI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) { ctx->error = "build_accel: no scene uploaded"; return NORI_ERR_NOT_READY; }
    /* auto: the device's builder.  With triangle splitting and parallel re-insertion (round 6, lbvh.hip) its trees trace within 1 % of the
       host's on the Cornell box, the pa5 table and the AO scene and within 3 % on the 10 M-triangle terrain, and are built in 19 / 28 / 33 /
       144 ms against 30 / 56 / 90 / 2 400 (profiles/r6_19_split_scale.txt); the host's builder if the device's gives up (a tree deeper than the
       traversal stack) */
    const bool was_auto = builder == NORI_ACCEL_AUTO;
    if (was_auto) builder = ctx->dev.n_triangles > 0 ? NORI_ACCEL_GPU_PLOC : NORI_ACCEL_HOST_SAH;
    if (builder != NORI_ACCEL_HOST_SAH && builder != NORI_ACCEL_GPU_LBVH && builder != NORI_ACCEL_GPU_PLOC) { ctx->error = "build_accel: unknown builder"; return NORI_ERR_INVALID_ARGUMENT; }
    DeviceGuard g(ctx->device);
    free_pool(ctx->allocs_accel);
    ctx->have_accel = false;
    int layout = ctx->accel_layout;
    if (const char *e = getenv("NORI_HIP_ACCEL_LAYOUT")) layout = std::string(e) == "bvh4q" ? 1 : (std::string(e) == "bvh2" ? 0 : -1);
    /* measured (wf_extend, BVH2 walked as 32-B records by the hand-written loop against wide nodes): 328 k triangles 9.7 against
       11.7 ms, 10 M triangles 32.2 against 28.9 ms */
    const bool want_wide = layout == 1 || (layout < 0 && ctx->dev.n_triangles >= (1u << 20));
    if (builder != NORI_ACCEL_HOST_SAH && ctx->dev.n_triangles > 0) {
        LbvhDeviceResult res;
        uint32_t ploc_radius = builder == NORI_ACCEL_GPU_PLOC ? 8u : 0u;      /* 8, 16, 32 give the same trees within 1 % (tools/builder_probe.py); 8 builds fastest */
        if (const char *e = getenv("NORI_HIP_PLOC_RADIUS")) if (ploc_radius) ploc_radius = (uint32_t) std::min(256, std::max(1, atoi(e)));
        std::string err = build_bvh_lbvh_device(ctx->dev, ctx->d_tri_mesh, res, want_wide, ploc_radius);
        if (err.empty() && res.wide && res.max_depth + 1 > 64) {      /* wide tree needs more stack than the kernels have: BVH2 nodes */
            if (res.d_nodes) (void) hipFree(res.d_nodes);
            if (res.d_tris) (void) hipFree(res.d_tris);
            err = build_bvh_lbvh_device(ctx->dev, ctx->d_tri_mesh, res, false, ploc_radius);
        }
        if (err.empty() && ploc_radius && res.max_depth + 1 > 64) {   /* a clustering of near-identical boxes can degenerate into a chain: the radix tree instead */
            if (res.d_nodes) (void) hipFree(res.d_nodes);
            if (res.d_tris) (void) hipFree(res.d_tris);
            err = build_bvh_lbvh_device(ctx->dev, ctx->d_tri_mesh, res, want_wide, 0u);
            if (err.empty() && res.wide && res.max_depth + 1 > 64) {
                if (res.d_nodes) (void) hipFree(res.d_nodes);
                if (res.d_tris) (void) hipFree(res.d_tris);
                err = build_bvh_lbvh_device(ctx->dev, ctx->d_tri_mesh, res, false, 0u);
            }
        }
        if (res.d_nodes) ctx->allocs_accel.push_back(res.d_nodes);
        if (res.d_tris) ctx->allocs_accel.push_back(res.d_tris);
        if (err.empty() && res.max_depth + 1 > 64) err = "LBVH deeper than the traversal stack (64); use NORI_ACCEL_HOST_SAH";
        if (!err.empty() && was_auto) { free_pool(ctx->allocs_accel); return nori_hip_build_accel(ctx, NORI_ACCEL_HOST_SAH); }
        if (!err.empty()) { ctx->error = err; free_pool(ctx->allocs_accel); return NORI_ERR_INTERNAL; }
        ctx->bvh = HostBvh();
        ctx->bvh.wide = res.wide;
        ctx->bvh.root = res.root; ctx->bvh.n_nodes = res.n_nodes; ctx->bvh.n_leaves = res.n_leaves;
        ctx->bvh.max_depth = res.max_depth; ctx->bvh.build_ms = res.build_ms; ctx->bvh.sah_cost = 0.0f;
        ctx->dev.nodes = res.d_nodes; ctx->dev.tris = res.d_tris;
        ctx->lbvh_bytes = (uint64_t) std::max<uint32_t>(res.n_nodes, 1) * kNodeQuads * 16 + (uint64_t) res.n_pairs * kPairQuads * 16;
        ctx->lbvh_refs = res.n_refs;
    } else {
        const bool wide = want_wide;
        std::string err = build_bvh_sah(ctx->host, 64, ctx->bvh, wide);
        if (!err.empty() && wide) err = build_bvh_sah(ctx->host, 64, ctx->bvh, false);      /* wide tree too deep for the stack: BVH2 */
        if (!err.empty()) { ctx->error = err; return NORI_ERR_INTERNAL; }
        int rc;
        if ((rc = upload(ctx, ctx->allocs_accel, ctx->bvh.nodes, &ctx->dev.nodes))) return rc;
        if ((rc = upload(ctx, ctx->allocs_accel, ctx->bvh.tris, &ctx->dev.tris))) return rc;
        ctx->lbvh_bytes = 0; ctx->lbvh_refs = 0;
    }
    if (const char *e = getenv("NORI_HIP_LAB_DEPTH_ADD")) ctx->bvh.max_depth += (uint32_t) std::max(0, atoi(e));      /* experiments: a tree priced as if it were deeper (the spilling stack, the smaller LDS image) */
    ctx->dev.root = ctx->bvh.root;
    ctx->dev.wide = ctx->bvh.wide ? 1u : 0u;
    ctx->dev.top_image = nullptr; ctx->dev.top_image_quads = 0u;
    ctx->dev.nodes_q = nullptr; ctx->dev.top_image_q = nullptr; ctx->dev.top_image_q_quads = 0u;
    const bool want_image = ctx->dev.n_triangles > 0 && getenv("NORI_HIP_NO_TOP_IMAGE") == nullptr;
    if (ctx->dev.n_triangles > 0 && !ctx->bvh.wide && ctx->bvh.root >= 0 && ctx->bvh.n_nodes > 0 && getenv("NORI_HIP_NO_NODEQ") == nullptr) {
        /* the nodes once more as 32-B records (rt_nodeq.h): what wf_extend's node loop reads */
        void *nq = nullptr, *aux = nullptr;
        HIP_TRY(ctx, hipMalloc(&nq, (size_t) ctx->bvh.n_nodes * kNodeqQuads * sizeof(f4)));
        ctx->allocs_accel.push_back(nq);
        HIP_TRY(ctx, hipMalloc(&aux, sizeof(NodeqGrid) + 2 * sizeof(uint32_t)));
        NodeqGrid *d_grid = reinterpret_cast<NodeqGrid *>(aux);
        uint32_t *d_status = reinterpret_cast<uint32_t *>(d_grid + 1);
        hipLaunchKernelGGL(nodeq_grid_kernel, dim3(1), dim3(64), 0, 0, ctx->dev, d_grid, d_status);
        hipLaunchKernelGGL(nodeq_convert_kernel, dim3((ctx->bvh.n_nodes + 255u) / 256u), dim3(256), 0, 0, ctx->dev.nodes, (uint32_t) ctx->bvh.n_nodes, d_grid, d_status,
                           reinterpret_cast<f4 *>(nq), d_status);
        struct { NodeqGrid g; uint32_t status[2]; } h;
        const hipError_t e1 = hipGetLastError(), e2 = hipMemcpy(&h, aux, sizeof(h), hipMemcpyDeviceToHost);      /* (synchronises) */
        (void) hipFree(aux);
        HIP_TRY(ctx, e1); HIP_TRY(ctx, e2);
        if (h.status[0] == 1u && h.status[1] == 0u) { ctx->dev.nodes_q = reinterpret_cast<const f4 *>(nq); ctx->dev.grid = h.g; }
    }
    if (want_image) {      /* what wf_extend keeps in LDS (rt_top.h): the hot records, with 64-B and -- if the tree has them -- 32-B nodes */
        int limit = kTopMaxNodes;
        if (const char *e = getenv("NORI_HIP_TOP_NODES")) limit = std::max(1, atoi(e));
        for (int q = 0; q < (ctx->dev.nodes_q ? 2 : 1); ++q) {
            void *img = nullptr;
            HIP_TRY(ctx, hipMalloc(&img, kTopImageMaxQuads * sizeof(f4)));
            ctx->allocs_accel.push_back(img);
            const int max_nodes = std::min(limit, wf_top_capacity(ctx->bvh.wide, q == 1, ctx->bvh.max_depth + 1 > 16));
            hipLaunchKernelGGL(top_image_kernel, dim3(1), dim3(64), 0, 0, ctx->dev, q == 1 ? ctx->dev.nodes_q : ctx->dev.nodes, top_layout(q == 1), max_nodes,
                               reinterpret_cast<f4 *>(img));
            HIP_TRY(ctx, hipGetLastError());
            f4 header;
            HIP_TRY(ctx, hipMemcpy(&header, img, sizeof(f4), hipMemcpyDeviceToHost));      /* (synchronises) */
            if (q == 1) { ctx->dev.top_image_q = reinterpret_cast<const f4 *>(img); ctx->dev.top_image_q_quads = f2u(header.w); }
            else { ctx->dev.top_image = reinterpret_cast<const f4 *>(img); ctx->dev.top_image_quads = f2u(header.w); }
        }
    }
    ctx->stack_depth = ctx->bvh.max_depth + 1 <= 32 ? 32 : 64;
    nori_accel_info &in = ctx->info;
    in.n_triangles = ctx->dev.n_triangles; in.n_nodes = ctx->bvh.n_nodes; in.n_leaves = ctx->bvh.n_leaves;
    in.max_depth = ctx->bvh.max_depth; in.node_bytes = kNodeQuads * 16; in.tri_bytes = kPairQuads * 16 / 2;
    in.total_bytes = ctx->lbvh_bytes ? ctx->lbvh_bytes : (uint64_t) ctx->bvh.nodes.size() * 16 + (uint64_t) ctx->bvh.tris.size() * 16;
    in.build_ms = ctx->bvh.build_ms; in.sah_cost = ctx->bvh.sah_cost;
    in.node_children = ctx->bvh.wide ? 4u : 2u; in.node_records_32b = ctx->dev.nodes_q != nullptr ? 1u : 0u;
    in.built_on_device = ctx->lbvh_bytes ? 1u : 0u; in.n_references = ctx->lbvh_refs;
    ctx->have_accel = true;
    return NORI_OK;
}

int nori_hip_debug_excursions(nori_hip_ctx *ctx, unsigned long long out[4], int reset) {
    if (!ctx || !out) return NORI_ERR_INVALID_ARGUMENT;
    out[0] = out[1] = out[2] = out[3] = 0ull;
#if defined(NORI_COUNT_EXCURSIONS)
    DeviceGuard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_nori_excursions), sizeof(h)));      /* this translation unit's kernels */
    for (int k = 0; k < 4; ++k) out[k] += h[k];
    if (reset) { const unsigned long long z[4] = {0, 0, 0, 0}; HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_nori_excursions), z, sizeof(z))); }
    if (!wavefront_excursions(out, reset != 0)) { ctx->error = "debug_excursions: reading the wavefront engine's counters failed"; return NORI_ERR_INTERNAL; }
    return NORI_OK;
#else
    (void) reset;
    ctx->error = "debug_excursions: this library was built without -DNORI_COUNT_EXCURSIONS (use libnori_hip_count.so)";
    return NORI_ERR_UNSUPPORTED;
#endif
}

int nori_hip_set_option(nori_hip_ctx *ctx, const char *key, const char *value) {
    if (!ctx || !key || !value) return NORI_ERR_INVALID_ARGUMENT;
    const std::string k(key), v(value);
    if (k == "engine") {
        if (v == "megakernel") ctx->engine = 0;
        else if (v == "wavefront") ctx->engine = 1;
        else if (v == "auto") ctx->engine = -1;
        else { ctx->error = "set_option: engine must be auto, megakernel or wavefront"; return NORI_ERR_INVALID_ARGUMENT; }
        return NORI_OK;
    }
    if (k == "wavefront_paths") {
        const long long n = atoll(value);
        if (n < 256) { ctx->error = "set_option: wavefront_paths must be >= 256"; return NORI_ERR_INVALID_ARGUMENT; }
        ctx->wavefront_paths = (size_t) n;
        return NORI_OK;
    }
    if (k == "wavefront_samples") {
        const long long n = atoll(value);
        if (n != 0 && (n < 256 || n > (1ll << 31))) { ctx->error = "set_option: wavefront_samples must be 0 (as wavefront_paths) or in [256, 2^31]"; return NORI_ERR_INVALID_ARGUMENT; }
        ctx->wavefront_samples = (size_t) n;
        return NORI_OK;
    }
    if (k == "film_order") {
        if (v == "fast") ctx->film_reference = false;
        else if (v == "reference") ctx->film_reference = true;
        else { ctx->error = "set_option: film_order must be fast or reference"; return NORI_ERR_INVALID_ARGUMENT; }
        return NORI_OK;
    }
    if (k == "accel_layout") {
        if (v == "bvh2") ctx->accel_layout = 0;
        else if (v == "bvh4q") ctx->accel_layout = 1;
        else if (v == "auto") ctx->accel_layout = -1;
        else { ctx->error = "set_option: accel_layout must be auto, bvh2 or bvh4q"; return NORI_ERR_INVALID_ARGUMENT; }
        return NORI_OK;
    }
    ctx->error = "set_option: unknown key " + k;
    return NORI_ERR_INVALID_ARGUMENT;
}

int nori_hip_get_option(const nori_hip_ctx *ctx, const char *key, char *value, size_t capacity) {
    if (!ctx || !key || !value || capacity == 0) return NORI_ERR_INVALID_ARGUMENT;
    const std::string k(key);
    std::string v;
    if (k == "engine") v = ctx->engine == 0 ? "megakernel" : ctx->engine == 1 ? "wavefront" : "auto";
    else if (k == "wavefront_paths") v = std::to_string(ctx->wavefront_paths);
    else if (k == "wavefront_samples") v = std::to_string(ctx->wavefront_samples);
    else if (k == "film_order") v = ctx->film_reference ? "reference" : "fast";
    else if (k == "accel_layout") v = ctx->accel_layout == 0 ? "bvh2" : ctx->accel_layout == 1 ? "bvh4q" : "auto";
    else return NORI_ERR_INVALID_ARGUMENT;
    if (v.size() + 1 > capacity) return NORI_ERR_INVALID_ARGUMENT;
    std::memcpy(value, v.c_str(), v.size() + 1);
    return NORI_OK;
}

int nori_hip_accel_info(const nori_hip_ctx *ctx, nori_accel_info *out) {
    if (!ctx || !out) return NORI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_accel) return NORI_ERR_NOT_READY;
    *out = ctx->info;
    return NORI_OK;
}

int nori_hip_border_size(const nori_hip_ctx *ctx) {
    if (!ctx || !ctx->have_scene) return NORI_ERR_NOT_READY;
    return ctx->host.filter.border;
}

#define REQUIRE_ACCEL(ctx)                                                                 \
    do {                                                                                   \
        if (!(ctx)) return NORI_ERR_INVALID_ARGUMENT;                                      \
        if (!(ctx)->have_accel) { (ctx)->error = "no acceleration structure built"; return NORI_ERR_NOT_READY; } \
    } while (0)

static int grid_for(size_t n) { return (int) std::min<size_t>((n + kBlock - 1) / kBlock, 8192); }

int nori_hip_intersect_device(nori_hip_ctx *ctx, const void *d_rays, void *d_its, size_t n, int shadow_ray, void *stream) {
    REQUIRE_ACCEL(ctx);
    if (n == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t) stream;
    if (ctx->stack_depth <= 32)
        hipLaunchKernelGGL(intersect_kernel<32>, dim3(grid_for(n)), dim3(kBlock), 32 * kBlock * sizeof(int), s, ctx->dev,
                           (const nori_ray *) d_rays, (nori_intersection *) d_its, n, shadow_ray);
    else
        hipLaunchKernelGGL(intersect_kernel<64>, dim3(grid_for(n)), dim3(kBlock), 64 * kBlock * sizeof(int), s, ctx->dev,
                           (const nori_ray *) d_rays, (nori_intersection *) d_its, n, shadow_ray);
    HIP_TRY(ctx, hipGetLastError());
    return NORI_OK;
}

/* small RAII device buffer for the host-buffer convenience entry points */
struct Scratch {
    void *p = nullptr;
    ~Scratch() { if (p) (void) hipFree(p); }
};
#define SCRATCH_IN(ctx, var, host, bytes)                                           \
    Scratch var;                                                                    \
    HIP_TRY(ctx, hipMalloc(&var.p, std::max<size_t>((bytes), 16)));                 \
    if (host) HIP_TRY(ctx, hipMemcpy(var.p, host, (bytes), hipMemcpyHostToDevice))
#define SCRATCH_OUT(ctx, var, bytes)                                                \
    Scratch var;                                                                    \
    HIP_TRY(ctx, hipMalloc(&var.p, std::max<size_t>((bytes), 16)))

int nori_hip_intersect(nori_hip_ctx *ctx, const nori_ray *rays, nori_intersection *its, size_t n, int shadow_ray) {
    REQUIRE_ACCEL(ctx);
    if (n == 0) return NORI_OK;
    if (!rays || !its) return NORI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, dr, rays, n * sizeof(nori_ray));
    SCRATCH_OUT(ctx, di, n * sizeof(nori_intersection));
    int rc = nori_hip_intersect_device(ctx, dr.p, di.p, n, shadow_ray, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(its, di.p, n * sizeof(nori_intersection), hipMemcpyDeviceToHost));
    return NORI_OK;
}

int nori_hip_sample_rays(nori_hip_ctx *ctx, const float *pixel_samples, size_t n, nori_ray *rays) {
    if (!ctx || !ctx->have_scene) return NORI_ERR_NOT_READY;
    if (n == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, dp, pixel_samples, n * 2 * sizeof(float));
    SCRATCH_OUT(ctx, dr, n * sizeof(nori_ray));
    hipLaunchKernelGGL(sample_rays_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, ctx->dev.camera,
                       (const float *) dp.p, n, (nori_ray *) dr.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(rays, dr.p, n * sizeof(nori_ray), hipMemcpyDeviceToHost));
    return NORI_OK;
}

} // extern "C"

template <int STACK>
static void launch_li(nori_hip_ctx *ctx, const nori_ray *r, size_t n, const uint64_t *ss, const uint64_t *sq, float *rgb) {
    const dim3 grid(grid_for(n)), block(kBlock);
    const size_t lds = STACK * kBlock * sizeof(int);
    switch (ctx->dev.integrator.type) {
#define LI_CASE(I) case I: hipLaunchKernelGGL((li_kernel<I, STACK>), grid, block, lds, 0, ctx->dev, r, n, ss, sq, rgb); break;
        LI_CASE(0) LI_CASE(1) LI_CASE(2) LI_CASE(3) LI_CASE(4) LI_CASE(5) LI_CASE(6)
#undef LI_CASE
    }
}

extern "C" {

int nori_hip_li(nori_hip_ctx *ctx, const nori_ray *rays, size_t n, const uint64_t *seed_state, const uint64_t *seed_seq, float *rgb) {
    REQUIRE_ACCEL(ctx);
    if (n == 0) return NORI_OK;
    if (!rays || !seed_state || !seed_seq || !rgb) return NORI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, dr, rays, n * sizeof(nori_ray));
    SCRATCH_IN(ctx, ds, seed_state, n * sizeof(uint64_t));
    SCRATCH_IN(ctx, dq, seed_seq, n * sizeof(uint64_t));
    SCRATCH_OUT(ctx, dc, n * 3 * sizeof(float));
    if (ctx->stack_depth <= 32) launch_li<32>(ctx, (const nori_ray *) dr.p, n, (const uint64_t *) ds.p, (const uint64_t *) dq.p, (float *) dc.p);
    else launch_li<64>(ctx, (const nori_ray *) dr.p, n, (const uint64_t *) ds.p, (const uint64_t *) dq.p, (float *) dc.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(rgb, dc.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return NORI_OK;
}

static Bsdf bsdf_from_desc(const nori_bsdf_desc &d) {
    Bsdf b;
    b.type = d.type; b.albedo = mk3(d.albedo[0], d.albedo[1], d.albedo[2]);
    b.alpha = d.alpha; b.int_ior = d.int_ior; b.ext_ior = d.ext_ior; b.ks = d.ks;
    return b;
}

static int bsdf_call(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf, int op, const float *wi, const float *wo_in,
                     const float *sample, size_t n, float *wo_out, float *value, float *eta, int32_t *measure) {
    if (!ctx || !bsdf || !wi) return NORI_ERR_INVALID_ARGUMENT;
    if (bsdf->type < 0 || bsdf->type > 3) { ctx->error = "unknown BSDF type"; return NORI_ERR_INVALID_ARGUMENT; }
    if (n == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, dwi, wi, n * 3 * sizeof(float));
    SCRATCH_IN(ctx, dwo, wo_in, n * 3 * sizeof(float));
    SCRATCH_IN(ctx, dsm, sample, n * 2 * sizeof(float));
    SCRATCH_OUT(ctx, dout_wo, n * 3 * sizeof(float));
    SCRATCH_OUT(ctx, dval, n * 3 * sizeof(float));
    SCRATCH_OUT(ctx, deta, n * sizeof(float));
    SCRATCH_OUT(ctx, dmeas, n * sizeof(int32_t));
    hipLaunchKernelGGL(bsdf_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, bsdf_from_desc(*bsdf), op,
                       (const float *) dwi.p, (const float *) dwo.p, (const float *) dsm.p, n, (float *) dout_wo.p,
                       (float *) dval.p, (float *) deta.p, (int32_t *) dmeas.p);
    HIP_TRY(ctx, hipGetLastError());
    if (op == 0) {
        HIP_TRY(ctx, hipMemcpy(wo_out, dout_wo.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(value, dval.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (eta) HIP_TRY(ctx, hipMemcpy(eta, deta.p, n * sizeof(float), hipMemcpyDeviceToHost));
        if (measure) HIP_TRY(ctx, hipMemcpy(measure, dmeas.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    } else {
        HIP_TRY(ctx, hipMemcpy(value, dval.p, n * (op == 1 ? 3 : 1) * sizeof(float), hipMemcpyDeviceToHost));
    }
    return NORI_OK;
}

int nori_hip_bsdf_sample(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf, const float *wi, const float *sample, size_t n,
                         float *wo, float *weight, float *eta, int32_t *measure) {
    if (!sample || !wo || !weight) return NORI_ERR_INVALID_ARGUMENT;
    return bsdf_call(ctx, bsdf, 0, wi, nullptr, sample, n, wo, weight, eta, measure);
}
int nori_hip_bsdf_eval(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *value) {
    if (!wo || !value) return NORI_ERR_INVALID_ARGUMENT;
    return bsdf_call(ctx, bsdf, 1, wi, wo, nullptr, n, nullptr, value, nullptr, nullptr);
}
int nori_hip_bsdf_pdf(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *pdf) {
    if (!wo || !pdf) return NORI_ERR_INVALID_ARGUMENT;
    return bsdf_call(ctx, bsdf, 2, wi, wo, nullptr, n, nullptr, pdf, nullptr, nullptr);
}

int nori_hip_warp(nori_hip_ctx *ctx, int warp, float param, const float *sample, size_t n, float *out) {
    if (!ctx || !sample || !out || warp < 0 || warp > 6) return NORI_ERR_INVALID_ARGUMENT;
    if (n == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, din, sample, n * 2 * sizeof(float));
    SCRATCH_OUT(ctx, dout, n * 3 * sizeof(float));
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, warp, param, 0, (const float *) din.p, n, (float *) dout.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return NORI_OK;
}

int nori_hip_warp_pdf(nori_hip_ctx *ctx, int warp, float param, const float *points, size_t n, float *pdf) {
    if (!ctx || !points || !pdf || warp < 0 || warp > 6) return NORI_ERR_INVALID_ARGUMENT;
    if (n == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, din, points, n * 3 * sizeof(float));
    SCRATCH_OUT(ctx, dout, n * sizeof(float));
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, warp, param, 1, (const float *) din.p, n, (float *) dout.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(pdf, dout.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return NORI_OK;
}

int nori_hip_pcg32_floats(nori_hip_ctx *ctx, const uint64_t *seed_state, const uint64_t *seed_seq, size_t n, uint32_t count, float *out) {
    return nori_hip_pcg32_floats_at(ctx, seed_state, seed_seq, 0, n, count, out);
}

int nori_hip_pcg32_floats_at(nori_hip_ctx *ctx, const uint64_t *seed_state, const uint64_t *seed_seq, uint64_t skip, size_t n, uint32_t count, float *out) {
    if (!ctx || !seed_state || !seed_seq || !out) return NORI_ERR_INVALID_ARGUMENT;
    if (n == 0 || count == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    SCRATCH_IN(ctx, ds, seed_state, n * sizeof(uint64_t));
    SCRATCH_IN(ctx, dq, seed_seq, n * sizeof(uint64_t));
    SCRATCH_OUT(ctx, dout, n * count * sizeof(float));
    hipLaunchKernelGGL(pcg32_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, (const uint64_t *) ds.p, (const uint64_t *) dq.p, skip, n, count, (float *) dout.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(out, dout.p, n * count * sizeof(float), hipMemcpyDeviceToHost));
    return NORI_OK;
}

static size_t frame_floats(const nori_hip_ctx *ctx) {
    const int b = ctx->host.filter.border;
    return (size_t) (ctx->host.camera.width + 2 * b) * (size_t) (ctx->host.camera.height + 2 * b) * 4;
}

int nori_hip_splat(nori_hip_ctx *ctx, const float *positions, const float *values, size_t n, float *rgbw) {
    if (!ctx || !ctx->have_scene) return NORI_ERR_NOT_READY;
    if (!positions || !values || !rgbw) return NORI_ERR_INVALID_ARGUMENT;
    if (n == 0) return NORI_OK;
    DeviceGuard g(ctx->device);
    const size_t fb = frame_floats(ctx) * sizeof(float);
    SCRATCH_IN(ctx, dp, positions, n * 2 * sizeof(float));
    SCRATCH_IN(ctx, dv, values, n * 3 * sizeof(float));
    SCRATCH_IN(ctx, df, rgbw, fb);
    hipLaunchKernelGGL(splat_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, ctx->dev.filter, ctx->d_filter,
                       ctx->host.camera.width, ctx->host.camera.height, (const float *) dp.p, (const float *) dv.p, n, (float *) df.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(rgbw, df.p, fb, hipMemcpyDeviceToHost));
    return NORI_OK;
}

} // extern "C"

/* ---------------------------------------------------------------- render */
template <int STACK>
static size_t render_lds_bytes(const RenderArgs &) {
    return sizeof(int) * STACK * kBlock + sizeof(unsigned int) * 16;       /* traversal stacks + counters */
}

template <int INTEG, int STACK, bool COUNT>
static hipError_t launch_render_one(nori_hip_ctx *ctx, const RenderArgs &a, const FilmStore &film, hipStream_t s) {
    const size_t lds = render_lds_bytes<STACK>(a);
    auto kern = render_kernel<INTEG, STACK, COUNT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.n_sel_tiles * a.n_chunks), dim3(kBlock), lds, s, ctx->dev, a, film, ctx->d_stats);
    return hipGetLastError();
}

/* spp chunking: one workgroup per (tile, chunk of samples); enough workgroups for the dispatcher
   to balance load (measured best around 16 k), at least 8 spp per chunk */
static void plan_chunks(RenderArgs &a) {
    uint32_t target_wgs = 16384;
    if (const char *e = getenv("NORI_HIP_TARGET_WGS")) target_wgs = (uint32_t) std::max(1, atoi(e));
    uint32_t n_chunks = 1;
    if (a.n_sel_tiles > 0 && a.n_sel_tiles < target_wgs) n_chunks = (target_wgs + a.n_sel_tiles - 1) / a.n_sel_tiles;
    n_chunks = std::max(1u, std::min(n_chunks, std::max(1u, a.spp_count / 8)));
    a.chunk_spp = (a.spp_count + n_chunks - 1) / std::max(1u, n_chunks);
    if (a.chunk_spp == 0) a.chunk_spp = 1;
    a.n_chunks = std::max(1u, (a.spp_count + a.chunk_spp - 1) / a.chunk_spp);
}

template <int STACK, bool COUNT>
static hipError_t launch_render(nori_hip_ctx *ctx, const RenderArgs &a, const FilmStore &d_rgbw, hipStream_t s) {
    switch (ctx->dev.integrator.type) {
    case 0: return launch_render_one<0, STACK, COUNT>(ctx, a, d_rgbw, s);
    case 1: return launch_render_one<1, STACK, COUNT>(ctx, a, d_rgbw, s);
    case 2: return launch_render_one<2, STACK, COUNT>(ctx, a, d_rgbw, s);
    case 3: return launch_render_one<3, STACK, COUNT>(ctx, a, d_rgbw, s);
    case 4: return launch_render_one<4, STACK, COUNT>(ctx, a, d_rgbw, s);
    case 5: return launch_render_one<5, STACK, COUNT>(ctx, a, d_rgbw, s);
    default: return launch_render_one<6, STACK, COUNT>(ctx, a, d_rgbw, s);
    }
}

/* nori_hip_render, and nori_hip_render_block_rows (share != nullptr: the block rows of a reference-order film; the accumulators
   of their blocks go to share->block_acc instead of a frame) */
static int render_impl(nori_hip_ctx *ctx, const nori_render_params *params, void *d_rgbw, nori_render_stats *stats, const FilmBlockRows *share) {
    REQUIRE_ACCEL(ctx);
    if (!params || (!d_rgbw && !share)) return NORI_ERR_INVALID_ARGUMENT;
    if (params->tile_mod == 0 || params->tile_rem >= params->tile_mod) { ctx->error = "render: bad tile_mod/tile_rem"; return NORI_ERR_INVALID_ARGUMENT; }
    if (share && (!ctx->film_reference || params->tile_mod != 1 || params->seed_mode != NORI_SEED_PER_SAMPLE || !share->block_acc)) {
        ctx->error = "render_block_rows: needs film_order = reference, tile_mod 1, NORI_SEED_PER_SAMPLE and an accumulator array";
        return NORI_ERR_INVALID_ARGUMENT;
    }
    if (params->seed_mode != NORI_SEED_PER_SAMPLE && params->seed_mode != NORI_SEED_NORI_BLOCK) { ctx->error = "render: unknown seed_mode"; return NORI_ERR_INVALID_ARGUMENT; }
    if (params->seed_mode == NORI_SEED_NORI_BLOCK && (params->tile_mod != 1 || params->spp_begin != 0)) {
        ctx->error = "render: NORI_SEED_NORI_BLOCK renders whole frames from sample 0 (a block's stream is serial: src/independent.cpp:36-41)";
        return NORI_ERR_UNSUPPORTED;
    }
    if (ctx->host.filter.border > 8) { ctx->error = "render: reconstruction filter radius too large for the LDS tile"; return NORI_ERR_UNSUPPORTED; }
    if (ctx->film_reference && params->tile_mod != 1) { ctx->error = "render: film_order = reference renders whole frames (a block's samples are added consecutively)"; return NORI_ERR_UNSUPPORTED; }
    DeviceGuard g(ctx->device);
    hipStream_t s = (hipStream_t) params->stream;

    RenderArgs a;
    a.spp_begin = params->spp_begin; a.spp_count = params->spp_count;
    a.tile_mod = params->tile_mod; a.tile_rem = params->tile_rem;
    a.tiles_x = (uint32_t) ((ctx->host.camera.width + kTile - 1) / kTile);
    a.tiles_y = (uint32_t) ((ctx->host.camera.height + kTile - 1) / kTile);
    const uint32_t n_tiles = a.tiles_x * a.tiles_y;
    a.n_sel_tiles = n_tiles > a.tile_rem ? (n_tiles - a.tile_rem + a.tile_mod - 1) / a.tile_mod : 0;
    if (share) {
        /* whole rows of 32x32 blocks = a contiguous range of tiles: tile_id = tile_rem + ordinal * 1 with tile_rem = the range's
           first tile (every kernel derives a tile from its ordinal this way) */
        const uint32_t byn = (uint32_t) ((ctx->host.camera.height + kNoriBlock - 1) / kNoriBlock);
        const uint32_t row0 = std::min(share->row_begin, byn), row1 = row0 + std::min(share->row_count, byn - row0);
        const uint32_t t0 = std::min(film_block_rows_first_tile(row0, a.tiles_x), n_tiles), t1 = std::min(film_block_rows_first_tile(row1, a.tiles_x), n_tiles);
        a.tile_rem = t0; a.n_sel_tiles = t1 - t0;
    }
    a.tile_w = kTile + 2 * ctx->host.filter.border;
    a.th_shade = 44; a.th_inner = 1; a.th_leaf = 1;
    a.debug_flags = getenv("NORI_HIP_NOSPLAT") ? 1u : 0u;
    if (const char *e = getenv("NORI_HIP_TH_SHADE")) a.th_shade = (uint32_t) std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NORI_HIP_TH_INNER")) a.th_inner = (uint32_t) std::min(64, std::max(1, atoi(e)));
    if (const char *e = getenv("NORI_HIP_TH_LEAF")) a.th_leaf = (uint32_t) std::min(64, std::max(1, atoi(e)));
    plan_chunks(a);
    unsigned long long n_invalid = 0; uint32_t n_workgroups = 0;

    struct EventPair {      /* destroyed on every return path */
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void) hipEventDestroy(a); if (b) (void) hipEventDestroy(b); }
    } evp;
    hipEvent_t &ev0 = evp.a, &ev1 = evp.b;
    if (stats) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 16 * sizeof(unsigned long long), s));
        HIP_TRY(ctx, hipEventCreate(&ev0)); HIP_TRY(ctx, hipEventCreate(&ev1));
        HIP_TRY(ctx, hipEventRecord(ev0, s));
    }
    WfStats wst;
    KernelTimer timer(stats && params->time_kernels != 0);
    int engine = ctx->engine;
    if (const char *e = getenv("NORI_HIP_ENGINE")) engine = std::string(e) == "wavefront" ? 1 : (std::string(e) == "megakernel" ? 0 : -1);
    /* auto: the wavefront engine wins once there are enough paths to keep its kernels full; small jobs avoid its launch train.
       Measured on the round-4 build (tools/engine_switch_probe.py, profiles/r4_11_engine_switch.txt): with a path loop -- every
       integrator but `normals` -- from 2^19 camera samples per call (Cornell box path_mis, megakernel / wavefront ms: 2^18 1.11 / 1.47,
       2^19 2.00 / 1.80, 2^22 3.90 / 3.08, 2^24 9.84 / 7.86, 2^26 33.1 / 19.4; AO on 328 k triangles 2^18 0.90 / 0.66); `normals` is one
       ray and no loop: one launch against four (2^20 samples: 0.17 / 0.44 ms), the old 2^24 stays */
    if (engine < 0) {
        const size_t samples = (size_t) a.n_sel_tiles * 256 * a.spp_count;
        engine = samples >= ((size_t) 1 << (ctx->dev.integrator.type == 0 ? 24 : 19)) ? 1 : 0;
    }
    if (ctx->dev.wide) engine = 1;      /* wide nodes are walked by the wavefront engine (and the batch twins) only */
    if (params->seed_mode == NORI_SEED_NORI_BLOCK && a.n_sel_tiles > 0 && a.spp_count > 0) {
        /* one lane per 32x32 block, the block's own pcg32 stream (render_block_serial_kernel) */
        const size_t n_samples = (size_t) a.n_sel_tiles * 256 * a.spp_count;
        if (n_samples > ((size_t) 1 << 30)) { ctx->error = "render: NORI_SEED_NORI_BLOCK keeps the whole frame's samples in the film store (limit 2^30)"; return NORI_ERR_UNSUPPORTED; }
        FilmStore film;
        std::string ferr = film_prepare(ctx->film, n_samples, a.n_sel_tiles, a.tile_w, s, film);
        if (!ferr.empty()) { ctx->error = ferr; return NORI_ERR_OUT_OF_MEMORY; }
        /* (slots of edge-tile pixels outside the image stay unwritten: film_gather never reads them) */
        const uint32_t bx = (uint32_t) ((ctx->host.camera.width + kNoriBlock - 1) / kNoriBlock), by = (uint32_t) ((ctx->host.camera.height + kNoriBlock - 1) / kNoriBlock);
        const uint32_t n_blocks = bx * by;
        const uint32_t need = ctx->bvh.max_depth + 1;
        timer.begin(KC_TRACE, s);
        switch (ctx->dev.integrator.type) {
#define SB(I) case I: if (need <= 32) hipLaunchKernelGGL((render_block_serial_kernel<I, 32>), dim3((n_blocks + 63) / 64), dim3(64), 32 * 64 * sizeof(int), s, ctx->dev, bx, n_blocks, a.spp_count, a.tiles_x, film, ctx->d_stats); \
                      else hipLaunchKernelGGL((render_block_serial_kernel<I, 64>), dim3((n_blocks + 63) / 64), dim3(64), 64 * 64 * sizeof(int), s, ctx->dev, bx, n_blocks, a.spp_count, a.tiles_x, film, ctx->d_stats); break;
            SB(0) SB(1) SB(2) SB(3) SB(4) SB(5) SB(6)
#undef SB
        }
        timer.end(s);
        HIP_TRY(ctx, hipGetLastError());
        FilmLaunch fl;
        fl.tile_first = 0; fl.store_tile_first = 0; fl.n_tiles = a.n_sel_tiles; fl.tile_mod = 1; fl.tile_rem = 0;
        fl.tiles_x = a.tiles_x; fl.tiles_y = a.tiles_y; fl.tile_w = a.tile_w; fl.n_spp = a.spp_count;
        timer.begin(KC_FILM, s);
        if (ctx->film_reference) {
            std::string rerr = film_reference_order(ctx->film, film, ctx->dev, ctx->d_filter, a.spp_count, a.tiles_x, nullptr, (float *) d_rgbw, s);
            if (!rerr.empty()) { ctx->error = rerr; return NORI_ERR_INTERNAL; }
        } else {
            film_gather(ctx->dev, ctx->d_filter, film, fl, s);
            film_resolve(ctx->dev, film, fl, (float *) d_rgbw, s);
        }
        timer.end(s);
        HIP_TRY(ctx, hipGetLastError());
        if (stats) n_invalid = film_invalid_count(film, s);
        engine = 2;      /* stats come from ctx->d_stats like the megakernel's */
    } else
    if (engine == 1 && a.n_sel_tiles > 0 && a.spp_count > 0) {
        WfLaunch wl;
        wl.spp_begin = a.spp_begin; wl.spp_count = a.spp_count; wl.tile_mod = a.tile_mod; wl.tile_rem = a.tile_rem;
        wl.tiles_x = a.tiles_x; wl.tiles_y = a.tiles_y; wl.n_sel_tiles = a.n_sel_tiles; wl.tile_w = a.tile_w;
        const uint32_t need = ctx->bvh.max_depth + 1;
        wl.stack_depth = (int) need;
        wl.count_traversal = params->count_traversal != 0;
        wl.time_kernels = stats && params->time_kernels != 0;
        /* paths in flight: the option, bounded by what this GPU has free right now (state already held by
           this context counts as free) -- a second context or another process may own part of the HBM */
        wl.film_reference = ctx->film_reference; wl.film_share = share;
        /* what the call can use at all: a batch never holds more samples than the call has, the pool never more paths than a batch */
        const size_t call_samples = (size_t) a.n_sel_tiles * 256 * a.spp_count;
        /* (reference film order: the frame is one batch whatever the options say -- the pool need not hold it) */
        const size_t opt_samples = ctx->film_reference ? call_samples : ctx->wavefront_samples ? ctx->wavefront_samples : ctx->wavefront_paths;
        wl.max_samples = std::max<size_t>(256, std::min(opt_samples, call_samples));
        wl.max_paths = std::max<size_t>(256, std::min(ctx->wavefront_paths, wl.max_samples));
        const size_t per_path = wavefront_bytes_per_path(), per_sample = wavefront_bytes_per_sample();
        size_t free_b = 0, total_b = 0;
        if (!getenv("NORI_HIP_WF_IGNORE_FREE") /* test hook: as if the free figure were stale */ && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t usable = (size_t) ((double) (free_b + wavefront_held_bytes(ctx->wf, ctx->film)) * 0.85);
            if (wl.max_paths * per_path + wl.max_samples * per_sample > usable) {
                /* reference film order: the batch is the frame, the pool takes what is left.  Else the pool keeps its size while the
                   sample store can still hold four pools' worth of samples; else both shrink */
                if (ctx->film_reference) wl.max_paths = std::max<size_t>(256, usable > wl.max_samples * per_sample ? (usable - wl.max_samples * per_sample) / per_path : 0);
                else if (usable >= wl.max_paths * (per_path + 4 * per_sample)) wl.max_samples = (usable - wl.max_paths * per_path) / per_sample;
                else { wl.max_paths = std::max<size_t>(256, usable / (per_path + 4 * per_sample)); wl.max_samples = std::min(wl.max_samples, 4 * wl.max_paths); }
            }
        }
        std::string err = wavefront_render(*ctx->wf, ctx->film, ctx->dev, ctx->d_filter, wl, (float *) d_rgbw, s, wst);
        /* The free figure above is a snapshot: another context on this device -- a group with a duplicated device list rendering
           in parallel threads, a second process -- may have claimed the same bytes in the meantime.  Nothing has been accumulated
           yet when the pool or the sample store cannot be allocated (they are allocated before the first launch), so the call is
           simply made again with less: what this context holds is given back first (a pool that fitted while the sample store did
           not must not survive the retry), then the bigger of the two is halved -- the POOL while it is, which leaves the batches,
           hence the bits of the frame, as they were. */
        while (!err.empty() && err.find("out of memory") != std::string::npos) {
            (void) hipGetLastError();
            wavefront_release_pool(ctx->wf);
            film_release(ctx->film);
            const bool pool_bigger = wl.max_paths * per_path >= wl.max_samples * per_sample || ctx->film_reference;
            if (pool_bigger && wl.max_paths > ((size_t) 1 << 20)) wl.max_paths /= 2;
            else if (ctx->film_reference) break;      /* (its one batch cannot shrink) */
            else if (wl.max_samples > ((size_t) 1 << 20)) { wl.max_samples /= 2; wl.max_paths = std::min(wl.max_paths, wl.max_samples); }
            else if (wl.max_paths > ((size_t) 1 << 20)) wl.max_paths /= 2;
            else break;
            err = wavefront_render(*ctx->wf, ctx->film, ctx->dev, ctx->d_filter, wl, (float *) d_rgbw, s, wst);
        }
        if (!err.empty()) {
            ctx->error = err;
            return err.find("out of memory") != std::string::npos ? NORI_ERR_OUT_OF_MEMORY : NORI_ERR_INTERNAL;
        }
    } else
    if (a.n_sel_tiles > 0 && a.spp_count > 0) {
        /* Samples go to the film store (24 B each); if a call produces more than the store
           budget it is cut into launches over sample sub-ranges, each followed by its splat. */
        size_t cap = (size_t) 1 << 29;
        if (const char *e = getenv("NORI_HIP_FILM_SAMPLES")) cap = (size_t) std::max(256ll, atoll(e));
        const uint32_t spp_per_launch = (uint32_t) std::max<size_t>(1, std::min<size_t>(a.spp_count, cap / ((size_t) a.n_sel_tiles * 256)));
        if (ctx->film_reference && spp_per_launch != a.spp_count) { ctx->error = "render: film_order = reference needs all samples of the frame in the film store at once"; return NORI_ERR_UNSUPPORTED; }
        FilmStore film;
        std::string ferr = film_prepare(ctx->film, (size_t) a.n_sel_tiles * 256 * spp_per_launch, a.n_sel_tiles, a.tile_w, s, film);
        if (!ferr.empty()) { ctx->error = ferr; return NORI_ERR_OUT_OF_MEMORY; }
        FilmLaunch fl;
        fl.tile_first = 0; fl.store_tile_first = 0; fl.n_tiles = a.n_sel_tiles; fl.tile_mod = a.tile_mod; fl.tile_rem = a.tile_rem;
        fl.tiles_x = a.tiles_x; fl.tiles_y = a.tiles_y; fl.tile_w = a.tile_w;
        const bool count = params->count_traversal != 0;
        const uint32_t need = ctx->bvh.max_depth + 1;
        for (uint32_t sb = 0; sb < a.spp_count; sb += spp_per_launch) {
            RenderArgs a2 = a;
            a2.spp_begin = a.spp_begin + sb; a2.spp_count = std::min(spp_per_launch, a.spp_count - sb);
            plan_chunks(a2);
            hipError_t e;
            timer.begin(KC_TRACE, s);
            /* the LDS stack is sized to the tree: fewer entries -> more workgroups per CU */
            if (need <= 16) e = count ? launch_render<16, true>(ctx, a2, film, s) : launch_render<16, false>(ctx, a2, film, s);
            else if (need <= 24) e = count ? launch_render<24, true>(ctx, a2, film, s) : launch_render<24, false>(ctx, a2, film, s);
            else if (need <= 32) e = count ? launch_render<32, true>(ctx, a2, film, s) : launch_render<32, false>(ctx, a2, film, s);
            else e = count ? launch_render<64, true>(ctx, a2, film, s) : launch_render<64, false>(ctx, a2, film, s);
            timer.end(s);
            HIP_TRY(ctx, e);
            fl.n_spp = a2.spp_count;
            timer.begin(KC_FILM, s);
            if (!(a.debug_flags & 1u) && !ctx->film_reference) film_gather(ctx->dev, ctx->d_filter, film, fl, s);
            timer.end(s);
            n_workgroups += a2.n_sel_tiles * a2.n_chunks;
        }
        timer.begin(KC_FILM, s);
        if (ctx->film_reference) {
            std::string rerr = film_reference_order(ctx->film, film, ctx->dev, ctx->d_filter, a.spp_count, a.tiles_x, share, (float *) d_rgbw, s);
            if (!rerr.empty()) { ctx->error = rerr; return NORI_ERR_INTERNAL; }
        } else film_resolve(ctx->dev, film, fl, (float *) d_rgbw, s);
        timer.end(s);
        HIP_TRY(ctx, hipGetLastError());
        if (stats) n_invalid = film_invalid_count(film, s);
    }
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ev1, s));
        HIP_TRY(ctx, hipEventSynchronize(ev1));
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev0, ev1));
        unsigned long long h[16];
        HIP_TRY(ctx, hipMemcpy(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost));
        memset(stats, 0, sizeof(*stats));
        stats->n_camera_samples = h[0]; stats->n_closest_rays = h[1]; stats->n_shadow_rays = h[2];
        stats->n_node_tests = h[3]; stats->n_tri_tests = h[4]; stats->n_invalid = n_invalid;
        stats->kernel_ms = ms;
        stats->engine = (uint32_t) engine;
        if (getenv("NORI_HIP_CENSUS") && h[8])
            fprintf(stderr, "[census] shade runs %llu lanes %.1f | inner runs %llu lanes %.1f | leaf runs %llu lanes %.1f\n", h[8], (double) h[9] / h[8], h[10], (double) h[11] / std::max(1ull, h[10]), h[12], (double) h[13] / std::max(1ull, h[12]));
        stats->n_workgroups = n_workgroups;
        float cms[KC_COUNT]; unsigned int cl[KC_COUNT];
        timer.collect(cms, cl);
        if (engine == 1) { for (int c = 0; c < KC_COUNT; ++c) { cms[c] = wst.class_ms[c]; cl[c] = wst.class_launches[c]; } }
        stats->trace_ms = cms[KC_TRACE]; stats->shade_ms = cms[KC_SHADE]; stats->film_ms = cms[KC_FILM]; stats->n_trace_launches = cl[KC_TRACE];
        if (engine == 1) {
            stats->n_camera_samples = wst.n_camera; stats->n_closest_rays = wst.n_closest; stats->n_shadow_rays = wst.n_shadow;
            stats->n_node_tests = wst.n_nodes; stats->n_tri_tests = wst.n_tris; stats->n_invalid = wst.n_invalid;
            stats->n_workgroups = wst.n_launches;
            stats->trace_cus = wst.trace_cus;
            stats->tail_ms = wst.class_ms[KC_TAIL]; stats->tail_cus = wst.tail_cus;
            if (getenv("NORI_HIP_CENSUS")) fprintf(stderr, "[wavefront] batches %u iterations %u launches %u state %.1f MB\n", wst.n_batches, wst.n_iterations, wst.n_launches, wst.state_bytes / 1048576.0);
        }
        const uint32_t need = ctx->bvh.max_depth + 1;
        stats->lds_bytes = (uint32_t) (need <= 16 ? render_lds_bytes<16>(a) : need <= 24 ? render_lds_bytes<24>(a) : need <= 32 ? render_lds_bytes<32>(a) : render_lds_bytes<64>(a));
    }
    return NORI_OK;
}

extern "C" {

int nori_hip_render(nori_hip_ctx *ctx, const nori_render_params *params, void *d_rgbw, nori_render_stats *stats) {
    return render_impl(ctx, params, d_rgbw, stats, nullptr);
}

int nori_hip_block_acc_floats(nori_hip_ctx *ctx, size_t *n_floats) {
    if (!ctx || !ctx->have_scene) return NORI_ERR_NOT_READY;
    if (!n_floats) return NORI_ERR_INVALID_ARGUMENT;
    *n_floats = film_block_acc_floats(ctx->dev);
    return NORI_OK;
}

int nori_hip_render_block_rows(nori_hip_ctx *ctx, const nori_render_params *params, uint32_t row_begin, uint32_t row_count,
                               void *d_block_acc, nori_render_stats *stats) {
    FilmBlockRows share;
    share.row_begin = row_begin; share.row_count = row_count; share.block_acc = (float *) d_block_acc;
    return render_impl(ctx, params, nullptr, stats, &share);
}

int nori_hip_resolve_blocks(nori_hip_ctx *ctx, const void *d_block_acc, void *d_rgbw, void *stream) {
    if (!ctx || !ctx->have_scene) return NORI_ERR_NOT_READY;
    if (!d_block_acc || !d_rgbw) return NORI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    const std::string err = film_resolve_blocks(ctx->film, ctx->dev, (const float *) d_block_acc, (float *) d_rgbw, (hipStream_t) stream);
    if (!err.empty()) { ctx->error = err; return NORI_ERR_INTERNAL; }
    return NORI_OK;
}

int nori_hip_render_host(nori_hip_ctx *ctx, const nori_render_params *params, float *rgbw, nori_render_stats *stats) {
    REQUIRE_ACCEL(ctx);
    if (!params || !rgbw) return NORI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    const size_t fb = frame_floats(ctx) * sizeof(float);
    SCRATCH_OUT(ctx, df, fb);
    HIP_TRY(ctx, hipMemset(df.p, 0, fb));
    nori_render_stats local;
    int rc = nori_hip_render(ctx, params, df.p, stats ? stats : &local);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(rgbw, df.p, fb, hipMemcpyDeviceToHost));
    return NORI_OK;
}

int nori_hip_develop(nori_hip_ctx *ctx, const void *d_rgbw, void *d_rgb, void *stream) {
    if (!ctx || !ctx->have_scene) return NORI_ERR_NOT_READY;
    if (!d_rgbw || !d_rgb) return NORI_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    const int w = ctx->host.camera.width, h = ctx->host.camera.height;
    hipLaunchKernelGGL(develop_kernel, dim3((w + 255) / 256, h), dim3(256), 0, (hipStream_t) stream, (const float4 *) d_rgbw,
                       (float *) d_rgb, w, h, ctx->host.filter.border);
    HIP_TRY(ctx, hipGetLastError());
    return NORI_OK;
}

} // extern "C"
