/*
 * wavefront.hip -- the wavefront engine of the render path.
 *
 * Same per-lane code as the megakernel (rt_path.h, rt_trace.h) and the same film (film.h), same
 * pcg32 streams, therefore the same radiance per camera sample; what changes is
 * how work is laid out for 64-wide waves.  In the megakernel a lane owns a
 * pixel and, when its path needs shading while its neighbours still traverse,
 * it waits.  Here paths live in HBM (sized for 288 GB: 216 B per path -- two state copies of 100 B, wf_records.h, and the
 * hit record --, up to 2^29 paths = 116 GB in flight) and every kernel runs with all lanes doing the SAME kind
 * of work:
 *
 *   loop until no path is alive (the first pass computes the camera sample of src/main.cpp:41-46
 *   in the kernels instead of reading path state):
 *     wf_extend  persistent waves pull paths from the dense state array; a lane traces
 *                the path's shadow ray, then its continuation ray, writes one hit
 *                record, and is refilled as soon as enough lanes of its wave are idle
 *                (__ballot + one atomic per chunk of paths)   -> Accel::rayIntersect
 *     wf_shade   one lane per path: consume the shadow result, shade the closest hit
 *                (Integrator::Li state machine), write the surviving path with its
 *                next rays COMPACTED into the second state copy (ping-pong), so
 *                every pass streams dense arrays however many paths have died
 *   wf_finish  once a batch has few live paths left a small persistent grid walks them to their ends (a lane pulls path after
 *              path); in a call of several batches it does so on its own CUs BESIDE the next batch (wavefront_render, "tail overlap")
 *   film_gather / film_resolve (film.hip)  the finished samples sit tile-major in the
 *                film's store; the reconstruction filter and the block merge
 *                (ImageBlock::put, src/block.cpp:62-102) run as gathers
 *
 * A path vertex costs ONE iteration: its shadow ray (slot B) and continuation
 * ray (slot A) are traced in the same wf_extend pass, and the next wf_shade
 * first adds the emitter sample if B was unoccluded, then shades A's hit --
 * the order in which the megakernel and the CPU oracle accumulate.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

/* The laboratory -- perturbations of the node step and of wf_shade, the cycle profile of a wave -- lives in wf_experiments.h and is
   compiled only into the variant libraries of tools/build_variant*.sh (-DNORI_LAB); the product build defines every hook empty. */
#ifdef NORI_LAB
#include "wf_experiments.h"
#else
#define NORI_EXP_Q_GLOBAL
#define NORI_EXP_Q_LDS
#define NORI_EXP_Q_ALU
#define NORI_PROF_DECL
#define NORI_PROF_MARK(acc)
#define NORI_PROF_STORE
#define NORI_PROF_REPORT(h, k)
#define NORI_LAB_SHADE_PAD
#define NORI_LAB_SHADE_VERTEX
#define NORI_LAB_SHADE_STORE
#endif

#include "film.h"
#include "ktimer.h"
#include "shade_tables.h"
#include "rt_path.h"
#include "wavefront.h"
#include "wf_records.h"

using namespace nrt;

namespace {

constexpr int kB = 256;

/* counters: path count and dynamic-chunk head per state copy (index + copy), overflow flag */
enum { C_N = 0, C_HEAD = 2, C_OVERFLOW = 4, C_TAIL_HEAD = 5, C_SAMPLE = 6, C_HEAD_FRESH = 8, C_COUNT = 12 };      /* C_SAMPLE + copy: the batch's next camera sample (regeneration); C_HEAD_FRESH + copy: chunk head of the launch that traces a pass's NEW paths */
enum { S_CAM = 0, S_CLOSEST = 1, S_SHADOW = 2, S_NODES = 3, S_TRIS = 4, S_INVALID = 5, S_COUNT = 8 };
/* census (COUNT builds, NORI_HIP_CENSUS): wave-level trips of wf_extend's loop and the lanes they used */
enum { Z_TRIPS = 0, Z_INNER_TRIPS = 1, Z_INNER_LANES = 2, Z_LEAF_TRIPS = 3, Z_LEAF_LANES = 4, Z_REFILLS = 5, Z_REFILL_LANES = 6, Z_COUNT = 8 };

constexpr uint32_t kShadeChunk = 4096u;     /* output space a wf_shade workgroup reserves per atomic */

/* The state of the paths in flight, one record per path, structure of arrays.  Two copies: wf_shade
   reads copy `cur` at index i and writes the surviving paths COMPACTED into copy `cur ^ 1`, so every
   kernel reads and writes dense, fully coalesced ranges [0, n) however many paths have died. */
struct WfState {      /* wf_records.h: how a record lies in HBM */
    P3 *o;          /* origin of both rays (their mint is kStoredMint) */
    f4 *dA;         /* (d.xyz of the continuation ray -- its maxt is inf --, bits: F_* | prev_measure << 4 | depth << 8; 0 = no path in this slot) */
    f4 *dB;         /* (d.xyz, maxt) shadow ray, same origin */
    f4 *T_eta, *L_pdf;
    P3 *Ld;
    uint32_t *sidx;    /* the path's camera sample: index into the film's sample store */
    unsigned long long *rng;
};

struct WfBuf {
    WfState st[2];
    f4 *hit;           /* (t, u, v, hit word) written by wf_extend for the paths of copy `cur` */
    f2 *samp_pos;
    P3 *samp_L;        /* the film's sample store (film.h) */
    uint32_t *ctr;
    unsigned long long *stats;
    uint32_t capacity; /* records per copy */
    int *stack_spill;  /* [entry beyond the LDS stack][lane of the wf_extend grid] */
    unsigned long long *census;   /* Z_* counters or null */
};

struct WfBatch {
    uint32_t tile_first, n_tiles;       /* range of selected-tile ordinals */
    uint32_t s_first, n_spp;            /* absolute first sample index, samples per pixel in this batch */
    uint32_t tile_mod, tile_rem, tiles_x;
    int32_t tile_w;
    int32_t inner_repeat;               /* wf_extend: node steps repeat while at least this many lanes are at inner nodes (65: never) */
    uint32_t flags;                     /* kBatch* */
    uint32_t pool;                      /* paths a pass works on: the stored survivors + as many NEW camera samples of the batch as fill it up */
};

/* Regeneration (constant population).  A pass over state copy `cur` works on the n_s survivors the last wf_shade stored plus
   n_fresh camera samples of the batch that have not been started yet -- path i >= n_f0 IS camera sample s0 + (i - n_f0): nothing of
   it is stored, both kernels compute its first vertex (first_vertex) as the batch's first pass always did.  So a batch of any
   size runs on a pool of bt.pool paths, every pass full until the batch's samples run out; with pool >= the batch's samples the
   first pass starts them all and the schedule is the shrinking one of rounds 2 - 5.  Which pass a sample starts in is not
   observable: its pcg32 stream is seeded by (pixel, sample), its radiance goes to the film store at its own index.
   MODE of a kernel: kStoredOnly it works on the stored paths [0, n_s) and carries no camera code; kFreshOnly on the new paths [n_f0, n) and
   reads no state; kMixed (wf_shade only) on both.  wf_extend exists in the two pure forms only -- one kernel with both refills needs 80 B of scratch
   and runs 17 % slower (profiles/r6_01_pool_sweep.txt, "mixed kernels") --: a pass that holds stored AND new paths is traced by TWO launches, the
   stored paths first, then the new ones, each with its own chunk counter; a batch's first pass has only the second, its passes after the last sample
   only the first. */
enum { kStoredOnly = 0, kFreshOnly = 1, kMixed = 2 };
struct PassShape { uint32_t n_s, s0, n_fresh, n_f0, n; };      /* stored paths [0, n_s), new paths [n_f0, n): n_f0 = n_s rounded up to a wf_shade round */
template <int MODE> __device__ __forceinline__ PassShape pass_shape(const uint32_t *ctr, int cur, const WfBatch &bt) {
    PassShape ps;
    ps.n_s = ctr[C_N + cur];      /* (zero in a batch's first pass: the counters are cleared when it starts) */
    ps.s0 = ctr[C_SAMPLE + cur];
    ps.n_f0 = (ps.n_s + 255u) & ~255u;
    const uint32_t total = bt.n_tiles * 256u * bt.n_spp;
    ps.n_fresh = MODE == kStoredOnly ? 0u : min(total - ps.s0, bt.pool > ps.n_f0 ? bt.pool - ps.n_f0 : 0u);
    ps.n = ps.n_fresh ? ps.n_f0 + ps.n_fresh : ps.n_s;
    return ps;
}
constexpr uint32_t kBatchNoAsmLoop = 1u;       /* wf_extend: the compiler's node loop instead of the hand-written one (A/B, tests) */
constexpr uint32_t kBatchCountQ = 2u;          /* wf_extend, COUNT builds: walk the 32-B node records (trav_inner_step_q, the C++ statement of the
                                                  hand-written loop) -- the node / triangle tests counted are those of the tree form the timed kernel walks */

/* Per-lane traversal stack in LDS ([entry][thread], bank = lane -> conflict free).
   Trees no deeper than DEPTH: one register (the address of the next free slot); entry 0 holds the
   "finished" marker, so popping the empty stack ends the traversal without an emptiness test.
   Deeper trees (SPILL): the first DEPTH entries in LDS, the rest -- rare -- in a per-lane column of
   a global buffer.  A small DEPTH keeps wf_extend at 8 waves/SIMD however deep the tree is (a
   64-entry LDS stack would allow 2), which is what hides the HBM latency of scenes that do not fit
   in L2. */
template <int DEPTH, bool SPILL, int BLOCK = kB>
struct LdsStackW {
    int *base; int sp;
    int *spill; uint32_t spill_stride;      /* wave-uniform base, lanes in flight */
    __device__ __forceinline__ void init(char *smem, int *spill_, uint32_t stride_) {
        base = reinterpret_cast<int *>(smem) + threadIdx.x; sp = 0; spill = spill_; spill_stride = stride_;
    }
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ void push(int v) {
        if (sp < DEPTH) base[sp * BLOCK] = v;
        else spill[(size_t) (sp - DEPTH) * spill_stride + blockIdx.x * BLOCK + threadIdx.x] = v;
        sp++;
    }
    __device__ __forceinline__ int pop_or(int empty_value) {
        if (sp == 0) return empty_value;
        sp--;
        if (sp < DEPTH) return base[sp * BLOCK];
        return spill[(size_t) (sp - DEPTH) * spill_stride + blockIdx.x * BLOCK + threadIdx.x];
    }
    static constexpr int kLdsEntries = DEPTH;
};

/* LDS as the hardware addresses it: 32-bit byte addresses, so that the walk's stack pointer is ONE register the hand-written
   node loop below can use as it is */
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ uint32_t lds_address(const void *generic) {
    return (uint32_t) reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) char *) generic);
}
__device__ __forceinline__ lds_int_t *lds_int_at(uint32_t address) { return reinterpret_cast<lds_int_t *>(address); }

template <int DEPTH, int BLOCK>
struct LdsStackW<DEPTH, false, BLOCK> {
    uint32_t top;         /* LDS address of the next free slot */
    uint32_t wave_base;   /* of slot 0 of the wave's lane 0 (wave-uniform): slot k of thread l lives at (k BLOCK + l) ints, bank = lane */
    __device__ __forceinline__ void init(char *smem, int *, uint32_t) {
        wave_base = (uint32_t) __builtin_amdgcn_readfirstlane((int) (lds_address(smem) + (threadIdx.x & ~63u) * 4u)); top = 0u;
    }
    /* the lane's column is found anew at every reset (two v_mbcnt in a volatile statement): as a value the compiler knows to be
       constant it would live in a register for the whole kernel -- the one it then spills */
    __device__ __forceinline__ void reset() {
        uint32_t lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const uint32_t base = wave_base + lane * 4u;
        *lds_int_at(base) = kTravDone; top = base + BLOCK * 4u;
    }
    __device__ __forceinline__ void push(int v) { *lds_int_at(top) = v; top += BLOCK * 4u; }
    __device__ __forceinline__ int pop_or(int) { top -= BLOCK * 4u; return *lds_int_at(top); }
    static constexpr int kLdsEntries = DEPTH + 1;
};

/* The stack of the hand-written node loop for trees deeper than DEPTH: the LDS stack above, and what does not fit goes to a
   per-lane column of a global buffer.  The loop itself only knows the LDS part (one register: `top`); a lane whose LDS part
   is full makes it hand the step to the C++ code (wf_extend), which pushes and pops through this class.  The number of
   entries a lane has in the global column is kept in the LDS slot behind its last stack slot -- the slot `top` points at
   exactly when the LDS part is full -- so it costs no register.  Full is a wave-uniform comparison: top >= limit. */
template <int DEPTH, int BLOCK>
struct LdsStackHybrid {
    uint32_t top;
    uint32_t wave_base, limit;      /* wave-uniform: LDS address of slot 0 of lane 0 / of the counter slot of lane 0 */
    int *spill; uint32_t spill_stride;
    __device__ __forceinline__ void init(char *smem, int *spill_, uint32_t stride_) {
        wave_base = (uint32_t) __builtin_amdgcn_readfirstlane((int) (lds_address(smem) + (threadIdx.x & ~63u) * 4u));
        limit = wave_base + (uint32_t) (DEPTH + 1) * BLOCK * 4u; top = 0u; spill = spill_; spill_stride = stride_;
    }
    __device__ __forceinline__ void reset() {
        uint32_t lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const uint32_t base = wave_base + lane * 4u;
        *lds_int_at(base) = kTravDone; *lds_int_at(base + (uint32_t) (DEPTH + 1) * BLOCK * 4u) = 0; top = base + BLOCK * 4u;
    }
    __device__ __forceinline__ void push(int v) {
        if (top < limit) { *lds_int_at(top) = v; top += BLOCK * 4u; }
        else { const int n = *lds_int_at(top); spill[(size_t) n * spill_stride + blockIdx.x * BLOCK + threadIdx.x] = v; *lds_int_at(top) = n + 1; }
    }
    __device__ __forceinline__ int pop_or(int) {
        if (top >= limit) {
            const int n = *lds_int_at(top);
            if (n > 0) { *lds_int_at(top) = n - 1; return spill[(size_t) (n - 1) * spill_stride + blockIdx.x * BLOCK + threadIdx.x]; }
        }
        top -= BLOCK * 4u; return *lds_int_at(top);
    }
    static constexpr int kLdsEntries = DEPTH + 2;
};
template <int STACK, bool SPILL, bool ASM, int BLOCK> struct ExtendStack { typedef LdsStackW<STACK, SPILL, BLOCK> type; };
template <int STACK, int BLOCK> struct ExtendStack<STACK, true, true, BLOCK> { typedef LdsStackHybrid<STACK, BLOCK> type; };

__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & 63u); }

/* Streaming accesses to the path state: every record is written once and read once per pass, tens of GB -- nothing of
   it is worth a line of L2 next to the tree.  Nontemporal where it measured faster (A/B on one box, ms per pass):
   level 1 = wf_extend's hit / sample-position stores and wf_shade's read of the hit record (wf_extend 47.9 -> 47.2),
   level 2 = wf_shade's state loads and stores (wf_shade 27.9 -> 27.1).  wf_extend's own state LOADS stay plain: nontemporal
   they cost it 2 ms (the 32 B a path re-reads after its shadow ray then miss). */
constexpr int kNontemporalLevel = 2;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <int LEVEL> __device__ __forceinline__ void st_f4(f4 *p, const f4 &v) {
    if (kNontemporalLevel >= LEVEL) { v4f_t t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f_t *>(p)); }
    else *p = v;
}
template <int LEVEL> __device__ __forceinline__ void st_f2(f2 *p, const f2 &v) {
    if (kNontemporalLevel >= LEVEL) { v2f_t t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<v2f_t *>(p)); }
    else *p = v;
}
template <int LEVEL> __device__ __forceinline__ f4 ld_f4(const f4 *p) {
    if (kNontemporalLevel >= LEVEL) { const v4f_t t = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p)); f4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r; }
    return *p;
}
typedef float v3f_t __attribute__((ext_vector_type(3)));
template <int LEVEL> __device__ __forceinline__ void st_p3(P3 *p, const P3 &v) {      /* 12 B, 4-B aligned: global_store_dwordx3 */
    if (kNontemporalLevel >= LEVEL) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); }
    else *p = v;
}
template <int LEVEL> __device__ __forceinline__ P3 ld_p3(const P3 *p) {
    if (kNontemporalLevel >= LEVEL) { P3 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); r.z = __builtin_nontemporal_load(&p->z); return r; }
    return *p;
}
template <int LEVEL, class T> __device__ __forceinline__ void st_w(T *p, T v) { if (kNontemporalLevel >= LEVEL) __builtin_nontemporal_store(v, p); else *p = v; }
template <int LEVEL, class T> __device__ __forceinline__ T ld_w(const T *p) { if (kNontemporalLevel >= LEVEL) return __builtin_nontemporal_load(p); return *p; }

/* The hot records of the tree (rt_top.h: the image is built once per acceleration structure) into the workgroup's LDS.
   Returns the link a walk starts with. */
__device__ int top_image_to_lds(const DevScene &sc, f4 *top, const f4 *image, uint32_t quads) {
    if (image == nullptr) {      /* no image: nothing cached, every link is a memory link */
        if (threadIdx.x == 0) { f4 h; h.x = __uint_as_float((uint32_t) sc.root); h.y = h.z = h.w = 0.0f; top[0] = h; }
    } else {
        for (uint32_t q = threadIdx.x; q < quads; q += blockDim.x) top[q] = image[q];
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane((int) __float_as_uint(top[0].x));
}

/* The first vertex of the batch's path p is never stored: its camera sample (renderBlock,
   src/main.cpp:41-46) is recomputed where it is needed -- by the first wf_extend (ray) and the first
   wf_shade (direction, pcg32 state) -- which costs ~100 instructions twice and saves writing and
   re-reading 80 B of state per path.  Returns false for pixels of edge tiles outside the image. */
__device__ __forceinline__ bool first_vertex(const DevScene &sc, const WfBatch &bt, uint32_t p, f2 &ps, RayIn &cam, Rng &rng) {
    const uint32_t per_tile = 256u * bt.n_spp;
    const uint32_t tsel = p / per_tile, rem = p - tsel * per_tile;
    const uint32_t sl = rem >> 8, pix = rem & 255u;
    const uint32_t tile_id = bt.tile_rem + (bt.tile_first + tsel) * bt.tile_mod;
    const int x0 = (int) (tile_id % bt.tiles_x) * kTile, y0 = (int) (tile_id / bt.tiles_x) * kTile;
    int px, py; film_tile_pixel((int) pix, x0, y0, px, py);
    if (px >= sc.camera.width || py >= sc.camera.height) return false;
    rng_seed(rng, (uint64_t) py * (uint64_t) sc.camera.width + (uint64_t) px, (uint64_t) (bt.s_first + sl));
    const f2 j = rng_next_2d(rng);
    ps = mk2((float) px + j.x, (float) py + j.y);
    (void) rng_next_2d(rng);          /* apertureSample: drawn, unused */
    camera_sample_ray(sc.camera, ps, cam);
    return true;
}

/* The BVH2 node loop of wf_extend, hand-written for gfx950, on the 32-B node records of rt_nodeq.h:
 *
 *     do { if (trav_at_inner(tv)) trav_inner_step_q(sc, stack, tv, tc, top); } while (lanes at inner nodes >= repeat);
 *
 * (rt_trace.h; BoundingBox::rayIntersect of include/nori/bbox.h:323-350 for both children).  With 64-B nodes wf_extend was bound by
 * the number of vector-memory instructions a CU issues -- each occupies its address path for ~16 cycles however few lanes take
 * part, while the arithmetic and scalar pipes were half idle.  Measured by adding instructions of each kind to the node step
 * (-DNORI_EXP_SENS=n, ms per frame): one more global_load_dwordx4 +6.3, a dwordx2 +5.2, a dword +2.8 -- the same with four lanes
 * active as with all of them --, sixteen v_mov +1.7, sixteen s_mov +0.4, the LDS reads over again +0.8.  So the loop reads a node
 * in TWO loads where the 64-B record takes four, and pays for it in arithmetic: per plane one v_cvt_f32_u32 (SDWA: a 16-bit half of
 * a dword) and one v_fma, per axis two v_bfi that pick the dword with the near planes by the sign of the ray's direction.  On this
 * loop the same experiment reads: one more dwordx4 +0.8, sixteen v_mov +1.8, 64 idle cycles +1.2 -- the kernel now sits against the
 * vector ALU (~80 % busy), which is where a ray tracer without ray-tracing hardware belongs.
 * Written as assembly because the child selection is two v_cndmask on a scalar-side mask and two exec-masked LDS operations (push
 * where both children are hit, pop where none is) where the compiler builds three nested regions with their save / restore /
 * skip-branch triples, because the loads must not be widened, merged or re-ordered, and because the loop's exit test can then look
 * ahead: it stays inside while the trip around the outer loop would run neither a triangle step nor a refill.
 * Same operations in the same order as trav_inner_step_q -- the CPU harness walks that one, against the linear scan.
 * gfx950 hazards honoured by hand: a VALU-written vcc is read by v_cndmask two instructions later at the earliest.
 * Records are addressed as base + (node << 5) in 32 bits: the caller takes this path for trees below 2^25 nodes only. */
#define NORI_SDWA(dst, src, half) "v_cvt_f32_u32_sdwa " dst ", " src " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_" half "\n\t"
/* CHECK_FULL (trees deeper than the LDS stack, LdsStackHybrid): before a step, lanes whose LDS stack is full (top >= stack_limit)
   end the loop -- returns true, and the caller does that step in C++. */
template <int STACK_STRIDE, bool CHECK_FULL>      /* STACK_STRIDE: bytes between a lane's stack slots */
__device__ __forceinline__ bool bvh2q_node_loop_asm(int &node_io, uint32_t &stack_top, float tmax, const NodeqRay &R, int mx, int my, int mz,
                                                    const f4 *nodes_q, uint32_t image_address, int repeat, int leaf_threshold, int busy_max,
                                                    uint32_t stack_limit) {
    int r_node = node_io;
    uint32_t r_sp = stack_top;
    unsigned long long all, inner, hl, hr, t, u;      /* lane masks (scalar register pairs the compiler picks) */
    int cnt, slow;
    /* v[32:35] = x lo, x hi, y lo, y hi; v[36:39] = z lo, z hi, left link, right link; v40 .. v45: the near / far dwords per axis */
#define NORI_QLOOP_FULL_CHECK "v_cmp_le_u32 vcc, %[limit], %[sp]\n\ts_cbranch_vccnz 4f\n\t"
#define NORI_QLOOP(FULL_CHECK) \
    asm volatile( \
        "s_mov_b64 %[all], exec\n\t" \
        "s_mov_b32 %[slow], 0\n\t" \
        "v_cmp_lt_i32 %[t], -1, %[node]\n"                        /* lanes at an inner node */ \
        "1:\n\t" \
        "s_and_b64 exec, %[all], %[t]\n\t" \
        "s_cbranch_execz 3f\n\t" \
        FULL_CHECK \
        /* fetch: the node's record from the LDS image (link carries kTopBit: quad offset in its low bits) or from memory */ \
        "v_cmp_lt_u32 vcc, 0x3fffffff, %[node]\n\t" \
        "s_mov_b64 %[inner], exec\n\t" \
        "s_and_b64 exec, %[inner], vcc\n\t" \
        "v_and_b32 v40, 0xffff, %[node]\n\t" \
        "v_lshl_add_u32 v40, v40, 4, %[image]\n\t" \
        "ds_read_b128 v[32:35], v40\n\t" \
        "ds_read_b128 v[36:39], v40 offset:16\n\t" \
        NORI_EXP_Q_LDS \
        "s_andn2_b64 exec, %[inner], vcc\n\t" \
        "s_cbranch_execz 2f\n\t" \
        "v_lshlrev_b32 v40, 5, %[node]\n\t"                       /* (other lanes than the ones that just used v40) */ \
        "global_load_dwordx4 v[32:35], v40, %[nodes]\n\t" \
        "global_load_dwordx4 v[36:39], v40, %[nodes] offset:16\n" \
        NORI_EXP_Q_GLOBAL \
        "2:\n\t" \
        "s_mov_b64 exec, %[inner]\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        NORI_EXP_Q_ALU \
        /* per axis the dword with the near planes and the one with the far planes (mask = sign of the direction) */ \
        "v_bfi_b32 v40, %[mx], v33, v32\n\t" \
        "v_bfi_b32 v41, %[mx], v32, v33\n\t" \
        "v_bfi_b32 v42, %[my], v35, v34\n\t" \
        "v_bfi_b32 v43, %[my], v34, v35\n\t" \
        "v_bfi_b32 v44, %[mz], v37, v36\n\t" \
        "v_bfi_b32 v45, %[mz], v36, v37\n\t" \
        /* t = q A + (B -+ S) per plane, near = max over the axes, far = min; the dwords of the record are free by now. \
           (Both children of an axis in one v_pk_fma_f32 -- six packed instead of twelve plain multiply-adds -- measured 3.5 ms \
           per frame slower: the coefficients as register pairs cost the kernel a spill.) */ \
        NORI_SDWA("v32", "v40", "0") NORI_SDWA("v33", "v42", "0") NORI_SDWA("v34", "v44", "0") \
        "v_fma_f32 v32, v32, %[ax], %[bnx]\n\t" \
        "v_fma_f32 v33, v33, %[ay], %[bny]\n\t" \
        "v_fma_f32 v34, v34, %[az], %[bnz]\n\t" \
        "v_max3_f32 v32, v32, v33, v34\n\t"                       /* v32: near, left child */ \
        NORI_SDWA("v33", "v41", "0") NORI_SDWA("v34", "v43", "0") NORI_SDWA("v35", "v45", "0") \
        "v_fma_f32 v33, v33, %[ax], %[bfx]\n\t" \
        "v_fma_f32 v34, v34, %[ay], %[bfy]\n\t" \
        "v_fma_f32 v35, v35, %[az], %[bfz]\n\t" \
        "v_min3_f32 v33, v33, v34, v35\n\t"                       /* v33: far, left child */ \
        NORI_SDWA("v34", "v40", "1") NORI_SDWA("v35", "v42", "1") NORI_SDWA("v36", "v44", "1") \
        "v_fma_f32 v34, v34, %[ax], %[bnx]\n\t" \
        "v_fma_f32 v35, v35, %[ay], %[bny]\n\t" \
        "v_fma_f32 v36, v36, %[az], %[bnz]\n\t" \
        "v_max3_f32 v34, v34, v35, v36\n\t"                       /* v34: near, right child */ \
        NORI_SDWA("v35", "v41", "1") NORI_SDWA("v36", "v43", "1") NORI_SDWA("v37", "v45", "1") \
        "v_fma_f32 v35, v35, %[ax], %[bfx]\n\t" \
        "v_fma_f32 v36, v36, %[ay], %[bfy]\n\t" \
        "v_fma_f32 v37, v37, %[az], %[bfz]\n\t" \
        "v_min3_f32 v35, v35, v36, v37\n\t"                       /* v35: far, right child */ \
        "v_cmp_le_f32 vcc, v32, v33\n\t" \
        "v_cmp_le_f32 %[t], v32, %[tm]\n\t" \
        "v_cmp_le_f32 %[hl], 0, v33\n\t" \
        "v_cmp_le_f32 %[hr], v34, v35\n\t" \
        "v_cmp_le_f32 %[u], v34, %[tm]\n\t" \
        "s_and_b64 %[hl], %[hl], vcc\n\t" \
        "v_cmp_le_f32 vcc, 0, v35\n\t" \
        "s_and_b64 %[hl], %[hl], %[t]\n\t"                        /* left child hit */ \
        "s_and_b64 %[hr], %[hr], %[u]\n\t" \
        "s_and_b64 %[hr], %[hr], vcc\n\t"                         /* right child hit */ \
        /* selection: both -> the nearer one next, the other pushed; one -> that one; none -> pop.  "Left next" as a lane mask \
           (scalar unit): both and left nearer, or left alone */ \
        "v_cmp_le_f32 vcc, v32, v34\n\t" \
        "s_and_b64 %[t], %[hl], %[hr]\n\t"                        /* both */ \
        "s_or_b64 %[u], %[hl], %[hr]\n\t"                         /* any */ \
        "s_and_b64 vcc, vcc, %[t]\n\t" \
        "s_andn2_b64 %[hl], %[hl], %[hr]\n\t" \
        "s_or_b64 vcc, vcc, %[hl]\n\t" \
        "v_cndmask_b32 %[node], v39, v38, vcc\n\t"              /* the child walked next ... */ \
        "v_cndmask_b32 v40, v38, v39, vcc\n\t"                  /* ... and the other one */ \
        "s_mov_b64 exec, %[t]\n\t" \
        "ds_write_b32 %[sp], v40\n\t" \
        "v_add_u32 %[sp], %[stride], %[sp]\n\t" \
        "s_andn2_b64 exec, %[inner], %[u]\n\t" \
        "v_subrev_u32 %[sp], %[stride], %[sp]\n\t" \
        /* (the pop requested at the START of the step, with the node's record -- its link is dead by then -- so that no LDS trip ends \
           the step: bit-identical and no faster, headline 41.4 against 41.7 ms, profiles/r6_04_early_pop_ab.txt) */ \
        "ds_read_b32 %[node], %[sp]\n\t" \
        "s_mov_b64 exec, %[all]\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        /* again at once while enough lanes are at inner nodes -- or while some are and the trip around the loop outside would do \
           nothing else: no triangle step (too few lanes at a leaf) and no refill (too few idle lanes, or nothing to hand out) */ \
        "v_cmp_lt_i32 %[t], -1, %[node]\n\t" \
        "s_bcnt1_i32_b64 vcc_lo, %[t]\n\t" \
        "s_cmp_ge_u32 vcc_lo, %[repeat]\n\t" \
        "s_cbranch_scc1 1b\n\t" \
        "s_cmp_eq_u32 vcc_lo, 0\n\t" \
        "s_cbranch_scc1 3f\n\t" \
        "s_mov_b32 %[cnt], vcc_lo\n\t"                                /* lanes at inner nodes */ \
        "v_cmp_lt_u32 vcc, 0x80000000, %[node]\n\t"             /* lanes at a leaf */ \
        "s_bcnt1_i32_b64 vcc_lo, vcc\n\t" \
        "s_cmp_ge_u32 vcc_lo, %[leafth]\n\t" \
        "s_cbranch_scc1 3f\n\t" \
        "s_add_u32 vcc_lo, vcc_lo, %[cnt]\n\t"                        /* lanes with a ray in flight */ \
        "s_cmp_le_i32 vcc_lo, %[busymax]\n\t"                     /* idle lanes >= the refill threshold (and there is something to refill with) */ \
        "s_cbranch_scc0 1b\n" \
        "s_branch 3f\n" \
        "4:\n\t" \
        "s_mov_b32 %[slow], 1\n" \
        "3:\n\t" \
        "s_mov_b64 exec, %[all]\n\t" \
        : [node] "+v"(r_node), [sp] "+v"(r_sp), [all] "=&s"(all), [inner] "=&s"(inner), [hl] "=&s"(hl), [hr] "=&s"(hr), [t] "=&s"(t), [u] "=&s"(u), [cnt] "=&s"(cnt), [slow] "=&s"(slow) \
        : [tm] "v"(tmax), [ax] "v"(R.A[0]), [ay] "v"(R.A[1]), [az] "v"(R.A[2]), [bnx] "v"(R.Bn[0]), [bny] "v"(R.Bn[1]), [bnz] "v"(R.Bn[2]), \
          [bfx] "v"(R.Bf[0]), [bfy] "v"(R.Bf[1]), [bfz] "v"(R.Bf[2]), [mx] "v"(mx), [my] "v"(my), [mz] "v"(mz), \
          [nodes] "s"(nodes_q), [image] "s"(image_address), [repeat] "s"(repeat), [leafth] "s"(leaf_threshold), [busymax] "s"(busy_max), [limit] "s"(stack_limit), [stride] "n"(STACK_STRIDE) \
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", \
          "vcc", "scc", "memory")
    if (CHECK_FULL) { NORI_QLOOP(NORI_QLOOP_FULL_CHECK); } else { NORI_QLOOP(""); }
#undef NORI_QLOOP
#undef NORI_QLOOP_FULL_CHECK
    node_io = r_node;
    stack_top = r_sp;
    return slow != 0;
}

/* hit_pack (wf_records.h) for wf_extend: the constants of "no closest hit" (t = inf, u = v = 0) are made from a zero the compiler
   cannot see through -- it otherwise keeps them as a register quad across the whole kernel, and with the register budget of 8 waves
   per SIMD that quad is what it spills (a scratch reload in front of every record store) */
__device__ __forceinline__ f4 hit_pack_here(const Hit *closest, bool shadow_occluded) {
    float z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    f4 h; h.x = u2f(0x7f800000u | f2u(z)); h.y = z; h.z = z;
    uint32_t w = kMissA;
    if (closest) { h.x = closest->t; h.y = closest->u; h.z = closest->v; if (closest->tri != kNoHit) w = closest->tri; }
    h.w = u2f(w | (shadow_occluded ? kOccludedB : 0u));
    return h;
}

/* Accel::rayIntersect for every path of copy `cur`: the shadow ray (if any) first, then the
   continuation ray, by the same lane; one 16-B hit record per path. */
/* BLOCK: threads per workgroup.  Every workgroup holds its own copy of the LDS image next to its stacks, so the bigger the
   workgroup the bigger the image can be: BVH2 trees run in workgroups of 1024 threads (two per CU: 2 x (68 KB of stacks +
   12 KB of image)), wide-node trees -- whose kernels need more than 64 registers, i.e. 5 or 6 waves per SIMD -- in workgroups of 256. */
template <int STACK, bool SPILL, bool COUNT, int MODE, bool WIDE, bool ASM, int BLOCK>
__global__ __launch_bounds__(BLOCK, WIDE ? (MODE != kStoredOnly ? 5 : 7) : 8) void wf_extend(DevScene sc, WfBuf b, int cur, int thresholds, WfBatch bt) {
    const int refill_threshold = thresholds & 0xff, leaf_threshold = (thresholds >> 8) & 0xff;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename ExtendStack<STACK, SPILL, ASM, BLOCK>::type Stack;
    Stack stack;
    stack.init(smem, b.stack_spill, gridDim.x * BLOCK);
    /* the first levels of the tree in LDS (rt_trace.h, node_fetch): behind the stacks */
    f4 *top = reinterpret_cast<f4 *>(smem + (size_t) Stack::kLdsEntries * BLOCK * sizeof(int));
    /* the kernel with the hand-written node loop walks the 32-B records: its image holds those (and so does the counting twin's) */
    const bool count_q = COUNT && !WIDE && (bt.flags & kBatchCountQ) != 0u;
    const int root_link = (ASM || count_q) ? top_image_to_lds(sc, top, sc.top_image_q, sc.top_image_q_quads) : top_image_to_lds(sc, top, sc.top_image, sc.top_image_quads);
    const TopNodesP top_lds = top_nodes_pointer(top);
    const uint32_t image_address = lds_address(smem) + (uint32_t) (Stack::kLdsEntries * BLOCK * sizeof(int));      /* of top */
    const WfState S = b.st[cur];
    static_assert(MODE == kStoredOnly || MODE == kFreshOnly, "wf_extend: stored paths and new paths are traced by separate launches");
    constexpr bool kFresh = MODE == kFreshOnly, kStored = MODE == kStoredOnly;
    const PassShape shape = pass_shape<MODE>(b.ctr, cur, bt);
    /* this launch's paths: the stored ones, indices [0, n_s), or the new ones, indices first + [0, n_fresh) */
    const uint32_t n = kFresh ? shape.n_fresh : shape.n_s, first = kFresh ? shape.n_f0 : 0u;
    /* the other copy's counters are free by now (its paths were consumed by the previous wf_shade):
       reset them for the wf_shade that follows this kernel and for the next wf_extend; the next pass starts its new camera
       samples where this one stops (a pass's stored-path launch passes the counter on unchanged, the launch of the new paths --
       behind it on the stream -- adds what it starts) */
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        b.ctr[C_N + (cur ^ 1)] = 0u; b.ctr[C_HEAD + (cur ^ 1)] = 0u; b.ctr[C_HEAD_FRESH + (cur ^ 1)] = 0u;
        b.ctr[C_SAMPLE + (cur ^ 1)] = shape.s0 + shape.n_fresh;
    }
    const int lane = lane_id();
    Trav tv; trav_idle(tv);
    uint32_t rid = 0;            /* path << 2 | continuation pending << 1 | shadow ray occluded */
    bool unsaved = false;        /* the lane's last query is answered (tv.hit) but its record is not written yet */
    bool exhausted = n == 0;
    /* Paths are handed to waves in chunks: every wave starts with a static chunk (no atomic), the
       rest -- about half, for load balance -- is claimed dynamically with one atomic per chunk.  The
       chunk size follows the number of paths: big chunks amortise the atomic, small ones keep all
       waves busy; in the long tail of a batch the static chunks cover everything and no wave
       touches the counter (a single hot word sustains only ~90 atomics/us --
       MI355X_MICROARCH.md, row "dequeue" -- which used to cost 0.2 ms per tail iteration). */
    /* wave_id is the same in all lanes of a wave: readfirstlane tells the compiler, so the chunk cursor lives in SGPRs */
    const uint32_t n_waves = gridDim.x * (BLOCK / 64u), wave_id = (uint32_t) __builtin_amdgcn_readfirstlane((int) (blockIdx.x * (BLOCK / 64u) + (threadIdx.x >> 6)));
    const uint32_t static_limit = (uint32_t) ((thresholds >> 16) & 0xfff) * n_waves;
    const uint32_t kChunk = n <= static_limit ? (((n + n_waves - 1u) / n_waves + 63u) & ~63u)      /* all static */
                                              : min(1024u, max(64u, (n / (n_waves * (uint32_t) ((thresholds >> 28) & 0xf))) & ~63u));
    const uint32_t dyn0 = n_waves * kChunk;        /* first dynamically claimed path */
    uint32_t chunk_pos = min(wave_id * kChunk, n), chunk_end = min(chunk_pos + kChunk, n);       /* wave-uniform */
    uint32_t nClosest = 0, nShadow = 0, nCam = 0;      /* wave-uniform: counted from ballots at the refill */
    uint32_t zc[Z_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};      /* wave-uniform census */
    TraversalCounters tc; tc.nodes = 0; tc.tris = 0;
    NORI_PROF_DECL
    while (true) {
        /* a lane is idle when it has no ray in flight: either it needs a new path, or its path's
           shadow ray is answered and the continuation ray is still to be traced (rid bit 1) */
        const bool pend = !trav_active(tv) && (rid & 2u) != 0u;
        const unsigned long long idle = __ballot(!trav_active(tv)), pending = __ballot(pend);
        const int nIdle = __popcll(idle);
        if ((!exhausted || pending != 0ull) && (nIdle >= refill_threshold || nIdle == 64)) {
            /* the wave owns [chunk_pos, chunk_end) and hands it out to its idle lanes */
            if (!exhausted && chunk_pos >= chunk_end) {
                uint32_t base = n;
                if (dyn0 < n) {
                    if (lane == 0) base = dyn0 + atomicAdd(&b.ctr[(kFresh ? C_HEAD_FRESH : C_HEAD) + cur], kChunk);
                    base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
                }
                chunk_pos = base; chunk_end = min(base + kChunk, n);
                if (base >= n) { exhausted = true; chunk_pos = chunk_end = 0u; }
            }
            /* Results are written HERE, not in the trip a query ends: on average five lanes of a wave finish per trip,
               so storing right away ran the record packing at 5 / 64 lanes nearly every trip; at a refill >= 32 lanes are
               idle and write together.  An idle lane keeps its answer in tv.hit until then. */
            if (unsaved && !trav_active(tv)) {
                if (pend) rid |= tv.hit.tri != kNoHit ? 1u : 0u;      /* the shadow ray is answered; the continuation ray of the same vertex is next */
                else st_f4<1>(&b.hit[rid >> 2], tv.any ? hit_pack_here(nullptr, tv.hit.tri != kNoHit) : hit_pack_here(&tv.hit, (rid & 1u) != 0u));
                unsaved = false;
            }
            const unsigned long long fresh = idle & ~pending;
            if (COUNT) { zc[Z_REFILLS]++; zc[Z_REFILL_LANES] += (uint32_t) nIdle; }
            const uint32_t avail = chunk_end - chunk_pos;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (fresh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) fresh, 0u));   /* set bits below this lane */
            bool startedA = false, startedB = false, startedCam = false;      /* this lane starts a closest-hit / a shadow query / a camera ray in this refill */
            const bool take = !pend && !trav_active(tv) && rank < avail;      /* this lane takes path chunk_pos + rank */
            if (kFresh && take) {      /* a new path: camera sample s0 + (i - n_f0) of the batch */
                const uint32_t i = first + chunk_pos + rank, sid = shape.s0 + chunk_pos + rank;
                f2 ps; RayIn ray; Rng rng;
                if (first_vertex(sc, bt, sid, ps, ray, rng)) {
                    st_f2<1>(&b.samp_pos[sid], ps);
                    rid = i << 2;
                    trav_begin<WIDE ? kLayoutWide : kLayoutBvh2>(sc, ray, false, stack, tv);
                    startedA = startedCam = true;
                    unsaved = trav_active(tv);
                    if (unsaved) tv.node = root_link;
                    if (!trav_active(tv)) st_f4<1>(&b.hit[i], hit_pack_here(nullptr, false));
                }
            } else if (kStored && (pend || take)) {
                const uint32_t i = pend ? (rid >> 2) : chunk_pos + rank;
                /* a fresh path: origin and both directions (the flags ride with the continuation direction) are requested
                   together -- one round trip to HBM instead of two (the direction a path needs first depends on its
                   flags).  A path whose shadow ray was just answered needs its continuation direction again (16 B; its
                   lane could not afford to keep it in registers during the shadow walk) -- the origin it still has. */
                f4 dB0; dB0.x = dB0.y = dB0.z = dB0.w = 0.0f;
                P3 o0; o0.x = tv.o.x; o0.y = tv.o.y; o0.z = tv.o.z;      /* a path whose shadow ray was just answered: the origin is still here */
                const f4 dA0 = S.dA[i];
                if (!pend) { o0 = S.o[i]; dB0 = S.dB[i]; }
                /* (the compiler sinks loads below the test of the flags -- a second trip to HBM per refill; naming them as inputs
                   of an empty asm statement pins all of them in front of it) */
                asm volatile("" :: "v"(o0.x), "v"(dA0.x), "v"(dB0.x));
                const uint32_t fl = pend ? F_HAS_A : state_flags(dA0);
                if (fl & (F_HAS_A | F_HAS_B)) {      /* 0: empty slot */
                    const bool any = (fl & F_HAS_B) != 0u;
                    const f4 d = any ? dB0 : dA0;
                    RayIn ray; ray.o = mk3(o0.x, o0.y, o0.z); ray.d = mk3(d.x, d.y, d.z);
                    ray.mint = kStoredMint; ray.maxt = any ? dB0.w : kInf;
                    rid = pend ? (rid & ~2u) : ((i << 2) | ((any && (fl & F_HAS_A)) ? 2u : 0u));
                    trav_begin<WIDE ? kLayoutWide : kLayoutBvh2>(sc, ray, any, stack, tv);
                    startedA = !any; startedB = any;
                    unsaved = trav_active(tv);
                    if (unsaved) tv.node = root_link;
                    if (!trav_active(tv)) {            /* empty scene: nothing occludes, nothing is hit */
                        if (rid & 2u) { startedA = true; rid &= ~2u; }      /* the continuation ray counts as traced, too */
                        st_f4<1>(&b.hit[i], hit_pack_here(nullptr, false));
                    }
                }
            }
            chunk_pos += min(avail, (uint32_t) __popcll(fresh));
            nClosest += (uint32_t) __popcll(__ballot(startedA));
            if (kFresh) nCam += (uint32_t) __popcll(__ballot(startedCam));
            if (kStored) nShadow += (uint32_t) __popcll(__ballot(startedB));
        }
        if (__ballot(trav_active(tv)) == 0ull) {
            if (exhausted && __ballot((rid & 2u) != 0u) == 0ull) break;
            continue;
        }
        NORI_PROF_MARK(prof_refill)
        /* inner-node steps run every trip, and again at once while enough lanes are at inner nodes (a tight loop without the
           refill test and the leaf vote: the trip's bookkeeping costs as much as a node step); the (rarer) triangle
           step only when enough lanes wait at a leaf or nobody has an inner node to test */
        if (COUNT) zc[Z_TRIPS]++;
        if constexpr (ASM) {
            static_assert(!WIDE && !COUNT, "the hand-written node loop: BVH2 nodes as 32-B records");
            /* the ray's plane coefficients (rt_nodeq.h), made anew for every pass through here: kept across the triangle step and
               the refill they would cost twelve registers the kernel does not have */
            f3 qo = tv.o, qr = tv.rcp;
            asm volatile("" : "+v"(qo.x), "+v"(qo.y), "+v"(qo.z), "+v"(qr.x), "+v"(qr.y), "+v"(qr.z));
            NodeqRay R;
            nodeq_ray(sc.grid, qo, qr, R);
            uint32_t stack_limit = 0xffffffffu;
            if constexpr (SPILL) stack_limit = stack.limit;
            const bool full = bvh2q_node_loop_asm<BLOCK * 4, SPILL>(tv.node, stack.top, tv.hit.t, R, (int) f2u(qr.x) >> 31, (int) f2u(qr.y) >> 31, (int) f2u(qr.z) >> 31,
                                           sc.nodes_q, image_address, bt.inner_repeat, leaf_threshold,
                                           /* lanes in flight at or below which the refill at the top of the loop would run: never (-1) if
                                              it has nothing to hand out */
                                           (!exhausted || __ballot((rid & 2u) != 0u) != 0ull) ? 64 - refill_threshold : -1, stack_limit);
            /* a lane's LDS stack is full (trees deeper than it): this step in C++, through the stack's global part */
            if (SPILL && full && trav_at_inner(tv)) trav_inner_step_q<false>(sc, stack, tv, tc, top_lds);
        } else
        do {
            if (COUNT) { const int ni = __popcll(__ballot(trav_at_inner(tv))); if (ni) { zc[Z_INNER_TRIPS]++; zc[Z_INNER_LANES] += (uint32_t) ni; } }
            if (trav_at_inner(tv)) {
                if (WIDE) trav_wide_step<COUNT>(sc, stack, tv, tc, top_lds);      /* BVH4, quantised boxes: scenes beyond the caches */
                else if (COUNT && count_q) trav_inner_step_q<COUNT>(sc, stack, tv, tc, top_lds);
                else trav_inner_step<COUNT>(sc, stack, tv, tc, top_lds);
            }
        } while (__popcll(__ballot(trav_at_inner(tv))) >= bt.inner_repeat);
        NORI_PROF_MARK(prof_node)
        const bool atLeaf = trav_at_leaf(tv);
        const int nLeaf = __popcll(__ballot(atLeaf));
        const bool innerLeft = __ballot(trav_at_inner(tv)) != 0ull;
        if (COUNT && nLeaf && (nLeaf >= leaf_threshold || !innerLeft)) { zc[Z_LEAF_TRIPS]++; zc[Z_LEAF_LANES] += (uint32_t) nLeaf; }
        if (atLeaf && (nLeaf >= leaf_threshold || !innerLeft)) trav_leaf_step<COUNT, false>(sc, stack, tv, tc, top_lds);
        NORI_PROF_MARK(prof_leaf)
    }
    /* the answers still held in registers when the wave ran out of paths (no lane is pending here: the loop ends only
       when no continuation ray is left): a path with a shadow ray only ends on it (tv.any); otherwise the closest hit
       + the shadow answer */
    if (unsaved) st_f4<1>(&b.hit[rid >> 2], tv.any ? hit_pack_here(nullptr, tv.hit.tri != kNoHit) : hit_pack_here(&tv.hit, (rid & 1u) != 0u));
    /* The ray counters: one atomic per WORKGROUP and counter.  One per wave -- 8192 waves x 2 - 3 atomics into one cache line, which takes
       ~90 of them per microsecond -- was a floor of ~0.18 ms under every launch whose waves end together: the small passes of a batch's tail,
       every pass of one device's share of eight.  The waves' stacks are free by now: the first words of the workgroup's LDS collect the sums. */
#ifdef NORI_LAB_WAVE_COUNTERS      /* A/B (variant builds): the counters per wave, as rounds 1 - 5 had them */
    if (lane == 0) {
        if (nClosest) atomicAdd(&b.stats[S_CLOSEST], (unsigned long long) nClosest);
        if (nShadow) atomicAdd(&b.stats[S_SHADOW], (unsigned long long) nShadow);
        if (kFresh && nCam) atomicAdd(&b.stats[S_CAM], (unsigned long long) nCam);
    }
#else
    __syncthreads();
    uint32_t *s_sum = reinterpret_cast<uint32_t *>(smem);
    if (threadIdx.x < 4u) s_sum[threadIdx.x] = 0u;
    __syncthreads();
    if (lane == 0) {
        if (nClosest) atomicAdd(&s_sum[0], nClosest);
        if (nShadow) atomicAdd(&s_sum[1], nShadow);
        if (kFresh && nCam) atomicAdd(&s_sum[2], nCam);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_sum[0]) atomicAdd(&b.stats[S_CLOSEST], (unsigned long long) s_sum[0]);
        if (s_sum[1]) atomicAdd(&b.stats[S_SHADOW], (unsigned long long) s_sum[1]);
        if (kFresh && s_sum[2]) atomicAdd(&b.stats[S_CAM], (unsigned long long) s_sum[2]);
    }
#endif
    /* (counting builds: per wave, they are not timed) */
    if (COUNT) for (int off = 32; off > 0; off >>= 1) { tc.nodes += (uint32_t) __shfl_down((int) tc.nodes, off); tc.tris += (uint32_t) __shfl_down((int) tc.tris, off); }
    if (lane == 0) {
        if (COUNT) { atomicAdd(&b.stats[S_NODES], (unsigned long long) tc.nodes); atomicAdd(&b.stats[S_TRIS], (unsigned long long) tc.tris); }
        NORI_PROF_STORE
        if (COUNT && b.census) for (int k = 0; k < Z_COUNT; ++k) if (zc[k]) atomicAdd(&b.census[k], (unsigned long long) zc[k]);
    }
}

/* Integrator::Li, one vertex: consume the shadow result, shade the closest hit, write the surviving
   path -- with its next shadow / continuation rays -- compacted into the other state copy.
   A workgroup owns a contiguous range of rounds (256 paths each) and reserves output space in
   chunks of <= kShadeChunk records with one atomic per chunk, sized by the survival rate it sees;
   what it leaves unused of its last chunk is marked empty (flags = 0) and dropped by the next pass. */
constexpr int kSB = 256;      /* threads per wf_shade workgroup (128 / 64: slower, DESIGN_HISTORY.md) */
static_assert(kSB == 256, "pass_shape starts the new paths of a pass at a multiple of 256");

/* wf_shade's workgroup barriers wait for LDS only: what the compaction exchanges lives there.  (__syncthreads() is also a fence,
   i.e. s_waitcnt vmcnt(0): a wave then sits out the trip of its own state STORES to L2 in every round.) */
__device__ __forceinline__ void shade_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* LDSTAB: the scene's small tables fit LDS (shade_tables.h) -- else they are read from global memory (SceneTables) */
template <bool LDSTAB> struct ShadeTab { typedef SceneTables type; static __device__ __forceinline__ type make(const DevScene &sc, uint4 *) { type t = {&sc}; return t; } };
template <> struct ShadeTab<true> { typedef LdsTables type; static __device__ __forceinline__ type make(const DevScene &sc, uint4 *s_tab) { return shade_tables_copy(sc, s_tab); } };

/* Software pipelining across rounds: the head of the next round's record (continuation direction with the flags, sample index,
   radiance, hit) is requested while this round computes, so that a round starts with its first dependent loads already answered
   (round 3: wf_shade 27.1 -> 26.0 ms; round 5, all-diffuse kernel: 23.7 -> 23.1).  Going further does not pay -- the WHOLE record
   of the next round sent straight into LDS with every gather of the vertex issued at the top of the round: bit-identical and
   0.5 - 1 ms slower (profiles/r4_02_shade_pipeline_ab.txt). */

/* MATSET (rt_path.h): the BSDF types the scene contains -- the kernel of an all-diffuse scene carries no mirror, dielectric or
   microfacet code (the all-diffuse kernel: 117 registers instead of 128; held to the 96 of five workgroups per CU it spills and is
   slower, profiles/r5_01_shade_matset_ab.txt) */
#ifndef NORI_SHADE_WGS
#define NORI_SHADE_WGS 4      /* workgroups per SIMD wf_shade is compiled for (variant builds: 5, 6) */
#endif
template <int INTEG, int MODE, bool LDSTAB, int MATSET>
__global__ __launch_bounds__(kSB, NORI_SHADE_WGS) void wf_shade(DevScene sc, WfBuf b, int cur, WfBatch bt) {
    NORI_LAB_SHADE_PAD
    const WfState S = b.st[cur], D = b.st[cur ^ 1];
    const uint32_t s_first = bt.s_first, n_spp = bt.n_spp;
    __shared__ uint4 s_tab[LDSTAB ? kShadeTabWords / 4 : 1];
    const typename ShadeTab<LDSTAB>::type tab = ShadeTab<LDSTAB>::make(sc, s_tab);      /* mesh / emitter tables: LDS instead of L2 round trips */
    constexpr bool kFresh = MODE != kStoredOnly, kStored = MODE != kFreshOnly;
    const PassShape shape = pass_shape<MODE>(b.ctr, cur, bt);      /* the pass wf_extend just traced: the same counters */
    const uint32_t n = shape.n, n_s = shape.n_s;
    const uint32_t rounds_total = (n + kSB - 1) / kSB;
    const uint32_t rounds_per_block = (rounds_total + gridDim.x - 1) / gridDim.x;
    const uint32_t r0 = blockIdx.x * rounds_per_block, r1 = min(rounds_total, r0 + rounds_per_block);
    /* rounds below r_mid hold stored paths, rounds from r_mid on new ones (pass_shape: the new paths start at a multiple of kSB) */
    const uint32_t r_mid = kStored ? min(r1, max(r0, shape.n_f0 / kSB)) : r0;
    __shared__ uint32_t s_wcnt[2][4], s_newbase;
    uint32_t out_base = 0u, out_used = 0u, out_len = 0u;     /* workgroup-uniform */
    bool overflow = false;
    const int wave = (int) (threadIdx.x >> 6), lane = lane_id();
    const uint32_t per_tile = 256u * n_spp;
    uint32_t pf_sidx = 0u; f4 pf_d, pf_L, pf_h;
    pf_d.x = pf_d.y = pf_d.z = pf_d.w = 0.0f; pf_L = pf_h = pf_d;
    /* One round: 256 paths, one vertex each.  FRESH rounds hold new paths -- camera sample s0 + (i - n_f0), nothing stored: the
       first vertex is recomputed from the sample index --, the others stored ones, whose record head was requested a round ahead.
       Two instantiations in two loops (a workgroup's range of rounds is stored rounds, then new ones): in ONE loop body the
       camera code and the prefetched head compete for registers and the kernel spills. */
    auto round = [&](auto fresh_tag, uint32_t r, uint32_t r_end) {
        constexpr bool FRESH = decltype(fresh_tag)::value;
        const uint32_t i = r * kSB + threadIdx.x;
        bool survive = false;
        const uint32_t c_sidx = pf_sidx; const f4 c_d = pf_d, c_L = pf_L, c_h = pf_h;
        if (!FRESH && r + 1u < r_end && i + kSB < n_s) {
            pf_d = ld_f4<2>(&S.dA[i + kSB]); pf_sidx = ld_w<2>(&S.sidx[i + kSB]); pf_L = ld_f4<2>(&S.L_pdf[i + kSB]); pf_h = ld_f4<1>(&b.hit[i + kSB]);
        }
        f4 n_o, n_dA, n_dB, n_T, n_L, n_Ld;
        uint32_t n_fl = 0u, sidx = 0u;
        unsigned long long n_rng = 0ull;
        if (FRESH ? i < n : i < n_s) {
            /* the path's state: from HBM, or -- first vertex -- recomputed from the sample index */
            uint32_t fl; f4 L4, d4, t4; Rng rng0; rng0.state = 0; rng0.inc = 0;
            if (FRESH) {
                f2 ps; RayIn cam;
                sidx = shape.s0 + (i - shape.n_f0);
                fl = first_vertex(sc, bt, sidx, ps, cam, rng0) ? (F_HAS_A | (2u << 4)) : 0u;      /* prev_measure = discrete, depth 0 */
                L4.x = L4.y = L4.z = L4.w = 0.0f;                  /* L = 0, pdf_mat = 0 */
                t4.x = t4.y = t4.z = t4.w = 1.0f;                  /* T = 1, eta = 1 */
                d4.x = cam.d.x; d4.y = cam.d.y; d4.z = cam.d.z; d4.w = 0.0f;
            } else {
                d4 = c_d;               /* the continuation direction and, in its w, the flags (wf_records.h) */
                fl = state_flags(d4);
            }
            if (fl & (F_HAS_A | F_HAS_B)) {
                if (!FRESH) { sidx = c_sidx; L4 = c_L; }
                const f4 h = FRESH ? ld_f4<1>(&b.hit[i]) : c_h;
                const uint32_t hw = __float_as_uint(h.w);
                bool done = false;
                NORI_LAB_SHADE_VERTEX
                if (fl & F_HAS_B) {                       /* path_on_shadow: add the emitter sample if unoccluded */
                    if (!(hw & kOccludedB)) {
                        const P3 ld = ld_p3<2>(&S.Ld[i]);
                        L4.x = L4.x + ld.x; L4.y = L4.y + ld.y; L4.z = L4.z + ld.z;
                    }
                    if (fl & F_END_AFTER_B) done = true;
                }
                PathState st;
                st.L = mk3(L4.x, L4.y, L4.z);
                if (!done) {
                    if (!FRESH) t4 = ld_f4<2>(&S.T_eta[i]);
                    Hit hit; bool found;
                    hit_unpack(sc, h, hit, found);
                    /* pcg32 stream of this camera sample: inc from the sample index, state from HBM */
                    const uint32_t sl = (sidx % per_tile) >> 8;
                    uint64_t rng_state = rng0.state;
                    if (!FRESH) rng_state = ld_w<2>(&S.rng[i]);
                    vertex_unpack(st, fl, L4, t4, rng_state, ((uint64_t) (s_first + sl) << 1u) | 1u);
                    done = path_on_closest<INTEG, typename ShadeTab<LDSTAB>::type, MATSET>(sc, tab, st, hit, found, mk3(d4.x, d4.y, d4.z));
                    if (!done) {
                        survive = true;
                        vertex_pack(st, n_o, n_dA, n_dB, n_T, n_L, n_Ld, n_fl);
                        n_rng = st.rng.state;
                    }
                }
                if (done) {
                    P3 out; out.x = st.L.x; out.y = st.L.y; out.z = st.L.z;
                    st_p3<2>(&b.samp_L[sidx], out);
                }
            }
        }
        /* compaction: rank of this survivor among the workgroup's survivors of the round */
        const unsigned long long mask = __ballot(survive);
        if (lane == 0) s_wcnt[r & 1u][wave] = (uint32_t) __popcll(mask);
        shade_barrier();
        const uint32_t c0 = s_wcnt[r & 1u][0], c1 = kSB > 64 ? s_wcnt[r & 1u][1] : 0u, c2 = kSB > 128 ? s_wcnt[r & 1u][2] : 0u, c3 = kSB > 192 ? s_wcnt[r & 1u][3] : 0u;
        const uint32_t c = c0 + c1 + c2 + c3;
        uint32_t off = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
        off += wave > 0 ? c0 : 0u; off += wave > 1 ? c1 : 0u; off += wave > 2 ? c2 : 0u;
        /* the round's survivors fill what is left of the current chunk, the rest opens a new one */
        const uint32_t room = out_len - out_used;
        uint32_t new_base = 0u, want = 0u;
        if (c > room) {                                  /* workgroup-uniform: reserve the next chunk */
            const uint32_t expect = c * (r1 - r);        /* survivors from here to the end of the range, at this round's rate */
            want = r + 1u == r1 ? c - room : min(kShadeChunk, max(c - room, ((expect + expect / 8u + 63u) & ~63u) - min(room, expect)));
            if (threadIdx.x == 0) s_newbase = atomicAdd(&b.ctr[C_N + (cur ^ 1)], want);
            shade_barrier();
            new_base = s_newbase;
            if (new_base + want > b.capacity) overflow = true;
        }
        if (survive && !overflow) {
            /* the record as it lies in HBM (wf_records.h): 100 B; the flags ride with the continuation direction, which is
               therefore always written */
            const uint32_t j = off < room ? out_base + out_used + off : new_base + (off - room);
            st_p3<2>(&D.o[j], p3_of(n_o)); st_f4<2>(&D.T_eta[j], n_T); st_f4<2>(&D.L_pdf[j], n_L);
            st_f4<2>(&D.dA[j], state_dA(n_dA, n_fl, (n_fl & F_HAS_A) != 0u)); st_w<2>(&D.rng[j], n_rng); st_w<2>(&D.sidx[j], sidx);
            if (n_fl & F_HAS_B) { st_f4<2>(&D.dB[j], n_dB); st_p3<2>(&D.Ld[j], p3_of(n_Ld)); }
            NORI_LAB_SHADE_STORE
        }
        if (c > room) { out_base = new_base; out_used = c - room; out_len = want; }
        else out_used += c;
    };
    if (kStored) {
        if (r0 < r_mid && r0 * kSB + threadIdx.x < n_s) {
            const uint32_t i0 = r0 * kSB + threadIdx.x;
            pf_d = ld_f4<2>(&S.dA[i0]); pf_sidx = ld_w<2>(&S.sidx[i0]); pf_L = ld_f4<2>(&S.L_pdf[i0]); pf_h = ld_f4<1>(&b.hit[i0]);
        }
        for (uint32_t r = r0; r < r_mid; ++r) round(std::false_type(), r, r_mid);
    }
    if (kFresh) for (uint32_t r = r_mid; r < r1; ++r) round(std::true_type(), r, r1);
    /* what is left of the last chunk holds no path: flags 0 */
    for (uint32_t k = out_used + threadIdx.x; k < out_len; k += kSB) { f4 z; z.x = z.y = z.z = z.w = 0.0f; D.dA[out_base + k] = z; }
    if (overflow && threadIdx.x == 0) b.ctr[C_OVERFLOW] = 1u;
}

/* The long tail of a batch: once only a few thousand paths are alive, every wf_extend / wf_shade pair
   is a launch that lasts as long as ONE path vertex takes (~0.1 ms) however few paths there are.
   wf_finish ends it with one launch: each lane takes a path and walks it to its end -- shadow ray,
   continuation ray, Li vertex, repeat -- the megakernel's loop started from stored path state.
   (On a device of its own the launch lasts as long as the LONGEST path, ~1300 vertices through glass at p = 0.99, one after the
   other, however its lanes are used: a kernel that re-compacts a workgroup's paths after every vertex keeps the lanes dense and
   is no faster, profiles/r4_04_tail_per_vertex_ab.txt.  Nor does walking through
   the LDS image of the hot records help here -- pa5 table, 64 spp: 44.9 against 42.0 ms of shade time, a 1/8 share of the Cornell
   box 3.99 against 3.56: 2048 workgroups each copy 12 KB for a few hundred paths, and the C++ form of the 32-B node step pays 24
   instructions per step for the ray's plane coefficients.  Round 6, on the small persistent grid, through the image of the 64-B nodes:
   bit-identical and no faster either -- a path inside the glass sphere walks the BOTTOM of the tree --, profiles/r6_06_finish_image_ab.txt.) */
/* The lanes of a small persistent grid pull paths one by one -- a lane whose path has ended takes the next unclaimed one at the top
   of the vertex loop (one atomic per wave and refill), so a wave's lanes stay busy while paths are left and all of them are at the
   same stage of a vertex.  (Round 4's form -- one lane per path, a grid of finish_paths lanes -- ran at 6.5 of 64 lanes: pa5 table
   scene 22 -> 19 ms per tail on all CUs, 223 -> 80 ms on 16 CUs, where the tail of a batch runs when the NEXT batch runs beside it:
   what it costs there is CU time, not the length of the longest path.  profiles/r5_06_tail_probe_c4.txt.)  Same arithmetic per
   path, hence the same radiance; which lane walks a path is not observable. */
template <int INTEG>
__global__ __launch_bounds__(kB) void wf_finish(DevScene sc, WfBuf b, int cur, WfBatch bt, int count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsStackW<16, true> stack;
    stack.init(smem, b.stack_spill, gridDim.x * kB);
    __shared__ uint4 s_tab[kShadeTabWords / 4];
    shade_tables_to_lds(sc, s_tab);
    const WfState S = b.st[cur];
    const uint32_t n = b.ctr[C_N + cur];
    const uint32_t per_tile = 256u * bt.n_spp;
    uint32_t nClosest = 0, nShadow = 0;
    TraversalCounters tc; tc.nodes = 0; tc.tris = 0;
    bool have = false, dry = false;      /* this lane walks a path / the paths have run out */
    uint32_t sidx = 0u, fl = 0u;
    f4 o, dA, dB, T, L, Ld;
    o.x = o.y = o.z = o.w = 0.0f; dA = dB = T = L = Ld = o;
    unsigned long long rng_state = 0ull;
    while (true) {
        const unsigned long long want = __ballot(!have && !dry);
        if (want != 0ull) {
            uint32_t base = 0u;
            if (lane_id() == 0) base = atomicAdd(&b.ctr[C_TAIL_HEAD], (uint32_t) __popcll(want));
            base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
            const uint32_t i = base + __builtin_amdgcn_mbcnt_hi((uint32_t) (want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) want, 0u));
            if (!have && !dry) {
                if (i >= n) dry = true;
                else {
                    dA = S.dA[i];
                    fl = state_flags(dA);
                    if (fl & (F_HAS_A | F_HAS_B)) {      /* (else: an empty slot -- the lane asks again in the next trip) */
                        have = true;
                        sidx = S.sidx[i]; dB = S.dB[i]; T = S.T_eta[i]; L = S.L_pdf[i];
                        const P3 o3 = S.o[i], l3 = S.Ld[i];
                        o.x = o3.x; o.y = o3.y; o.z = o3.z; o.w = kStoredMint; Ld.x = l3.x; Ld.y = l3.y; Ld.z = l3.z; Ld.w = 0.0f;
                        dA.w = kInf;
                        rng_state = S.rng[i];
                    }
                }
            }
        }
        if (__ballot(have) == 0ull) { if (__ballot(!dry) == 0ull) break; continue; }
        if (have) {      /* one vertex */
            bool fin = false;
            if (fl & F_HAS_B) {
                RayIn ray; ray.o = mk3(o.x, o.y, o.z); ray.d = mk3(dB.x, dB.y, dB.z); ray.mint = kEpsilon; ray.maxt = dB.w;
                Hit sh; ++nShadow;
                if (!traverse<true>(sc, ray, true, stack, sh, tc)) { L.x = L.x + Ld.x; L.y = L.y + Ld.y; L.z = L.z + Ld.z; }
                fin = (fl & F_END_AFTER_B) != 0u;
            }
            if (!fin) {
                RayIn ray; ray.o = mk3(o.x, o.y, o.z); ray.d = mk3(dA.x, dA.y, dA.z); ray.mint = o.w; ray.maxt = dA.w;
                Hit hit; ++nClosest;
                const bool found = traverse<true>(sc, ray, false, stack, hit, tc);
                if (found) hit.mesh = f2u(sc.shade_tris[(size_t) hit.tri * kShadeQuads].w);
                PathState st;
                vertex_unpack(st, fl, L, T, rng_state, ((uint64_t) (bt.s_first + ((sidx % per_tile) >> 8)) << 1u) | 1u);
                fin = path_on_closest<INTEG>(sc, st, hit, found, ray.d);
                L.x = st.L.x; L.y = st.L.y; L.z = st.L.z;
                if (!fin) { vertex_pack(st, o, dA, dB, T, L, Ld, fl); rng_state = st.rng.state; }
            }
            if (fin) { P3 out; out.x = L.x; out.y = L.y; out.z = L.z; b.samp_L[sidx] = out; have = false; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        nClosest += (uint32_t) __shfl_down((int) nClosest, off);
        nShadow += (uint32_t) __shfl_down((int) nShadow, off);
        tc.nodes += (uint32_t) __shfl_down((int) tc.nodes, off); tc.tris += (uint32_t) __shfl_down((int) tc.tris, off);
    }
    /* one atomic per workgroup and counter (as wf_extend) */
    __shared__ uint32_t s_sum[4];
    __syncthreads();
    if (threadIdx.x < 4u) s_sum[threadIdx.x] = 0u;
    __syncthreads();
    if (lane_id() == 0) {
        if (nClosest) atomicAdd(&s_sum[0], nClosest);
        if (nShadow) atomicAdd(&s_sum[1], nShadow);
        if (count) { atomicAdd(&s_sum[2], tc.nodes); atomicAdd(&s_sum[3], tc.tris); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_sum[0]) atomicAdd(&b.stats[S_CLOSEST], (unsigned long long) s_sum[0]);
        if (s_sum[1]) atomicAdd(&b.stats[S_SHADOW], (unsigned long long) s_sum[1]);
        if (count) { atomicAdd(&b.stats[S_NODES], (unsigned long long) s_sum[2]); atomicAdd(&b.stats[S_TRIS], (unsigned long long) s_sum[3]); }
    }
}

/* Tail overlap (wavefront_render): the records of a batch's last live paths move out of the state pool -- which the next batch
   is about to overwrite -- into the engine's side copy, where wf_finish walks them on its own CUs. */
__global__ __launch_bounds__(kB) void wf_tail_copy(WfBuf b, int cur, WfState D, uint32_t *d_ctr) {
    const WfState S = b.st[cur];
    const uint32_t n = b.ctr[C_N + cur];
    for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB) {
        D.o[i] = S.o[i]; D.dA[i] = S.dA[i]; D.dB[i] = S.dB[i]; D.T_eta[i] = S.T_eta[i]; D.L_pdf[i] = S.L_pdf[i]; D.Ld[i] = S.Ld[i];
        D.sidx[i] = S.sidx[i]; D.rng[i] = S.rng[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { d_ctr[C_N] = n; d_ctr[C_TAIL_HEAD] = 0u; }
}

/* ----------------------------------------------------------- host driver */
constexpr int kShadeGridMax = 4096;
/* CUs of the shading side when the device is split (wavefront_render), and the job size from which it is */
#ifndef NORI_SPLIT_CUS_DEFAULT
#define NORI_SPLIT_CUS_DEFAULT 0
#endif
constexpr int kSplitCusDefault = NORI_SPLIT_CUS_DEFAULT;
constexpr size_t kSplitMinSamples = (size_t) 1 << 24;

int shade_grid(size_t paths) { return (int) std::min<size_t>(kShadeGridMax, std::max<size_t>(64, (paths + kShadeChunk - 1) / kShadeChunk)); }

/* records per state copy for a batch of `paths`: the survivors plus the unused tail of every
   wf_shade workgroup's last chunk (a round that does not fit is split across two chunks, so
   nothing else is ever left empty) */
size_t state_capacity(size_t paths) {
    return paths + (size_t) kShadeChunk * (size_t) shade_grid(paths) * (size_t) (kB / kSB) + 1024;
}

struct Pool {
    std::vector<void *> allocs;
    size_t capacity = 0;       /* records per copy, all pipes */
    size_t bytes = 0;
    WfBuf buf;
    uint32_t *h_ctr = nullptr; /* pinned */
    int *spill = nullptr;      /* traversal-stack overflow of wf_extend, one column per lane and pipe */
    size_t spill_ints = 0;
    void release() {
        for (void *p : allocs) (void) hipFree(p);
        allocs.clear(); capacity = 0; bytes = 0;
        if (spill) { (void) hipFree(spill); spill = nullptr; spill_ints = 0; }
        if (h_ctr) { (void) hipHostFree(h_ctr); h_ctr = nullptr; }
    }
};

#define WF_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); } while (0)

template <class T> std::string pool_alloc(Pool &pool, T **out, size_t count) {
    void *p = nullptr;
    WF_TRY(hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)));
    pool.allocs.push_back(p); pool.bytes += count * sizeof(T);
    *out = reinterpret_cast<T *>(p);
    return std::string();
}

/* bytes of path state per record: two copies of the SoA state + the hit record */
constexpr size_t kStateBytesPerRecord = 2 * (4 * sizeof(f4) + 2 * sizeof(P3) + sizeof(uint32_t) + sizeof(unsigned long long)) + sizeof(f4);

std::string ensure_pool(Pool &pool, size_t records) {
    if (pool.capacity >= records) return std::string();
    pool.release();
    WfBuf &b = pool.buf;
    std::string e;
#define A(field, count) if (!(e = pool_alloc(pool, &b.field, (count))).empty()) { pool.release(); return e; }
    for (int k = 0; k < 2; ++k) {
        A(st[k].o, records); A(st[k].dA, records); A(st[k].dB, records);
        A(st[k].T_eta, records); A(st[k].L_pdf, records); A(st[k].Ld, records);
        A(st[k].sidx, records); A(st[k].rng, records);
    }
    A(hit, records);
    A(ctr, (size_t) 2 * C_COUNT); A(stats, (size_t) 2 * S_COUNT);
#undef A
    WF_TRY(hipHostMalloc((void **) &pool.h_ctr, 2 * C_COUNT * sizeof(uint32_t)));
    pool.capacity = records;
    return std::string();
}

/* threads per wf_extend workgroup and workgroups per CU the LDS image is sized for, per node layout (see wf_extend) */
constexpr int kExtendBlockBvh2 = 1024, kExtendBlockWide = kB;
constexpr int kExtendWgsBvh2 = 2048 / kExtendBlockBvh2, kExtendWgsWide = 7, kExtendWgsWideFirst = 5;
constexpr size_t kLdsPerCu = 160 * 1024;
template <int STACK, bool SPILL, bool COUNT, int MODE>
void launch_extend(const DevScene &sc, const WfBuf &b, int cur, int refill, int grid, const WfBatch &bt, hipStream_t s) {
    const size_t image = (size_t) std::max(1u, sc.top_image_quads) * sizeof(f4);
    if (sc.wide) {
        /* wide trees push up to three children per step: always the spilling stack */
        const size_t lds = (size_t) LdsStackW<STACK, SPILL, kExtendBlockWide>::kLdsEntries * kExtendBlockWide * sizeof(int) + image;
        if (SPILL) hipLaunchKernelGGL((wf_extend<STACK, true, COUNT, MODE, true, false, kExtendBlockWide>), dim3(grid), dim3(kExtendBlockWide), lds, s, sc, b, cur, refill, bt);
    } else {
        /* the hand-written node loop (bvh2q_node_loop_asm) walks the tree's 32-B records (rt_nodeq.h: trees without unbounded boxes),
           addresses them by 32-bit byte offsets (trees below 2^25 nodes -- a BVH2 tree has fewer nodes than triangles) and keeps its
           stack in LDS; everything else takes the compiler's loop over the 64-B nodes */
        constexpr bool kAsm = !COUNT;
        const bool use_asm = kAsm && sc.nodes_q != nullptr && sc.n_triangles < (1u << 25) && (bt.flags & kBatchNoAsmLoop) == 0u;
        const bool image_q = use_asm || (COUNT && (bt.flags & kBatchCountQ) != 0u);
        const size_t lds = (use_asm ? (size_t) ExtendStack<STACK, SPILL, kAsm, kExtendBlockBvh2>::type::kLdsEntries : (size_t) LdsStackW<STACK, SPILL, kExtendBlockBvh2>::kLdsEntries) *
                               kExtendBlockBvh2 * sizeof(int) + (size_t) std::max(1u, image_q ? sc.top_image_q_quads : sc.top_image_quads) * sizeof(f4);
        if (use_asm)
            hipLaunchKernelGGL((wf_extend<STACK, SPILL, COUNT, MODE, false, kAsm, kExtendBlockBvh2>), dim3(grid), dim3(kExtendBlockBvh2), lds, s, sc, b, cur, refill, bt);
        else
            hipLaunchKernelGGL((wf_extend<STACK, SPILL, COUNT, MODE, false, false, kExtendBlockBvh2>), dim3(grid), dim3(kExtendBlockBvh2), lds, s, sc, b, cur, refill, bt);
    }
}

/* lds_stack: entries kept in LDS (16 / 24 / 32); spill: the tree is deeper than that;
   mode: kStoredOnly (the pass's stored paths) / kFreshOnly (its new ones) */
void launch_extend_dyn(const DevScene &sc, const WfBuf &b, int cur, int refill, int lds_stack, bool spill, bool count, int mode,
                       int grid, const WfBatch &bt, hipStream_t s) {
#define G(S, P, C) if (mode == kFreshOnly) launch_extend<S, P, C, kFreshOnly>(sc, b, cur, refill, grid, bt, s); \
                   else launch_extend<S, P, C, kStoredOnly>(sc, b, cur, refill, grid, bt, s)
#define E(S, P) if (count) { G(S, P, true); } else { G(S, P, false); }
#define F(S) if (spill) { E(S, true); } else { E(S, false); }
    if (lds_stack <= 16) { F(16); } else if (lds_stack <= 24) { F(24); } else { F(32); }
#undef F
#undef E
#undef G
}

void launch_shade(const DevScene &sc, const WfBuf &b, int cur, const WfBatch &bt, int mode, int grid_, hipStream_t s) {
    const dim3 grid(grid_ * (kB / kSB)), block(kSB);
    const bool lds_tables = shade_tables_fit(sc);
    /* the material set the kernel is compiled for: all-diffuse scenes (the Cornell box of the headline), scenes without a
       microfacet BSDF, any scene; integrators that never ask a BSDF (normals, ao, simple) have one kernel */
    int matset = sc.integrator.type < INT_WHITTED ? kAnyBsdf : sc.bsdf_mask == 1u ? 1 : (sc.bsdf_mask & 8u) == 0u ? 7 : kAnyBsdf;
    if (getenv("NORI_HIP_SHADE_ANY_BSDF")) matset = kAnyBsdf;      /* A/B: the general kernel whatever the scene holds */
    switch (sc.integrator.type) {
#define SH3(I, F, M) if (lds_tables) hipLaunchKernelGGL((wf_shade<I, F, true, M>), grid, block, 0, s, sc, b, cur, bt); \
                     else hipLaunchKernelGGL((wf_shade<I, F, false, M>), grid, block, 0, s, sc, b, cur, bt)
#define SH2(I, F) if (matset == 1) { SH3(I, F, 1); } else if (matset == 7) { SH3(I, F, 7); } else { SH3(I, F, kAnyBsdf); }
#define SH1(I, F) SH3(I, F, kAnyBsdf)
#define SH(I, W) case I: if (mode == kFreshOnly) { W(I, kFreshOnly); } else if (mode == kMixed) { W(I, kMixed); } else { W(I, kStoredOnly); } break;
        SH(0, SH1) SH(1, SH1) SH(2, SH1) SH(3, SH2) SH(4, SH2) SH(5, SH2) SH(6, SH2)
#undef SH
#undef SH1
#undef SH2
#undef SH3
    }
}

void launch_finish(const DevScene &sc, const WfBuf &b, int cur, const WfBatch &bt, bool count, int grid_, hipStream_t s) {
    const dim3 grid(grid_), block(kB);
    const size_t lds = (size_t) LdsStackW<16, true>::kLdsEntries * kB * sizeof(int);
    switch (sc.integrator.type) {
#define FN(I) case I: hipLaunchKernelGGL((wf_finish<I>), grid, block, lds, s, sc, b, cur, bt, (int) count); break;
        FN(0) FN(1) FN(2) FN(3) FN(4) FN(5) FN(6)
#undef FN
    }
}

} // namespace

namespace nrt {

int wf_top_capacity(bool wide_nodes, bool records_32b, bool deeper_than_lds_stack) {
    /* 16 stack entries per lane in LDS, + 1: the "done" marker of the stacks that do not spill, + 1: the counter of the stack
       that spills behind the hand-written loop (LdsStackHybrid); the image in what is left */
    const size_t per_wg = kLdsPerCu / (wide_nodes ? kExtendWgsWide : kExtendWgsBvh2);
    const size_t stacks = (size_t) (deeper_than_lds_stack && records_32b ? 18 : 17) * (wide_nodes ? kExtendBlockWide : kExtendBlockBvh2) * sizeof(int);
    const int stride = records_32b ? kTopStrideQuadsQ : kTopStrideQuads;
    int n = 0;
    while (n < kTopMaxNodes && stacks + (size_t) top_image_quads(n + 1, stride) * sizeof(f4) <= per_wg) ++n;
    return n;
}


/* Everything a context keeps between render calls: the path-state pool, the pipes' streams and
   events, what the device offers.  One per nori_hip_ctx (= per GPU); nothing here is process-global,
   so contexts on different devices, or two contexts on one device, never share or free each
   other's buffers. */
struct WfEngine {
    Pool pool;
    hipStream_t streams[2] = {nullptr, nullptr};
    hipEvent_t events[3] = {nullptr, nullptr, nullptr};
    int n_cus = 0;                  /* hipDeviceProp_t::multiProcessorCount of the context's device */
    /* the CUs of the device shared out between the two kinds of work (wavefront_render, "split"): [0] the stream of the
       traversal kernels, on all CUs but split_cus, [1] the stream of the shading and film kernels, on split_cus CUs */
    hipStream_t split_streams[2] = {nullptr, nullptr};
    int split_cus = 0;
    hipEvent_t pipe_events[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      /* per pipe: its traversal / its shading is done */
    /* tail overlap: the side copy of a batch's last live paths (wf_tail_copy), walked by wf_finish on split_streams[1] */
    struct TailSide {
        WfState st; uint32_t *ctr = nullptr; size_t capacity = 0; std::vector<void *> allocs;
        int *spill = nullptr; size_t spill_ints = 0;
        void release() { for (void *q : allocs) (void) hipFree(q); allocs.clear(); capacity = 0; ctr = nullptr; if (spill) { (void) hipFree(spill); spill = nullptr; spill_ints = 0; } }
    } tail;
    hipEvent_t tail_events[3] = {nullptr, nullptr, nullptr};      /* the side copy is filled / the tail is done / the tail is about to be dispatched */
    hipStream_t tail_stream = nullptr;
    int tail_stream_cus = 0;
};

/* The stream of the overlapped tails: it owns the CUs of the first `cus` mask bits.  The driver hands bit i of a queue's mask to
   XCD i mod 8 and walks an XCD's shader engines with the bits it gets, so 32 bits are one CU of every shader engine of every XCD:
   what is left for the other streams' kernels is even over XCDs (a kernel's workgroups go round the XCDs, b mod 8, whatever their
   CUs hold) and over the shader engines of an XCD (taking CUs from one shader engine only costs the bulk kernels twice the CUs'
   share -- pa5 table scene, bulk on the complement of 8 / 16 / 32 CUs of the FIRST shader engines: wf_extend +6 / +14 / +37 %,
   profiles/r5_07_tail_overlap_masks.txt). */
bool ensure_tail_stream(WfEngine &e, int cus) {
    if (e.tail_stream && e.tail_stream_cus == cus) return true;
    if (e.tail_stream) { (void) hipStreamDestroy(e.tail_stream); e.tail_stream = nullptr; e.tail_stream_cus = 0; }
    if (cus < 8 || cus >= e.n_cus) return false;
    const uint32_t words = (uint32_t) ((e.n_cus + 31) / 32);
    std::vector<uint32_t> mask(words, 0u);
    for (int i = 0; i < cus; ++i) mask[(size_t) i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(&e.tail_stream, words, mask.data()) != hipSuccess) { (void) hipGetLastError(); e.tail_stream = nullptr; return false; }
    e.tail_stream_cus = cus;
    return true;
}

std::string ensure_tail_side(WfEngine &e, size_t records, size_t spill_ints) {
    WfEngine::TailSide &t = e.tail;
    if (t.capacity < records) {
        t.release();
#define A(field, T) { void *q = nullptr; WF_TRY(hipMalloc(&q, records * sizeof(T))); t.allocs.push_back(q); t.st.field = reinterpret_cast<T *>(q); }
        A(o, P3) A(dA, f4) A(dB, f4) A(T_eta, f4) A(L_pdf, f4) A(Ld, P3) A(sidx, uint32_t) A(rng, unsigned long long)
#undef A
        { void *q = nullptr; WF_TRY(hipMalloc(&q, C_COUNT * sizeof(uint32_t))); t.allocs.push_back(q); t.ctr = reinterpret_cast<uint32_t *>(q); }
        t.capacity = records;
    }
    if (t.spill_ints < spill_ints) {
        if (t.spill) (void) hipFree(t.spill);
        t.spill = nullptr; t.spill_ints = 0;
        WF_TRY(hipMalloc((void **) &t.spill, spill_ints * sizeof(int)));
        t.spill_ints = spill_ints;
    }
    for (hipEvent_t &ev : e.tail_events) if (!ev) WF_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return std::string();
}

/* Two streams whose kernels run on disjoint sets of CUs (hipExtStreamCreateWithCUMask).  The shading side gets `cus` CUs (a
   multiple of 8) as whole ROWS of eight consecutive mask bits, rows spread evenly over the mask: the driver hands bit i of a
   queue's mask to XCD i mod 8 and walks an XCD's shader engines with the bits it gets, so a row is one CU of every XCD and the
   two sets are even over XCDs and shader engines.  Returns false (and leaves the engine unsplit) if the runtime refuses. */
bool ensure_split_streams(WfEngine &e, int cus) {
    if (e.split_cus == cus && e.split_streams[0] && e.split_streams[1]) return true;
    for (hipStream_t &st : e.split_streams) if (st) { (void) hipStreamDestroy(st); st = nullptr; }
    e.split_cus = 0;
    const int rows_total = e.n_cus / 8, rows = cus / 8;
    if (rows < 1 || rows >= rows_total) return false;
    const uint32_t words = (uint32_t) ((e.n_cus + 31) / 32);
    std::vector<uint32_t> shade(words, 0u), extend(words, 0u);
    std::vector<char> taken((size_t) rows_total, 0);
    for (int j = 0; j < rows; ++j) taken[(size_t) ((long long) j * rows_total / rows)] = 1;
    for (int i = 0; i < e.n_cus; ++i) {
        const bool in_row = i / 8 < rows_total && taken[(size_t) (i / 8)];
        (in_row ? shade : extend)[(size_t) i / 32] |= 1u << (i % 32);
    }
    if (hipExtStreamCreateWithCUMask(&e.split_streams[0], words, extend.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&e.split_streams[1], words, shade.data()) != hipSuccess) {
        (void) hipGetLastError();
        for (hipStream_t &st : e.split_streams) if (st) { (void) hipStreamDestroy(st); st = nullptr; }
        return false;
    }
    e.split_cus = cus;
    return true;
}

WfEngine *wavefront_create() {
    WfEngine *e = new WfEngine();
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) e->n_cus = prop.multiProcessorCount;
    if (e->n_cus <= 0) e->n_cus = 256;
    return e;
}

void wavefront_destroy(WfEngine *e) {
    if (!e) return;
    e->pool.release();
    for (hipStream_t &st : e->streams) if (st) { (void) hipStreamDestroy(st); st = nullptr; }
    for (hipStream_t &st : e->split_streams) if (st) { (void) hipStreamDestroy(st); st = nullptr; }
    for (hipEvent_t &ev : e->events) if (ev) { (void) hipEventDestroy(ev); ev = nullptr; }
    for (auto &pe : e->pipe_events) for (hipEvent_t &ev : pe) if (ev) { (void) hipEventDestroy(ev); ev = nullptr; }
    for (hipEvent_t &ev : e->tail_events) if (ev) { (void) hipEventDestroy(ev); ev = nullptr; }
    if (e->tail_stream) { (void) hipStreamDestroy(e->tail_stream); e->tail_stream = nullptr; }
    e->tail.release();
    delete e;
}

bool wavefront_excursions(unsigned long long out[4], bool reset) {
#if defined(NORI_COUNT_EXCURSIONS)
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_nori_excursions), sizeof(h)) != hipSuccess) return false;
    for (int k = 0; k < 4; ++k) out[k] += h[k];
    if (reset) { const unsigned long long z[4] = {0, 0, 0, 0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_nori_excursions), z, sizeof(z)); }
    return true;
#else
    (void) out; (void) reset;
    return false;
#endif
}

size_t wavefront_bytes_per_path() { return kStateBytesPerRecord; }
size_t wavefront_bytes_per_sample() { return 2 * (sizeof(f2) + sizeof(P3)); }      /* (two halves of the sample store: tail overlap) */
void wavefront_release_pool(WfEngine *e) { if (e) { e->pool.release(); e->tail.release(); } }

size_t wavefront_held_bytes(const WfEngine *e, const FilmStore &film) {
    return (e ? e->pool.bytes : 0) + film.capacity * (sizeof(f2) + sizeof(P3));
}

/* A pipe works through its own list of batches with its own slice of the state pool, on its own
   HIP stream.  Two pipes interleave so that one pipe's wf_shade (HBM-bound) overlaps the other
   pipe's wf_extend (VALU-bound); the persistent extend grid is sized to leave room for it. */
struct Pipe {
    hipStream_t stream = nullptr;            /* shading, film, counters: everything but ... */
    hipStream_t extend_stream = nullptr;     /* ... the traversal kernels (the same stream unless the CUs are split) */
    hipEvent_t ev_extend = nullptr, ev_shade = nullptr;      /* split: the hand-over between the two streams */
    WfBuf b;
    FilmStore film;
    uint32_t *h_ctr = nullptr;
    uint32_t tile_lo = 0, tile_hi = 0;      /* selected-tile ordinals owned by this pipe */
    uint32_t tiles_b = 0, spp_b = 0;        /* batch geometry */
    uint32_t t0 = 0, s0 = 0;                /* next batch */
    bool active = false, finished = false;
    int mode = kFreshOnly;                   /* of the next pass's kernels */
    uint32_t batch_samples = 0;              /* camera samples of the current batch */
    uint32_t batch_rounds = 0;               /* host-loop rounds spent on the current batch */
    WfBatch bt;
    int cur = 0;
};

static WfBuf slice(const WfBuf &b, size_t off, size_t records, int k) {
    WfBuf v = b;
    for (int c = 0; c < 2; ++c) {
        WfState &t = v.st[c];
        t.o += off; t.dA += off; t.dB += off; t.T_eta += off; t.L_pdf += off; t.Ld += off;
        t.sidx += off; t.rng += off;
    }
    v.hit += off;
    v.ctr += (size_t) k * C_COUNT; v.stats += (size_t) k * S_COUNT;
    v.capacity = (uint32_t) records;
    return v;
}

std::string wavefront_render(WfEngine &eng, FilmStore &film_store, const DevScene &sc, const float *d_filter_table, const WfLaunch &L,
                             float *d_rgbw, void *stream_, WfStats &stats) {
    hipStream_t s = (hipStream_t) stream_;
    Pool &g_pool = eng.pool;
    hipStream_t (&g_streams)[2] = eng.streams;
    hipEvent_t (&g_events)[3] = eng.events;
    stats = WfStats();
    if (L.n_sel_tiles == 0 || L.spp_count == 0) return std::string();

    int n_pipes = 1;        /* measured: 2 pipes are 5 % slower (wf_extend wants all 8 waves/SIMD); NORI_HIP_WF_PIPES=2 to try */
    if (const char *e = getenv("NORI_HIP_WF_PIPES")) n_pipes = std::min(2, std::max(1, atoi(e)));
    /* Split: the device's CUs shared out between the two kinds of work.  wf_extend is bound by the vector ALU (its time follows
       the number of CUs it runs on), wf_shade and the film by the memory system (theirs does not, down to a few dozen CUs): run
       one after the other, each leaves idle what the other needs.  Two pipes -- each half of the tiles -- with the traversal
       kernels of both on a stream that owns all CUs but `split_cus`, shading and film on a stream that owns those: while one
       pipe's paths are traversed the other pipe's are shaded, both kernels at their full occupancy on their own CUs (two pipes
       that SHARE every CU halve each kernel's waves per SIMD and gain nothing: profiles/r4_05).  NORI_HIP_WF_SPLIT_CUS: CUs of
       the shading side (a multiple of 8; 0: no split). */
    int split_cus = kSplitCusDefault;
    if (const char *e = getenv("NORI_HIP_WF_SPLIT_CUS")) split_cus = std::max(0, atoi(e)) & ~7;
    if (getenv("NORI_HIP_WF_PIPES")) split_cus = 0;
    size_t split_min = kSplitMinSamples;
    if (const char *e = getenv("NORI_HIP_WF_SPLIT_MIN")) split_min = (size_t) std::max(0ll, atoll(e));
    if (L.n_sel_tiles < 2 || (size_t) L.n_sel_tiles * 256 * L.spp_count < split_min || L.film_reference || L.count_traversal) split_cus = 0;
    if (L.n_sel_tiles < 2 || (size_t) L.n_sel_tiles * 256 * L.spp_count < ((size_t) 1 << 22)) n_pipes = 1;
    if (split_cus >= eng.n_cus) split_cus = 0;
    if (split_cus > 0 && ensure_split_streams(eng, split_cus)) n_pipes = 2; else split_cus = 0;
    const bool split = split_cus > 0;

    /* per pipe: a batch = a range of the pipe's tiles x as many samples per pixel as the film's sample store may hold
       (max_samples); the paths in flight -- the state pool -- are bounded separately (max_paths): a batch bigger than the pool
       starts its samples pass by pass, in the slots its finished paths leave (regeneration, WfBatch::pool) */
    const size_t budget = std::min<size_t>(std::max<size_t>(std::max(L.max_samples, L.max_paths) / n_pipes, 256), (size_t) 1 << 31);
    Pipe pipes[2];
    size_t need[2] = {0, 0};
    for (int k = 0; k < n_pipes; ++k) {
        Pipe &P = pipes[k];
        P.tile_lo = (uint32_t) ((uint64_t) L.n_sel_tiles * k / n_pipes);
        P.tile_hi = (uint32_t) ((uint64_t) L.n_sel_tiles * (k + 1) / n_pipes);
        const uint32_t nt = P.tile_hi - P.tile_lo;
        if ((size_t) nt * 256 > budget) { P.tiles_b = (uint32_t) (budget / 256); P.spp_b = 1; }
        else { P.tiles_b = nt; P.spp_b = (uint32_t) std::min<size_t>(L.spp_count, std::max<size_t>(1, budget / ((size_t) nt * 256))); }
        need[k] = (size_t) P.tiles_b * 256 * P.spp_b;
        P.t0 = P.tile_lo; P.s0 = 0;
    }
    /* Tail overlap.  A batch ends with wf_finish walking its last <= finish_paths live paths to their ends: a launch as long as the
       batch's longest path (inside glass ~1300 vertices, one after the other) during which the device is next to idle -- 12 % of
       the pa5 table scene at 2048^2 x 1024 spp.  In a call of two or more batches the tail of batch k runs BESIDE the kernels of
       batch k + 1 instead: its records are copied out of the state pool (wf_tail_copy) and the persistent form of wf_finish walks
       them on a stream that owns `tail_cus` CUs (ensure_tail_stream) while the bulk of the next batch goes on on the caller's
       stream -- on the CUs the tail leaves it while it lasts (46 ms on 32 CUs), on all of them afterwards; the film gather of
       batch k -- which needs the tail's radiance -- is queued behind the bulk of batch k + 1.  The sample store has two halves,
       one per batch in flight.  Gathers stay in batch order, so the frame keeps its bits.  NORI_HIP_WF_TAIL_CUS (a multiple of 8;
       0: tails on the bulk's stream, as a call of one batch runs them). */
    int tail_cus = 32;
    if (const char *e = getenv("NORI_HIP_WF_TAIL_CUS")) tail_cus = std::max(0, atoi(e)) & ~7;
    {
        const Pipe &P0 = pipes[0];
        const uint64_t nt0 = P0.tile_hi - P0.tile_lo;
        const uint64_t batches = ((nt0 + P0.tiles_b - 1) / std::max(1u, P0.tiles_b)) * ((L.spp_count + P0.spp_b - 1) / std::max(1u, P0.spp_b));
        if (n_pipes != 1 || split || batches < 2 || L.film_reference || L.count_traversal || tail_cus >= eng.n_cus) tail_cus = 0;
        if (getenv("NORI_HIP_WF_FINISH") && atoi(getenv("NORI_HIP_WF_FINISH")) == 0) tail_cus = 0;
        if (tail_cus > 0 && !ensure_tail_stream(eng, tail_cus)) tail_cus = 0;
    }
    const bool overlap = tail_cus > 0;
    if (L.film_reference && (n_pipes != 1 || pipes[0].tiles_b != L.n_sel_tiles || pipes[0].spp_b != L.spp_count || L.tile_mod != 1))
        return "wavefront: film_order = reference needs the whole frame in one batch (tile_mod 1, wavefront_samples >= pixels x samples)";
    const size_t per_pipe = std::max(need[0], need[1]);         /* camera samples of a batch */
    size_t pool_paths = std::min(per_pipe, std::max<size_t>(L.max_paths / n_pipes, 256));      /* paths in flight */
    if (const char *e = getenv("NORI_HIP_WF_POOL")) pool_paths = std::min(per_pipe, (size_t) std::max(256ll, atoll(e)));      /* experiments */
    const size_t records = state_capacity(pool_paths);          /* per pipe, per state copy */
    const int sh_grid = shade_grid(pool_paths);
    if (records >= ((size_t) 1 << 30)) return "wavefront: wavefront_paths too large (state index is 30 bits)";
    std::string err = ensure_pool(g_pool, records * n_pipes);
    if (!err.empty()) return err;
    FilmStore film;
    err = film_prepare(film_store, per_pipe * (overlap ? 2 : n_pipes), L.n_sel_tiles, L.tile_w, s, film);
    if (!err.empty()) return err;
    stats.state_bytes = g_pool.bytes + per_pipe * (overlap ? 2 : n_pipes) * (sizeof(f2) + sizeof(P3));
    WF_TRY(hipMemsetAsync(g_pool.buf.stats, 0, 2 * S_COUNT * sizeof(unsigned long long), s));

    for (int k = 0; k < 3; ++k) if (!g_events[k]) WF_TRY(hipEventCreateWithFlags(&g_events[k], hipEventDisableTiming));
    for (int k = 0; k < n_pipes; ++k) {
        Pipe &P = pipes[k];
        if (split) {
            P.extend_stream = eng.split_streams[0]; P.stream = eng.split_streams[1];
            for (hipEvent_t &ev : eng.pipe_events[k]) if (!ev) WF_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            P.ev_extend = eng.pipe_events[k][0]; P.ev_shade = eng.pipe_events[k][1];
        } else if (n_pipes == 1 && !overlap) P.stream = P.extend_stream = s;
        else {      /* (overlap: the bulk on a stream of the engine's own, too -- the caller's may be the legacy default stream, which
                       runs nothing beside the work of another stream) */
            if (!g_streams[k]) WF_TRY(hipStreamCreateWithFlags(&g_streams[k], hipStreamNonBlocking));
            P.stream = P.extend_stream = g_streams[k];
        }
        P.b = slice(g_pool.buf, records * k, records, k);
        P.film = film; P.film.pos += per_pipe * k; P.film.L += per_pipe * k;
        P.b.samp_pos = P.film.pos; P.b.samp_L = P.film.L;
        P.h_ctr = g_pool.h_ctr + (size_t) k * C_COUNT;
    }
    const bool own_streams = n_pipes > 1 || overlap;
    hipStream_t tail_stream = overlap ? eng.tail_stream : nullptr;
    if (own_streams) {      /* the pipes start after whatever the caller queued on its stream */
        WF_TRY(hipEventRecord(g_events[2], s));
        for (int k = 0; k < n_pipes; ++k) { WF_TRY(hipStreamWaitEvent(pipes[k].stream, g_events[2], 0)); if (split) WF_TRY(hipStreamWaitEvent(pipes[k].extend_stream, g_events[2], 0)); }
    }

    int refill = 32;
    if (const char *e = getenv("NORI_HIP_WF_REFILL")) refill = std::min(64, std::max(1, atoi(e)));
    /* lanes at a leaf before a triangle step runs: 16 on BVH2 trees; 8 on wide trees, whose node steps are twice as long and whose
       waves hold fewer lanes per step (terrain, trace ms at 16 / 8 / 4: 54.2 / 53.3 / 53.2; the AO scene's BVH2 walk: 62.1 / 62.6 at 8;
       profiles/r4_05_sweeps.txt) */
    int leaf_th = sc.wide ? 8 : 16;
    if (const char *e = getenv("NORI_HIP_WF_LEAF")) leaf_th = std::min(64, std::max(1, atoi(e)));
    int static_per_wave = 1024, dyn_div = 8;      /* chunking of wf_extend: all-static below static_per_wave paths per wave, else n / (waves * dyn_div) per claim */
    if (const char *e = getenv("NORI_HIP_WF_STATIC")) static_per_wave = std::min(4095, std::max(0, atoi(e)));
    if (const char *e = getenv("NORI_HIP_WF_DYNDIV")) dyn_div = std::min(15, std::max(1, atoi(e)));
    const int thresholds = refill | (leaf_th << 8) | (static_per_wave << 16) | (dyn_div << 28);
    /* persistent extend grid: fill the CUs, but with two pipes leave half of the wave slots to
       the other pipe's kernels */
    /* traversal stack: what the tree needs, at most `lds_stack` entries of it in LDS */
    /* 16 entries in LDS, deeper walks spill to HBM: a stack of 24 entries costs three of the eight workgroups per CU, and walks
       rarely hold more than 16 deferred subtrees however deep the tree (measured, trace ms at 16 / 24 entries: table scene,
       depth 19: 77.5 / 83.0; Cornell box with the device builders' deeper trees: 61.6 / 74.8 and 73.3 / 82.1) */
    const bool no_asm_loop = getenv("NORI_HIP_WF_NO_ASM_LOOP") != nullptr && atoi(getenv("NORI_HIP_WF_NO_ASM_LOOP")) != 0;
    int lds_stack = 16;
    if (const char *e = getenv("NORI_HIP_WF_STACK")) lds_stack = atoi(e) <= 16 ? 16 : atoi(e) <= 24 ? 24 : 32;
    int finish_paths = 524288;           /* fewer live paths than this: wf_finish ends the batch */
    if (const char *e = getenv("NORI_HIP_WF_FINISH_PATHS")) finish_paths = std::max(256, atoi(e)) & ~255;
    const int finish_grid = finish_paths / kB;
    /* LDS per workgroup: stack entries (+1: the "done" marker of the non-spilling stack) + the top-node cache */
    const int extend_block = sc.wide ? kExtendBlockWide : kExtendBlockBvh2;
    /* which node loop wf_extend will run (launch_extend): the hand-written one on 32-B records, or the compiler's on 64-B nodes */
    const bool walk_q = !sc.wide && sc.nodes_q != nullptr && sc.n_triangles < (1u << 25) && !no_asm_loop;      /* 32-B node records */
    const bool node_loop_q = walk_q && !L.count_traversal;         /* ... by the hand-written loop */
    const bool count_q = walk_q && L.count_traversal;              /* ... by its C++ statement, counting (the same tree form as the timed kernel's) */
    bool spill = false;
    size_t lds_per_wg = 0;
    while (true) {
        spill = L.stack_depth > lds_stack || sc.wide != 0;
        const int stack_entries = node_loop_q ? lds_stack + (spill ? 2 : 1) : lds_stack + (spill ? 0 : 1);      /* kLdsEntries of the kernel's stack class */
        lds_per_wg = (size_t) stack_entries * extend_block * sizeof(int) + (size_t) std::max(1u, (node_loop_q || count_q) ? sc.top_image_q_quads : sc.top_image_quads) * sizeof(f4);
        /* the LDS image was sized at build_accel for 16-entry stacks (wf_top_capacity): a bigger stack asked for at render time
           (NORI_HIP_WF_STACK, an experiment knob) must not push a workgroup beyond its share of the CU's LDS -- fewer workgroups
           per CU than the kernel is built for, or a launch that fails: back to the stack the image was sized for */
        if (lds_stack > 16 && lds_per_wg > kLdsPerCu / (size_t) (sc.wide ? kExtendWgsWide : kExtendWgsBvh2)) { lds_stack = 16; continue; }
        break;
    }
    int per_cu = std::max(1, std::min(2048 / extend_block, (int) (kLdsPerCu / lds_per_wg)));      /* workgroups per CU */
    /* every workgroup of the persistent grid must be resident from the start (a workgroup that starts late owns a
       static share of the paths and works it off alone): the wide-node kernels are built for 7 waves per SIMD, their
       first-pass variant (camera rays computed in the kernel: 77 - 87 registers) for 5.  (Terrain, 10 M triangles, wf_extend ms per
       128 spp at 4 / 5 / 6 / 7 / 8 workgroups per CU: 65.7 / 57.8 / 53.9 / 53.6 / 54.2 -- the last two with the first pass still on the
       same grid; how many records the LDS image holds does not matter there: 113, 40 or none, profiles/r4_07_c5_occupancy.txt.) */
    if (sc.wide) per_cu = std::min(per_cu, kExtendWgsWide);
    if (n_pipes > 1 && !split) per_cu = std::max(1, per_cu / 2);
    if (const char *e = getenv("NORI_HIP_WF_EXTEND_WGS_PER_CU")) per_cu = std::min(2048 / extend_block, std::max(1, atoi(e)));
    const int extend_cus = eng.n_cus - split_cus;      /* the persistent grid fills the CUs its stream owns (a tail beside it: some of its
                                                          workgroups start when the tail's have left -- handing them their first chunk in
                                                          the order they start instead of by position was measured and is no faster,
                                                          profiles/r5_11_tail_overlap_c4.txt) */
    const int extend_grid_first = extend_cus * (sc.wide ? std::min(per_cu, kExtendWgsWideFirst) : per_cu);
    const int extend_grid = extend_cus * per_cu;
    stats.trace_cus = (uint32_t) extend_cus; stats.tail_cus = (uint32_t) tail_cus;
    if (L.stack_depth > 16) {      /* wf_finish keeps 16 entries in LDS */
        const size_t per_pipe_ints = (size_t) (L.stack_depth - 16) * std::max(extend_grid * extend_block, finish_grid * kB), ints = per_pipe_ints * n_pipes;
        if (g_pool.spill_ints < ints) {
            if (g_pool.spill) (void) hipFree(g_pool.spill);
            g_pool.spill = nullptr; g_pool.spill_ints = 0;
            WF_TRY(hipMalloc((void **) &g_pool.spill, ints * sizeof(int)));
            g_pool.spill_ints = ints;
        }
        for (int k = 0; k < n_pipes; ++k) pipes[k].b.stack_spill = g_pool.spill + per_pipe_ints * k;
    }

    /* wf_finish: the persistent form (a lane pulls path after path) on four workgroups per CU of the stream it runs on -- the pa5
       table scene's tail 22 -> 19 ms on all CUs, 223 -> 80 ms on 16 (profiles/r5_06_tail_probe_c4.txt) */
    const int finish_grid_main = std::min(finish_grid, std::max(1, extend_cus * 4));
    const int finish_grid_side = std::min(finish_grid, std::max(1, tail_cus * 4));
    if (overlap) {
        err = ensure_tail_side(eng, (size_t) finish_paths, L.stack_depth > 16 ? (size_t) (L.stack_depth - 16) * finish_grid_side * kB : 0);
        if (!err.empty()) return err;
    }
    struct { bool active = false; FilmLaunch fl; FilmStore film; } pend;      /* overlap: the batch whose tail is in flight -- its film gather is still to come */

    /* node steps repeat in a tight loop while >= 24 lanes of the wave are at inner nodes -- without the refill test, the leaf
       vote and the rest of the trip's bookkeeping, which cost as much as the node step itself (measured, wf_extend: Cornell
       box 51.7 -> 49.7 ms, 328 k-triangle AO 14.2 -> 12.8, table 77.7 -> 73.5, terrain 37.6 -> 34.7; the same loop
       around the triangle step compiles with 56 B of scratch and runs 50 % slower) */
    int inner_repeat = 24;
    if (const char *e = getenv("NORI_HIP_WF_INNER_REPEAT")) inner_repeat = std::min(65, std::max(1, atoi(e)));
    int sync_every = 6;      /* path-loop iterations between two readbacks of the path count */
    if (const char *e = getenv("NORI_HIP_WF_SYNC_EVERY")) sync_every = std::min(64, std::max(1, atoi(e)));
    const bool census = getenv("NORI_HIP_CENSUS") != nullptr;
    bool use_finish = true;
    if (const char *e = getenv("NORI_HIP_WF_FINISH")) use_finish = atoi(e) != 0;
    const bool force_mixed = getenv("NORI_HIP_WF_FORCE_MIXED") != nullptr && atoi(getenv("NORI_HIP_WF_FORCE_MIXED")) != 0;      /* A/B, tests: every pass but the first as a pass that may hold new paths (both wf_extend launches, the kMixed wf_shade) */

    KernelTimer timer(L.time_kernels);
    unsigned long long *d_census = nullptr;
    if (census && L.count_traversal) {
        WF_TRY(hipMalloc((void **) &d_census, Z_COUNT * sizeof(unsigned long long)));
        WF_TRY(hipMemsetAsync(d_census, 0, Z_COUNT * sizeof(unsigned long long), s));
    }
    for (int k = 0; k < n_pipes; ++k) pipes[k].b.census = d_census;
    FilmLaunch fl;
    fl.tile_mod = L.tile_mod; fl.tile_rem = L.tile_rem; fl.tiles_x = L.tiles_x; fl.tiles_y = L.tiles_y; fl.tile_w = L.tile_w;

    while (true) {
        bool any = false;
        for (int k = 0; k < n_pipes; ++k) {         /* start the next batch of idle pipes */
            Pipe &P = pipes[k];
            if (P.active || P.finished) { any |= P.active; continue; }
            if (P.t0 >= P.tile_hi) { P.finished = true; continue; }
            const uint32_t nt = std::min(P.tiles_b, P.tile_hi - P.t0), ns = std::min(P.spp_b, L.spp_count - P.s0);
            P.bt.tile_first = P.t0; P.bt.n_tiles = nt; P.bt.s_first = L.spp_begin + P.s0; P.bt.n_spp = ns;
            P.bt.tile_mod = L.tile_mod; P.bt.tile_rem = L.tile_rem; P.bt.tiles_x = L.tiles_x; P.bt.tile_w = L.tile_w;
            P.bt.inner_repeat = inner_repeat;
            P.bt.flags = (no_asm_loop ? kBatchNoAsmLoop : 0u) | (count_q ? kBatchCountQ : 0u);
            P.bt.pool = (uint32_t) pool_paths;
            P.batch_samples = nt * 256u * ns;
            WF_TRY(hipMemsetAsync(P.b.ctr, 0, C_COUNT * sizeof(uint32_t), P.stream));
            if (split) WF_TRY(hipEventRecord(P.ev_shade, P.stream));      /* (and the film of the pipe's last batch, which read the sample store) */
            if (overlap) {      /* the sample store's halves alternate: the previous batch's tail and gather still use the other one */
                const size_t half = (size_t) (stats.n_batches & 1u) * per_pipe;
                P.film = film; P.film.pos += half; P.film.L += half;
                P.b.samp_pos = P.film.pos; P.b.samp_L = P.film.L;
            }
            stats.n_batches++;
            P.cur = 0; P.mode = kFreshOnly; P.active = true; any = true; P.batch_rounds = 0;
        }
        if (!any) break;
        /* the path count is read back every sync_every iterations, more often once it is small
           (the readback costs ~20 us, an iteration on a few paths ~100 us) */
        uint32_t n_max = 0;
        for (int k = 0; k < n_pipes; ++k) if (pipes[k].active) n_max = std::max(n_max, pipes[k].mode != kStoredOnly ? 0xffffffffu : pipes[k].h_ctr[C_N + pipes[k].cur]);
        const int iters = n_max > (16u << 20) ? sync_every : std::min(sync_every, 2);
        for (int it = 0; it < iters; ++it)
            for (int k = 0; k < n_pipes; ++k) {
                Pipe &P = pipes[k];
                if (!P.active) continue;
                if (split) WF_TRY(hipStreamWaitEvent(P.extend_stream, P.ev_shade, 0));
                /* the pass's stored paths, then its new ones (pass_shape): two launches of the two pure kernels */
                if (P.mode != kFreshOnly) {
                    timer.begin(KC_TRACE, P.extend_stream);
                    launch_extend_dyn(sc, P.b, P.cur, thresholds, lds_stack, spill, L.count_traversal, kStoredOnly, extend_grid, P.bt, P.extend_stream);
                    WF_TRY(hipGetLastError());      /* a launch that did not fit (LDS, registers) must not pass for an empty pass */
                    timer.end(P.extend_stream);
                    stats.n_launches++;
                }
                if (P.mode != kStoredOnly) {
                    timer.begin(KC_TRACE, P.extend_stream);
                    launch_extend_dyn(sc, P.b, P.cur, thresholds, lds_stack, spill, L.count_traversal, kFreshOnly, extend_grid_first, P.bt, P.extend_stream);
                    WF_TRY(hipGetLastError());
                    timer.end(P.extend_stream);
                    stats.n_launches++;
                }
                if (split) { WF_TRY(hipEventRecord(P.ev_extend, P.extend_stream)); WF_TRY(hipStreamWaitEvent(P.stream, P.ev_extend, 0)); }
                timer.begin(KC_SHADE, P.stream);
                launch_shade(sc, P.b, P.cur, P.bt, P.mode, sh_grid, P.stream);
                timer.end(P.stream);
                if (split) WF_TRY(hipEventRecord(P.ev_shade, P.stream));
                /* the first pass started min(pool, samples) camera samples: a batch that fits the pool has none left */
                if (P.mode == kFreshOnly) P.mode = P.batch_samples <= P.bt.pool && !force_mixed ? kStoredOnly : kMixed;
                P.cur ^= 1;
                stats.n_launches++;
                if (k == 0) stats.n_iterations++;
            }
        for (int k = 0; k < n_pipes; ++k)
            if (pipes[k].active) WF_TRY(hipMemcpyAsync(pipes[k].h_ctr, pipes[k].b.ctr, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, pipes[k].stream));
        for (int k = 0; k < n_pipes; ++k)
            if (pipes[k].active) WF_TRY(hipStreamSynchronize(pipes[k].stream));
        for (int k = 0; k < n_pipes; ++k) {
            Pipe &P = pipes[k];
            if (P.active && P.h_ctr[C_OVERFLOW] != 0) return "wavefront: path state pool overflow";
            /* camera samples of the batch no pass has started yet (the counter the next pass would read) */
            const uint32_t samples_left = P.active ? P.batch_samples - std::min(P.batch_samples, P.h_ctr[C_SAMPLE + P.cur]) : 0u;
            if (P.active && samples_left == 0u && P.mode == kMixed && !force_mixed) P.mode = kStoredOnly;
            if (census && P.active) fprintf(stderr, "[wavefront] pipe %d iteration %u: %u path slots, %u samples to start\n", k, stats.n_iterations, P.h_ctr[C_N + P.cur], samples_left);
            /* the film gather of the batch whose tail ran beside this one: behind the tail, in front of everything that follows on
               this stream (gathers stay in batch order) */
            auto flush_pending = [&]() -> std::string {
                if (!pend.active) return std::string();
                WF_TRY(hipStreamWaitEvent(P.stream, eng.tail_events[1], 0));
                timer.begin(KC_FILM, P.stream);
                film_gather(sc, d_filter_table, pend.film, pend.fl, P.stream);
                timer.end(P.stream);
                stats.n_launches++;
                pend.active = false;
                return std::string();
            };
            bool gather_deferred = false;
            if (P.active && samples_left == 0u && P.h_ctr[C_N + P.cur] != 0 && P.h_ctr[C_N + P.cur] <= (uint32_t) finish_paths && use_finish) {
                const bool last_batch = P.s0 + P.bt.n_spp >= L.spp_count && P.t0 + P.bt.n_tiles >= P.tile_hi;
                if (overlap && !last_batch) {
                    err = flush_pending();      /* (also: the previous tail is done with the side copy) */
                    if (!err.empty()) return err;
                    hipLaunchKernelGGL(wf_tail_copy, dim3(std::max(1u, (P.h_ctr[C_N + P.cur] + kB - 1) / kB)), dim3(kB), 0, P.stream, P.b, P.cur, eng.tail.st, eng.tail.ctr);
                    WF_TRY(hipEventRecord(eng.tail_events[0], P.stream));
                    WF_TRY(hipStreamWaitEvent(tail_stream, eng.tail_events[0], 0));
                    /* the bulk goes on once the tail is about to be dispatched: its workgroups take their CUs first */
                    WF_TRY(hipEventRecord(eng.tail_events[2], tail_stream));
                    WF_TRY(hipStreamWaitEvent(P.stream, eng.tail_events[2], 0));
                    WfBuf tb = P.b;
                    tb.st[0] = eng.tail.st; tb.ctr = eng.tail.ctr; tb.stack_spill = eng.tail.spill; tb.capacity = (uint32_t) eng.tail.capacity;
                    timer.begin(KC_TAIL, tail_stream);
                    launch_finish(sc, tb, 0, P.bt, false, finish_grid_side, tail_stream);
                    timer.end(tail_stream);
                    WF_TRY(hipEventRecord(eng.tail_events[1], tail_stream));
                    pend.active = true; pend.film = P.film;
                    pend.fl = fl; pend.fl.tile_first = P.bt.tile_first; pend.fl.store_tile_first = P.bt.tile_first; pend.fl.n_tiles = P.bt.n_tiles; pend.fl.n_spp = P.bt.n_spp;
                    gather_deferred = true;
                    stats.n_launches += 2;
                } else {
                    timer.begin(KC_SHADE, P.stream);
                    launch_finish(sc, P.b, P.cur, P.bt, L.count_traversal, finish_grid_main, P.stream);
                    timer.end(P.stream);
                    stats.n_launches++;
                }
                P.h_ctr[C_N + P.cur] = 0;
            }
            if (!P.active || P.h_ctr[C_N + P.cur] != 0 || samples_left != 0u) continue;
            /* batch done: splat its samples (each pipe owns its tiles' accumulators) */
            if (!gather_deferred) {
                err = flush_pending();
                if (!err.empty()) return err;
                fl.tile_first = P.bt.tile_first; fl.store_tile_first = P.bt.tile_first; fl.n_tiles = P.bt.n_tiles; fl.n_spp = P.bt.n_spp;
                timer.begin(KC_FILM, P.stream);
                if (!L.film_reference) film_gather(sc, d_filter_table, P.film, fl, P.stream);
                timer.end(P.stream);
                stats.n_launches++;
            }
            P.active = false;
            P.s0 += P.bt.n_spp;
            if (P.s0 >= L.spp_count) { P.s0 = 0; P.t0 += P.bt.n_tiles; }
        }
        /* a batch's paths die geometrically (Russian roulette from depth 3): thousands of iterations on ONE
           batch mean the loop is stuck; the bound is per batch, however many batches the call needs */
        for (int k = 0; k < n_pipes; ++k)
            if (pipes[k].active && ++pipes[k].batch_rounds > 20000) return "wavefront: path loop did not terminate";
    }
    if (own_streams)        /* back to the caller's stream */
        for (int k = 0; k < n_pipes; ++k) {
            WF_TRY(hipEventRecord(g_events[k], pipes[k].stream));
            WF_TRY(hipStreamWaitEvent(s, g_events[k], 0));
        }
    timer.begin(KC_FILM, s);
    if (L.film_reference) {
        err = film_reference_order(film_store, film, sc, d_filter_table, L.spp_count, L.tiles_x, L.film_share, d_rgbw, s);
        if (!err.empty()) return err;
    } else film_resolve(sc, film, fl, d_rgbw, s);
    timer.end(s);
    stats.n_launches++;
    WF_TRY(hipGetLastError());
    unsigned long long h[2 * S_COUNT];
    WF_TRY(hipMemcpyAsync(h, g_pool.buf.stats, sizeof(h), hipMemcpyDeviceToHost, s));
    WF_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 2; ++k) {
        stats.n_camera += h[k * S_COUNT + S_CAM]; stats.n_closest += h[k * S_COUNT + S_CLOSEST]; stats.n_shadow += h[k * S_COUNT + S_SHADOW];
        stats.n_nodes += h[k * S_COUNT + S_NODES]; stats.n_tris += h[k * S_COUNT + S_TRIS];
        NORI_PROF_REPORT(h, k)
    }
    stats.n_invalid = film_invalid_count(film, s);
    timer.collect(stats.class_ms, stats.class_launches);
    if (d_census) {
        unsigned long long z[Z_COUNT];
        WF_TRY(hipMemcpy(z, d_census, sizeof(z), hipMemcpyDeviceToHost));
        (void) hipFree(d_census);
        fprintf(stderr, "[wavefront census] trips %llu | node steps %llu, lanes/step %.1f | triangle steps %llu, lanes/step %.1f | refills %llu, idle lanes/refill %.1f\n",
                z[Z_TRIPS], z[Z_INNER_TRIPS], (double) z[Z_INNER_LANES] / std::max(1ull, z[Z_INNER_TRIPS]), z[Z_LEAF_TRIPS],
                (double) z[Z_LEAF_LANES] / std::max(1ull, z[Z_LEAF_TRIPS]), z[Z_REFILLS], (double) z[Z_REFILL_LANES] / std::max(1ull, z[Z_REFILLS]));
    }
    return std::string();
}

} // namespace nrt
