/*
 * nori_hip.h -- C ABI of the MI355X (gfx950) hot path of the Nori renderer.
 *
 * This is the drop-in boundary: the host keeps Nori's
 * NoriObject/XML object graph, flattens it into the POD `nori_scene_desc`
 * below, and drives everything that Nori does per sample -- Accel build and
 * rayIntersect, Mesh::rayIntersect, Integrator::Li, BSDF::sample/eval/pdf,
 * Warp::*, Independent/pcg32, PerspectiveCamera::sampleRay, ImageBlock::put --
 * through the entry points declared here.  Plain pointers and sizes only; no
 * C++ types, no torch types.  Every function returns 0 on success or a
 * negative `nori_status`; nothing throws across the boundary.  A context is
 * bound to one GPU and is not thread safe (one host thread per context).  A
 * context owns every device buffer it uses -- scene, tree, the wavefront
 * engine's path-state pool and streams, the film's sample store -- so any
 * number of contexts may live in one process (one per GPU, or several on one
 * GPU) without sharing or freeing each other's memory; nori_hip_destroy frees
 * exactly its context's.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the wjakob/nori tree).
 *
 * The same POD structs are consumed by the CPU oracle (oracle/oracle.h), which
 * is test infrastructure only and is never linked into this library.
 */
#ifndef NORI_HIP_H
#define NORI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The structs below that the LIBRARY writes (nori_render_stats, nori_accel_info) grow with the library: they carry no size
 * field, the library fills sizeof(its own struct).  Header and library must therefore be of ONE version: a caller compares
 * nori_hip_abi_version() -- what the loaded library was built from -- with the NORI_HIP_ABI_VERSION it was compiled against
 * before its first call that passes such a struct, and refuses to go on if they differ (the host library and the Python
 * bindings of this repository do).  Bumped whenever a struct of this header changes size or layout, an enum value changes
 * meaning, or an entry point changes its signature. */
#define NORI_HIP_ABI_VERSION 7      /* round 6: nori_render_stats as of round 5 (tail_ms, tail_cus); option "wavefront_samples"; nori_accel_info: built_on_device, n_references */
int nori_hip_abi_version(void);

/* ------------------------------------------------------------------ enums */

typedef enum nori_status {
    NORI_OK = 0,
    NORI_ERR_INVALID_ARGUMENT = -1,
    NORI_ERR_NO_DEVICE = -2,       /* no HIP device / HIP runtime error      */
    NORI_ERR_OUT_OF_MEMORY = -3,
    NORI_ERR_NOT_READY = -4,       /* e.g. render before build_accel         */
    NORI_ERR_UNSUPPORTED = -5,
    NORI_ERR_INTERNAL = -6
} nori_status;

/* BSDF plugin names: src/diffuse.cpp:90, src/mirror.cpp:50,
 * src/dielectric.cpp:49, src/microfacet.cpp:90 (NORI_REGISTER_CLASS) */
typedef enum nori_bsdf_type {
    NORI_BSDF_DIFFUSE = 0,
    NORI_BSDF_MIRROR = 1,
    NORI_BSDF_DIELECTRIC = 2,
    NORI_BSDF_MICROFACET = 3
} nori_bsdf_type;

/* Integrator plugin names used by scenes/pa1..pa5 (no implementation ships
 * in the reference; behaviour spec: DESIGN.md §Integrators) */
typedef enum nori_integrator_type {
    NORI_INTEGRATOR_NORMALS = 0,
    NORI_INTEGRATOR_AO = 1,
    NORI_INTEGRATOR_SIMPLE = 2,
    NORI_INTEGRATOR_WHITTED = 3,
    NORI_INTEGRATOR_PATH_MATS = 4,
    NORI_INTEGRATOR_PATH_EMS = 5,
    NORI_INTEGRATOR_PATH_MIS = 6
} nori_integrator_type;

/* src/rfilter.cpp:110-113 */
typedef enum nori_rfilter_type {
    NORI_RFILTER_GAUSSIAN = 0,
    NORI_RFILTER_MITCHELL = 1,
    NORI_RFILTER_TENT = 2,
    NORI_RFILTER_BOX = 3
} nori_rfilter_type;

/* include/nori/common.h:179-183 (EMeasure) */
typedef enum nori_measure {
    NORI_MEASURE_UNKNOWN = 0,
    NORI_MEASURE_SOLID_ANGLE = 1,
    NORI_MEASURE_DISCRETE = 2
} nori_measure;

/* include/nori/warp.h:18-57; numbering follows src/warptest.cpp:79-82 */
typedef enum nori_warp_type {
    NORI_WARP_SQUARE = 0,
    NORI_WARP_TENT = 1,
    NORI_WARP_DISK = 2,
    NORI_WARP_UNIFORM_SPHERE = 3,
    NORI_WARP_UNIFORM_HEMISPHERE = 4,
    NORI_WARP_COSINE_HEMISPHERE = 5,
    NORI_WARP_BECKMANN = 6
} nori_warp_type;

/* How the pcg32 stream of a camera sample is seeded.
 *  PER_SAMPLE : seed(initstate = y*width + x, initseq = sample index) before
 *               every camera sample -- order independent, what a GPU can
 *               reproduce; the mode in which device and oracle images are
 *               compared sample for sample.
 *  NORI_BLOCK : the reference's scheme, src/independent.cpp:36-41 -- one
 *               stream per 32x32 block seeded with (offset.x, offset.y) and
 *               consumed serially over y, x, sample.  On the device this runs
 *               one lane per block (whole frames from sample 0 only; slow by
 *               design): every camera sample then carries exactly the
 *               radiance of a render with the reference's own sampler. */
typedef enum nori_seed_mode {
    NORI_SEED_PER_SAMPLE = 0,
    NORI_SEED_NORI_BLOCK = 1
} nori_seed_mode;

typedef enum nori_accel_builder {
    NORI_ACCEL_HOST_SAH = 0,   /* binned SAH on the host, uploaded            */
    NORI_ACCEL_GPU_LBVH = 1,   /* built on the device: Morton order + radix tree (1.5 ms per million triangles)   */
    NORI_ACCEL_AUTO = 2,       /* GPU_PLOC; HOST_SAH if the device's tree comes out deeper than the traversal stack */
    NORI_ACCEL_GPU_PLOC = 3    /* built on the device: triangles whose boxes are several times the scene's typical one and hold other
                                  geometry enter as several parts (spatial splits up front), Morton order, nearest-neighbour
                                  clustering (PLOC, 3 ms per million triangles), treelet restructuring (a wave per treelet, 2 - 3 ms
                                  per million triangles and sweep), parallel re-insertion (1 ms per million candidates).
                                  wf_extend against HOST_SAH's tree: Cornell box +0.8 %, pa5 table +0.7 %, AO scene +-1 %, terrain of
                                  10 M triangles +2.7 %; built in 19 / 28 / 33 / 144 ms against 30 / 56 / 90 / 2 400 */
} nori_accel_builder;

/* ------------------------------------------------------ scene description */

/* BSDF parameters; defaults as in the reference constructors
 * (src/diffuse.cpp:18-20, src/dielectric.cpp:17-23, src/microfacet.cpp:17-36).
 * `albedo` holds Diffuse::m_albedo or Microfacet::m_kd. */
typedef struct nori_bsdf_desc {
    int32_t type;       /* nori_bsdf_type */
    float albedo[3];
    float alpha;
    float int_ior;
    float ext_ior;
    float ks;           /* 1 - max(kd), src/microfacet.cpp:35 */
} nori_bsdf_desc;

/* One <mesh>: the buffers of include/nori/mesh.h:160-163 (m_V, m_N, m_UV, m_F;
 * column-major 3xN == xyz interleaved per vertex), already in world space
 * (src/obj.cpp:52,62 applies toWorld at load), plus its BSDF and optional
 * area emitter (mesh.h:128-134). */
typedef struct nori_mesh_desc {
    uint32_t n_vertices;
    uint32_t n_triangles;
    const float *positions;    /* 3 * n_vertices                           */
    const float *normals;      /* 3 * n_vertices or NULL                   */
    const float *texcoords;    /* 2 * n_vertices or NULL                   */
    const uint32_t *indices;   /* 3 * n_triangles                          */
    nori_bsdf_desc bsdf;
    int32_t is_emitter;        /* <emitter type="area">                    */
    float radiance[3];
} nori_mesh_desc;

/* src/perspective.cpp:22-39; to_world is row-major 4x4 (camera -> world). */
typedef struct nori_camera_desc {
    int32_t width, height;
    float fov;                 /* horizontal, degrees */
    float near_clip, far_clip;
    float to_world[16];
} nori_camera_desc;

/* src/rfilter.cpp constructors. radius is ignored for tent (1) and box (.5) */
typedef struct nori_rfilter_desc {
    int32_t type;              /* nori_rfilter_type */
    float radius;
    float stddev;              /* gaussian */
    float B, C;                /* mitchell */
} nori_rfilter_desc;

typedef struct nori_integrator_desc {
    int32_t type;              /* nori_integrator_type */
    float position[3];         /* simple: point light position             */
    float energy[3];           /* simple: point light power                */
} nori_integrator_desc;

typedef struct nori_scene_desc {
    uint32_t n_meshes;
    const nori_mesh_desc *meshes;
    nori_camera_desc camera;
    nori_rfilter_desc rfilter;
    nori_integrator_desc integrator;
    int32_t sample_count;      /* Independent::m_sampleCount               */
} nori_scene_desc;

/* ------------------------------------------------------- per-query records */

/* include/nori/ray.h:30-34 without dRcp (recomputed on the device). 32 B. */
typedef struct nori_ray {
    float o[3];
    float d[3];
    float mint, maxt;
} nori_ray;

/* include/nori/mesh.h:23-35.  mesh = index into nori_scene_desc::meshes,
 * 0xFFFFFFFF when nothing was hit (Intersection::mesh == nullptr).
 * `tri` is the triangle index inside that mesh (the `f` of accel.cpp:25). */
typedef struct nori_intersection {
    float p[3];
    float t;
    float uv[2];
    float sh_s[3], sh_t[3], sh_n[3];
    float geo_s[3], geo_t[3], geo_n[3];
    uint32_t mesh;
    uint32_t tri;
} nori_intersection;

#define NORI_NO_HIT 0xFFFFFFFFu

/* What to render: the frame is cut into fixed tiles (NORI_TILE_SIZE^2 px,
 * the device analogue of NORI_BLOCK_SIZE, include/nori/block.h:17).  A call
 * renders the tiles whose raster index i satisfies i % tile_mod == tile_rem,
 * with camera samples [spp_begin, spp_begin+spp_count) per pixel, and ADDS the
 * weighted samples into the caller's RGBW accumulation buffer, laid out like
 * the full-frame ImageBlock of src/main.cpp:67: (height+2b) rows x
 * (width+2b) columns x 4 floats, b = nori_hip_border_size().  Summing the
 * buffers of several calls / ranks is exactly ImageBlock::put(ImageBlock&)
 * (src/block.cpp:93-102). */
typedef struct nori_render_params {
    uint32_t spp_begin;
    uint32_t spp_count;
    uint32_t tile_mod;      /* >= 1 */
    uint32_t tile_rem;      /* <  tile_mod */
    int32_t seed_mode;      /* nori_seed_mode */
    int32_t count_traversal;/* != 0: also count node/triangle tests (slower) */
    int32_t time_kernels;   /* != 0: HIP events around every kernel launch, summed per
                               kernel class into nori_render_stats.*_ms          */
    void *stream;           /* hipStream_t or NULL for the default stream   */
} nori_render_params;

#define NORI_TILE_SIZE 16

/* Work counters of one render call.  Ray queries are counted where the
 * reference would pass through Scene::rayIntersect (include/nori/scene.h:63
 * closest hit, :82 shadow). */
typedef struct nori_render_stats {
    uint64_t n_camera_samples;
    uint64_t n_closest_rays;
    uint64_t n_shadow_rays;
    uint64_t n_node_tests;   /* only when count_traversal != 0 */
    uint64_t n_tri_tests;    /* only when count_traversal != 0 */
    uint64_t n_invalid;      /* samples dropped by the isValid() guard,
                                src/block.cpp:63-67 */
    float kernel_ms;         /* HIP-event time of the render kernel on the
                                stream it was launched on */
    uint32_t n_workgroups;   /* launch geometry of the render kernel: one
                                workgroup per (tile, spp chunk)              */
    uint32_t lds_bytes;      /* dynamic LDS per workgroup                    */
    /* time_kernels != 0: summed HIP-event durations, on the launch stream, of
       the ray-query kernels (wf_extend; render_kernel for the megakernel, which
       also shades), the Li kernels (wf_shade, wf_finish) and the film kernels */
    float trace_ms, shade_ms, film_ms;
    uint32_t n_trace_launches;
    uint32_t engine;         /* what rendered this call: 0 = megakernel (render_kernel), 1 = wavefront (wf_extend / wf_shade),
                                2 = the block-serial kernel of NORI_SEED_NORI_BLOCK.  The option "engine" is a request: "auto"
                                picks by job size, and trees with wide nodes are walked by the wavefront engine only */
    uint32_t trace_cus;      /* CUs the ray-query kernels ran on: all of the device's, or -- wavefront engine, big jobs -- all but
                                the CUs set aside for the shading and film kernels, which then run BESIDE the ray queries of
                                the other half of the tiles (trace_ms, shade_ms and film_ms overlap in that case) */
    float tail_ms;           /* time_kernels != 0: summed durations of the wf_finish launches that ran BESIDE the next batch's kernels, on
                                their own `tail_cus` CUs (wavefront engine, calls of two or more batches); not part of shade_ms */
    uint32_t tail_cus;
} nori_render_stats;

typedef struct nori_accel_info {
    uint32_t n_triangles;
    uint32_t n_nodes;
    uint32_t n_leaves;
    uint32_t max_depth;
    uint32_t node_bytes;     /* size of one node record as laid out */
    uint32_t tri_bytes;      /* size of one leaf triangle record     */
    uint64_t total_bytes;    /* nodes + leaf triangles in HBM        */
    float build_ms;
    float sah_cost;
    uint32_t node_children;  /* boxes one node record holds: 2 (BVH2) or 4 (wide nodes, quantised boxes) */
    uint32_t node_records_32b;  /* 1: the tree also exists as 32-B node records (two children's boxes as 16-bit planes on one
                                grid + the links), which the wavefront engine's hand-written node loop walks when the tree's
                                traversal stack fits LDS (depth <= 16); BVH2 trees without unbounded boxes only */
    uint32_t built_on_device;   /* 1: the tree was built by HIP kernels (lbvh.hip), 0: by the host's SAH builder and uploaded */
    uint32_t n_references;      /* device builder: (triangle, box) references the tree holds -- the triangles, plus the parts of those that
                                   were split; 0: built on the host */
} nori_accel_info;

typedef struct nori_hip_ctx nori_hip_ctx;

/* ------------------------------------------------------------ life cycle */

/* Bind a context to HIP device `device`. */
int nori_hip_create(int device, nori_hip_ctx **out);
void nori_hip_destroy(nori_hip_ctx *ctx);
/* Message of the last failing call on this context (or on creation when
 * ctx == NULL).  Stands in for NoriException::what(), common.h:135-140. */
const char *nori_hip_last_error(const nori_hip_ctx *ctx);

/* Copy the scene into HBM (SoA vertex/index buffers, material and emitter
 * tables, camera, filter table).  Replaces the load-time half of
 * Scene::addChild / Accel::addMesh (src/scene.cpp:48-52, src/accel.cpp:12-17),
 * the ImageBlock filter tabulation (src/block.cpp:18-27) and
 * PerspectiveCamera::activate (src/perspective.cpp:41-74). */
int nori_hip_upload_scene(nori_hip_ctx *ctx, const nori_scene_desc *scene);

/* Accel::build (src/accel.cpp:19-21 -- a no-op in the reference, a BVH here) */
int nori_hip_build_accel(nori_hip_ctx *ctx, int builder /* nori_accel_builder */);
int nori_hip_accel_info(const nori_hip_ctx *ctx, nori_accel_info *out);

/* Diagnostics of the arithmetic the device substitutes for the reference's `/`, `1 / x` and std::sqrt (Eigen's
 * operator/ and normalized() in src/common.cpp:248-257, include/nori/frame.h, the BSDFs): out[0..2] = operands that left
 * the domain on which the short reciprocal / quotient / square-root sequences are verified bit-identical to the IEEE
 * operations, out[3] = results that came out NaN or infinite (rt_types.h), since the last reset.  Only the
 * build with -DNORI_COUNT_EXCURSIONS (libnori_hip_count.so) counts; the product build returns NORI_ERR_UNSUPPORTED. */
int nori_hip_debug_excursions(nori_hip_ctx *ctx, unsigned long long out[4], int reset);

/* Tuning / engine selection (no reference counterpart).  Keys:
 *   "engine"          "auto" (default: wavefront for >= 2^19 camera samples per call -- 2^24 for the `normals`
 *                     integrator --, else megakernel) | "megakernel" | "wavefront"
 *   "wavefront_paths" paths in flight in the wavefront engine: records of its state pool (default 2^29, 216 B of HBM each;
 *                     bounded by 85 % of the free memory)
 *   "wavefront_samples" camera samples per batch: what the film's sample store holds at a time, 20 B each (twice when a call
 *                     has several batches).  "0" (default): as many as wavefront_paths -- a batch starts all its samples in its
 *                     first pass, the fastest schedule.  A batch bigger than the pool starts its samples pass by pass in the
 *                     slots finished paths leave (regeneration): same frame, bit for bit, at +2 .. 4 % for half to a quarter of the
 *                     pool (more below that); it is what lets an out-of-memory retry shrink the pool and keep the frame, and
 *                     film_order = reference run on any pool
 *   "accel_layout"    node layout of the NEXT nori_hip_build_accel: "bvh2" (64-B node = two full-precision child
 *                     boxes; the wavefront engine walks a second, 32-B form of them, see nori_accel_info) | "bvh4q"
 *                     (64-B node = four child boxes quantised to 8 bits: half the node fetches, for trees that do
 *                     not fit the caches) | "auto" (default: bvh4q from 2^20 triangles)
 *   "film_order"      "fast" (default: the film adds a pixel's samples round by round in LDS tiles) | "reference" (the
 *                     samples are added in the order of renderBlock / ImageBlock::put / BlockGenerator,
 *                     src/main.cpp:33-53, src/block.cpp:62-152: whole frames -- or whole rows of 32x32 blocks through
 *                     nori_hip_render_block_rows --, slower; the frame is then bit-identical to a single-threaded
 *                     render of the same samples by the reference's loops)
 * Unknown keys return NORI_ERR_INVALID_ARGUMENT. */
int nori_hip_set_option(nori_hip_ctx *ctx, const char *key, const char *value);
/* The current value of an option as set_option would take it ("engine", "wavefront_paths", "wavefront_samples", "film_order", "accel_layout"),
 * NUL-terminated into value[capacity]; NORI_ERR_INVALID_ARGUMENT for unknown keys or a buffer too small. */
int nori_hip_get_option(const nori_hip_ctx *ctx, const char *key, char *value, size_t capacity);

/* ImageBlock::m_borderSize for the uploaded filter (src/block.cpp:20). */
int nori_hip_border_size(const nori_hip_ctx *ctx);

/* --------------------------------------------------- operator-level twins */

/* Accel::rayIntersect(ray, its, shadowRay) (src/accel.cpp:23-99) for a batch
 * of n rays held in HOST memory.  shadow_ray != 0: only `mesh` is meaningful
 * (0 = some occluder, NORI_NO_HIT = none), as the reference returns early. */
int nori_hip_intersect(nori_hip_ctx *ctx, const nori_ray *rays,
                       nori_intersection *its, size_t n, int shadow_ray);

/* Same on DEVICE buffers (rays: n x nori_ray, its: n x nori_intersection),
 * asynchronous on `stream`. */
int nori_hip_intersect_device(nori_hip_ctx *ctx, const void *d_rays,
                              void *d_its, size_t n, int shadow_ray,
                              void *stream);

/* PerspectiveCamera::sampleRay (src/perspective.cpp:76-97) for n film
 * positions (pixel_samples: 2n floats, fractional pixel coordinates). */
int nori_hip_sample_rays(nori_hip_ctx *ctx, const float *pixel_samples,
                         size_t n, nori_ray *rays);

/* Integrator::Li (include/nori/integrator.h:42) for n rays; path k draws its
 * random numbers from pcg32.seed(seed_state[k], seed_seq[k]).  rgb: 3n. */
int nori_hip_li(nori_hip_ctx *ctx, const nori_ray *rays, size_t n,
                const uint64_t *seed_state, const uint64_t *seed_seq,
                float *rgb);

/* BSDF::sample / eval / pdf (include/nori/bsdf.h:59,70,87) in the local
 * frame.  wi, wo: 3n floats; sample: 2n; weight, value: 3n; eta, pdf: n. */
int nori_hip_bsdf_sample(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf,
                         const float *wi, const float *sample, size_t n,
                         float *wo, float *weight, float *eta,
                         int32_t *measure);
int nori_hip_bsdf_eval(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf,
                       const float *wi, const float *wo, size_t n,
                       float *value);
int nori_hip_bsdf_pdf(nori_hip_ctx *ctx, const nori_bsdf_desc *bsdf,
                      const float *wi, const float *wo, size_t n, float *pdf);

/* Warp::squareToX / squareToXPdf (include/nori/warp.h:18-57).  sample: 2n.
 * out: 3n (2D warps leave z = 0).  pdf takes points in the same 3n layout. */
int nori_hip_warp(nori_hip_ctx *ctx, int warp /* nori_warp_type */,
                  float param, const float *sample, size_t n, float *out);
int nori_hip_warp_pdf(nori_hip_ctx *ctx, int warp, float param,
                      const float *points, size_t n, float *pdf);

/* pcg32::nextFloat stream (Independent::next1D, src/independent.cpp:46-48):
 * out[k*count + j] = j-th float of pcg32.seed(seed_state[k], seed_seq[k]). */
int nori_hip_pcg32_floats(nori_hip_ctx *ctx, const uint64_t *seed_state,
                          const uint64_t *seed_seq, size_t n, uint32_t count,
                          float *out);
/* The same streams `skip` draws further (pcg32::advance): out[k*count + j] = float number
 * skip + j of pcg32.seed(seed_state[k], seed_seq[k]).  This is how the host's Independent sampler
 * continues ONE stream seeded by prepare(block) = seed(offset.x, offset.y)
 * (src/independent.cpp:36-41) across refills of its buffer. */
int nori_hip_pcg32_floats_at(nori_hip_ctx *ctx, const uint64_t *seed_state,
                             const uint64_t *seed_seq, uint64_t skip, size_t n,
                             uint32_t count, float *out);

/* ImageBlock::put(pos, value) (src/block.cpp:62-91) of n samples into the
 * full-frame RGBW buffer `rgbw` (HOST memory, layout as in
 * nori_render_params; accumulated into). positions: 2n, values: 3n. */
int nori_hip_splat(nori_hip_ctx *ctx, const float *positions,
                   const float *values, size_t n, float *rgbw);

/* ----------------------------------------------------------- the hot path */

/* renderBlock + render (src/main.cpp:27-119) for the tiles / samples named in
 * `params`, accumulating into the DEVICE buffer d_rgbw (see
 * nori_render_params).  Asynchronous on params->stream unless `stats` is
 * non-NULL, in which case the call synchronises the stream and fills it. */
int nori_hip_render(nori_hip_ctx *ctx, const nori_render_params *params,
                    void *d_rgbw, nori_render_stats *stats);

/* film_order = "reference" shared out over devices or ranks.  The reference adds a 32x32 block's samples consecutively
 * into the block's own ImageBlock (renderBlock, src/main.cpp:27-55) and the blocks into the frame in BlockGenerator's order
 * (src/block.cpp:93-152): the smallest share that keeps the order is a block, the one handed out here is a ROW of blocks.
 * nori_hip_render_block_rows renders block rows [row_begin, row_begin + row_count) (clipped to the frame) with samples
 * [spp_begin, spp_begin + spp_count) and writes the bordered accumulators of THOSE blocks into d_block_acc, the array for
 * ALL blocks of the frame: nori_hip_block_acc_floats floats, block b = by * blocks_x + bx at b * (32 + 2 border)^2 * 4,
 * zeroed by the caller.  The arrays of disjoint shares sum exactly in any order (x + 0 = x), e.g. by one ncclReduce;
 * nori_hip_resolve_blocks then ADDS the blocks into the RGBW frame in BlockGenerator's order -- the bits of the frame one
 * device renders with film_order = reference, whatever the number of shares.  params: tile_mod 1, tile_rem 0,
 * NORI_SEED_PER_SAMPLE; the context's film_order must be "reference". */
int nori_hip_block_acc_floats(nori_hip_ctx *ctx, size_t *n_floats);
int nori_hip_render_block_rows(nori_hip_ctx *ctx, const nori_render_params *params, uint32_t row_begin, uint32_t row_count,
                               void *d_block_acc, nori_render_stats *stats);
int nori_hip_resolve_blocks(nori_hip_ctx *ctx, const void *d_block_acc, void *d_rgbw, void *stream);

/* Convenience: zero a host RGBW buffer, render into a scratch device buffer
 * and copy back (synchronous). */
int nori_hip_render_host(nori_hip_ctx *ctx, const nori_render_params *params,
                         float *rgbw, nori_render_stats *stats);

/* ImageBlock::toBitmap (src/block.cpp:45-51): rgb = rgbw.rgb / w (0 if w==0)
 * without the border, on the device.  d_rgb: height*width*3 floats. */
int nori_hip_develop(nori_hip_ctx *ctx, const void *d_rgbw, void *d_rgb,
                     void *stream);

/* ------------------------------------------------------ all GPUs of one node
 *
 * render() of the reference spreads the image blocks over TBB workers (src/main.cpp:85-113) and merges every worker's
 * block into the frame under a mutex, ImageBlock::put(ImageBlock&) (src/block.cpp:93-102).  A group does the same over
 * the GPUs of a node: one context and one host thread per device, every device renders its share of the frame into its
 * own RGBW buffer, ONE merge on the first device (RCCL over xGMI), the merged frame to the caller's host buffer.
 * Shares: NORI_SPLIT_TILE = 16x16 tiles round-robin over the devices, NORI_SPLIT_SAMPLE = a range of the per-pixel
 * sample indices each.  Merges: NORI_MERGE_REDUCE = ncclReduce(sum) of the whole frames, NORI_MERGE_GATHER (tile split,
 * tile columns divisible by the device count) = every device sends only the column strips its tiles touched.
 * A device list that names a device more than once is served without RCCL (peer copies) -- for tests on one GPU; so is a
 * node whose librccl cannot be loaded (nori_hip_group_warning says so).  A fresh RCCL group proves its communicators before
 * nori_hip_group_create returns, on a buffer the size of a 2052^2 frame (67 MB): a sum-reduce of a known pattern, then the
 * gather merge's grouped ncclSend / ncclRecv, both checked on the device.  With film_order = reference on its contexts a
 * group hands out rows of 32x32 blocks instead (whatever `split` says), merges the blocks' accumulators (a reduce of disjoint
 * arrays: exact) and adds them into the frame in BlockGenerator's order on the first device: the frame has the bits of the
 * one-device frame for any number of devices.  NORI_SEED_NORI_BLOCK renders whole frames on one device: a group of more
 * than one refuses it. */
typedef struct nori_hip_group nori_hip_group;
typedef enum nori_group_split { NORI_SPLIT_TILE = 0, NORI_SPLIT_SAMPLE = 1 } nori_group_split;
typedef enum nori_group_merge { NORI_MERGE_REDUCE = 0, NORI_MERGE_GATHER = 1 } nori_group_merge;

/* NORI_ERR_NO_DEVICE when a listed device does not exist ("device 1 not found ...", nori_hip_group_last_error(NULL)) */
int nori_hip_group_create(const int *devices, int n_devices, nori_hip_group **out);
void nori_hip_group_destroy(nori_hip_group *group);
int nori_hip_group_size(const nori_hip_group *group);
/* context of device i of the list (owned by the group): options, accel_info, the batch operators */
nori_hip_ctx *nori_hip_group_ctx(nori_hip_group *group, int i);
const char *nori_hip_group_last_error(const nori_hip_group *group);
/* "rccl" or "copy" */
const char *nori_hip_group_transport(const nori_hip_group *group);
/* "" or why the group merges over peer copies although RCCL was wanted (librccl not loadable on this box) */
const char *nori_hip_group_warning(const nori_hip_group *group);
/* nori_render_stats::engine of every device's share of the last frame ("auto" picks by the size of a share, so the devices
 * may differ); returns the number of entries written (<= capacity) or a negative nori_status */
int nori_hip_group_engines(const nori_hip_group *group, uint32_t *engines, int capacity);
/* nori_hip_upload_scene + nori_hip_build_accel on every device, side by side (Scene::activate per device) */
int nori_hip_group_upload_scene(nori_hip_group *group, const nori_scene_desc *scene, int builder /* nori_accel_builder */);
/* The render loop of src/main.cpp:78-119 over the group: samples [spp_begin, spp_begin + spp_count) of every pixel of the
 * width x height frame (params->tile_mod must be 1: the group shares the tiles out itself), merged into the HOST buffer
 * `rgbw` (layout of nori_render_params; overwritten).  stats: counters summed over the devices, times = the slowest
 * device's; merge_ms: HIP-event time of the merge on the first device's stream. */
int nori_hip_group_render_host(nori_hip_group *group, const nori_render_params *params, int split, int merge,
                               int width, int height, float *rgbw, nori_render_stats *stats, float *merge_ms);

#ifdef __cplusplus
}
#endif
#endif /* NORI_HIP_H */
