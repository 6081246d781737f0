/*
 * oracle.h -- C ABI of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY: a CPU restatement of the reference's per-sample
 * path (see oracle_scene.h for the file:line map), used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker and
 * the reported CPU baseline -- never by the product.  The entry points mirror
 * include/nori_hip.h one to one, on host buffers.
 *
 * Pinning status: the reference cannot be built here (all
 * ext/ submodules are empty) and ships no integrators, emitters or warps, so
 * the oracle is pinned by the reference's own analytic goldens --
 * tests/test_oracle_goldens.py runs scenes/pa4/tests/*.xml and
 * scenes/pa5/tests/*.xml (Student-t and chi^2) and the warptest chi^2 cases
 * against it.  Unpinned by the reference (no test touches them): triangle /
 * Accel outputs, camera rays, ImageBlock::put weights, normals/ao/simple,
 * mirror, dielectric, raw pcg32 output (checked against the published PCG
 * known answers instead).
 */
#ifndef NORI_ORACLE_H
#define NORI_ORACLE_H

#include "../include/nori_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

int oracle_create(const nori_scene_desc *scene, oracle_ctx **out);
void oracle_destroy(oracle_ctx *ctx);
/* 0: brute force over all triangles, exactly src/accel.cpp:30-40 (default);
 * 1: SAH BVH returning the same hits (CPU baseline speed). */
int oracle_set_accel(oracle_ctx *ctx, int use_bvh);
int oracle_border_size(const oracle_ctx *ctx);
int oracle_filter_table(const oracle_ctx *ctx, float *table33);

int oracle_intersect(oracle_ctx *ctx, const nori_ray *rays, nori_intersection *its,
                     size_t n, int shadow_ray);
int oracle_sample_rays(oracle_ctx *ctx, const float *pixel_samples, size_t n, nori_ray *rays);
int oracle_li(oracle_ctx *ctx, const nori_ray *rays, size_t n, const uint64_t *seed_state,
              const uint64_t *seed_seq, float *rgb);
int oracle_bsdf_sample(const nori_bsdf_desc *bsdf, const float *wi, const float *sample, size_t n,
                       float *wo, float *weight, float *eta, int32_t *measure);
int oracle_bsdf_eval(const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *value);
int oracle_bsdf_pdf(const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *pdf);
int oracle_warp(int warp, float param, const float *sample, size_t n, float *out);
int oracle_warp_pdf(int warp, float param, const float *points, size_t n, float *pdf);
int oracle_pcg32_floats(const uint64_t *seed_state, const uint64_t *seed_seq, size_t n,
                        uint32_t count, float *out);
/* raw 32-bit outputs of pcg32.seed(state, seq) (or the default-constructed
 * generator when use_default != 0) for the PCG known-answer test */
int oracle_pcg32_uints(uint64_t seed_state, uint64_t seed_seq, int use_default, uint32_t count, uint32_t *out);
int oracle_splat(oracle_ctx *ctx, const float *positions, const float *values, size_t n, float *rgbw);
float oracle_fresnel(float cos_theta_i, float ext_ior, float int_ior);
/* sin (0) / cos (1) / log (2) / exp (3) as the oracle evaluates them (oracle_libm.h) */
int oracle_libm_eval(int op, const float *x, size_t n, float *out);

/* renderBlock + render (src/main.cpp:27-119) with `threads` std::thread
 * workers pulling 32x32 blocks in the reference's spiral order.  Same
 * params / buffer layout / accumulate-into semantics as nori_hip_render_host.
 * seed_mode NORI_SEED_NORI_BLOCK reproduces src/independent.cpp:36-41
 * (requires spp_begin == 0, tile_mod == 1); NORI_SEED_PER_SAMPLE matches the
 * device.  In PER_SAMPLE mode the tile_mod/tile_rem selection uses the same
 * NORI_TILE_SIZE raster tiles as the device.  stats->kernel_ms = wall ms. */
int oracle_render(oracle_ctx *ctx, const nori_render_params *params, float *rgbw,
                  nori_render_stats *stats, int threads);
int oracle_develop(const oracle_ctx *ctx, const float *rgbw, float *rgb);

#ifdef __cplusplus
}
#endif
#endif
