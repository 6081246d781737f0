/*
 * oracle.cpp -- C ABI + render driver of the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * render(): src/main.cpp:58-119 with tbb::parallel_for replaced by
 * std::thread workers (TBB headers are absent); renderBlock(): src/main.cpp:27-56.
 */
#include "oracle.h"
#include "oracle_scene.h"

#include <chrono>

using namespace oracle;

struct oracle_ctx {
    std::unique_ptr<Scene> scene;
};

static inline Ray toRay(const nori_ray &r) {
    return Ray(Vec3(r.o[0], r.o[1], r.o[2]), Vec3(r.d[0], r.d[1], r.d[2]), r.mint, r.maxt);
}
static inline void put3(float *dst, const Vec3 &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

extern "C" {

int oracle_create(const nori_scene_desc *scene, oracle_ctx **out) {
    if (!scene || !out) return NORI_ERR_INVALID_ARGUMENT;
    oracle_ctx *c = new oracle_ctx();
    c->scene.reset(new Scene(*scene));
    *out = c;
    return NORI_OK;
}

void oracle_destroy(oracle_ctx *ctx) { delete ctx; }

int oracle_set_accel(oracle_ctx *ctx, int use_bvh) {
    if (!ctx) return NORI_ERR_INVALID_ARGUMENT;
    Accel &a = ctx->scene->accel;
    if (use_bvh && a.nodes.empty()) a.build();
    a.useBVH = use_bvh != 0;
    return NORI_OK;
}

int oracle_border_size(const oracle_ctx *ctx) {
    return (int) std::ceil(ctx->scene->rfilter.radius - 0.5f);
}

int oracle_filter_table(const oracle_ctx *ctx, float *table33) {
    ImageBlock b(1, 1, &ctx->scene->rfilter);
    for (int i = 0; i <= kFilterResolution; ++i) table33[i] = b.filter[i];
    return NORI_OK;
}

int oracle_intersect(oracle_ctx *ctx, const nori_ray *rays, nori_intersection *out, size_t n, int shadow_ray) {
    if (!ctx || !rays || !out) return NORI_ERR_INVALID_ARGUMENT;
    const Scene &sc = *ctx->scene;
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            for (size_t i = t; i < n; i += nt) {
                Intersection its;
                nori_intersection &o = out[i];
                std::memset(&o, 0, sizeof(o));
                bool hit = sc.accel.rayIntersect(toRay(rays[i]), its, shadow_ray != 0);
                o.mesh = NORI_NO_HIT; o.tri = NORI_NO_HIT;
                if (!hit) continue;
                if (shadow_ray) { o.mesh = 0; continue; }
                put3(o.p, its.p); o.t = its.t; o.uv[0] = its.uv.x; o.uv[1] = its.uv.y;
                put3(o.sh_s, its.shFrame.s); put3(o.sh_t, its.shFrame.t); put3(o.sh_n, its.shFrame.n);
                put3(o.geo_s, its.geoFrame.s); put3(o.geo_t, its.geoFrame.t); put3(o.geo_n, its.geoFrame.n);
                o.mesh = its.mesh->index; o.tri = its.tri;
            }
        });
    }
    for (auto &t : th) t.join();
    return NORI_OK;
}

int oracle_sample_rays(oracle_ctx *ctx, const float *ps, size_t n, nori_ray *rays) {
    if (!ctx || !ps || !rays) return NORI_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        Ray r;
        ctx->scene->camera.sampleRay(r, Vec2(ps[2 * i], ps[2 * i + 1]));
        put3(rays[i].o, r.o); put3(rays[i].d, r.d); rays[i].mint = r.mint; rays[i].maxt = r.maxt;
    }
    return NORI_OK;
}

int oracle_li(oracle_ctx *ctx, const nori_ray *rays, size_t n, const uint64_t *seed_state,
              const uint64_t *seed_seq, float *rgb) {
    if (!ctx || !rays || !rgb || !seed_state || !seed_seq) return NORI_ERR_INVALID_ARGUMENT;
    const Scene &sc = *ctx->scene;
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    if (n < 1024) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            RayCounter rc;
            Integrator integ(&sc, &rc);
            for (size_t i = t; i < n; i += nt) {
                Sampler s; s.rng.seed(seed_state[i], seed_seq[i]);
                Color3 L = integ.Li(s, toRay(rays[i]));
                put3(rgb + 3 * i, L);
            }
        });
    }
    for (auto &t : th) t.join();
    return NORI_OK;
}

int oracle_bsdf_sample(const nori_bsdf_desc *bsdf, const float *wi, const float *sample, size_t n,
                       float *wo, float *weight, float *eta, int32_t *measure) {
    if (!bsdf) return NORI_ERR_INVALID_ARGUMENT;
    BSDF b(*bsdf);
    for (size_t i = 0; i < n; ++i) {
        BSDFQueryRecord rec(Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
        rec.wo = Vec3(0.0f);
        Color3 w = b.sample(rec, Vec2(sample[2 * i], sample[2 * i + 1]));
        put3(wo + 3 * i, rec.wo); put3(weight + 3 * i, w);
        if (eta) eta[i] = rec.eta;
        if (measure) measure[i] = rec.measure;
    }
    return NORI_OK;
}

int oracle_bsdf_eval(const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *value) {
    if (!bsdf) return NORI_ERR_INVALID_ARGUMENT;
    BSDF b(*bsdf);
    for (size_t i = 0; i < n; ++i) {
        BSDFQueryRecord rec(Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]),
                            Vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), NORI_MEASURE_SOLID_ANGLE);
        put3(value + 3 * i, b.eval(rec));
    }
    return NORI_OK;
}

int oracle_bsdf_pdf(const nori_bsdf_desc *bsdf, const float *wi, const float *wo, size_t n, float *pdf) {
    if (!bsdf) return NORI_ERR_INVALID_ARGUMENT;
    BSDF b(*bsdf);
    for (size_t i = 0; i < n; ++i) {
        BSDFQueryRecord rec(Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]),
                            Vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), NORI_MEASURE_SOLID_ANGLE);
        pdf[i] = b.pdf(rec);
    }
    return NORI_OK;
}

int oracle_warp(int warp, float param, const float *sample, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) {
        Vec2 s(sample[2 * i], sample[2 * i + 1]);
        Vec3 r(0.0f);
        switch (warp) {
        case NORI_WARP_SQUARE: { Vec2 p = Warp::squareToUniformSquare(s); r = Vec3(p.x, p.y, 0); } break;
        case NORI_WARP_TENT: { Vec2 p = Warp::squareToTent(s); r = Vec3(p.x, p.y, 0); } break;
        case NORI_WARP_DISK: { Vec2 p = Warp::squareToUniformDisk(s); r = Vec3(p.x, p.y, 0); } break;
        case NORI_WARP_UNIFORM_SPHERE: r = Warp::squareToUniformSphere(s); break;
        case NORI_WARP_UNIFORM_HEMISPHERE: r = Warp::squareToUniformHemisphere(s); break;
        case NORI_WARP_COSINE_HEMISPHERE: r = Warp::squareToCosineHemisphere(s); break;
        case NORI_WARP_BECKMANN: r = Warp::squareToBeckmann(s, param); break;
        default: return NORI_ERR_INVALID_ARGUMENT;
        }
        put3(out + 3 * i, r);
    }
    return NORI_OK;
}

int oracle_warp_pdf(int warp, float param, const float *pts, size_t n, float *pdf) {
    for (size_t i = 0; i < n; ++i) {
        Vec3 v(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        Vec2 p(v.x, v.y);
        switch (warp) {
        case NORI_WARP_SQUARE: pdf[i] = Warp::squareToUniformSquarePdf(p); break;
        case NORI_WARP_TENT: pdf[i] = Warp::squareToTentPdf(p); break;
        case NORI_WARP_DISK: pdf[i] = Warp::squareToUniformDiskPdf(p); break;
        case NORI_WARP_UNIFORM_SPHERE: pdf[i] = Warp::squareToUniformSpherePdf(v); break;
        case NORI_WARP_UNIFORM_HEMISPHERE: pdf[i] = Warp::squareToUniformHemispherePdf(v); break;
        case NORI_WARP_COSINE_HEMISPHERE: pdf[i] = Warp::squareToCosineHemispherePdf(v); break;
        case NORI_WARP_BECKMANN: pdf[i] = Warp::squareToBeckmannPdf(v, param); break;
        default: return NORI_ERR_INVALID_ARGUMENT;
        }
    }
    return NORI_OK;
}

int oracle_pcg32_floats(const uint64_t *seed_state, const uint64_t *seed_seq, size_t n, uint32_t count, float *out) {
    for (size_t i = 0; i < n; ++i) {
        Pcg32 r; r.seed(seed_state[i], seed_seq[i]);
        for (uint32_t j = 0; j < count; ++j) out[i * count + j] = r.nextFloat();
    }
    return NORI_OK;
}

int oracle_pcg32_uints(uint64_t seed_state, uint64_t seed_seq, int use_default, uint32_t count, uint32_t *out) {
    Pcg32 r;
    if (!use_default) r.seed(seed_state, seed_seq);
    for (uint32_t j = 0; j < count; ++j) out[j] = r.nextUInt();
    return NORI_OK;
}

float oracle_fresnel(float c, float e, float i) { return fresnel(c, e, i); }

/* op 0: sin, 1: cos, 2: log, 3: exp -- the oracle's specified transcendental functions (oracle_libm.h) */
int oracle_libm_eval(int op, const float *x, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) {
        float s, c;
        switch (op) {
        case 0: oracle_libm::sincos(x[i], &s, &c); out[i] = s; break;
        case 1: oracle_libm::sincos(x[i], &s, &c); out[i] = c; break;
        case 2: out[i] = oracle_libm::log(x[i]); break;
        case 3: out[i] = oracle_libm::exp(x[i]); break;
        default: return -1;
        }
    }
    return 0;
}

int oracle_splat(oracle_ctx *ctx, const float *positions, const float *values, size_t n, float *rgbw) {
    if (!ctx) return NORI_ERR_INVALID_ARGUMENT;
    const Scene &sc = *ctx->scene;
    ImageBlock result(sc.camera.width, sc.camera.height, &sc.rfilter);
    for (size_t i = 0; i < n; ++i)
        result.put(Vec2(positions[2 * i], positions[2 * i + 1]),
                   Color3(values[3 * i], values[3 * i + 1], values[3 * i + 2]));
    for (size_t i = 0; i < result.px.size(); ++i) rgbw[i] += result.px[i];
    return NORI_OK;
}

int oracle_render(oracle_ctx *ctx, const nori_render_params *params, float *rgbw,
                  nori_render_stats *stats, int threads) {
    if (!ctx || !params || !rgbw) return NORI_ERR_INVALID_ARGUMENT;
    if (params->tile_mod == 0 || params->tile_rem >= params->tile_mod) return NORI_ERR_INVALID_ARGUMENT;
    Scene &sc = *ctx->scene;
    const bool noriMode = params->seed_mode == NORI_SEED_NORI_BLOCK;
    if (noriMode && (params->spp_begin != 0 || params->tile_mod != 1)) return NORI_ERR_UNSUPPORTED;
    const int W = sc.camera.width, H = sc.camera.height;
    const uint32_t tilesX = (uint32_t) ((W + NORI_TILE_SIZE - 1) / NORI_TILE_SIZE);

    BlockGenerator blockGenerator(W, H, kBlockSize);
    ImageBlock result(W, H, &sc.rfilter);
    result.clear();
    sc.accel.countTests = params->count_traversal != 0;
    sc.accel.nodeTests = 0; sc.accel.triTests = 0;

    std::atomic<uint64_t> nCamera{0}, nClosest{0}, nShadow{0};
    if (threads <= 0) threads = (int) std::max(1u, std::thread::hardware_concurrency());
    auto t0 = std::chrono::steady_clock::now();

    auto worker = [&]() {
        ImageBlock block(kBlockSize, kBlockSize, &sc.rfilter);
        Sampler sampler;
        RayCounter rc;
        Integrator integ(&sc, &rc);
        uint64_t cam = 0;
        while (blockGenerator.next(block)) {
            if (noriMode) sampler.rng.seed((uint64_t) block.offX, (uint64_t) block.offY);
            /* renderBlock, src/main.cpp:27-56 */
            block.clear();
            block.invalid = 0;
            for (int y = 0; y < block.sizeY; ++y) {
                for (int x = 0; x < block.sizeX; ++x) {
                    int px = x + block.offX, py = y + block.offY;
                    if (!noriMode && params->tile_mod > 1) {
                        uint32_t tile = (uint32_t) (py / NORI_TILE_SIZE) * tilesX + (uint32_t) (px / NORI_TILE_SIZE);
                        if (tile % params->tile_mod != params->tile_rem) continue;
                    }
                    for (uint32_t i = 0; i < params->spp_count; ++i) {
                        if (!noriMode)
                            sampler.rng.seed((uint64_t) py * (uint64_t) W + (uint64_t) px,
                                             (uint64_t) (params->spp_begin + i));
                        Vec2 j = sampler.next2D();
                        Vec2 pixelSample((float) px + j.x, (float) py + j.y);
                        Vec2 apertureSample = sampler.next2D();
                        (void) apertureSample;
                        Ray ray;
                        sc.camera.sampleRay(ray, pixelSample);
                        Color3 value = integ.Li(sampler, ray);
                        block.put(pixelSample, value);
                        ++cam;
                    }
                }
            }
            result.put(block);
        }
        nCamera += cam; nClosest += rc.closest; nShadow += rc.shadow;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();

    for (size_t i = 0; i < result.px.size(); ++i) rgbw[i] += result.px[i];
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n_camera_samples = nCamera; stats->n_closest_rays = nClosest; stats->n_shadow_rays = nShadow;
        stats->n_node_tests = sc.accel.nodeTests; stats->n_tri_tests = sc.accel.triTests;
        stats->n_invalid = result.invalid;
        stats->kernel_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    }
    sc.accel.countTests = false;
    return NORI_OK;
}

/* src/block.cpp:45-51 + color.h:100-105 */
int oracle_develop(const oracle_ctx *ctx, const float *rgbw, float *rgb) {
    const Scene &sc = *ctx->scene;
    int W = sc.camera.width, H = sc.camera.height, b = oracle_border_size(ctx);
    int cols = W + 2 * b;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *p = rgbw + ((size_t) (y + b) * cols + (x + b)) * 4;
            float *o = rgb + ((size_t) y * W + x) * 3;
            if (p[3] != 0) { o[0] = p[0] / p[3]; o[1] = p[1] / p[3]; o[2] = p[2] / p[3]; }
            else { o[0] = o[1] = o[2] = 0.0f; }
        }
    return NORI_OK;
}

} // extern "C"
