/*
 * rt_sampling.h -- pcg32, warps, Fresnel and the four BSDFs, per lane.
 *
 * Reference interfaces: Independent/pcg32 (src/independent.cpp:36-55,
 * ext/pcg32 -- un-vendored, PCG-XSH-RR 64/32), Warp::* (include/nori/warp.h:18-57),
 * fresnel (src/common.cpp:259-288), BSDF::sample/eval/pdf (include/nori/bsdf.h:59-87),
 * Diffuse (src/diffuse.cpp:23-71), Mirror (src/mirror.cpp:17-43), Dielectric and
 * Microfacet parameter sets (src/dielectric.cpp:17-23, src/microfacet.cpp:17-36).
 */
#pragma once
#include "rt_math.h"
#include "rt_types.h"

namespace nrt {

/* ------------------------------------------------------------------ pcg32 */
struct Rng {
    uint64_t state, inc;
};
NORI_HD uint32_t rng_next_uint(Rng &r) {
    uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
NORI_HD void rng_seed(Rng &r, uint64_t initstate, uint64_t initseq) {
    r.state = 0u;
    r.inc = (initseq << 1u) | 1u;
    rng_next_uint(r);
    r.state += initstate;
    rng_next_uint(r);
}
/* pcg32::advance(delta): the LCG state `delta` draws further, in O(log delta) (Brown, "Random number
   generation with arbitrary strides"): state' = a^delta * state + c (a^delta - 1) / (a - 1) */
NORI_HD void rng_advance(Rng &r, uint64_t delta) {
    uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = r.inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1u) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1u) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1u;
    }
    r.state = acc_mult * r.state + acc_plus;
}
/* Independent::next1D -> pcg32::nextFloat: [0,1) from the top 23 bits */
NORI_HD float rng_next_float(Rng &r) {
    return u2f((rng_next_uint(r) >> 9) | 0x3f800000u) - 1.0f;
}
NORI_HD f2 rng_next_2d(Rng &r) {
    float a = rng_next_float(r);
    float b = rng_next_float(r);
    return mk2(a, b);
}

/* ------------------------------------------------------------------ warps */
NORI_HD float warp_tent_1d(float xi) {
    return xi < 0.5f ? exact_sqrt(2.0f * xi) - 1.0f : 1.0f - exact_sqrt(2.0f - 2.0f * xi);
}
NORI_HD f2 square_to_tent(f2 s) { return mk2(warp_tent_1d(s.x), warp_tent_1d(s.y)); }
NORI_HD float square_to_tent_pdf(f2 p) {
    float ax = fabsf(p.x), ay = fabsf(p.y);
    if (ax > 1.0f || ay > 1.0f) return 0.0f;
    return (1.0f - ax) * (1.0f - ay);
}
NORI_HD f2 square_to_uniform_disk(f2 s) {
    float r = exact_sqrt(s.x);
    float sp, cp;
    det_sincosf(2.0f * kPi * s.y, &sp, &cp);
    return mk2(r * cp, r * sp);
}
NORI_HD float square_to_uniform_disk_pdf(f2 p) {
    return (p.x * p.x + p.y * p.y <= 1.0f) ? kInvPi : 0.0f;
}
NORI_HD f3 square_to_uniform_sphere(f2 s) {
    float z = 1.0f - 2.0f * s.x;
    float r = exact_sqrt(fmaxf(0.0f, 1.0f - z * z));
    float sp, cp;
    det_sincosf(2.0f * kPi * s.y, &sp, &cp);
    return mk3(r * cp, r * sp, z);
}
NORI_HD f3 square_to_uniform_hemisphere(f2 s) {
    float z = s.x;
    float r = exact_sqrt(fmaxf(0.0f, 1.0f - z * z));
    float sp, cp;
    det_sincosf(2.0f * kPi * s.y, &sp, &cp);
    return mk3(r * cp, r * sp, z);
}
NORI_HD f3 square_to_cosine_hemisphere(f2 s) {
    f2 d = square_to_uniform_disk(s);
    float z = exact_sqrt(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return mk3(d.x, d.y, z);
}
NORI_HD float square_to_cosine_hemisphere_pdf(f3 v) { return v.z > 0.0f ? v.z * kInvPi : 0.0f; }
NORI_HD f3 square_to_beckmann(f2 s, float alpha) {
    float sp, cp;
    det_sincosf(2.0f * kPi * s.x, &sp, &cp);
    float tan2 = -alpha * alpha * det_logf(1.0f - s.y);
    float cosTheta = exact_rcp(exact_sqrt(1.0f + tan2));
    float sinTheta = exact_sqrt(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    return mk3(sinTheta * cp, sinTheta * sp, cosTheta);
}
NORI_HD float square_to_beckmann_pdf(f3 m, float alpha) {
    if (m.z <= 0.0f) return 0.0f;
    float cos2 = m.z * m.z;
    float tan2 = exact_div(1.0f - cos2, cos2);
    float a2 = alpha * alpha;
    return det_expf(exact_div(-tan2, a2)) / (kPi * a2 * cos2 * m.z);      /* plain division: the numerator reaches the denormals */
}

NORI_HD f3 warp_dispatch(int warp, float param, f2 s) {
    switch (warp) {
    case 0: return mk3(s.x, s.y, 0.0f);
    case 1: { f2 p = square_to_tent(s); return mk3(p.x, p.y, 0.0f); }
    case 2: { f2 p = square_to_uniform_disk(s); return mk3(p.x, p.y, 0.0f); }
    case 3: return square_to_uniform_sphere(s);
    case 4: return square_to_uniform_hemisphere(s);
    case 5: return square_to_cosine_hemisphere(s);
    default: return square_to_beckmann(s, param);
    }
}
NORI_HD float warp_pdf_dispatch(int warp, float param, f3 v) {
    switch (warp) {
    case 0: return (v.x >= 0.0f && v.x <= 1.0f && v.y >= 0.0f && v.y <= 1.0f) ? 1.0f : 0.0f;
    case 1: return square_to_tent_pdf(mk2(v.x, v.y));
    case 2: return square_to_uniform_disk_pdf(mk2(v.x, v.y));
    case 3: return kInvFourPi;
    case 4: return v.z >= 0.0f ? kInvTwoPi : 0.0f;
    case 5: return square_to_cosine_hemisphere_pdf(v);
    default: return square_to_beckmann_pdf(v, param);
    }
}

/* src/common.cpp:259-288 */
NORI_HD float fresnel(float cosThetaI, float extIOR, float intIOR) {
    float etaI = extIOR, etaT = intIOR;
    if (extIOR == intIOR) return 0.0f;
    if (cosThetaI < 0.0f) {
        float tmp = etaI; etaI = etaT; etaT = tmp;
        cosThetaI = -cosThetaI;
    }
    float eta = exact_div(etaI, etaT), sinThetaTSqr = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSqr > 1.0f) return 1.0f;
    float cosThetaT = exact_sqrt(1.0f - sinThetaTSqr);
    float Rs = exact_div(etaI * cosThetaI - etaT * cosThetaT, etaI * cosThetaI + etaT * cosThetaT);
    float Rp = exact_div(etaT * cosThetaI - etaI * cosThetaT, etaT * cosThetaI + etaI * cosThetaT);
    return (Rs * Rs + Rp * Rp) / 2.0f;
}

/* ------------------------------------------------------------------ BSDFs */
struct Bsdf {
    int32_t type;     /* nori_bsdf_type: 0 diffuse 1 mirror 2 dielectric 3 microfacet */
    f3 albedo;
    float alpha, int_ior, ext_ior, ks;
};
NORI_HD Bsdf bsdf_from_mesh(const MeshRec &m) {
    Bsdf b;
    b.type = m.bsdf_type;
    b.albedo = mk3(m.albedo[0], m.albedo[1], m.albedo[2]);
    b.alpha = m.alpha; b.int_ior = m.int_ior; b.ext_ior = m.ext_ior; b.ks = m.ks;
    return b;
}
NORI_HD bool bsdf_is_diffuse(int32_t type) { return type == 0 || type == 3; }

NORI_HD float frame_tan_theta(f3 v) {   /* include/nori/frame.h:68-73 */
    float temp = 1.0f - v.z * v.z;
    if (temp <= 0.0f) return 0.0f;
    return exact_div(exact_sqrt(temp), v.z);
}

/* Beckmann shadowing-masking, rational approximation */
NORI_HD float microfacet_g1(f3 wv, f3 wh, float alpha) {
    if (dot(wv, wh) / wv.z <= 0.0f) return 0.0f;
    float tanTheta = frame_tan_theta(wv);
    if (tanTheta == 0.0f) return 1.0f;
    float b = exact_rcp(alpha * tanTheta);
    if (b >= 1.6f) return 1.0f;
    float b2 = b * b;
    return exact_div(3.535f * b + 2.181f * b2, 1.0f + 2.276f * b + 2.577f * b2);
}

/* BSDF::eval for measure == ESolidAngle */
NORI_HD f3 bsdf_eval(const Bsdf &b, f3 wi, f3 wo) {
    if (b.type == 0) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return mk3(0.0f);
        return b.albedo * kInvPi;
    }
    if (b.type == 3) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return mk3(0.0f);
        f3 wh = normalized(wi + wo);
        float D = square_to_beckmann_pdf(wh, b.alpha);
        float F = fresnel(dot(wh, wi), b.ext_ior, b.int_ior);
        float G = microfacet_g1(wi, wh, b.alpha) * microfacet_g1(wo, wh, b.alpha);
        float spec = b.ks * D * F * G / (4.0f * wi.z * wo.z * wh.z);
        return b.albedo * kInvPi + mk3(spec);
    }
    return mk3(0.0f);
}

NORI_HD float bsdf_pdf(const Bsdf &b, f3 wi, f3 wo) {
    if (b.type == 0) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        return kInvPi * wo.z;
    }
    if (b.type == 3) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        f3 wh = normalized(wi + wo);
        float D = square_to_beckmann_pdf(wh, b.alpha);
        float Jh = exact_rcp(4.0f * dot(wh, wo));
        return b.ks * D * Jh + (1.0f - b.ks) * wo.z * kInvPi;
    }
    return 0.0f;
}

/* BSDF::sample: returns the importance weight, fills wo / eta / measure */
NORI_HD f3 bsdf_sample(const Bsdf &b, f3 wi, f2 s, f3 &wo, float &eta, int &measure) {
    wo = mk3(0.0f); eta = 1.0f; measure = 0;
    switch (b.type) {
    case 0:
        if (wi.z <= 0.0f) return mk3(0.0f);
        measure = 1;
        wo = square_to_cosine_hemisphere(s);
        return b.albedo;
    case 1:
        if (wi.z <= 0.0f) return mk3(0.0f);
        wo = mk3(-wi.x, -wi.y, wi.z);
        measure = 2;
        return mk3(1.0f);
    case 2: {
        float cosThetaI = wi.z;
        float F = fresnel(cosThetaI, b.ext_ior, b.int_ior);
        measure = 2;
        if (s.x < F) {
            wo = mk3(-wi.x, -wi.y, wi.z);
            return mk3(1.0f);
        }
        bool entering = cosThetaI > 0.0f;
        float etaI = entering ? b.ext_ior : b.int_ior;
        float etaT = entering ? b.int_ior : b.ext_ior;
        float e = exact_div(etaI, etaT);
        float sinThetaTSqr = e * e * (1.0f - cosThetaI * cosThetaI);
        float cosThetaT = exact_sqrt(fmaxf(0.0f, 1.0f - sinThetaTSqr));
        wo = mk3(-e * wi.x, -e * wi.y, entering ? -cosThetaT : cosThetaT);
        eta = exact_div(etaT, etaI);
        return mk3(1.0f);
    }
    default: {
        if (wi.z <= 0.0f) return mk3(0.0f);
        measure = 1;
        if (s.x < b.ks) {
            f3 n = square_to_beckmann(mk2(exact_div(s.x, b.ks), s.y), b.alpha);
            wo = 2.0f * dot(wi, n) * n - wi;
        } else {
            wo = square_to_cosine_hemisphere(mk2(exact_div(s.x - b.ks, 1.0f - b.ks), s.y));
        }
        if (wo.z <= 0.0f) return mk3(0.0f);
        float p = bsdf_pdf(b, wi, wo);
        if (!(p > 0.0f)) return mk3(0.0f);
        return div_ieee(bsdf_eval(b, wi, wo) * wo.z, p);      /* plain division: with kd = 0 the numerator reaches the denormals */
    }
    }
}

} // namespace nrt
