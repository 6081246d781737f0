/*
 * rt_path.h -- Integrator::Li as a per-lane state machine.
 *
 * The reference declares Li (include/nori/integrator.h:42) but ships no
 * integrator; behaviour follows the spec pinned by its test scenes
 * (scenes/pa4/tests/*.xml, scenes/pa5/tests/test-{furnace,direct}.xml; DESIGN.md
 * §Integrators).  A recursive / loop-with-two-traces CPU formulation becomes a
 * machine that issues exactly ONE ray query per step -- closest-hit or shadow
 * -- so that all lanes of a wave re-converge at the traversal loop:
 *
 *     NEW --camera ray--> CLOSEST --shade--> [SHADOW -->] CLOSEST ... --> done
 *
 * At a path vertex every random number of that vertex (Russian roulette,
 * emitter sample, BSDF sample) is drawn and both the shadow ray and the
 * continuation ray are prepared before either is traced; the draw order is
 * RR (1), emitter (1+1+2), BSDF (2) -- the order the CPU oracle consumes them.
 */
#pragma once
#include "rt_sampling.h"
#include "rt_trace.h"

namespace nrt {

enum { PH_NEW = 0, PH_CLOSEST = 1, PH_SHADOW = 2 };
enum { INT_NORMALS = 0, INT_AO = 1, INT_SIMPLE = 2, INT_WHITTED = 3, INT_MATS = 4, INT_EMS = 5, INT_MIS = 6 };

struct PathState {
    Rng rng;
    RayIn ray;          /* the query to issue next */
    f3 T, L;
    f3 Ld;              /* added to L if the pending shadow ray is unoccluded */
    f3 cont_d;          /* continuation direction (origin = ray.o) */
    float eta;
    float pdf_mat;      /* solid-angle pdf of the last BSDF sample (MIS) */
    int32_t prev_measure;
    int32_t depth;
    int32_t phase;
    int32_t end_after_shadow;
};

NORI_HD void path_begin(PathState &st, const RayIn &camRay) {
    st.ray = camRay;
    st.T = mk3(1.0f); st.L = mk3(0.0f); st.Ld = mk3(0.0f); st.cont_d = mk3(0.0f);
    st.eta = 1.0f; st.pdf_mat = 0.0f; st.prev_measure = 2; st.depth = 0;
    st.phase = PH_CLOSEST; st.end_after_shadow = 0;
}

/* Where the small per-scene tables -- mesh records, the emitter list, the emitters' triangle CDFs -- are read from.  This one
   goes through DevScene's own pointers (global memory; every engine and twin).  wf_shade keeps a copy of the tables in LDS and
   reads it with LDS instructions (shade_tables.h, LdsTables): a load through a generic pointer is a FLAT instruction, which
   counts as a vector-memory AND an LDS access and makes the wave wait for everything it has in flight of either kind. */
struct SceneTables {
    const DevScene *sc;
    NORI_HD MeshRec mesh(uint32_t i) const { return sc->meshes[i]; }
    NORI_HD uint32_t emitter(uint32_t i) const { return sc->emitters[i]; }
    NORI_HD float cdf(uint32_t i) const { return sc->emitter_cdf[i]; }
};

/* DiscretePDF::sample, include/nori/dpdf.h:106-111 (std::lower_bound); the CDF is entries [first, first + n] of the table */
template <class Tab>
NORI_HD uint32_t cdf_sample(const Tab &tab, uint32_t first, uint32_t n, float v) {
    uint32_t lo = 0, len = n + 1;
    while (len > 0) {
        uint32_t half = len >> 1, mid = lo + half;
        if (tab.cdf(first + mid) < v) { lo = mid + 1; len -= half + 1; }
        else len = half;
    }
    uint32_t index = lo > 0 ? lo - 1 : 0;
    return index < n - 1 ? index : n - 1;
}

/* the point of an emitter triangle that the sample xi selects, and the normal there: uniform barycentrics
   (alpha = 1 - sqrt(1 - xi1), beta = xi2 sqrt(1 - xi1)); interpolated vertex normal if the mesh has normals, else geometric */
NORI_HD void emitter_point(f2 xi, f3 p0, f3 p1, f3 p2, bool has_normals, f3 n0, f3 n1, f3 n2, f3 &p, f3 &n) {
    const float su = exact_sqrt(1.0f - xi.x);
    const float alpha = 1.0f - su, beta = xi.y * su;
    const float gamma = 1.0f - alpha - beta;
    p = (alpha * p0 + beta * p1) + gamma * p2;
    if (has_normals) n = normalized((alpha * n0 + beta * n1) + gamma * n2);
    else n = normalized(cross(p1 - p0, p2 - p0));
}

/* the emitter triangle the sample (xiE, xiT) selects: uniform emitter, triangle by area (DiscretePDF::sample).
   Returns the emitter's mesh index; tri = triangle within that mesh. */
template <class Tab>
NORI_HD uint32_t emitter_pick(const Tab &tab, uint32_t nE, float xiE, float xiT, MeshRec &m, uint32_t &tri) {
    uint32_t ei = (uint32_t) (xiE * (float) nE);
    if (ei > nE - 1) ei = nE - 1;
    const uint32_t mi = tab.emitter(ei);
    m = tab.mesh(mi);
    tri = cdf_sample(tab, m.cdf_offset, m.n_triangles, xiT);
    return mi;
}

struct NeeResult {
    f3 Ld;          /* f * Le * cosX * cosY / (dist^2 * pdfA) */
    f3 dir;
    float maxt;
    float pdf_em, pdf_bsdf;
};

/* Emitter sampling at surface `s`: uniform emitter, triangle by area, uniform
 * barycentrics (alpha = 1 - sqrt(1 - xi1), beta = xi2 sqrt(1 - xi1)).
 * Always consumes 4 random numbers.  Returns true if a shadow ray is needed. */
template <class Tab>
NORI_HD bool sample_direct(const DevScene &sc, const Tab &tab, Rng &rng, const Surface &s, const Frame &fr,
                           const Bsdf &bsdf, f3 wi, NeeResult &out) {
    const float xiE = rng_next_float(rng);
    const float xiT = rng_next_float(rng);
    const f2 xi = rng_next_2d(rng);
    const uint32_t nE = sc.n_emitters;
    if (nE == 0) return false;
    MeshRec m; uint32_t tri;
    (void) emitter_pick(tab, nE, xiE, xiT, m, tri);
    const float pdfPick = exact_rcp((float) nE);
    const f4 *rec = sc.shade_tris + (size_t) (m.tri_offset + tri) * kShadeQuads;
    const bool hn = (m.flags & kMeshHasNormals) != 0u;
    const f3 z = mk3(0.0f);
    f3 p, n;
    emitter_point(xi, xyz(rec[0]), xyz(rec[1]), xyz(rec[2]), hn, hn ? xyz(rec[3]) : z, hn ? xyz(rec[4]) : z, hn ? xyz(rec[5]) : z, p, n);
    const f3 dvec = p - s.p;
    const float dist2 = dot(dvec, dvec);
    const float dist = exact_sqrt(dist2);
    const f3 dir = dvec / dist;
    const float cosY = dot(n, -dir);
    if (!(cosY > 0.0f)) return false;
    const f3 wo = to_local(fr, dir);
    const f3 f = bsdf_eval(bsdf, wi, wo);
    if (is_zero(f)) return false;
    const float pdfA = m.inv_area * pdfPick;
    out.pdf_em = exact_div(pdfA * dist2, cosY);
    out.pdf_bsdf = bsdf_pdf(bsdf, wi, wo);
    const float cosX = wo.z;
    const f3 rad = mk3(m.radiance[0], m.radiance[1], m.radiance[2]);
    out.Ld = f * rad * exact_div(cosX * cosY, dist2 * pdfA);
    out.dir = dir;
    out.maxt = dist - kEpsilon;
    return true;
}

/* Consume the result of a closest-hit query.  Returns true when the path is
 * complete (st.L is the radiance estimate); otherwise st.ray / st.phase name
 * the next query. */
/* src/accel.cpp:45-96 (surface_fill, rt_trace.h) from the quads of the triangle's shading record: barycentric position, and the
   interpolated vertex normal if the mesh has normals, else the geometric one */
NORI_HD void surface_from_record(bool has_normals, float u, float v, f3 p0, f3 p1, f3 p2, f3 n0, f3 n1, f3 n2, Surface &s) {
    const float b0 = 1.0f - (u + v), b1 = u, b2 = v;
    s.p = (b0 * p0 + b1 * p1) + b2 * p2;
    if (has_normals) s.ns = normalized((b0 * n0 + b1 * n1) + b2 * n2);
    else s.ns = normalized(cross(p1 - p0, p2 - p0));
}

/* MATSET: the BSDF types the scene contains, bit t = nori_bsdf_type t (DevScene::bsdf_mask).  A kernel instantiated for a
   subset never compiles the other BSDFs' sample / eval / pdf (include/nori/bsdf.h:59-87): wf_shade of an all-diffuse scene
   (src/diffuse.cpp:23-71) is a smaller kernel with more waves per SIMD.  kAnyBsdf: every type. */
constexpr int kAnyBsdf = 0xf;
template <int MATSET> NORI_HD int32_t bsdf_type_in_set(int32_t type) {
    if (MATSET == 1) return 0;                                      /* diffuse only: a constant */
    if (MATSET == 8) return 3;
    if (!(MATSET & 8) && type == 3) type = 0;                         /* (never taken: the type is in the set; tells the compiler) */
    if (!(MATSET & 6) && type != 0 && type != 3) type = 0;
    return type;
}
template <int INTEG, class Tab, int MATSET = kAnyBsdf>
NORI_HD bool path_on_closest(const DevScene &sc, const Tab &tab, PathState &st, const Hit &hit, bool found, const f3 d) {
    if (!found) return true;
    const MeshRec m = tab.mesh(hit.mesh);
    Surface sf;
    {
        const f4 *rec = sc.shade_tris + (size_t) hit.tri * kShadeQuads;
        const bool hn = (m.flags & kMeshHasNormals) != 0u;
        const f3 z = mk3(0.0f);
        surface_from_record(hn, hit.u, hit.v, xyz(rec[0]), xyz(rec[1]), xyz(rec[2]), hn ? xyz(rec[3]) : z, hn ? xyz(rec[4]) : z, hn ? xyz(rec[5]) : z, sf);
    }

    if (INTEG == INT_NORMALS) {
        st.L = mk3(fabsf(sf.ns.x), fabsf(sf.ns.y), fabsf(sf.ns.z));
        return true;
    }
    const Frame fr = make_frame(sf.ns);

    if (INTEG == INT_AO) {
        const f3 wo = square_to_cosine_hemisphere(rng_next_2d(st.rng));
        st.Ld = mk3(1.0f);
        st.ray.o = sf.p; st.ray.d = to_world(fr, wo); st.ray.mint = kEpsilon; st.ray.maxt = kInf;
        st.phase = PH_SHADOW; st.end_after_shadow = 1;
        return false;
    }
    if (INTEG == INT_SIMPLE) {
        const f3 lp = mk3(sc.integrator.position[0], sc.integrator.position[1], sc.integrator.position[2]);
        const f3 energy = mk3(sc.integrator.energy[0], sc.integrator.energy[1], sc.integrator.energy[2]);
        const f3 dvec = lp - sf.p;
        const float dist2 = dot(dvec, dvec), dist = exact_sqrt(dist2);
        const f3 dir = dvec / dist;
        const float cosTheta = dot(sf.ns, dir);
        if (!(cosTheta > 0.0f)) return true;
        st.Ld = energy * exact_div((kInvPi * kInvPi * 0.25f) * cosTheta, dist2);
        st.ray.o = sf.p; st.ray.d = dir; st.ray.mint = kEpsilon; st.ray.maxt = dist;
        st.phase = PH_SHADOW; st.end_after_shadow = 1;
        return false;
    }

    const bool emitter = (m.flags & kMeshEmitter) != 0;
    const f3 rad = mk3(m.radiance[0], m.radiance[1], m.radiance[2]);
    Bsdf bsdf = bsdf_from_mesh(m);
    bsdf.type = bsdf_type_in_set<MATSET>(bsdf.type);
    const f3 wi = to_local(fr, -d);

    if (INTEG == INT_WHITTED) {
        if (emitter && dot(sf.ns, -d) > 0.0f) st.L = st.L + st.T * rad;
        if (bsdf_is_diffuse(bsdf.type)) {
            NeeResult nee;
            if (!sample_direct(sc, tab, st.rng, sf, fr, bsdf, wi, nee)) return true;
            st.Ld = st.T * nee.Ld;
            st.ray.o = sf.p; st.ray.d = nee.dir; st.ray.mint = kEpsilon; st.ray.maxt = nee.maxt;
            st.phase = PH_SHADOW; st.end_after_shadow = 1;
            return false;
        }
        if (!(rng_next_float(st.rng) < 0.95f)) return true;
        f3 wo; float e; int measure;
        const f3 f = bsdf_sample(bsdf, wi, rng_next_2d(st.rng), wo, e, measure);
        if (is_zero(f)) return true;
        st.T = st.T * (f * (1.0f / 0.95f));
        st.ray.o = sf.p; st.ray.d = to_world(fr, wo); st.ray.mint = kEpsilon; st.ray.maxt = kInf;
        st.phase = PH_CLOSEST;
        return false;
    }

    /* path_mats / path_ems / path_mis */
    constexpr bool EMS = (INTEG == INT_EMS || INTEG == INT_MIS);
    constexpr bool MIS = (INTEG == INT_MIS);

    /* weight of emission reached by BSDF sampling */
    float wMat = 1.0f;
    if (EMS && st.prev_measure != 2) {
        if (!MIS) {
            wMat = 0.0f;
        } else if (emitter) {
            float pdfEm = 0.0f;
            const float cosY = dot(sf.ns, -d);
            if (cosY > 0.0f) {
                const float pdfA = exact_div(m.inv_area, (float) sc.n_emitters);
                pdfEm = exact_div(pdfA * hit.t * hit.t, cosY);
            }
            wMat = (st.pdf_mat + pdfEm) > 0.0f ? exact_div(st.pdf_mat, st.pdf_mat + pdfEm) : 0.0f;
        }
    }
    if (emitter && wMat > 0.0f) {
        const f3 le = dot(sf.ns, -d) > 0.0f ? rad : mk3(0.0f);
        st.L = st.L + st.T * le * wMat;
    }
    if (st.depth >= 3) {
        const float p = fminf(max3(st.T) * st.eta * st.eta, 0.99f);
        if (!(rng_next_float(st.rng) < p)) return true;
        st.T = st.T / p;
    }
    bool needShadow = false;
    NeeResult nee;
    if (EMS && bsdf_is_diffuse(bsdf.type)) {
        needShadow = sample_direct(sc, tab, st.rng, sf, fr, bsdf, wi, nee);
        if (needShadow) {
            float w = 1.0f;
            if (MIS) w = (nee.pdf_em + nee.pdf_bsdf) > 0.0f ? exact_div(nee.pdf_em, nee.pdf_em + nee.pdf_bsdf) : 0.0f;
            st.Ld = st.T * nee.Ld * w;
        }
    }
    f3 wo; float e; int measure;
    const f3 f = bsdf_sample(bsdf, wi, rng_next_2d(st.rng), wo, e, measure);
    const bool dead = is_zero(f);
    if (!dead) {
        st.T = st.T * f;
        st.eta = st.eta * e;
        st.cont_d = to_world(fr, wo);
        st.prev_measure = measure;
        st.pdf_mat = (MIS && measure != 2) ? bsdf_pdf(bsdf, wi, wo) : 0.0f;
        st.depth++;
    }
    st.ray.o = sf.p; st.ray.mint = kEpsilon;
    if (needShadow) {
        st.ray.d = nee.dir; st.ray.maxt = nee.maxt;
        st.phase = PH_SHADOW; st.end_after_shadow = dead ? 1 : 0;
        return false;
    }
    if (dead) return true;
    st.ray.d = st.cont_d; st.ray.maxt = kInf;
    st.phase = PH_CLOSEST;
    return false;
}

/* (every engine but wf_shade: the tables behind DevScene's pointers) */
template <int INTEG>
NORI_HD bool path_on_closest(const DevScene &sc, PathState &st, const Hit &hit, bool found, const f3 d) {
    const SceneTables tab = {&sc};
    return path_on_closest<INTEG>(sc, tab, st, hit, found, d);
}

/* Consume the result of a shadow query (`o` = origin of the shadow ray, which
 * is also the origin of the continuation). */
NORI_HD bool path_on_shadow(PathState &st, bool occluded, const f3 o) {
    if (!occluded) st.L = st.L + st.Ld;
    if (st.end_after_shadow) return true;
    st.ray.o = o; st.ray.d = st.cont_d; st.ray.mint = kEpsilon; st.ray.maxt = kInf;
    st.phase = PH_CLOSEST;
    return false;
}

} // namespace nrt
