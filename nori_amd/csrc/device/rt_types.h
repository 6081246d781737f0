/*
 * rt_types.h -- POD value types and device scene layout of the gfx950 path
 * tracer.  Per-lane code in rt_*.h is plain scalar C++ annotated NORI_HD; the
 * kernels that drive it (LDS stacks, tile accumulation, wave reductions) are
 * HIP-only and live in nori_hip.hip.
 *
 * All arithmetic is IEEE float32 without FMA contraction (build flag
 * -ffp-contract=off) in the reference's evaluation order, so that triangle
 * hits are bit-identical to Mesh::rayIntersect (src/mesh.cpp:39-76).
 *   dot(a,b)  = a.x*b.x + (a.y*b.y + a.z*b.z)      (Eigen's unrolled redux)
 *   cross     = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 */
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NORI_HD __host__ __device__ __forceinline__
#else
#define NORI_HD inline
#endif

namespace nrt {

constexpr float kEpsilon = 1e-4f;              /* include/nori/common.h:38 */
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr float kInvFourPi = 0.07957747154594766788f;
constexpr float kInf = __builtin_huge_valf();
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr int kFilterRes = 32;                 /* include/nori/rfilter.h:12 */
constexpr int kTile = 16;                      /* NORI_TILE_SIZE */

struct f2 { float x, y; };
struct P3 { float x, y, z; };      /* three floats as they lie in memory, 12 B apart (f3 is the arithmetic type) */
struct f3 { float x, y, z; };
struct alignas(16) f4 { float x, y, z, w; };

NORI_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
NORI_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

/* ---- IEEE-correct reciprocal, quotient and square root in fewer instructions than the compiler's expansions.
 *
 * hipcc expands a / b into v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup plus VCC hazard nops (~13 issue slots,
 * ~47 cycles per wave64) and sqrtf(x) into 16 instructions + 4 nops: both carry scaling for operands near the ends of the
 * exponent range and fix-ups for inf / NaN / 0.  A path vertex of the render loop divides ~22 times and takes ~6 square
 * roots: a third of wf_shade's instructions.  The forms below give the SAME bits as the IEEE operations on the domain
 * stated with each (checked on gfx950: tools/ubench_recip.hip exhaustively, tools/ubench_divsqrt.hip exhaustively for the
 * square root and on 2^36 pairs -- exponents of both operands within +-60, hard mantissa patterns included -- for the
 * quotient; results in profiles/r2_exact_rcp.txt and profiles/r2_exact_div_sqrt.txt), and the render path's operands
 * -- lengths, cosines, pdfs, areas of scenes whose unit is not 1e-15 of their size -- lie well inside it.  The CPU twins
 * (emulation harness, host code) use the plain operators.
 *
 *   exact_rcp(x)    = 1.0f / x   for 2^-126 <= |x| < 2^126: r0 = v_rcp_f32(x) (1 ulp), one Newton step with two fma.
 *   exact_div(a, b) = a / b      y = exact_rcp(b) is the correctly rounded reciprocal, q = a y is within 2 ulp, the
 *                                residual r = a - b q is exact (fma), and RN(q + r y) is the correctly rounded quotient
 *                                (Markstein 1990).  Domain: b as for exact_rcp; a = 0, or a and a / b normal numbers with
 *                                |a| >= 2^-100 (so that r does not underflow).  Three numerators over one denominator
 *                                (a vector / its length) share y: 12 instructions instead of 39.
 *   exact_sqrt(x)   = sqrtf(x)   s = v_sqrt_f32(x) (1 ulp); the residuals x - (s -+ 1 ulp) s, exact in one fma each,
 *                                say whether a neighbour is the correctly rounded root (the compiler's own correction
 *                                step without its scaling).  Domain: x = 0, +inf, or x >= 2^-104 (below that the residuals
 *                                underflow; the exhaustive run finds its first mismatch at 2^-105). */
/* The sequences carry no test and no fallback (measured: one compare + branch per division costs wf_shade 0.9 ms and the
 * first wf_extend 1 ms of an 80 ms frame).  Outside the stated domains the last bit is unspecified, and a zero, infinite
 * or denormal divisor or an overflowing quotient gives NaN where IEEE gives inf or 0 -- in this renderer either value
 * ends as a sample the film drops (src/block.cpp:63-67), e.g. a light sample at distance 0.  The traversal uses
 * exact_rcp_raw where an operand outside the domain is expected and harmless (rt_trace.h: slab_rcp clamps what comes back,
 * a triangle's 1 / det is only used after |det| >= 1e-8 was checked).
 * Builds with -DNORI_COUNT_EXCURSIONS (libnori_hip_count.so) count every operand of the shading code outside the verified
 * domains; the GPU suite renders the golden scenes through that build and asserts zeros
 * (test_shading_arithmetic_stays_inside_its_verified_domain). */
#if defined(NORI_COUNT_EXCURSIONS) && defined(__HIP__)
static __device__ unsigned long long g_nori_excursions[4];      /* rcp, div, sqrt operands outside the domain; results that are NaN or infinite (per translation unit) */
#endif
#if defined(NORI_COUNT_EXCURSIONS) && defined(__HIP_DEVICE_COMPILE__)
#define NORI_EXCURSION(k, cond) do { if (cond) atomicAdd(&g_nori_excursions[k], 1ull); } while (0)
#else
#define NORI_EXCURSION(k, cond) ((void) 0)
#endif

NORI_HD float exact_rcp_raw(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
#else
    return 1.0f / x;
#endif
}
NORI_HD float exact_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    NORI_EXCURSION(0, !(fabsf(x) >= 1.17549435e-38f && fabsf(x) < 8.5070592e37f));
    const float r = exact_rcp_raw(x);
    NORI_EXCURSION(3, !(fabsf(r) < kInf));
    return r;
#else
    return 1.0f / x;
#endif
}
NORI_HD float exact_div_by(float a, float b, float rcp_b) {      /* rcp_b = exact_rcp_raw(b) or exact_rcp(b) */
#if defined(__HIP_DEVICE_COMPILE__)
    NORI_EXCURSION(1, !(fabsf(b) >= 1.17549435e-38f && fabsf(b) < 8.5070592e37f) ||
                      (a != 0.0f && !(fabsf(a) >= 7.8886091e-31f && fabsf(a * rcp_b) >= 1.17549435e-38f && fabsf(a * rcp_b) < kInf)));
    const float q = a * rcp_b;
    const float r = __builtin_fmaf(__builtin_fmaf(-b, q, a), rcp_b, q);
    NORI_EXCURSION(3, !(fabsf(r) < kInf));
    return r;
#else
    (void) rcp_b;
    return a / b;
#endif
}
NORI_HD float exact_div(float a, float b) { return exact_div_by(a, b, exact_rcp_raw(b)); }
NORI_HD float exact_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    NORI_EXCURSION(2, x > 0.0f && x < 4.9303807e-32f);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = u2f(f2u(s) - 1u), hi = u2f(f2u(s) + 1u);
    const float r_lo = __builtin_fmaf(-lo, s, x), r_hi = __builtin_fmaf(-hi, s, x);
    float r = r_lo <= 0.0f ? lo : s;
    r = r_hi > 0.0f ? hi : r;
    return r;
#else
    return sqrtf(x);
#endif
}

NORI_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
NORI_HD f3 mk3(float v) { return mk3(v, v, v); }
NORI_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
NORI_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
NORI_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
NORI_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
NORI_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
NORI_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
NORI_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
NORI_HD f3 operator/(f3 a, float s) { const float y = exact_rcp_raw(s); return mk3(exact_div_by(a.x, s, y), exact_div_by(a.y, s, y), exact_div_by(a.z, s, y)); }
NORI_HD f3 div_ieee(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }      /* the compiler's full-range division */
NORI_HD float dot(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
NORI_HD f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
NORI_HD f3 normalized(f3 a) {
    float z = dot(a, a);
    if (z > 0.0f) return a / exact_sqrt(z);
    return a;
}
NORI_HD float max3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
NORI_HD bool is_zero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
NORI_HD f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }

/* src/common.cpp:198-205 */
NORI_HD bool color_valid(f3 c) {
    return c.x >= 0.0f && c.y >= 0.0f && c.z >= 0.0f &&
           c.x < kInf && c.y < kInf && c.z < kInf;    /* NaN fails every compare */
}

/* src/common.cpp:248-257 */
NORI_HD void coordinate_system(f3 a, f3 &b, f3 &c) {
    if (fabsf(a.x) > fabsf(a.y)) {
        float invLen = exact_rcp(exact_sqrt(a.x * a.x + a.z * a.z));
        c = mk3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = exact_rcp(exact_sqrt(a.y * a.y + a.z * a.z));
        c = mk3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

/* include/nori/frame.h:20-50 */
struct Frame {
    f3 s, t, n;
};
NORI_HD Frame make_frame(f3 n) { Frame f; f.n = n; coordinate_system(n, f.s, f.t); return f; }
NORI_HD f3 to_local(const Frame &f, f3 v) { return mk3(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
NORI_HD f3 to_world(const Frame &f, f3 v) { return (f.s * v.x + f.t * v.y) + f.n * v.z; }

/* ------------------------------------------------------------ scene in HBM */

/* Padding of a triangle's box in the BVH builders.  The node test must never cull a triangle whose
 * Moeller-Trumbore test (src/mesh.cpp:39-76, float arithmetic) would accept the ray.  For a well-shaped
 * triangle the accepted region exceeds the exact triangle by rounding only, covered by `pad`
 * (kBoxPadRel = 2e-6 x scene diagonal).  The rounding error of det = e1 . (d x e2) is ~2^-23 |e1||e2|, so relative
 * to det it grows like 1 / sin(angle between the edges): a sliver's accepted region is wider by that
 * factor, and for (numerically) collinear vertices u, v, t are noise and the reference's linear scan
 * can report a hit for a ray that passes nowhere near the triangle -- even outside the scene's box.
 * The pad therefore scales with 1 / sin; below 1e-4 the triangle is `unbounded`: the builders keep such
 * triangles out of the spatial hierarchy and hang them under the root in a subtree whose boxes are
 * (-kBoxInf, kBoxInf)^3, so every ray tests them, exactly as the scan does.  Triangles with a
 * zero-length edge have det == 0 and are never hit. */
constexpr float kBoxInf = 3.0e38f;
/* pad of a well-shaped triangle's box, relative to the scene diagonal; NORI_HIP_BOX_PAD overrides (experiments) */
constexpr float kBoxPadRel = 2e-6f;
inline float box_pad_rel() {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (const char *e = std::getenv("NORI_HIP_BOX_PAD")) return (float) std::atof(e);
#endif
    return kBoxPadRel;
}

NORI_HD float tri_box_pad(f3 e1, f3 e2, float pad, bool &unbounded) {
    unbounded = false;
    const float l1 = dot(e1, e1), l2 = dot(e2, e2);
    if (!(l1 > 0.0f) || !(l2 > 0.0f)) return pad;
    const f3 c = cross(e1, e2);
    const float s2 = dot(c, c) / (l1 * l2);             /* sin^2 */
    if (!(s2 >= 1e-8f)) { unbounded = true; return pad; }
    return s2 >= 1.0f ? pad : pad / sqrtf(s2);
}

/* One BVH2 node = 64 B = 4 x dwordx4: both child boxes + both child links, so one fetch decides both children.
 * A child box is stored as CENTRE c and HALF-EXTENT h per axis: for a ray o + t d with r = 1 / d the slab interval of
 * an axis is  (c - o) r -+ h |r|  -- near and far come out of ONE multiply and two fused multiply-adds whose |r| is a free
 * source modifier, with no min / max to sort the two planes (on gfx950 v_min / v_max issue at half the rate of
 * v_fma; the min / max form cost 16 of them per node, rt_trace.h slab_two).
 *   q0 = (lc.x, lc.y, rc.x, rc.y)      the x / y centres pair up with (o.x, o.y) / (r.x, r.y): packed f32 subtract, multiply
 *   q1 = (lc.z, rc.z, lh.z, rh.z)
 *   q2 = (lh.x, lh.y, rh.x, rh.y)
 *   q3 = (bits left, bits right, 0, 0)
 * [c - h, c + h] contains the child's (padded) box: c = the rounded midpoint, h rounded up from max(hi - c, c - lo)
 * evaluated in binary64.  An unbounded box (numerically collinear triangles, tri_box_pad) is c = 0, h = kBoxHalfInf.
 * child link >= 0: inner node index; < 0: leaf, ~link = (first_tri << 3) | (count - 1)
 */
constexpr int kNodeQuads = 4;
constexpr float kBoxHalfInf = 5.76460752303423488e17f;      /* 2^59: with |r| <= 2^60 (slab_rcp) the products stay finite */

NORI_HD void box_centre_half(float mn, float mx, float &c, float &h) {
    if (!(fabsf(mn) < 1e37f) || !(fabsf(mx) < 1e37f)) { c = 0.0f; h = kBoxHalfInf; return; }
    c = 0.5f * mn + 0.5f * mx;
    const double lo = (double) c - (double) mn, hi = (double) mx - (double) c;
    const double hd = lo > hi ? lo : hi;
    h = (float) hd;
    if ((double) h < hd) h = u2f(f2u(h) + 1u);      /* next float up (h >= 0) */
    if (!(h >= 0.0f)) h = 0.0f;
}

NORI_HD void node_pack(const float lmn[3], const float lmx[3], const float rmn[3], const float rmx[3],
                       int32_t left, int32_t right, f4 q[4]) {
    float lc[3], lh[3], rc[3], rh[3];
    for (int a = 0; a < 3; ++a) { box_centre_half(lmn[a], lmx[a], lc[a], lh[a]); box_centre_half(rmn[a], rmx[a], rc[a], rh[a]); }
    q[0].x = lc[0]; q[0].y = lc[1]; q[0].z = rc[0]; q[0].w = rc[1];
    q[1].x = lc[2]; q[1].y = rc[2]; q[1].z = lh[2]; q[1].w = rh[2];
    q[2].x = lh[0]; q[2].y = lh[1]; q[2].z = rh[0]; q[2].w = rh[1];
    q[3].x = u2f((uint32_t) left); q[3].y = u2f((uint32_t) right); q[3].z = 0.0f; q[3].w = 0.0f;
}
constexpr int kMaxLeafTris = 8;      /* = 4 pairs */

/* WIDE node = BVH4 with quantised child boxes, also 64 B = 4 x dwordx4: the layout for scenes whose tree does not
 * fit L2 + Infinity Cache (traversal is bound by HBM bytes and dependent fetches there): half the
 * node visits of the BVH2 at the same bytes per visit.
 *   q0 = (origin.x, origin.y, origin.z, bits meta)   meta = ex | ey << 8 | ez << 16 | axis << 24 | kWideAllHit
 *                                                     scale of axis a = 2^(e_a - 128); axis = the node's widest axis
 *   q1 = (bits loX, bits loY, bits loZ, bits hiX)     one dword per plane set: child k's 8-bit coordinate in byte k
 *   q2 = (bits hiY, bits hiZ, 0, 0)
 *   q3 = (bits link0, link1, link2, link3)            child links as in the BVH2 node (>= 0 inner, < 0 leaf)
 * Child k covers [origin_a + lo_a[k] 2^(e_a-128), origin_a + hi_a[k] 2^(e_a-128)] on axis a, a superset of its true
 * (padded) box: origin = the node box's lower corner, lo rounded down, hi rounded up (rt_wide.h, wide_pack).
 * The children are stored in ascending order of their centre along `axis`: slot order = front-to-back order for a
 * ray that travels in +axis, back-to-front otherwise.  An unused slot has lo = 255, hi = 0 (never hit) and link
 * kWideEmpty = the leaf code of pair record 0, which wide trees reserve as an all-zero pair (never hit).
 * kWideAllHit: a child box is unbounded (the subtree of numerically collinear triangles, tri_box_pad): no box test,
 * every child is visited. */
constexpr uint32_t kWideAllHit = 1u << 26;
constexpr int32_t kWideEmpty = -1;           /* ~((0 << 3) | 0): first_pair 0, one pair */
constexpr float kWideInfinite = 1e37f;       /* |coordinate| beyond this: the box counts as unbounded */

/* Leaf triangles are stored in PAIRS, 96 B = 6 x dwordx4 per pair, de-indexed and pre-gathered in leaf
 * order, so that one leaf step is one contiguous read and tests two triangles (a, b) with packed-f32
 * math (rt_trace.h, tri_pair_test).  e1 = p1 - p0, e2 = p2 - p0: the IEEE subtraction mesh.cpp:43
 * performs per ray.
 *   q0 = (p0.x a, p0.x b, p0.y a, p0.y b)     q3 = (e2.x a, e2.x b, e2.y a, e2.y b)
 *   q1 = (p0.z a, p0.z b, e1.x a, e1.x b)     q4 = (e2.z a, e2.z b, bits global_tri a, bits global_tri b)
 *   q2 = (e1.y a, e1.y b, e1.z a, e1.z b)     q5 = (bits mesh a, bits mesh b, 0, 0)
 * A leaf with an odd number of triangles pads its last pair with an all-zero triangle (det = 0: never
 * hit) whose global id is kNoHit.  Leaf link: ~link = (first_pair << 3) | (n_pairs - 1).
 */
constexpr int kPairQuads = 6;
constexpr uint32_t kNoTriangle = 0xffffffffu;

/* slot k (0 = a, 1 = b) of a pair record */
NORI_HD void pair_pack(f4 q[6], int k, const float p0[3], const float e1[3], const float e2[3], uint32_t gid, uint32_t mesh) {
    float *f = &q[0].x;      /* 24 consecutive floats */
    f[0 + k] = p0[0]; f[2 + k] = p0[1]; f[4 + k] = p0[2];
    f[6 + k] = e1[0]; f[8 + k] = e1[1]; f[10 + k] = e1[2];
    f[12 + k] = e2[0]; f[14 + k] = e2[1]; f[16 + k] = e2[2];
    f[18 + k] = u2f(gid); f[20 + k] = u2f(mesh);
}

/* Shading record of one global triangle = 96 B = 6 x dwordx4: what accel.cpp:58-95 fetches through
 * the index buffer after the closest hit is known (and what the area light samples), gathered once
 * at upload so that shading costs ONE dependent fetch instead of index -> vertex:
 *   q0..q2 = (p0, bits mesh), (p1, 0), (p2, 0)      q3..q5 = (n0, 0), (n1, 0), (n2, 0) (zero without normals)
 * Same values as positions[indices[..]] / normals[indices[..]]: bit-identical results. */
constexpr int kShadeQuads = 6;

struct MeshRec {
    uint32_t tri_offset;      /* first global triangle                */
    uint32_t vtx_offset;      /* first global vertex                  */
    uint32_t n_triangles;
    uint32_t flags;           /* bit0 normals, bit1 uv, bit2 emitter  */
    int32_t bsdf_type;
    float albedo[3];          /* diffuse albedo / microfacet kd       */
    float alpha, int_ior, ext_ior, ks;
    float radiance[3];
    float inv_area;           /* DiscretePDF::getNormalization()      */
    uint32_t cdf_offset;      /* into emitter_cdf (n_triangles+1)     */
    uint32_t pad[3];
};
constexpr uint32_t kMeshHasNormals = 1u, kMeshHasUV = 2u, kMeshEmitter = 4u;

struct CameraRec {
    float sample_to_camera[16];   /* row-major */
    float camera_to_world[16];
    float inv_w, inv_h;
    float near_clip, far_clip;
    int32_t width, height;
};

struct FilterRec {
    float table[kFilterRes + 1];  /* src/block.cpp:23-26 */
    float radius;
    float lookup_factor;          /* src/block.cpp:27 */
    int32_t border;               /* src/block.cpp:20 */
};

struct IntegratorRec {
    int32_t type;
    float position[3];
    float energy[3];
};

/* grid of the 32-B node records (rt_nodeq.h): plane position = mn + q scale per axis */
struct NodeqGrid { float mn[3], scale[3]; };

struct DevScene {
    const f4 *nodes;
    const f4 *tris;
    const f4 *positions;        /* global vertex arrays, xyz_ */
    const f4 *normals;          /* valid where the mesh has normals */
    const f2 *texcoords;
    const uint32_t *indices;    /* 3 per global triangle, GLOBAL vertex ids */
    const MeshRec *meshes;
    const float *emitter_cdf;
    const uint32_t *emitters;   /* mesh ids of emitters */
    const uint32_t *tri_mesh;   /* mesh id per global triangle */
    const f4 *shade_tris;       /* kShadeQuads per global triangle: the shading data pre-gathered */
    const f4 *top_image;        /* the hot records of the tree as wf_extend keeps them in LDS (rt_top.h), or null */
    const f4 *nodes_q;          /* the tree's nodes as 32-B records (rt_nodeq.h), same indices as `nodes`; null: the tree does not qualify */
    const f4 *top_image_q;      /* the LDS image with 32-B node records (valid with nodes_q) */
    uint32_t n_emitters;
    uint32_t n_meshes;
    uint32_t n_triangles;
    uint32_t n_cdf;             /* entries of emitter_cdf */
    int32_t root;               /* child-link code of the root */
    uint32_t wide;              /* nodes are WIDE nodes (BVH4, quantised boxes) instead of BVH2 nodes */
    uint32_t top_image_quads;   /* quads of top_image in use (its header's .w) */
    uint32_t top_image_q_quads;
    uint32_t bsdf_mask;         /* bit t: some mesh has a BSDF of nori_bsdf_type t (wf_shade is instantiated per material set) */
    NodeqGrid grid;             /* of nodes_q */
    CameraRec camera;
    FilterRec filter;
    IntegratorRec integrator;
};

struct Hit {
    float t, u, v;
    uint32_t tri;     /* global triangle id or kNoHit */
    uint32_t mesh;
};

struct TraversalCounters {
    uint32_t nodes, tris;
};

} // namespace nrt
