/*
 * rt_wide.h -- packing of one WIDE node (BVH4, child boxes quantised to 8 bits; layout in rt_types.h), shared by
 * the host builder (scene_prep.cpp), the device builder (lbvh.hip) and the test harness.
 *
 * Quantise up to four child boxes against their union: origin = the union's lower corner, scale 2^(e - 128) per
 * axis with the smallest e that lets 255 steps span the union (kept within [2^-100, 2^100]); a child's lower
 * coordinate is rounded DOWN and its upper coordinate UP, checked in binary64 against the dequantised value
 * origin + q 2^e (exact there), so the box a ray is tested against always contains the child's true, padded box.
 * Slots are filled in ascending order of the children's centres along the union's widest axis; unused slots get
 * lo = 255, hi = 0 and the link kWideEmpty.  A child with an unbounded box (numerically collinear triangles,
 * tri_box_pad) makes the node kWideAllHit: no box test, every child is visited.
 */
#pragma once
#include "rt_types.h"

namespace nrt {

NORI_HD void wide_pack(int n, const float (*mn)[3], const float (*mx)[3], const int32_t *link, f4 q[4]) {
    uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};          /* plane dwords: child k in byte k */
    int32_t links[4] = {kWideEmpty, kWideEmpty, kWideEmpty, kWideEmpty};
    bool unbounded = false;
    float umn[3] = {kInf, kInf, kInf}, umx[3] = {-kInf, -kInf, -kInf};
    for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) {
            if (!(fabsf(mn[k][a]) < kWideInfinite) || !(fabsf(mx[k][a]) < kWideInfinite)) unbounded = true;
            umn[a] = fminf(umn[a], mn[k][a]); umx[a] = fmaxf(umx[a], mx[k][a]);
        }
    if (unbounded || n == 0) {
        for (int k = 0; k < n; ++k) links[k] = link[k];
        q[0].x = q[0].y = q[0].z = 0.0f; q[0].w = u2f(kWideAllHit | 128u | (128u << 8) | (128u << 16));
        q[1].x = q[1].y = q[1].z = u2f(0u); q[1].w = u2f(0xffffffffu);
        q[2].x = q[2].y = u2f(0xffffffffu); q[2].z = q[2].w = 0.0f;
        q[3].x = u2f((uint32_t) links[0]); q[3].y = u2f((uint32_t) links[1]); q[3].z = u2f((uint32_t) links[2]); q[3].w = u2f((uint32_t) links[3]);
        return;
    }
    /* slots in ascending order of the children's centres along the widest axis of the union (stable insertion sort) */
    int axis = 0;
    { const float e0 = umx[0] - umn[0], e1 = umx[1] - umn[1], e2 = umx[2] - umn[2]; axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2); }
    int order[4] = {0, 1, 2, 3};
    for (int i = 1; i < n; ++i) {
        const int v = order[i];
        const float key = mn[v][axis] + mx[v][axis];
        int j = i - 1;
        while (j >= 0 && mn[order[j]][axis] + mx[order[j]][axis] > key) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    uint32_t ebits[3] = {128u, 128u, 128u};
    for (int a = 0; a < 3; ++a) {
        const double origin = umn[a], extent = (double) umx[a] - origin;
        int e = 28;                                                  /* scale 2^(e - 128), kept within [2^-100, 2^100] */
        if (extent > 0.0) { int ex; (void) __builtin_frexp(extent / 255.0, &ex); e = ex + 128 < 28 ? 28 : (ex + 128 > 228 ? 228 : ex + 128); }
        for (;; ++e) {                                               /* at most a couple of rounds */
            const double scale = __builtin_ldexp(1.0, e - 128);
            bool ok = true;
            uint32_t l = 0, h = 0;
            for (int s = 0; s < 4 && ok; ++s) {
                if (s >= n) { l |= 255u << (8 * s); continue; }      /* unused slot: lo = 255, hi = 0 */
                const int k = order[s];
                double ql = __builtin_floor(((double) mn[k][a] - origin) / scale), qh = __builtin_ceil(((double) mx[k][a] - origin) / scale);
                ql = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql); qh = qh < 0.0 ? 0.0 : qh;
                while (ql > 0.0 && origin + ql * scale > (double) mn[k][a]) ql -= 1.0;
                while (origin + qh * scale < (double) mx[k][a]) qh += 1.0;
                if (qh > 255.0) { ok = false; break; }
                l |= (uint32_t) ql << (8 * s); h |= (uint32_t) qh << (8 * s);
            }
            if (ok) { lo[a] = l; hi[a] = h; ebits[a] = (uint32_t) e; break; }
        }
    }
    for (int s = 0; s < n; ++s) links[s] = link[order[s]];
    q[0].x = umn[0]; q[0].y = umn[1]; q[0].z = umn[2];
    q[0].w = u2f(ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | ((uint32_t) axis << 24));
    q[1].x = u2f(lo[0]); q[1].y = u2f(lo[1]); q[1].z = u2f(lo[2]); q[1].w = u2f(hi[0]);
    q[2].x = u2f(hi[1]); q[2].y = u2f(hi[2]); q[2].z = q[2].w = 0.0f;
    q[3].x = u2f((uint32_t) links[0]); q[3].y = u2f((uint32_t) links[1]); q[3].z = u2f((uint32_t) links[2]); q[3].w = u2f((uint32_t) links[3]);
}

} // namespace nrt
