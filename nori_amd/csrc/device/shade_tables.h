/*
 * shade_tables.h -- the scene's small tables in LDS for the kernels that run Integrator::Li.
 *
 * The per-mesh records (BSDF, radiance, area pdf), the emitter list and the emitters' triangle CDFs sit
 * in the middle of every path vertex's dependent chain
 *     hit -> shading record -> mesh record -> emitter -> its mesh record -> CDF -> emitter triangle
 * and are a few KB for the scenes Nori ships.  A workgroup copies them to LDS once and redirects the
 * DevScene pointers, turning four of those round trips to L2 into LDS reads (measured on the Cornell
 * box: wf_shade 31.3 -> 28.5 ms per frame).  Scenes whose tables do not fit keep reading global memory.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "rt_types.h"

namespace nrt {

/* capacities in 32-bit words: 32 mesh records, 64 emitters, 960 CDF entries (6 KB) */
constexpr uint32_t kShadeMeshWords = 32u * (uint32_t) (sizeof(MeshRec) / 4), kShadeEmitterWords = 64u, kShadeCdfWords = 960u;
constexpr uint32_t kShadeTabWords = kShadeMeshWords + kShadeEmitterWords + kShadeCdfWords;

/* call from every thread of the workgroup (contains a barrier); s_tab: kShadeTabWords / 4 uint4 in LDS */
__device__ __forceinline__ void shade_tables_to_lds(DevScene &sc, uint4 *s_tab) {
    const bool fit = sc.n_meshes * (uint32_t) (sizeof(MeshRec) / 4) <= kShadeMeshWords && sc.n_emitters <= kShadeEmitterWords &&
                     sc.n_cdf <= kShadeCdfWords;
    if (!fit) return;                   /* workgroup-uniform */
    const uint32_t nm = sc.n_meshes * (uint32_t) (sizeof(MeshRec) / 16), ne = (sc.n_emitters + 3u) / 4u, nc = (sc.n_cdf + 3u) / 4u;
    const uint4 *gm = reinterpret_cast<const uint4 *>(sc.meshes), *ge = reinterpret_cast<const uint4 *>(sc.emitters),
                *gc = reinterpret_cast<const uint4 *>(sc.emitter_cdf);       /* uploads are padded to 16 B */
    for (uint32_t k = threadIdx.x; k < nm; k += blockDim.x) s_tab[k] = gm[k];
    for (uint32_t k = threadIdx.x; k < ne; k += blockDim.x) s_tab[kShadeMeshWords / 4 + k] = ge[k];
    for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) s_tab[(kShadeMeshWords + kShadeEmitterWords) / 4 + k] = gc[k];
    __syncthreads();
    sc.meshes = reinterpret_cast<const MeshRec *>(s_tab);
    sc.emitters = reinterpret_cast<const uint32_t *>(s_tab) + kShadeMeshWords;
    sc.emitter_cdf = reinterpret_cast<const float *>(s_tab) + kShadeMeshWords + kShadeEmitterWords;
}

/* The same copy read with LDS instructions (rt_path.h, SceneTables: why) -- for scenes whose tables fit (shade_tables_fit, decided
   on the host: wf_shade is instantiated for either policy).  The copy is filled by shade_tables_copy -- call it from every thread
   of the workgroup (barrier). */
typedef const __attribute__((address_space(3))) uint32_t *lds_words_t;
struct LdsTables {
    lds_words_t w;
    __device__ __forceinline__ MeshRec mesh(uint32_t i) const {
        MeshRec m;
        uint32_t *dst = reinterpret_cast<uint32_t *>(&m);
        const lds_words_t src = w + i * (uint32_t) (sizeof(MeshRec) / 4);
#pragma unroll
        for (int k = 0; k < (int) (sizeof(MeshRec) / 4); ++k) dst[k] = src[k];      /* (only the fields a caller uses survive) */
        return m;
    }
    __device__ __forceinline__ uint32_t emitter(uint32_t i) const { return w[kShadeMeshWords + i]; }
    __device__ __forceinline__ float cdf(uint32_t i) const { return __uint_as_float(w[kShadeMeshWords + kShadeEmitterWords + i]); }
};

inline bool shade_tables_fit(const DevScene &sc) {
    return sc.n_meshes * (uint32_t) (sizeof(MeshRec) / 4) <= kShadeMeshWords && sc.n_emitters <= kShadeEmitterWords && sc.n_cdf <= kShadeCdfWords;
}

__device__ __forceinline__ LdsTables shade_tables_copy(const DevScene &sc, uint4 *s_tab) {
    const uint32_t nm = sc.n_meshes * (uint32_t) (sizeof(MeshRec) / 16), ne = (sc.n_emitters + 3u) / 4u, nc = (sc.n_cdf + 3u) / 4u;
    const uint4 *gm = reinterpret_cast<const uint4 *>(sc.meshes), *ge = reinterpret_cast<const uint4 *>(sc.emitters),
                *gc = reinterpret_cast<const uint4 *>(sc.emitter_cdf);       /* uploads are padded to 16 B */
    for (uint32_t k = threadIdx.x; k < nm; k += blockDim.x) s_tab[k] = gm[k];
    for (uint32_t k = threadIdx.x; k < ne; k += blockDim.x) s_tab[kShadeMeshWords / 4 + k] = ge[k];
    for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) s_tab[(kShadeMeshWords + kShadeEmitterWords) / 4 + k] = gc[k];
    __syncthreads();
    LdsTables t; t.w = (lds_words_t) reinterpret_cast<const uint32_t *>(s_tab);
    return t;
}

} // namespace nrt
