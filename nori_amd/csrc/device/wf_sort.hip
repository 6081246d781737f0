/* wf_sort.hip -- keys and the radix sort of wf_sort.h (its own translation unit: hipCUB's sort takes a while to compile). */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "wf_records.h"
#include "wf_sort.h"

namespace nrt {

namespace {

__device__ __forceinline__ uint32_t spread3(uint32_t v) {      /* 10 bits -> every third bit */
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void wf_sort_keys_kernel(DevScene sc, const P3 *o, const f4 *dA, uint32_t n, int kind, int bits, uint32_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f4 d = dA[i];
    const uint32_t fl = state_flags(d);
    const uint32_t live_bits = (uint32_t) (3 * bits + 3 + (kind == 2 ? 1 : 0));
    uint32_t key = 1u << live_bits;                   /* no path in this slot: behind every live key */
    if (fl & (F_HAS_A | F_HAS_B)) {
        const P3 p = o[i];
        const float q[3] = {(p.x - sc.bounds_lo[0]) * sc.bounds_inv[0], (p.y - sc.bounds_lo[1]) * sc.bounds_inv[1], (p.z - sc.bounds_lo[2]) * sc.bounds_inv[2]};
        uint32_t c[3];
        const float cells = (float) (1 << bits);
        for (int a = 0; a < 3; ++a) c[a] = (uint32_t) fminf(cells - 1.0f, fmaxf(0.0f, q[a] * cells));
        const uint32_t cell = spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
        /* the direction the path's LONG ray takes: the continuation ray, or -- none -- the shadow ray (dA is zero then: octant 0) */
        const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
        if (kind == 3) key = (oct << (3 * bits)) | cell;
        else key = (cell << 3) | oct;
        if (kind == 2 && (fl & F_HAS_B)) key |= 1u << (3 * bits + 3);
    }
    keys[i] = key; vals[i] = i;
}

} // namespace

void WfSortBuffers::release() {
    for (int k = 0; k < 2; ++k) { if (keys[k]) (void) hipFree(keys[k]); if (vals[k]) (void) hipFree(vals[k]); keys[k] = vals[k] = nullptr; }
    if (temp) (void) hipFree(temp);
    temp = nullptr; temp_bytes = 0; capacity = 0;
}

#define WS_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e__); } while (0)

std::string wf_sort_pass(WfSortBuffers &b, const DevScene &sc, const void *o, const void *dA, uint32_t n, const WfSortParams &p, void *stream_) {
    hipStream_t s = (hipStream_t) stream_;
    if (n == 0) return std::string();
    const int bits = std::min(9, std::max(0, p.cell_bits));
    const int key_bits = 3 * bits + 3 + (p.kind == 2 ? 1 : 0) + 1;      /* + 1: the key of empty slots */
    if (b.capacity < n) {
        const size_t cap = (size_t) n + n / 8 + 1024;
        b.release();
        for (int k = 0; k < 2; ++k) { WS_TRY(hipMalloc((void **) &b.keys[k], cap * sizeof(uint32_t))); WS_TRY(hipMalloc((void **) &b.vals[k], cap * sizeof(uint32_t))); }
        size_t bytes = 0;
        WS_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, b.keys[0], b.keys[1], b.vals[0], b.vals[1], (int) cap, 0, 32, s));
        WS_TRY(hipMalloc(&b.temp, bytes));
        b.temp_bytes = bytes; b.capacity = cap;
    }
    hipLaunchKernelGGL(wf_sort_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sc, (const P3 *) o, (const f4 *) dA, n, p.kind, bits, b.keys[0], b.vals[0]);
    size_t bytes = b.temp_bytes;
    WS_TRY(hipcub::DeviceRadixSort::SortPairs(b.temp, bytes, b.keys[0], b.keys[1], b.vals[0], b.vals[1], (int) n, 0, key_bits, s));
    return std::string();
}

} // namespace nrt
