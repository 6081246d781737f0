// Checks of rt_types.h's exact_div / exact_sqrt on the device against the compiler's IEEE-correct a / b and sqrtf(x):
//   sqrt: EXHAUSTIVE over all non-negative floats, mismatches per biased exponent;
//   div : 2^36 pseudo-random pairs; numerator and denominator exponents drawn from [-E, E] for E = 30, 60, 100 plus
//         mantissa patterns near all-zeros / all-ones (the hard cases of Markstein's theorem); numerator zero included.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I nori_amd/csrc/device tools/ubench_divsqrt.hip -o /tmp/ubench_divsqrt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#include "rt_types.h"

using namespace nrt;

__global__ void check_sqrt(unsigned long long *bad, unsigned int *per_exp) {
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t) gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (uint64_t i = tid; i < (1ull << 31); i += stride) {
        const uint32_t u = (uint32_t) i;
        if (((u >> 23) & 255u) == 255u && (u & 0x7fffffu)) continue;      // NaN
        const float x = __builtin_bit_cast(float, u);
        const float ref = sqrtf(x), got = exact_sqrt(x);
        if (__builtin_bit_cast(uint32_t, ref) != __builtin_bit_cast(uint32_t, got)) { ++b; atomicAdd(&per_exp[(u >> 23) & 255u], 1u); }
    }
    atomicAdd(bad, b);
}

__device__ inline uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

__global__ void check_div(int E, unsigned long long per_thread, unsigned long long *bad, unsigned int *first) {
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b = 0;
    for (unsigned long long k = 0; k < per_thread; ++k) {
        const uint64_t r = mix(tid * per_thread + k), r2 = mix(r);
        uint32_t ma = (uint32_t) r & 0x7fffffu, mb = (uint32_t) (r >> 23) & 0x7fffffu;
        const int ea = -E + (int) (r2 % (uint64_t) (2 * E + 1)), eb = -E + (int) ((r2 >> 20) % (uint64_t) (2 * E + 1));
        const uint32_t sel = (uint32_t) (r2 >> 60);
        if (sel == 0u) mb = 0x7fffffu - ((uint32_t) (r2 >> 50) & 15u);
        if (sel == 1u) mb = (uint32_t) (r2 >> 50) & 15u;
        if (sel == 2u) ma = 0x7fffffu - ((uint32_t) (r2 >> 40) & 15u);
        if (sel == 3u) ma = (uint32_t) (r2 >> 40) & 15u;
        float a = __builtin_bit_cast(float, ((uint32_t) (ea + 127) << 23) | ma | ((uint32_t) (r >> 63) << 31));
        const float bb = __builtin_bit_cast(float, ((uint32_t) (eb + 127) << 23) | mb | ((uint32_t) ((r >> 62) & 1u) << 31));
        if (sel == 4u && ((r2 >> 40) & 3u) == 0u) a = 0.0f;
        const float ref = a / bb, got = exact_div(a, bb);
        if (__builtin_bit_cast(uint32_t, ref) != __builtin_bit_cast(uint32_t, got)) { ++b; atomicMin(first, __builtin_bit_cast(uint32_t, bb)); }
    }
    atomicAdd(bad, b);
}

int main() {
    unsigned long long *d; unsigned int *pe, *f;
    hipMalloc(&d, 8); hipMalloc(&pe, 1024); hipMalloc(&f, 4);
    hipMemset(d, 0, 8); hipMemset(pe, 0, 1024);
    hipLaunchKernelGGL(check_sqrt, dim3(4096), dim3(256), 0, 0, d, pe);
    unsigned long long h; unsigned int hp[256];
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(hp, pe, 1024, hipMemcpyDeviceToHost);
    printf("exact_sqrt vs sqrtf over all 2^31 non-negative floats: %llu mismatches\n", h);
    for (int e = 0; e < 256; ++e) if (hp[e]) printf("  biased exponent %3d (x ~ 2^%d): %u mismatches\n", e, e - 127, hp[e]);
    const int Es[3] = {30, 60, 100};
    for (int E : Es) {
        hipMemset(d, 0, 8); hipMemset(f, 0xff, 4);
        const unsigned long long per_thread = 1ull << 16;                  // 2^20 threads x 2^16 = 2^36 pairs
        hipLaunchKernelGGL(check_div, dim3(4096), dim3(256), 0, 0, E, per_thread, d, f);
        unsigned int hf; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
        printf("exact_div vs a / b, 2^36 pairs, exponents of a and b in [-%d, %d]: %llu mismatches (smallest bad denominator bits 0x%08x)\n", E, E, h, hf);
    }
    return 0;
}
