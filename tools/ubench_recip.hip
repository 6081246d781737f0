// Exhaustive check over ALL 2^32 floats: is  r0 = v_rcp_f32(x); r1 = fma(fma(-x, r0, 1), r0, r0)  (rt_trace.h, exact_rcp)
// equal to the IEEE-correct 1.0f / x?  Reports the mismatches per biased exponent of x.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_recip.hip -o build/ubench_recip && build/ubench_recip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void check(unsigned long long *bad1, unsigned long long *bad2, unsigned int *first_bad, unsigned int *per_exp) {
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    unsigned long long b1 = 0, b2 = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const uint32_t u = (uint32_t) i;
        const uint32_t ex = (u >> 23) & 255u;
        if (ex == 255u) continue;                          // inf / NaN
        const float x = __builtin_bit_cast(float, u);
        const float ref = 1.0f / x;                       // IEEE correctly rounded (hipcc default)
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e0 = __builtin_fmaf(-x, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        const float e1 = __builtin_fmaf(-x, r1, 1.0f);
        const float r2 = __builtin_fmaf(e1, r1, r1);
        if (__builtin_bit_cast(uint32_t, r1) != __builtin_bit_cast(uint32_t, ref)) { ++b1; atomicAdd(&per_exp[ex], 1u); }
        if (__builtin_bit_cast(uint32_t, r2) != __builtin_bit_cast(uint32_t, ref)) { ++b2; atomicMin(first_bad, u); }
    }
    atomicAdd(bad1, b1); atomicAdd(bad2, b2);
}

int main() {
    unsigned long long *d; unsigned int *f, *pe;
    hipMalloc(&d, 16); hipMemset(d, 0, 16); hipMalloc(&f, 4); hipMemset(f, 0xff, 4); hipMalloc(&pe, 1024); hipMemset(pe, 0, 1024);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d + 1, f, pe);
    unsigned long long h[2]; unsigned int hf;
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
    printf("one Newton step: %llu mismatches; two steps: %llu mismatches (first 0x%08x)\n", h[0], h[1], hf);
    unsigned int hp[256]; hipMemcpy(hp, pe, 1024, hipMemcpyDeviceToHost);
    for (int e = 0; e < 255; ++e) if (hp[e]) printf("  biased exponent %3d (|x| ~ 2^%d): %u mismatches\n", e, e - 127, hp[e]);
    return 0;
}
