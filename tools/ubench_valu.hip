// ubench_valu.hip -- issue cost of the VALU instructions wf_extend is made of, on gfx950.
// Every kernel runs 8 waves/SIMD of independent instruction chains (16 registers round-robin, so
// dependent-issue latency is hidden) and reports cycles per wave-instruction per SIMD
// (= waves_per_simd * clock * time / instructions).  Build + run (GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((vector_size(8)));

constexpr int kIters = 4096;

// BODY = 16 independent instructions on registers a0..a15 (scalar) or p0..p15 (packed)
#define KERNEL_SCALAR(name, INSTR)                                                                     \
    __global__ __launch_bounds__(256) void name(float *out, float s) {                                 \
        float a[16];                                                                                   \
        for (int k = 0; k < 16; ++k) a[k] = s + (float) (threadIdx.x + k);                             \
        float b = s * 1.0001f, c = s + 0.5f;                                                           \
        for (int it = 0; it < kIters; ++it) {                                                          \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(INSTR : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                              \
        float r = 0.0f;                                                                                \
        for (int k = 0; k < 16; ++k) r += a[k];                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                       \
    }

#define KERNEL_PACKED(name, INSTR)                                                                     \
    __global__ __launch_bounds__(256) void name(float *out, float s) {                                 \
        v2f a[16];                                                                                     \
        for (int k = 0; k < 16; ++k) { a[k][0] = s + (float) (threadIdx.x + k); a[k][1] = s - (float) k; } \
        v2f b = {s * 1.0001f, s * 0.9999f}, c = {s + 0.5f, s - 0.5f};                                  \
        for (int it = 0; it < kIters; ++it) {                                                          \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(INSTR : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                              \
        float r = 0.0f;                                                                                \
        for (int k = 0; k < 16; ++k) r += a[k][0] + a[k][1];                                           \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                       \
    }

KERNEL_SCALAR(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL_SCALAR(k_add, "v_add_f32 %0, %0, %1")
KERNEL_SCALAR(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL_SCALAR(k_min, "v_min_f32 %0, %0, %1")
KERNEL_SCALAR(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL_SCALAR(k_minmax_e64, "v_min_f32_e64 %0, %0, %1 clamp")
KERNEL_SCALAR(k_rcp, "v_rcp_f32 %0, %0")
KERNEL_SCALAR(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL_SCALAR(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL_SCALAR(k_mov, "v_mov_b32 %0, %1")
KERNEL_SCALAR(k_and, "v_and_b32 %0, %0, %1")
KERNEL_SCALAR(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_PACKED(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL_PACKED(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL_PACKED(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_PACKED(k_pk_mov, "v_pk_mov_b32 %0, %1, %1")

KERNEL_SCALAR(k_cvt_u32, "v_cvt_f32_u32 %0, %0")
KERNEL_SCALAR(k_cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL_SCALAR(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %1")
KERNEL_SCALAR(k_cvt_f16, "v_cvt_f32_f16 %0, %1")
KERNEL_SCALAR(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL_SCALAR(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
KERNEL_SCALAR(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_SCALAR(k_lshl_add, "v_lshl_add_u32 %0, %0, 4, %1")
KERNEL_SCALAR(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL_SCALAR(k_fma_mix, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]")
KERNEL_SCALAR(k_cmp_vcc, "v_cmp_le_f32 vcc, %0, %1")
KERNEL_SCALAR(k_cmp_sgpr, "v_cmp_le_f32 s[20:21], %0, %1")
KERNEL_SCALAR(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_SCALAR(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_SCALAR(k_lshlrev, "v_lshlrev_b32 %0, 5, %0")
KERNEL_SCALAR(k_bfe, "v_bfe_u32 %0, %0, 16, 16")
KERNEL_SCALAR(k_sad, "v_sad_u32 %0, %0, %1, %2")
KERNEL_SCALAR(k_fma_abs, "v_fma_f32 %0, -|%0|, %1, %2")

struct Entry { const char *name; void (*fn)(float *, float); int per_iter; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int blocks = cus * 8;      /* 8 blocks of 256 = 8 waves per SIMD */
    float *out;
    CHECK(hipMalloc(&out, (size_t) blocks * 256 * sizeof(float)));
    std::vector<Entry> es = {
        {"v_mul_f32", k_mul, 16}, {"v_add_f32", k_add, 16}, {"v_fma_f32", k_fma, 16}, {"v_min_f32", k_min, 16},
        {"v_max3_f32", k_max3, 16}, {"v_min_f32_e64", k_minmax_e64, 16}, {"v_rcp_f32", k_rcp, 16}, {"v_sqrt_f32", k_sqrt, 16},
        {"v_cmp+v_cndmask (pair)", k_cmp_cnd, 32}, {"v_mov_b32", k_mov, 16}, {"v_and_b32", k_and, 16}, {"v_mul_lo_u32", k_mul_lo_u32, 16},
        {"v_pk_mul_f32", k_pk_mul, 16}, {"v_pk_add_f32", k_pk_add, 16}, {"v_pk_fma_f32", k_pk_fma, 16}, {"v_pk_mov_b32", k_pk_mov, 16},
        {"v_cvt_f32_u32", k_cvt_u32, 16}, {"v_cvt_f32_u32_sdwa WORD_1", k_cvt_sdwa, 16}, {"v_cvt_f32_ubyte1", k_cvt_ubyte, 16}, {"v_cvt_f32_f16", k_cvt_f16, 16},
        {"v_perm_b32", k_perm, 16}, {"v_bfi_b32", k_bfi, 16}, {"v_and_or_b32", k_and_or, 16}, {"v_lshl_add_u32", k_lshl_add, 16}, {"v_med3_f32", k_med3, 16},
        {"v_fma_mix_f32", k_fma_mix, 16}, {"v_cmp_le_f32 -> vcc", k_cmp_vcc, 16}, {"v_cmp_le_f32 -> sgpr", k_cmp_sgpr, 16}, {"v_cndmask_b32 vcc", k_cndmask, 16},
        {"v_add_u32", k_add_u32, 16}, {"v_lshlrev_b32", k_lshlrev, 16}, {"v_bfe_u32", k_bfe, 16}, {"v_sad_u32", k_sad, 16}, {"v_fma_f32 -|a|", k_fma_abs, 16},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (auto &e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);      /* warm */
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5.0f;
        const double wave_instr_per_simd = 8.0 * (double) kIters * e.per_iter;      /* 8 waves on each SIMD */
        const double ns_per = ms * 1e6 / wave_instr_per_simd;
        printf("%-26s %8.3f ms  %6.3f ns per wave-instruction per SIMD  = %5.2f cycles @2.4GHz  (%.1f G wave-instr/s chip)\n",
               e.name, ms, ns_per, ns_per * 2.4, wave_instr_per_simd * cus * 4 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
