// ubench_fetch.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access patterns of THIS renderer, against byte
// counts that are known (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"):
//   k_stream_read     coalesced 16 B per lane over the buffer            (the pattern the guide calibrated: FETCH_SIZE = 1/2)
//   k_gather64        every LANE reads one random 64-B record as four global_load_dwordx4 (a wide-node / 64-B node fetch of wf_extend)
//   k_gather32        ... a random 32-B record as two loads (the 32-B node records)
//   k_gather96        ... a random 96-B pair record as six loads (a leaf step)
//   k_stream_write    coalesced 16-B nontemporal stores
//   k_scatter16       every lane stores 16 B nontemporally at a random 16-B slot (hit records written at a refill)
//   k_scatter16_dense  64 lanes store 16 B each at 64 consecutive slots starting at a random multiple of 64 slots (coalesced 1 KB)
// Buffer 2 GiB (8 x the Infinity Cache); the random indices come from a hash of the global thread id, so the byte counts are exact.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_fetch.hip -o /tmp/ubench_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/uf_r -o f -- /tmp/ubench_fetch
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/uf_w -o w -- /tmp/ubench_fetch
//   python tools/ubench_fetch_summary.py /tmp/uf_r /tmp/uf_w      (the program itself prints bytes and GB/s per kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int kPerThread = 16;

__global__ __launch_bounds__(256) void k_stream_read(const v4f *buf, size_t quads, float *out) {
    const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x, stride = (size_t) gridDim.x * 256;
    v4f acc = {0, 0, 0, 0};
    for (size_t i = t; i < quads; i += stride) acc += __builtin_nontemporal_load(&buf[i]);
    if (acc.x == 123.456f) out[t] = acc.y + acc.z + acc.w;
}

template <int QUADS>
__global__ __launch_bounds__(256) void k_gather(const v4f *buf, uint32_t records, float *out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t r = hash32(t * kPerThread + k) % records;
        const v4f *p = buf + (size_t) r * QUADS;
#pragma unroll
        for (int q = 0; q < QUADS; ++q) acc += p[q];
    }
    if (acc.x == 123.456f) out[t] = acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(256) void k_stream_write(v4f *buf, size_t quads) {
    const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x, stride = (size_t) gridDim.x * 256;
    const v4f v = {1.0f, 2.0f, 3.0f, (float) threadIdx.x};
    for (size_t i = t; i < quads; i += stride) __builtin_nontemporal_store(v, &buf[i]);
}

template <bool DENSE>
__global__ __launch_bounds__(256) void k_scatter16(v4f *buf, uint32_t slots) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const v4f v = {1.0f, 2.0f, 3.0f, (float) threadIdx.x};
    for (int k = 0; k < kPerThread; ++k) {
        uint32_t s;
        if (DENSE) s = (hash32((t >> 6) * kPerThread + k) % (slots >> 6)) * 64u + (t & 63u);      /* a wave writes 1 KB in one piece */
        else s = hash32(t * kPerThread + k) % slots;
        __builtin_nontemporal_store(v, &buf[s]);
    }
}

int main() {
    const size_t bytes = (size_t) 2 << 30, quads = bytes / 16;
    v4f *buf; float *out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, (size_t) 64 << 20));
    CHECK(hipMemset(buf, 0, bytes));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const uint32_t threads = 1u << 24;      /* 16 M threads x 16 accesses */
    auto run = [&](const char *name, double algorithmic, auto &&launch) {
        launch();                                /* warm */
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-18s algorithmic %8.3f GB per launch  %7.3f ms  %7.1f GB/s\n", name, algorithmic / 1e9, ms, algorithmic / ms / 1e6);
    };
    run("k_stream_read", (double) bytes, [&] { hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, 0, buf, quads, out); });
    run("k_gather<4>", (double) threads * kPerThread * 64, [&] { hipLaunchKernelGGL(k_gather<4>, dim3(threads / 256), dim3(256), 0, 0, buf, (uint32_t) (bytes / 64), out); });
    run("k_gather<2>", (double) threads * kPerThread * 32, [&] { hipLaunchKernelGGL(k_gather<2>, dim3(threads / 256), dim3(256), 0, 0, buf, (uint32_t) (bytes / 32), out); });
    run("k_gather<6>", (double) threads * kPerThread * 96, [&] { hipLaunchKernelGGL(k_gather<6>, dim3(threads / 256), dim3(256), 0, 0, buf, (uint32_t) (bytes / 96), out); });
    run("k_stream_write", (double) bytes, [&] { hipLaunchKernelGGL(k_stream_write, dim3(256 * 8), dim3(256), 0, 0, buf, quads); });
    run("k_scatter16<0>", (double) threads * kPerThread * 16, [&] { hipLaunchKernelGGL(k_scatter16<false>, dim3(threads / 256), dim3(256), 0, 0, buf, (uint32_t) quads); });
    run("k_scatter16<1>", (double) threads * kPerThread * 16, [&] { hipLaunchKernelGGL(k_scatter16<true>, dim3(threads / 256), dim3(256), 0, 0, buf, (uint32_t) quads); });
    return 0;
}
