/*
 * group.hip -- all GPUs of one node behind ONE host call: nori_hip_group_* of include/nori_hip.h.
 *
 * What the reference does with TBB workers over image blocks (src/main.cpp:85-113) and a mutex around
 * ImageBlock::put(ImageBlock&) (src/block.cpp:93-102), a node of MI355Xs does with one host thread and one
 * nori_hip_ctx per GPU (a context is bound to its device and not thread-safe: one thread each) and ONE merge over xGMI:
 *
 *   render   thread k renders share k (group_merge.h: tiles k, k + N, ... or a range of the sample indices) into device
 *            k's own zero-initialised RGBW frame -- no traffic between devices
 *   merge    reduce: ncclReduce(sum) of the N frames into device 0's (ring over the xGMI links; 17 MB at 1024^2)
 *            gather: every device packs the column strips its tiles touched (pack_strips_kernel), device 0 receives them
 *                    (ncclSend / ncclRecv in one group) and adds them (add_strips_kernel): 1/N of a frame per link
 *   output   device 0's frame -> the caller's host buffer
 *
 * RCCL runs single-process here (ncclCommInitAll: one communicator per device, all owned by this process) and is
 * loaded with dlopen at the first group of more than one distinct device: a one-GPU process never maps librccl.
 * Transport "copy" (hipMemcpyPeerAsync + the same kernels) serves device lists that name a device twice -- which RCCL
 * refuses -- i.e. the tests that drive the whole group path, threads and merges included, on a one-GPU box.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/nori_hip.h"
#include "group_merge.h"
#include "rccl_abi.h"

using namespace nrt;
/* The handful of RCCL types and entry points the group uses are declared by hand in rccl_abi.h (the library itself is dlopen'ed at
   the first group of more than one device: a one-GPU build needs neither the RCCL headers nor the library) and held against the
   installed <rccl/rccl.h> at build-test time (tests/abi/rccl_abi_check.cpp); the version is checked at load time. */
using nori_rccl::ncclComm_t; using nori_rccl::ncclResult_t; using nori_rccl::ncclSuccess; using nori_rccl::ncclFloat; using nori_rccl::ncclSum;

namespace {

struct Rccl {
    void *lib = nullptr;
    nori_rccl::CommInitAll_t CommInitAll = nullptr;
    nori_rccl::CommDestroy_t CommDestroy = nullptr;
    nori_rccl::GetErrorString_t GetErrorString = nullptr;
    nori_rccl::GetVersion_t GetVersion = nullptr;
    nori_rccl::Reduce_t Reduce = nullptr;
    nori_rccl::Send_t Send = nullptr;
    nori_rccl::Recv_t Recv = nullptr;
    nori_rccl::GroupStart_t GroupStart = nullptr;
    nori_rccl::GroupEnd_t GroupEnd = nullptr;
    std::string load() {
        if (lib) return std::string();
        std::string why;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
            const char *e = dlerror();      /* ONE call: dlerror() clears the state it returns */
            if (e && why.empty()) why = e;
        }
        if (!lib) return "librccl not found: " + why;
#define SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { const std::string m = std::string("librccl lacks ") + name; dlclose(lib); lib = nullptr; return m; }
        SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(GetErrorString, "ncclGetErrorString") SYM(GetVersion, "ncclGetVersion")
        SYM(Reduce, "ncclReduce") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
        int version = 0;
        if (GetVersion(&version) != ncclSuccess || version < nori_rccl::kMinVersion) { dlclose(lib); lib = nullptr; return "librccl: version " + std::to_string(version) + " (the enum values used here are those of 2.x)"; }
        return std::string();
    }
};

Rccl g_rccl;      /* process-wide: the library handle only; communicators belong to their group */

/* frame: rows x cols x RGBW; x: frame column of each packed column or -1 (group_merge.h) */
__global__ void pack_strips_kernel(const float4 *frame, int rows, int cols, const int32_t *x, int w, float4 *pack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (i >= w || y >= rows) return;
    const int32_t c = x[i];
    pack[(size_t) y * w + i] = c >= 0 ? frame[(size_t) y * cols + c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__global__ void add_strips_kernel(float4 *frame, int rows, int cols, const int32_t *x, int w, const float4 *pack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (i >= w || y >= rows) return;
    const int32_t c = x[i];
    if (c < 0) return;                 /* columns of one list are distinct: no two threads add to the same pixel */
    float4 a = frame[(size_t) y * cols + c];
    const float4 b = pack[(size_t) y * w + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    frame[(size_t) y * cols + c] = a;
}
__global__ void add_frames_kernel(float4 *dst, const float4 *src, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = dst[i]; const float4 b = src[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    dst[i] = a;
}

} // namespace

struct nori_hip_group {
    std::vector<int> devices;
    std::vector<nori_hip_ctx *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<float *> d_frame;          /* per device, sized for the last frame geometry */
    std::vector<float *> d_pack;           /* per device: its packed strips */
    std::vector<int32_t *> d_x;            /* per device: its strip column list; on device 0 also every other device's */
    std::vector<float *> d_recv;           /* on device 0: one receive buffer per other device (strips or a whole frame) */
    std::vector<int32_t *> d_x_root;       /* on device 0: column list of device k */
    std::vector<ncclComm_t> comms;
    size_t frame_floats = 0, pack_floats = 0, x_ints = 0, recv_floats = 0;      /* what the buffers are sized for */
    /* film_order = reference (group_render_reference): per device its block accumulators; on device 0 a receive buffer (peer-copy
       transport) and the frame -- kept between frames like the others */
    std::vector<float *> d_ref_acc;
    float *d_ref_recv = nullptr, *d_ref_frame = nullptr;
    size_t ref_acc_floats = 0, ref_frame_floats = 0;
    void free_reference_buffers() {
        for (size_t k = 0; k < d_ref_acc.size() && k < devices.size(); ++k) if (d_ref_acc[k]) { (void) hipSetDevice(devices[k]); (void) hipFree(d_ref_acc[k]); }
        if (!devices.empty()) (void) hipSetDevice(devices[0]);
        if (d_ref_recv) (void) hipFree(d_ref_recv);
        if (d_ref_frame) (void) hipFree(d_ref_frame);
        d_ref_acc.clear(); d_ref_recv = d_ref_frame = nullptr; ref_acc_floats = ref_frame_floats = 0;
    }
    bool rccl = false;
    std::string error, warning;
    std::vector<uint32_t> engines;         /* what rendered each device's share of the last frame (nori_render_stats::engine) */
    void free_buffers() {
        for (size_t k = 0; k < devices.size(); ++k) {
            (void) hipSetDevice(devices[k]);
            if (k < d_frame.size() && d_frame[k]) (void) hipFree(d_frame[k]);
            if (k < d_pack.size() && d_pack[k]) (void) hipFree(d_pack[k]);
            if (k < d_x.size() && d_x[k]) (void) hipFree(d_x[k]);
        }
        if (!devices.empty()) (void) hipSetDevice(devices[0]);
        for (float *p : d_recv) if (p) (void) hipFree(p);
        for (int32_t *p : d_x_root) if (p) (void) hipFree(p);
        d_frame.clear(); d_pack.clear(); d_x.clear(); d_recv.clear(); d_x_root.clear();
        frame_floats = pack_floats = x_ints = recv_floats = 0;
        free_reference_buffers();
    }
};

#define GRP_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { g->error = std::string(#expr) + ": " + hipGetErrorString(e__); return NORI_ERR_INTERNAL; } } while (0)
#define GRP_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) { g->error = std::string(#expr) + ": " + g_rccl.GetErrorString(r__); return NORI_ERR_INTERNAL; } } while (0)

static std::string g_group_create_error;

/* The communicators proven before a frame depends on them, on a buffer the size of the biggest frame the bench merges (a 2052^2
   RGBW frame, 67 MB -- a token-sized message takes other protocols and channels inside RCCL than a frame does):
     1. ncclReduce(sum) to rank 0, rank k contributing (k + 1) * (i % 251 + 1): exact in binary32 for <= 64 ranks, so rank 0 must
        read n (n + 1) / 2 * (i % 251 + 1) bit for bit;
     2. the gather merge's exchange: in ONE group every other rank ncclSends its buffer to rank 0, which ncclRecvs each into its
        own slot (a group of one rank sends to and receives from itself) and must find rank k's pattern in slot k.
   Returns "" or what went wrong. */
constexpr size_t kCheckFloats = (size_t) 2052 * 2052 * 4;
__global__ void check_fill_kernel(float *p, size_t n, float scale) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = scale * (float) (i % 251 + 1);
}
/* first element of p[0, n) that is not scale * (i % 251 + 1), or n */
__global__ void check_verify_kernel(const float *p, size_t n, float scale, unsigned long long *first_bad) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] != scale * (float) (i % 251 + 1)) atomicMin(first_bad, (unsigned long long) i);
}
static std::string rccl_self_check(nori_hip_group *g) {
    const int n = (int) g->devices.size();
    const unsigned grid = (unsigned) ((kCheckFloats + 255) / 256);
    std::vector<float *> buf((size_t) n, nullptr), slot((size_t) n, nullptr);      /* slot[k]: on device 0, what rank k sent */
    unsigned long long *d_bad = nullptr;
    std::string err;
    auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e); return e == hipSuccess; };
    auto fill = [&]() {
        for (int k = 0; k < n && err.empty(); ++k) {
            if (!hip_ok(hipSetDevice(g->devices[(size_t) k]), "hipSetDevice")) break;
            hipLaunchKernelGGL(check_fill_kernel, dim3(grid), dim3(256), 0, g->streams[(size_t) k], buf[(size_t) k], kCheckFloats, (float) (k + 1));
            hip_ok(hipGetLastError(), "check_fill_kernel");
        }
    };
    auto sync_all = [&]() { for (int k = 0; k < n && err.empty(); ++k) { hip_ok(hipSetDevice(g->devices[(size_t) k]), "hipSetDevice"); hip_ok(hipStreamSynchronize(g->streams[(size_t) k]), "hipStreamSynchronize"); } };
    /* on device 0: p must hold scale * pattern */
    auto verify = [&](const float *p, float scale, const char *what) {
        if (!err.empty() || !hip_ok(hipSetDevice(g->devices[0]), "hipSetDevice")) return;
        const unsigned long long none = ~0ull;
        unsigned long long bad = none;
        if (!hip_ok(hipMemcpyAsync(d_bad, &none, sizeof(none), hipMemcpyHostToDevice, g->streams[0]), "hipMemcpyAsync")) return;
        hipLaunchKernelGGL(check_verify_kernel, dim3(grid), dim3(256), 0, g->streams[0], p, kCheckFloats, scale, d_bad);
        if (!hip_ok(hipGetLastError(), "check_verify_kernel") || !hip_ok(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, g->streams[0]), "hipMemcpyAsync") ||
            !hip_ok(hipStreamSynchronize(g->streams[0]), "hipStreamSynchronize")) return;
        if (bad != none) {
            float got = 0.0f;
            (void) hipMemcpy(&got, p + bad, sizeof(float), hipMemcpyDeviceToHost);
            err = std::string(what) + ": rank 0 holds " + std::to_string(got) + " at element " + std::to_string(bad) + " of " + std::to_string(kCheckFloats) +
                  ", expected " + std::to_string(scale * (float) (bad % 251 + 1));
        }
    };
    for (int k = 0; k < n && err.empty(); ++k)
        if (!hip_ok(hipSetDevice(g->devices[(size_t) k]), "hipSetDevice") || !hip_ok(hipMalloc((void **) &buf[(size_t) k], kCheckFloats * sizeof(float)), "hipMalloc")) break;
    if (err.empty() && hip_ok(hipSetDevice(g->devices[0]), "hipSetDevice")) {
        hip_ok(hipMalloc((void **) &d_bad, sizeof(unsigned long long)), "hipMalloc");
        for (int k = (n == 1 ? 0 : 1); k < n && err.empty(); ++k) hip_ok(hipMalloc((void **) &slot[(size_t) k], kCheckFloats * sizeof(float)), "hipMalloc");
    }
    /* 1. reduce */
    fill();
    if (err.empty()) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int k = 0; k < n && r == ncclSuccess; ++k) r = g_rccl.Reduce(buf[(size_t) k], buf[(size_t) k], kCheckFloats, ncclFloat, ncclSum, 0, g->comms[(size_t) k], g->streams[(size_t) k]);
        const ncclResult_t e = g_rccl.GroupEnd();      /* always closed, whatever happened inside */
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) err = std::string("ncclReduce: ") + g_rccl.GetErrorString(r);
    }
    sync_all();
    verify(buf[0], (float) (n * (n + 1) / 2), "ncclReduce");
    /* 2. send / receive, grouped as the gather merge groups them */
    fill();
    sync_all();
    if (err.empty()) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int k = (n == 1 ? 0 : 1); k < n && r == ncclSuccess; ++k) {
            r = g_rccl.Send(buf[(size_t) k], kCheckFloats, ncclFloat, 0, g->comms[(size_t) k], g->streams[(size_t) k]);
            if (r == ncclSuccess) r = g_rccl.Recv(slot[(size_t) k], kCheckFloats, ncclFloat, k, g->comms[0], g->streams[0]);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) err = std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(r);
    }
    sync_all();
    for (int k = (n == 1 ? 0 : 1); k < n; ++k) verify(slot[(size_t) k], (float) (k + 1), "ncclSend / ncclRecv");
    for (int k = 0; k < n; ++k) if (buf[(size_t) k]) { (void) hipSetDevice(g->devices[(size_t) k]); (void) hipFree(buf[(size_t) k]); }
    (void) hipSetDevice(g->devices[0]);
    for (float *p : slot) if (p) (void) hipFree(p);
    if (d_bad) (void) hipFree(d_bad);
    return err;
}

/* stats of a frame from the devices' shares: counters summed, times = the slowest device's */
static void group_sum_stats(nori_hip_group *g, const std::vector<nori_render_stats> &st, nori_render_stats *stats) {
    const int n = (int) st.size();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (int k = 0; k < n; ++k) {
            const nori_render_stats &s = st[(size_t) k];
            stats->n_camera_samples += s.n_camera_samples; stats->n_closest_rays += s.n_closest_rays; stats->n_shadow_rays += s.n_shadow_rays;
            stats->n_node_tests += s.n_node_tests; stats->n_tri_tests += s.n_tri_tests; stats->n_invalid += s.n_invalid;
            stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms);       /* the devices render side by side: the slowest counts */
            stats->trace_ms = std::max(stats->trace_ms, s.trace_ms); stats->shade_ms = std::max(stats->shade_ms, s.shade_ms); stats->film_ms = std::max(stats->film_ms, s.film_ms);
            stats->n_workgroups += s.n_workgroups; stats->n_trace_launches = std::max(stats->n_trace_launches, s.n_trace_launches);
            stats->lds_bytes = std::max(stats->lds_bytes, s.lds_bytes);
        }
        /* "auto" picks the engine by the size of a share: the devices may differ -- the summed stats name the first device's,
           nori_hip_group_engines() every device's */
        stats->engine = st[0].engine;
    }
    g->engines.resize((size_t) n);
    for (int k = 0; k < n; ++k) g->engines[(size_t) k] = st[(size_t) k].engine;
}

/* film_order = reference over the group (nori_hip.h: nori_hip_render_block_rows): every device renders its rows of 32x32 blocks
   into a zeroed array of block accumulators, the arrays -- disjoint, so their sum is exact in any order -- are reduced to the
   first device, which adds the blocks into the frame in BlockGenerator's order */
static int group_render_reference(nori_hip_group *g, const nori_render_params *params, int width, int height, float *rgbw,
                                  nori_render_stats *stats, float *merge_ms) {
    const int n = (int) g->ctx.size();
    const int border = nori_hip_border_size(g->ctx[0]);
    const size_t frame_floats = (size_t) (height + 2 * border) * (size_t) (width + 2 * border) * 4;
    size_t acc_floats = 0;
    if (nori_hip_block_acc_floats(g->ctx[0], &acc_floats) != NORI_OK) { g->error = std::string("group_render: ") + nori_hip_last_error(g->ctx[0]); return NORI_ERR_NOT_READY; }
    const uint32_t block_rows = (uint32_t) ((height + 31) / 32);
    struct { std::vector<float *> &acc; float *&recv, *&frame; } buf = {g->d_ref_acc, g->d_ref_recv, g->d_ref_frame};
    if (g->ref_acc_floats != acc_floats || g->ref_frame_floats != frame_floats || (int) g->d_ref_acc.size() != n || (!g->rccl && !g->d_ref_recv)) {
        g->free_reference_buffers();
        g->d_ref_acc.assign((size_t) n, nullptr);
        for (int k = 0; k < n; ++k) { GRP_HIP(hipSetDevice(g->devices[(size_t) k])); GRP_HIP(hipMalloc((void **) &g->d_ref_acc[(size_t) k], acc_floats * sizeof(float))); }
        GRP_HIP(hipSetDevice(g->devices[0]));
        GRP_HIP(hipMalloc((void **) &g->d_ref_frame, frame_floats * sizeof(float)));
        if (!g->rccl) GRP_HIP(hipMalloc((void **) &g->d_ref_recv, acc_floats * sizeof(float)));
        g->ref_acc_floats = acc_floats; g->ref_frame_floats = frame_floats;
    }

    std::vector<int> rc((size_t) n, NORI_OK);
    std::vector<nori_render_stats> st((size_t) n);
    std::vector<std::string> errs((size_t) n);
    std::vector<std::thread> th;
    for (int k = 0; k < n; ++k) th.emplace_back([&, k] {
        const size_t i = (size_t) k;
        if (hipSetDevice(g->devices[i]) != hipSuccess || hipMemsetAsync(buf.acc[i], 0, acc_floats * sizeof(float), g->streams[i]) != hipSuccess) { rc[i] = NORI_ERR_INTERNAL; errs[i] = "clearing the block accumulators failed"; return; }
        const GroupRows rows = group_block_rows(k, n, block_rows);
        nori_render_params p = *params;
        p.stream = g->streams[i];
        std::memset(&st[i], 0, sizeof(st[i]));
        rc[i] = nori_hip_render_block_rows(g->ctx[i], &p, rows.row_begin, rows.row_count, buf.acc[i], &st[i]);      /* synchronises its stream (stats requested) */
        if (rc[i] != NORI_OK) errs[i] = nori_hip_last_error(g->ctx[i]);
        else if (hipStreamSynchronize(g->streams[i]) != hipSuccess) { rc[i] = NORI_ERR_INTERNAL; errs[i] = "hipStreamSynchronize failed"; }      /* (an empty share launched nothing but the memset) */
    });
    for (auto &t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (rc[(size_t) k] != NORI_OK) { g->error = "device " + std::to_string(g->devices[(size_t) k]) + ": " + errs[(size_t) k]; return rc[(size_t) k]; }

    struct Events {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void) hipEventDestroy(e0); if (e1) (void) hipEventDestroy(e1); }
    } ev;
    GRP_HIP(hipSetDevice(g->devices[0]));
    GRP_HIP(hipEventCreate(&ev.e0)); GRP_HIP(hipEventCreate(&ev.e1));
    GRP_HIP(hipEventRecord(ev.e0, g->streams[0]));
    if (g->rccl) {
        GRP_NCCL(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int k = 0; k < n && r == ncclSuccess; ++k) r = g_rccl.Reduce(buf.acc[(size_t) k], buf.acc[(size_t) k], acc_floats, ncclFloat, ncclSum, 0, g->comms[(size_t) k], g->streams[(size_t) k]);
        const ncclResult_t closed = g_rccl.GroupEnd();
        GRP_NCCL(r); GRP_NCCL(closed);
        for (int k = 1; k < n; ++k) { GRP_HIP(hipSetDevice(g->devices[(size_t) k])); GRP_HIP(hipStreamSynchronize(g->streams[(size_t) k])); }
        GRP_HIP(hipSetDevice(g->devices[0]));
    } else {
        for (int k = 1; k < n; ++k) {      /* one receive buffer, the copies and adds in stream order */
            GRP_HIP(hipMemcpyPeerAsync(buf.recv, g->devices[0], buf.acc[(size_t) k], g->devices[(size_t) k], acc_floats * sizeof(float), g->streams[0]));
            hipLaunchKernelGGL(add_frames_kernel, dim3((unsigned) ((acc_floats / 4 + 255) / 256)), dim3(256), 0, g->streams[0], reinterpret_cast<float4 *>(buf.acc[0]), reinterpret_cast<const float4 *>(buf.recv), acc_floats / 4);
            GRP_HIP(hipGetLastError());
        }
    }
    GRP_HIP(hipMemsetAsync(buf.frame, 0, frame_floats * sizeof(float), g->streams[0]));
    if (nori_hip_resolve_blocks(g->ctx[0], buf.acc[0], buf.frame, g->streams[0]) != NORI_OK) { g->error = std::string("group_render: ") + nori_hip_last_error(g->ctx[0]); return NORI_ERR_INTERNAL; }
    GRP_HIP(hipEventRecord(ev.e1, g->streams[0]));
    GRP_HIP(hipMemcpyAsync(rgbw, buf.frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, g->streams[0]));
    GRP_HIP(hipStreamSynchronize(g->streams[0]));
    float ms = 0.0f;
    GRP_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (merge_ms) *merge_ms = ms;
    group_sum_stats(g, st, stats);
    return NORI_OK;
}

extern "C" {

int nori_hip_group_create(const int *devices, int n_devices, nori_hip_group **out) {
    if (!out) return NORI_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) { g_group_create_error = "group_create: 1 .. 64 devices"; return NORI_ERR_INVALID_ARGUMENT; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { g_group_create_error = "no HIP device available"; return NORI_ERR_NO_DEVICE; }
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= count) {
            g_group_create_error = "device " + std::to_string(devices[k]) + " not found (this node has " + std::to_string(count) + ")";
            return NORI_ERR_NO_DEVICE;
        }
    nori_hip_group *g = new nori_hip_group();
    g->devices.assign(devices, devices + n_devices);
    bool distinct = true;
    for (int a = 0; a < n_devices; ++a) for (int b = a + 1; b < n_devices; ++b) distinct &= devices[a] != devices[b];
    const char *tr = getenv("NORI_GROUP_TRANSPORT");
    g->rccl = distinct && (n_devices > 1 || (tr && std::string(tr) == "rccl")) && !(tr && std::string(tr) == "copy");
    for (int k = 0; k < n_devices; ++k) {
        nori_hip_ctx *c = nullptr;
        const int rc = nori_hip_create(devices[k], &c);
        if (rc != NORI_OK) { g_group_create_error = std::string("device ") + std::to_string(devices[k]) + ": " + nori_hip_last_error(nullptr); nori_hip_group_destroy(g); return rc; }
        g->ctx.push_back(c);
        hipStream_t s = nullptr;
        if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { g_group_create_error = "hipStreamCreate failed"; nori_hip_group_destroy(g); return NORI_ERR_INTERNAL; }
        g->streams.push_back(s);
    }
    if (g->rccl) {
        std::string e = g_rccl.load();
        if (!e.empty()) {
            /* no RCCL on this box: the merge still works over peer copies (hipMemcpyPeerAsync + the same kernels), only slower --
               unless the caller insisted on RCCL */
            if (tr && std::string(tr) == "rccl") { g_group_create_error = e; nori_hip_group_destroy(g); return NORI_ERR_UNSUPPORTED; }
            g->rccl = false; g->warning = e + " -- merging over peer copies";
            fprintf(stderr, "[nori_hip_group] %s\n", g->warning.c_str());
        }
    }
    if (g->rccl) {
        g->comms.assign((size_t) n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
        if (r != ncclSuccess) { g_group_create_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); g->comms.clear(); nori_hip_group_destroy(g); return NORI_ERR_INTERNAL; }
        /* RCCL with more than one rank runs for the first time on the user's node: prove the communicators before a frame
           depends on them -- every rank contributes a known pattern, rank 0 must hold the sum */
        const std::string check = rccl_self_check(g);
        if (!check.empty()) { g_group_create_error = "RCCL self-check: " + check; nori_hip_group_destroy(g); return NORI_ERR_INTERNAL; }
    }
    (void) hipSetDevice(devices[0]);
    *out = g;
    return NORI_OK;
}

void nori_hip_group_destroy(nori_hip_group *g) {
    if (!g) return;
    g->free_buffers();
    for (ncclComm_t c : g->comms) if (c) (void) g_rccl.CommDestroy(c);
    for (size_t k = 0; k < g->streams.size(); ++k) { (void) hipSetDevice(g->devices[k]); if (g->streams[k]) (void) hipStreamDestroy(g->streams[k]); }
    for (nori_hip_ctx *c : g->ctx) nori_hip_destroy(c);
    delete g;
}

int nori_hip_group_size(const nori_hip_group *g) { return g ? (int) g->ctx.size() : 0; }
nori_hip_ctx *nori_hip_group_ctx(nori_hip_group *g, int i) { return (g && i >= 0 && i < (int) g->ctx.size()) ? g->ctx[(size_t) i] : nullptr; }
const char *nori_hip_group_last_error(const nori_hip_group *g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }
const char *nori_hip_group_transport(const nori_hip_group *g) { return !g ? "" : g->rccl ? "rccl" : "copy"; }
const char *nori_hip_group_warning(const nori_hip_group *g) { return g ? g->warning.c_str() : ""; }
int nori_hip_group_engines(const nori_hip_group *g, uint32_t *engines, int capacity) {
    if (!g || !engines || capacity < 0) return NORI_ERR_INVALID_ARGUMENT;
    const int n = (int) std::min<size_t>(g->engines.size(), (size_t) capacity);
    for (int k = 0; k < n; ++k) engines[k] = g->engines[(size_t) k];
    return n;
}

int nori_hip_group_upload_scene(nori_hip_group *g, const nori_scene_desc *scene, int builder) {
    if (!g || !scene) return NORI_ERR_INVALID_ARGUMENT;
    const size_t n = g->ctx.size();
    std::vector<int> rc(n, NORI_OK);
    std::vector<std::thread> th;
    for (size_t k = 0; k < n; ++k) th.emplace_back([&, k] {
        rc[k] = nori_hip_upload_scene(g->ctx[k], scene);
        if (rc[k] == NORI_OK) rc[k] = nori_hip_build_accel(g->ctx[k], builder);
    });
    for (auto &t : th) t.join();
    for (size_t k = 0; k < n; ++k)
        if (rc[k] != NORI_OK) { g->error = "device " + std::to_string(g->devices[k]) + ": " + nori_hip_last_error(g->ctx[k]); return rc[k]; }
    return NORI_OK;
}

int nori_hip_group_render_host(nori_hip_group *g, const nori_render_params *params, int split, int merge, int width, int height,
                               float *rgbw, nori_render_stats *stats, float *merge_ms) {
    if (!g || !params || !rgbw) return NORI_ERR_INVALID_ARGUMENT;
    if (params->tile_mod != 1 || params->tile_rem != 0) { g->error = "group_render: the group shares the frame out itself (tile_mod 1, tile_rem 0)"; return NORI_ERR_INVALID_ARGUMENT; }
    if (split != kSplitTile && split != kSplitSample) { g->error = "group_render: split must be tile (0) or sample (1)"; return NORI_ERR_INVALID_ARGUMENT; }
    if (merge != kMergeReduce && merge != kMergeGather) { g->error = "group_render: merge must be reduce (0) or gather (1)"; return NORI_ERR_INVALID_ARGUMENT; }
    const int n = (int) g->ctx.size();
    const int border = nori_hip_border_size(g->ctx[0]);
    if (border < 0) { g->error = std::string("group_render: ") + nori_hip_last_error(g->ctx[0]); return NORI_ERR_NOT_READY; }
    const int rows = height + 2 * border, cols = width + 2 * border;
    const uint32_t tiles_x = (uint32_t) ((width + NORI_TILE_SIZE - 1) / NORI_TILE_SIZE);
    if (merge == kMergeGather && (split != kSplitTile || tiles_x % (uint32_t) n != 0u)) {
        g->error = "group_render: the gather merge needs the tile split and a tile-column count (" + std::to_string(tiles_x) + ") divisible by the number of devices (" + std::to_string(n) + "); use the reduce merge";
        return NORI_ERR_INVALID_ARGUMENT;
    }
    /* film_order = reference promises the reference's summation order over the WHOLE frame: a sum of the devices' frames would lose
       it, so the group hands out block rows and merges the blocks' accumulators instead (group_render_reference) */
    if (n > 1) {
        int n_ref = 0;
        for (int k = 0; k < n; ++k) {
            char v[32] = "";
            if (nori_hip_get_option(g->ctx[(size_t) k], "film_order", v, sizeof(v)) == NORI_OK && std::string(v) == "reference") ++n_ref;
        }
        if (n_ref != 0 && n_ref != n) { g->error = "group_render: film_order = reference on some devices of the group only"; return NORI_ERR_INVALID_ARGUMENT; }
        if (n_ref == n) {
            if (params->seed_mode == NORI_SEED_NORI_BLOCK) { g->error = "group_render: NORI_SEED_NORI_BLOCK renders whole frames on one device"; return NORI_ERR_UNSUPPORTED; }
            /* the reference's order leaves no choice of split or merge: rows of 32x32 blocks, the blocks' accumulators reduced */
            {
                const char *note = "group_render: film_order = reference renders rows of 32x32 blocks per device and reduces their accumulators; the split and merge arguments do not apply";
                if (g->warning.find(note) == std::string::npos) g->warning += (g->warning.empty() ? "" : "; ") + std::string(note);
            }
            return group_render_reference(g, params, width, height, rgbw, stats, merge_ms);
        }
    }
    if (params->seed_mode == NORI_SEED_NORI_BLOCK && n > 1) { g->error = "group_render: NORI_SEED_NORI_BLOCK renders whole frames on one device"; return NORI_ERR_UNSUPPORTED; }

    /* buffers for this frame geometry */
    const size_t frame_floats = (size_t) rows * cols * 4;
    std::vector<std::vector<int32_t>> xs((size_t) n);
    size_t pack_floats = 0;
    if (merge == kMergeGather)
        for (int k = 0; k < n; ++k) { xs[(size_t) k] = group_strip_columns(k, n, tiles_x, border, cols); pack_floats = std::max(pack_floats, xs[(size_t) k].size() * (size_t) rows * 4); }
    size_t x_ints = 0;
    for (int k = 0; k < n; ++k) x_ints = std::max(x_ints, xs[(size_t) k].size());
    const size_t recv_floats = n == 1 ? 0 : merge == kMergeGather ? pack_floats : (g->rccl ? 0 : frame_floats);
    if (g->frame_floats != frame_floats || g->pack_floats < pack_floats || g->x_ints < x_ints || g->recv_floats < recv_floats) {
        g->free_buffers();
        g->d_frame.assign((size_t) n, nullptr); g->d_pack.assign((size_t) n, nullptr); g->d_x.assign((size_t) n, nullptr);
        g->d_recv.assign((size_t) n, nullptr); g->d_x_root.assign((size_t) n, nullptr);
        for (int k = 0; k < n; ++k) {
            GRP_HIP(hipSetDevice(g->devices[(size_t) k]));
            GRP_HIP(hipMalloc((void **) &g->d_frame[(size_t) k], frame_floats * sizeof(float)));
            if (pack_floats) GRP_HIP(hipMalloc((void **) &g->d_pack[(size_t) k], pack_floats * sizeof(float)));
            if (x_ints) GRP_HIP(hipMalloc((void **) &g->d_x[(size_t) k], x_ints * sizeof(int32_t)));
        }
        GRP_HIP(hipSetDevice(g->devices[0]));
        for (int k = 1; k < n; ++k) {      /* device 0 receives: strips (gather), or whole frames when the transport is a plain copy */
            if (recv_floats) GRP_HIP(hipMalloc((void **) &g->d_recv[(size_t) k], recv_floats * sizeof(float)));
            if (x_ints) GRP_HIP(hipMalloc((void **) &g->d_x_root[(size_t) k], x_ints * sizeof(int32_t)));
        }
        g->frame_floats = frame_floats; g->pack_floats = pack_floats; g->x_ints = x_ints; g->recv_floats = recv_floats;
    }
    if (merge == kMergeGather)
        for (int k = 1; k < n; ++k) {
            GRP_HIP(hipSetDevice(g->devices[(size_t) k]));
            GRP_HIP(hipMemcpy(g->d_x[(size_t) k], xs[(size_t) k].data(), xs[(size_t) k].size() * sizeof(int32_t), hipMemcpyHostToDevice));
            GRP_HIP(hipSetDevice(g->devices[0]));
            GRP_HIP(hipMemcpy(g->d_x_root[(size_t) k], xs[(size_t) k].data(), xs[(size_t) k].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }

    /* render: one host thread per device */
    std::vector<int> rc((size_t) n, NORI_OK);
    std::vector<nori_render_stats> st((size_t) n);
    std::vector<std::string> errs((size_t) n);
    std::vector<std::thread> th;
    for (int k = 0; k < n; ++k) th.emplace_back([&, k] {
        const size_t i = (size_t) k;
        if (hipSetDevice(g->devices[i]) != hipSuccess || hipMemsetAsync(g->d_frame[i], 0, frame_floats * sizeof(float), g->streams[i]) != hipSuccess) { rc[i] = NORI_ERR_INTERNAL; errs[i] = "clearing the frame failed"; return; }
        const GroupShare sh = group_share(split, k, n, params->spp_begin, params->spp_count);
        nori_render_params p = *params;
        p.spp_begin = sh.spp_begin; p.spp_count = sh.spp_count; p.tile_mod = sh.tile_mod; p.tile_rem = sh.tile_rem; p.stream = g->streams[i];
        std::memset(&st[i], 0, sizeof(st[i]));
        rc[i] = nori_hip_render(g->ctx[i], &p, g->d_frame[i], &st[i]);      /* synchronises its stream (stats requested) */
        if (rc[i] != NORI_OK) errs[i] = nori_hip_last_error(g->ctx[i]);
    });
    for (auto &t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (rc[(size_t) k] != NORI_OK) { g->error = "device " + std::to_string(g->devices[(size_t) k]) + ": " + errs[(size_t) k]; return rc[(size_t) k]; }

    /* merge on device 0: ImageBlock::put(ImageBlock&) across devices */
    struct Events {      /* destroyed on every way out */
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void) hipEventDestroy(e0); if (e1) (void) hipEventDestroy(e1); }
    } ev;
    hipEvent_t &e0 = ev.e0, &e1 = ev.e1;
    GRP_HIP(hipSetDevice(g->devices[0]));
    GRP_HIP(hipEventCreate(&e0)); GRP_HIP(hipEventCreate(&e1));
    GRP_HIP(hipEventRecord(e0, g->streams[0]));
    const dim3 blk(256);
    if ((n > 1 || g->rccl) && merge == kMergeReduce) {      /* (a forced one-rank RCCL group still calls ncclReduce: a smoke test of the library on one GPU) */
        if (g->rccl) {
            GRP_NCCL(g_rccl.GroupStart());
            ncclResult_t r = ncclSuccess;      /* a failure inside leaves through GroupEnd: an open group would poison every later RCCL call of the process */
            for (int k = 0; k < n && r == ncclSuccess; ++k) r = g_rccl.Reduce(g->d_frame[(size_t) k], g->d_frame[(size_t) k], frame_floats, ncclFloat, ncclSum, 0, g->comms[(size_t) k], g->streams[(size_t) k]);
            const ncclResult_t closed = g_rccl.GroupEnd();
            GRP_NCCL(r); GRP_NCCL(closed);
        } else {
            for (int k = 1; k < n; ++k) {
                GRP_HIP(hipMemcpyPeerAsync(g->d_recv[(size_t) k], g->devices[0], g->d_frame[(size_t) k], g->devices[(size_t) k], frame_floats * sizeof(float), g->streams[0]));
                hipLaunchKernelGGL(add_frames_kernel, dim3((unsigned) ((frame_floats / 4 + 255) / 256)), blk, 0, g->streams[0], reinterpret_cast<float4 *>(g->d_frame[0]), reinterpret_cast<const float4 *>(g->d_recv[(size_t) k]), frame_floats / 4);
            }
        }
    } else if (n > 1) {
        for (int k = 1; k < n; ++k) {      /* every other device packs its strips */
            const int w = (int) xs[(size_t) k].size();
            GRP_HIP(hipSetDevice(g->devices[(size_t) k]));
            hipLaunchKernelGGL(pack_strips_kernel, dim3((unsigned) ((w + 255) / 256), (unsigned) rows), blk, 0, g->streams[(size_t) k], reinterpret_cast<const float4 *>(g->d_frame[(size_t) k]), rows, cols, g->d_x[(size_t) k], w, reinterpret_cast<float4 *>(g->d_pack[(size_t) k]));
            GRP_HIP(hipGetLastError());
        }
        if (g->rccl) {
            GRP_NCCL(g_rccl.GroupStart());
            ncclResult_t r = ncclSuccess;
            for (int k = 1; k < n && r == ncclSuccess; ++k) {
                const size_t cnt = xs[(size_t) k].size() * (size_t) rows * 4;
                r = g_rccl.Send(g->d_pack[(size_t) k], cnt, ncclFloat, 0, g->comms[(size_t) k], g->streams[(size_t) k]);
                if (r == ncclSuccess) r = g_rccl.Recv(g->d_recv[(size_t) k], cnt, ncclFloat, k, g->comms[0], g->streams[0]);
            }
            const ncclResult_t closed = g_rccl.GroupEnd();
            GRP_NCCL(r); GRP_NCCL(closed);
        } else {
            for (int k = 1; k < n; ++k) {
                GRP_HIP(hipSetDevice(g->devices[(size_t) k]));
                GRP_HIP(hipStreamSynchronize(g->streams[(size_t) k]));      /* the strips are packed */
                GRP_HIP(hipMemcpyPeerAsync(g->d_recv[(size_t) k], g->devices[0], g->d_pack[(size_t) k], g->devices[(size_t) k], xs[(size_t) k].size() * (size_t) rows * 4 * sizeof(float), g->streams[0]));
            }
        }
        GRP_HIP(hipSetDevice(g->devices[0]));
        for (int k = 1; k < n; ++k) {
            const int w = (int) xs[(size_t) k].size();
            hipLaunchKernelGGL(add_strips_kernel, dim3((unsigned) ((w + 255) / 256), (unsigned) rows), blk, 0, g->streams[0], reinterpret_cast<float4 *>(g->d_frame[0]), rows, cols, g->d_x_root[(size_t) k], w, reinterpret_cast<const float4 *>(g->d_recv[(size_t) k]));
            GRP_HIP(hipGetLastError());
        }
    }
    if (g->rccl && n > 1) for (int k = 1; k < n; ++k) { GRP_HIP(hipSetDevice(g->devices[(size_t) k])); GRP_HIP(hipStreamSynchronize(g->streams[(size_t) k])); }
    GRP_HIP(hipSetDevice(g->devices[0]));
    GRP_HIP(hipEventRecord(e1, g->streams[0]));
    GRP_HIP(hipMemcpyAsync(rgbw, g->d_frame[0], frame_floats * sizeof(float), hipMemcpyDeviceToHost, g->streams[0]));
    GRP_HIP(hipStreamSynchronize(g->streams[0]));
    float ms = 0.0f;
    GRP_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (merge_ms) *merge_ms = ms;
    group_sum_stats(g, st, stats);
    return NORI_OK;
}

} // extern "C"
