/* TOOLING ONLY -- the counters behind tools/trav_histogram.py: link into the CPU harness built with -DNORI_TRAV_HISTOGRAM
 *   g++ -O2 -std=c++17 -fPIC -ffp-contract=off -pthread -shared -DNORI_TRAV_HISTOGRAM=1 -o /tmp/libnori_emu_hist.so \
 *       tests/emu/emu.cpp nori_amd/csrc/device/scene_prep.cpp tools/trav_hist_impl.cpp                                      */
#include <atomic>
#include <cstdint>
#include <cstring>
static constexpr uint32_t kSlots = 1u << 22;
static std::atomic<uint64_t> g_hist[2][kSlots];
extern "C" void nori_trav_hist(int kind, uint32_t index) {
    if (index < kSlots) g_hist[kind & 1][index].fetch_add(1, std::memory_order_relaxed);
}
extern "C" void nori_trav_hist_get(int kind, uint64_t *out, uint32_t n) {
    for (uint32_t i = 0; i < n && i < kSlots; ++i) out[i] = g_hist[kind & 1][i].load(std::memory_order_relaxed);
}
extern "C" void nori_trav_hist_reset() { for (auto &k : g_hist) for (auto &c : k) c.store(0, std::memory_order_relaxed); }
