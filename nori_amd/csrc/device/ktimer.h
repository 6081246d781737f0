/* ktimer.h -- per-kernel-class timing with HIP events on the launch stream.
 *
 * bench.py's roofline needs the duration of the dominant kernel measured live, on the stream the
 * kernel runs on.  When enabled, every launch is bracketed by a pair of events; after the stream
 * is synchronised the elapsed times are summed per class.  Disabled: every call is a no-op. */
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace nrt {

enum KernelClass { KC_TRACE = 0, KC_SHADE = 1, KC_FILM = 2, KC_TAIL = 3, KC_COUNT = 4 };      /* KC_TAIL: wf_finish launches that ran BESIDE the next batch (wavefront.hip, tail overlap) */

class KernelTimer {
public:
    explicit KernelTimer(bool enabled) : m_enabled(enabled) {}
    ~KernelTimer() { for (hipEvent_t e : m_events) (void) hipEventDestroy(e); }
    void begin(KernelClass c, hipStream_t s) {
        if (!m_enabled) return;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { m_enabled = false; return; }
        m_events.push_back(a); m_events.push_back(b); m_class.push_back(c);
        (void) hipEventRecord(a, s);
    }
    void end(hipStream_t s) {
        if (!m_enabled || m_events.empty()) return;
        (void) hipEventRecord(m_events.back(), s);
    }
    /* call after the stream(s) have been synchronised */
    void collect(float ms[KC_COUNT], unsigned int launches[KC_COUNT]) const {
        for (int c = 0; c < KC_COUNT; ++c) { ms[c] = 0.0f; launches[c] = 0; }
        for (size_t i = 0; i < m_class.size(); ++i) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, m_events[2 * i], m_events[2 * i + 1]) == hipSuccess) { ms[m_class[i]] += t; launches[m_class[i]]++; }
        }
    }
    bool enabled() const { return m_enabled; }
private:
    bool m_enabled;
    std::vector<hipEvent_t> m_events;
    std::vector<KernelClass> m_class;
};

} // namespace nrt
