/*
 * nori/device.h -- host-side handle on libnori_hip (include/nori_hip.h).
 * Every per-sample virtual of the plugin surface (BSDF::sample, Integrator::Li,
 * Accel::rayIntersect, Sampler::next1D, ...) forwards here; there is no CPU
 * implementation behind it -- without a GPU these calls throw NoriException.
 */
#pragma once
#include <nori/common.h>
#include "../../../../include/nori_hip.h"

NORI_NAMESPACE_BEGIN

class Device {
public:
    /* `device` < 0: $NORI_DEVICE or 0 */
    explicit Device(int device = -1);
    ~Device();
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;

    nori_hip_ctx *ctx() const { return m_ctx; }
    int index() const { return m_device; }
    /* throws NoriException carrying nori_hip_last_error on rc != 0 */
    void check(int rc, const char *what) const;

    /* process-wide context for scene-less operators (BSDF / warp / pcg32 twins) */
    static Device &shared();
private:
    nori_hip_ctx *m_ctx = nullptr;
    int m_device = 0;
};

NORI_NAMESPACE_END
