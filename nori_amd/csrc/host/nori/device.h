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

/* All GPUs of the node the render is shared over (nori_hip_group_*, include/nori_hip.h): what tbb::parallel_for over the
 * image blocks (src/main.cpp:85-113) and the mutex-guarded ImageBlock::put(ImageBlock&) (src/block.cpp:93-102) are in the
 * reference.  One context and one host thread per device inside the library; one RCCL merge over xGMI. */
class DeviceGroup {
public:
    /* devices 0 .. n-1 of the node; throws NoriException ("device 1 not found ...") when one is missing */
    explicit DeviceGroup(int n);
    ~DeviceGroup();
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;
    nori_hip_group *group() const { return m_group; }
    int size() const { return nori_hip_group_size(m_group); }
    void check(int rc, const char *what) const;
private:
    nori_hip_group *m_group = nullptr;
};

NORI_NAMESPACE_END
