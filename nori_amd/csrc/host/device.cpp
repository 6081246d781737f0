#include <nori/device.h>
#include <cstdlib>
#include <memory>

NORI_NAMESPACE_BEGIN

Device::Device(int device) {
    if (device < 0) {
        const char *e = std::getenv("NORI_DEVICE");
        device = e ? std::atoi(e) : 0;
    }
    m_device = device;
    int rc = nori_hip_create(device, &m_ctx);
    if (rc != NORI_OK)
        throw NoriException("Unable to create a HIP context on device %i: %s (the MI355X path has no CPU fallback)",
                            device, nori_hip_last_error(nullptr));
}

Device::~Device() { nori_hip_destroy(m_ctx); }

void Device::check(int rc, const char *what) const {
    if (rc != NORI_OK) throw NoriException("%s failed (%i): %s", what, rc, nori_hip_last_error(m_ctx));
}

Device &Device::shared() {
    static std::unique_ptr<Device> dev;
    if (!dev) dev.reset(new Device());
    return *dev;
}

NORI_NAMESPACE_END
