#include <nori/device.h>
#include <cstdlib>
#include <memory>
#include <vector>

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

DeviceGroup::DeviceGroup(int n) {
    std::vector<int> devices;
    for (int k = 0; k < n; ++k) devices.push_back(k);
    int rc = nori_hip_group_create(devices.data(), n, &m_group);
    if (rc != NORI_OK)
        throw NoriException("Unable to set up %i GPUs: %s (the MI355X path has no CPU fallback)", n, nori_hip_group_last_error(nullptr));
}

DeviceGroup::~DeviceGroup() { nori_hip_group_destroy(m_group); }

void DeviceGroup::check(int rc, const char *what) const {
    if (rc != NORI_OK) throw NoriException("%s failed (%i): %s", what, rc, nori_hip_group_last_error(m_group));
}

NORI_NAMESPACE_END
