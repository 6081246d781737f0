#include <nori/device.h>
#include <cstdlib>
#include <memory>
#include <vector>

NORI_NAMESPACE_BEGIN

/* NORI_FILM_ORDER=fast|reference (`nori --film-order ...`): the option film_order of include/nori_hip.h on every context --
   "reference" adds the samples in the order of renderBlock / ImageBlock::put / BlockGenerator (src/main.cpp:33-53,
   src/block.cpp:62-152): the frame of a single-threaded run of the reference's loops, bit for bit, from one GPU or several */
static void applyEnvironmentOptions(nori_hip_ctx *ctx) {
    if (const char *e = std::getenv("NORI_FILM_ORDER"))
        if (nori_hip_set_option(ctx, "film_order", e) != NORI_OK)
            throw NoriException("NORI_FILM_ORDER / --film-order expects \"fast\" or \"reference\", got \"%s\"", e);
}

Device::Device(int device) {
    if (device < 0) {
        const char *e = std::getenv("NORI_DEVICE");
        device = e ? std::atoi(e) : 0;
    }
    m_device = device;
    /* the structs the library fills carry no size field: header and library must be of one version (include/nori_hip.h) */
    if (nori_hip_abi_version() != NORI_HIP_ABI_VERSION)
        throw NoriException("libnori_hip reports ABI version %i, this host was compiled against %i (include/nori_hip.h): rebuild both",
                            nori_hip_abi_version(), NORI_HIP_ABI_VERSION);
    int rc = nori_hip_create(device, &m_ctx);
    if (rc != NORI_OK)
        throw NoriException("Unable to create a HIP context on device %i: %s (the MI355X path has no CPU fallback)",
                            device, nori_hip_last_error(nullptr));
    try { applyEnvironmentOptions(m_ctx); } catch (...) { nori_hip_destroy(m_ctx); m_ctx = nullptr; throw; }
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
    try {
        for (int k = 0; k < n; ++k) applyEnvironmentOptions(nori_hip_group_ctx(m_group, k));
    } catch (...) { nori_hip_group_destroy(m_group); m_group = nullptr; throw; }
}

DeviceGroup::~DeviceGroup() { nori_hip_group_destroy(m_group); }

void DeviceGroup::check(int rc, const char *what) const {
    if (rc != NORI_OK) throw NoriException("%s failed (%i): %s", what, rc, nori_hip_group_last_error(m_group));
}

NORI_NAMESPACE_END
