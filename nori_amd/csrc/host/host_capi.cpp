/* host_capi.cpp -- implementation of include/nori_host.h. */
#include <nori/bitmap.h>
#include <nori/plugins.h>
#include <nori/testobjects.h>

#include "../../../include/nori_host.h"

using namespace nori;

struct nori_host_root {
    std::unique_ptr<NoriObject> root;
    std::string str;
};

static thread_local std::string g_error;

template <typename F> static int guarded(F &&f) {
    try {
        return f();
    } catch (const std::exception &e) {
        g_error = e.what();
        return NORI_ERR_INVALID_ARGUMENT;
    }
}

extern "C" {

const char *nori_host_last_error(void) { return g_error.c_str(); }

int nori_host_load_xml(const char *path, int flags, nori_host_root **out) {
    if (!path || !out) return NORI_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    return guarded([&] {
        std::string p(path);
        size_t slash = p.find_last_of('/');
        getFileResolver()->prepend(slash == std::string::npos ? std::string(".") : p.substr(0, slash));
        const bool defer = TestBase::s_defer, verbose = Scene::s_verbose;
        TestBase::s_defer = (flags & NORI_HOST_DEFER_TESTS) != 0;
        Scene::s_verbose = (flags & NORI_HOST_QUIET) == 0;
        std::unique_ptr<nori_host_root> r(new nori_host_root());
        try {
            r->root.reset(loadFromXML(p));
        } catch (...) {
            TestBase::s_defer = defer; Scene::s_verbose = verbose;
            throw;
        }
        TestBase::s_defer = defer; Scene::s_verbose = verbose;
        r->str = r->root->toString();
        *out = r.release();
        return (int) NORI_OK;
    });
}

void nori_host_free(nori_host_root *root) { delete root; }

int nori_host_root_type(const nori_host_root *root) { return root ? (int) root->root->getClassType() : -1; }
const char *nori_host_root_string(const nori_host_root *root) { return root ? root->str.c_str() : ""; }

int nori_host_scene_desc(const nori_host_root *root, nori_scene_desc *out) {
    if (!root || !out || root->root->getClassType() != NoriObject::EScene) return NORI_ERR_INVALID_ARGUMENT;
    return guarded([&] { *out = static_cast<const Scene *>(root->root.get())->getDesc(); return (int) NORI_OK; });
}

int nori_host_test_info_get(const nori_host_root *root, nori_host_test_info *out) {
    if (!root || !out || root->root->getClassType() != NoriObject::ETest) return NORI_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    if (auto *t = dynamic_cast<const StudentsTTest *>(root->root.get())) {
        out->kind = 0; out->significance_level = t->m_significanceLevel; out->sample_count = t->m_sampleCount;
        out->n_angles = (uint32_t) t->m_angles.size(); out->n_references = (uint32_t) t->m_references.size();
        out->n_bsdfs = (uint32_t) t->m_bsdfs.size(); out->n_scenes = (uint32_t) t->m_scenes.size();
        out->angles = t->m_angles.data(); out->references = t->m_references.data();
        return NORI_OK;
    }
    if (auto *t = dynamic_cast<const ChiSquareTest *>(root->root.get())) {
        out->kind = 1; out->significance_level = t->m_significanceLevel; out->sample_count = t->m_sampleCount;
        out->test_count = t->m_testCount; out->resolution = t->m_cosThetaResolution; out->min_exp_frequency = t->m_minExpFrequency;
        out->n_bsdfs = (uint32_t) t->m_bsdfs.size();
        return NORI_OK;
    }
    return NORI_ERR_UNSUPPORTED;
}

int nori_host_test_bsdf(const nori_host_root *root, uint32_t index, nori_bsdf_desc *out) {
    if (!root || !out) return NORI_ERR_INVALID_ARGUMENT;
    const std::vector<BSDF *> *v = nullptr;
    if (auto *t = dynamic_cast<const StudentsTTest *>(root->root.get())) v = &t->m_bsdfs;
    if (auto *t = dynamic_cast<const ChiSquareTest *>(root->root.get())) v = &t->m_bsdfs;
    if (!v || index >= v->size()) return NORI_ERR_INVALID_ARGUMENT;
    (*v)[index]->fill(*out);
    return NORI_OK;
}

int nori_host_test_scene_desc(const nori_host_root *root, uint32_t index, nori_scene_desc *out) {
    if (!root || !out) return NORI_ERR_INVALID_ARGUMENT;
    auto *t = dynamic_cast<const StudentsTTest *>(root->root.get());
    if (!t || index >= t->m_scenes.size()) return NORI_ERR_INVALID_ARGUMENT;
    return guarded([&] { *out = t->m_scenes[index]->getDesc(); return (int) NORI_OK; });
}

int nori_host_test_run(nori_host_root *root) {
    if (!root) return NORI_ERR_INVALID_ARGUMENT;
    auto *t = dynamic_cast<TestBase *>(root->root.get());
    if (!t) return NORI_ERR_INVALID_ARGUMENT;
    try {
        t->run();
        return 0;
    } catch (const NoriException &e) {
        g_error = e.what();
        return NORI_ERR_INTERNAL;
    } catch (const std::runtime_error &e) {   /* "Some tests failed :(" */
        g_error = e.what();
        return 1;
    }
}

int nori_host_render(nori_host_root *root, float *rgbw, nori_render_stats *stats) {
    if (!root || !rgbw || root->root->getClassType() != NoriObject::EScene) return NORI_ERR_INVALID_ARGUMENT;
    return guarded([&] {
        std::unique_ptr<ImageBlock> b = renderScene(static_cast<Scene *>(root->root.get()), stats);
        std::memcpy(rgbw, b->data(), sizeof(float) * (size_t) b->rows() * b->cols() * 4);
        return (int) NORI_OK;
    });
}

int nori_host_save_images(const char *basename, const float *rgb, int width, int height) {
    if (!basename || !rgb || width <= 0 || height <= 0) return NORI_ERR_INVALID_ARGUMENT;
    return guarded([&] {
        Bitmap bmp(Vector2i(width, height));
        std::memcpy(bmp.data(), rgb, sizeof(float) * (size_t) width * height * 3);
        bmp.saveEXR(basename);
        bmp.savePNG(basename);
        return (int) NORI_OK;
    });
}

int nori_host_load_exr(const char *path, float **rgb, int *width, int *height) {
    if (!path || !rgb || !width || !height) return NORI_ERR_INVALID_ARGUMENT;
    return guarded([&] {
        Bitmap bmp{std::string(path)};
        *width = bmp.cols(); *height = bmp.rows();
        size_t n = (size_t) bmp.cols() * bmp.rows() * 3;
        *rgb = (float *) std::malloc(sizeof(float) * n);
        std::memcpy(*rgb, bmp.data(), sizeof(float) * n);
        return (int) NORI_OK;
    });
}

void nori_host_free_buffer(void *p) { std::free(p); }

} // extern "C"
