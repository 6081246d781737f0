/*
 * nori_host.h -- C ABI of libnori_host.so, the C++ host that keeps Nori's
 * NoriObject / XML plugin surface (loadFromXML, src/parser.cpp:16).  It lets a
 * non-C++ caller (the Python tests and bench.py) load an unmodified Nori scene
 * file and obtain the flattened nori_scene_desc that include/nori_hip.h
 * consumes, or enumerate the contents of a <test> file.
 *
 * Load-time only; no per-sample work happens behind these calls.
 */
#ifndef NORI_HOST_H
#define NORI_HOST_H

#include "nori_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nori_host_root nori_host_root;

/* flags for nori_host_load_xml */
#define NORI_HOST_DEFER_TESTS 1   /* do not run <test> objects inside activate() */
#define NORI_HOST_QUIET 2         /* no "Configuration: ..." / OBJ loader chatter  */

/* Parse `path` (resources resolve relative to its directory, src/main.cpp:190).
 * Returns 0 or a negative nori_status; message via nori_host_last_error(). */
int nori_host_load_xml(const char *path, int flags, nori_host_root **out);
void nori_host_free(nori_host_root *root);
const char *nori_host_last_error(void);

/* NoriObject::EClassType of the root (0 = scene, 9 = test, ...) and its toString() */
int nori_host_root_type(const nori_host_root *root);
const char *nori_host_root_string(const nori_host_root *root);

/* Root is a <scene>: borrow its flattened description (valid until free). */
int nori_host_scene_desc(const nori_host_root *root, nori_scene_desc *out);

/* Root is a <test>: kind 0 = ttest, 1 = chi2test. */
typedef struct nori_host_test_info {
    int32_t kind;
    float significance_level;
    int32_t sample_count;
    int32_t test_count;          /* chi2test */
    int32_t resolution;          /* chi2test: cos(theta) cells; phi cells = 2x */
    int32_t min_exp_frequency;   /* chi2test */
    uint32_t n_angles, n_references, n_bsdfs, n_scenes;
    const float *angles;
    const float *references;
} nori_host_test_info;
int nori_host_test_info_get(const nori_host_root *root, nori_host_test_info *out);
int nori_host_test_bsdf(const nori_host_root *root, uint32_t index, nori_bsdf_desc *out);
int nori_host_test_scene_desc(const nori_host_root *root, uint32_t index, nori_scene_desc *out);
/* Run the test on the GPU now (what activate() does when not deferred);
 * 0 = all passed, 1 = some failed (report on stdout), < 0 = error. */
int nori_host_test_run(nori_host_root *root);

/* render() of src/main.cpp:58-148 for a <scene> root on the GPU: fills the
 * caller's RGBW frame ((h+2b) x (w+2b) x 4 floats) and optional stats. */
int nori_host_render(nori_host_root *root, float *rgbw, nori_render_stats *stats);
/* Write rgb (h x w x 3 floats) as `<basename>.exr` and `<basename>.png`
 * (Bitmap::saveEXR / savePNG, src/bitmap.cpp:69-122). */
int nori_host_save_images(const char *basename, const float *rgb, int width, int height);
/* Read an OpenEXR file written by Nori / this library into rgb (caller frees
 * with nori_host_free_buffer). */
int nori_host_load_exr(const char *path, float **rgb, int *width, int *height);
void nori_host_free_buffer(void *p);

#ifdef __cplusplus
}
#endif
#endif
