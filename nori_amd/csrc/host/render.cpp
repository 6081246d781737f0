/*
 * render.cpp -- ImageBlock (host view) and the render driver.
 *
 * renderScene() stands where render() of the reference's src/main.cpp:58-148
 * does: instead of tbb::parallel_for over 32x32 blocks calling renderBlock
 * (main.cpp:27-56) per block, one call hands the whole frame to
 * nori_hip_render -- block scheduling, sampling, Li and the filtered splat all
 * run in the render kernel -- and the full-frame RGBW block comes back for
 * toBitmap (src/block.cpp:45-51).
 *
 * NORI_GPUS=N in the environment (`nori scene.xml --gpus N`) shares the frame over the first N GPUs of the node: the
 * blocks the reference's parallel_for hands to TBB workers (src/main.cpp:85-113) go to devices -- 16x16 tiles round
 * robin (NORI_SPLIT=tile, default) or a range of the sample indices each (NORI_SPLIT=sample) -- and the mutex-guarded
 * ImageBlock::put(ImageBlock&) (src/block.cpp:93-102) becomes one RCCL merge on the first device (NORI_MERGE=reduce,
 * default, or gather): nori_hip_group_render_host, one host thread and one context per device inside the library.
 */
#include <nori/bitmap.h>
#include <nori/plugins.h>

NORI_NAMESPACE_BEGIN

ImageBlock::ImageBlock(const Vector2i &size, const ReconstructionFilter *filter) : m_offset(0, 0), m_size(size) {
    if (filter) m_borderSize = (int) std::ceil(filter->getRadius() - 0.5f);     /* src/block.cpp:20 */
    m_data.assign((size_t) rows() * cols() * 4, 0.0f);
}

Bitmap *ImageBlock::toBitmap() const {
    Bitmap *result = new Bitmap(m_size);
    for (int y = 0; y < m_size.y(); ++y)
        for (int x = 0; x < m_size.x(); ++x) {
            const float *p = &m_data[((size_t) (y + m_borderSize) * cols() + (x + m_borderSize)) * 4];
            /* Color4f::divideByFilterWeight, include/nori/color.h:100-105 */
            if (p[3] != 0) result->set(y, x, Color3f(p[0] / p[3], p[1] / p[3], p[2] / p[3]));
            else result->set(y, x, Color3f(0.0f));
        }
    return result;
}

void ImageBlock::put(const ImageBlock &b) {
    if (b.rows() != rows() || b.cols() != cols()) throw NoriException("ImageBlock::put(): block sizes differ");
    for (size_t i = 0; i < m_data.size(); ++i) m_data[i] += b.m_data[i];
}

std::string ImageBlock::toString() const { return format("ImageBlock[offset=%s, size=%s]]", m_offset.toString(), m_size.toString()); }

std::unique_ptr<ImageBlock> renderScene(Scene *scene, nori_render_stats *stats) {
    const Camera *camera = scene->getCamera();
    scene->getIntegrator()->preprocess(scene);
    std::unique_ptr<ImageBlock> result(new ImageBlock(camera->getOutputSize(), camera->getReconstructionFilter()));
    result->clear();

    nori_render_params params;
    std::memset(&params, 0, sizeof(params));
    params.spp_begin = 0;
    params.spp_count = (uint32_t) scene->getSampler()->getSampleCount();
    params.tile_mod = 1; params.tile_rem = 0;
    /* sampler streams: one pcg32 stream per camera sample (default), or -- NORI_SEED=block in the environment, `--seed block`
       on the nori command line -- the reference's own scheme, one serial stream per 32x32 block seeded by
       Independent::prepare (src/independent.cpp:36-41), which the device then runs one lane per block */
    const char *seed = std::getenv("NORI_SEED");
    params.seed_mode = (seed && std::string(seed) == "block") ? NORI_SEED_NORI_BLOCK : NORI_SEED_PER_SAMPLE;
    nori_render_stats local;
    if (const char *gpus = std::getenv("NORI_GPUS")) {
        const int n = std::atoi(gpus);
        if (n < 1) throw NoriException("NORI_GPUS / --gpus expects a positive number of GPUs, got \"%s\"", gpus);
        const char *sp = std::getenv("NORI_SPLIT"), *mg = std::getenv("NORI_MERGE");
        const std::string split = sp ? sp : "tile", merge = mg ? mg : "reduce";
        if (split != "tile" && split != "sample") throw NoriException("NORI_SPLIT / --split expects \"tile\" or \"sample\", got \"%s\"", split);
        if (merge != "reduce" && merge != "gather") throw NoriException("NORI_MERGE / --merge expects \"reduce\" or \"gather\", got \"%s\"", merge);
        DeviceGroup &grp = scene->deviceGroup(n);
        float merge_ms = 0.0f;
        grp.check(nori_hip_group_render_host(grp.group(), &params, split == "sample" ? NORI_SPLIT_SAMPLE : NORI_SPLIT_TILE,
                                             merge == "gather" ? NORI_MERGE_GATHER : NORI_MERGE_REDUCE, camera->getOutputSize().x(), camera->getOutputSize().y(),
                                             result->data(), stats ? stats : &local, &merge_ms), "nori_hip_group_render_host");
        if (Scene::s_verbose) cout << "[" << n << " GPUs, " << split << " split, " << merge << " merge over " << nori_hip_group_transport(grp.group()) << ": " << merge_ms << " ms] ";
        return result;
    }
    Device &dev = scene->device();          /* uploads the scene and builds the BVH on first use */
    dev.check(nori_hip_render_host(dev.ctx(), &params, result->data(), stats ? stats : &local), "nori_hip_render_host");
    return result;
}

NORI_NAMESPACE_END
