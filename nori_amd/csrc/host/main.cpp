/*
 * main.cpp -- `nori <scene.xml> [--no-gui] [--threads N] [--seed sample|block] [--gpus N] [--split tile|sample] [--merge reduce|gather] [--film-order fast|reference]`
 * Command line of the reference (src/main.cpp:150-246).  There is no GUI on a
 * compute node: --no-gui is accepted and implied; --threads is accepted for
 * compatibility (the work runs on the GPU).  --seed block renders with the
 * reference's sampler streams (one pcg32 stream per 32x32 block,
 * src/independent.cpp:36-41; one GPU lane per block, slow by design) instead of
 * one stream per camera sample.  --gpus N shares the frame over the first N GPUs
 * of the node (render.cpp: what TBB workers are in the reference; --split / --merge
 * choose how the work is cut and how the frames come together).  A <test> root runs during parsing
 * (its activate()), as in the reference; failures exit with -1.
 */
#include <nori/bitmap.h>
#include <nori/plugins.h>

using namespace nori;

int main(int argc, char **argv) {
    if (argc < 2) {
        cerr << "Syntax: " << argv[0] << " <scene.xml> [--no-gui] [--threads N] [--seed sample|block] [--gpus N] [--split tile|sample] [--merge reduce|gather] [--film-order fast|reference]" << endl;
        return -1;
    }
    std::string sceneName;
    for (int i = 1; i < argc; ++i) {
        std::string token(argv[i]);
        if (token == "-t" || token == "--threads") {
            if (i + 1 >= argc || atoi(argv[i + 1]) <= 0) {
                cerr << "\"--threads\" argument expects a positive integer following it." << endl;
                return -1;
            }
            ++i;
            continue;
        } else if (token == "--no-gui") {
            continue;
        } else if (token == "--seed") {
            if (i + 1 >= argc || (std::string(argv[i + 1]) != "sample" && std::string(argv[i + 1]) != "block")) {
                cerr << "\"--seed\" expects \"sample\" or \"block\"." << endl;
                return -1;
            }
            setenv("NORI_SEED", argv[++i], 1);
            continue;
        } else if (token == "--gpus") {
            if (i + 1 >= argc || atoi(argv[i + 1]) <= 0) {
                cerr << "\"--gpus\" argument expects a positive integer following it." << endl;
                return -1;
            }
            setenv("NORI_GPUS", argv[++i], 1);
            continue;
        } else if (token == "--split" || token == "--merge" || token == "--film-order") {
            if (i + 1 >= argc) {
                cerr << "\"" << token << "\" expects a value." << endl;
                return -1;
            }
            setenv(token == "--split" ? "NORI_SPLIT" : token == "--merge" ? "NORI_MERGE" : "NORI_FILM_ORDER", argv[++i], 1);
            continue;
        }
        if (endsWith(token, ".xml")) {
            sceneName = token;
            size_t slash = token.find_last_of('/');
            getFileResolver()->prepend(slash == std::string::npos ? std::string(".") : token.substr(0, slash));
        } else if (endsWith(token, ".exr")) {
            cerr << "The EXR viewer needs a display; this build has no GUI." << endl;
            return -1;
        } else {
            cerr << "Fatal error: unknown file \"" << token << "\", expected an extension of type .xml or .exr" << endl;
        }
    }
    if (sceneName.empty()) {
        cerr << "Please provide the path to a .xml (or .exr) file." << endl;
        return -1;
    }
    try {
        std::unique_ptr<NoriObject> root(loadFromXML(sceneName));
        if (root->getClassType() == NoriObject::EScene) {
            Scene *scene = static_cast<Scene *>(root.get());
            cout << "Rendering .. ";
            cout.flush();
            Timer timer;
            nori_render_stats st;
            std::unique_ptr<ImageBlock> result = renderScene(scene, &st);
            cout << "done. (took " << timer.elapsedString() << "; kernel " << timeString(st.kernel_ms, true) << ", "
                 << (double) (st.n_closest_rays + st.n_shadow_rays) / (st.kernel_ms * 1e3) << " Mrays/s)" << endl;
            std::unique_ptr<Bitmap> bitmap(result->toBitmap());
            std::string outputName = sceneName;
            size_t lastdot = outputName.find_last_of(".");
            if (lastdot != std::string::npos) outputName.erase(lastdot, std::string::npos);
            bitmap->saveEXR(outputName);
            bitmap->savePNG(outputName);
        }
    } catch (const std::exception &e) {
        cerr << e.what() << endl;
        return -1;
    }
    return 0;
}
