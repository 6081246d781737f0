/*
 * warptest.cpp -- command-line chi^2 test of the warps:
 *     warptest <name> [param] [param2]        exit code 0 = pass, 1 = fail
 * CLI mode of the reference's src/warptest.cpp:956-995 (the GUI mode needs a
 * display and is out of scope).  Names as in warptest.cpp:79-82: square tent
 * disk uniform_sphere uniform_hemisphere cosine_hemisphere beckmann
 * microfacet_brdf.  Test recipe of WarpTest::run (warptest.cpp:109-215):
 * 51x51 bins (102x51 on the sphere), 1000 samples per bin, expected counts by
 * integrating the pdf over each bin, chi2_test(minExpFrequency 5, alpha 0.01).
 * Warp points and pdf values are computed ON THE DEVICE (Warp::warpBatch /
 * pdfBatch, BSDF::sampleBatch / pdfBatch).
 */
#include <nori/hypothesis.h>
#include <nori/plugins.h>

using namespace nori;

static const char *kNames[] = {"square", "tent", "disk", "uniform_sphere", "uniform_hemisphere",
                               "cosine_hemisphere", "beckmann", "microfacet_brdf"};

int main(int argc, char **argv) {
    if (argc <= 1) {
        cerr << "Syntax: " << argv[0] << " <warp> [param] [param2]   (the GUI mode is not available)" << endl;
        return -1;
    }
    int type = -1;
    for (int i = 0; i < 8; ++i) if (std::string(argv[1]) == kNames[i]) type = i;
    if (type < 0) { cerr << "Unknown warp \"" << argv[1] << "\"" << endl; return -1; }
    float param = argc > 2 ? std::stof(argv[2]) : 0.f, param2 = argc > 3 ? std::stof(argv[3]) : 0.f;
    try {
        std::unique_ptr<BSDF> bsdf;
        Vector3f wi;
        if (type == 7) {        /* create_microfacet_bsdf(alpha, kd, angle 0), warptest.cpp:289-301 */
            PropertyList list;
            list.setFloat("alpha", param);
            list.setColor("kd", Color3f(param2));
            bsdf.reset((BSDF *) NoriObjectFactory::createInstance("microfacet", list));
            wi = Vector3f(std::sin(0.f), 0.f, std::max(std::cos(0.f), 1e-4f)).normalized();
        }
        cout << format("Testing warp %s, parameter value = %f%s", kNames[type], param,
                       param2 > 0 ? format(", second parameter value = %f", param2) : std::string()) << endl;

        const bool planar = type <= 2;
        int xres = 51, yres = 51;
        if (!planar) xres *= 2;
        const int res = xres * yres;
        const size_t n = (size_t) 1000 * res;

        /* 1. sample points (pcg32 stream seed(42, 54), the PCG demo seed) */
        std::vector<float> u(2 * n), pts(3 * n), w(3 * n, 1.0f);
        { Device &d = Device::shared(); uint64_t st = 42, sq = 54;
          d.check(nori_hip_pcg32_floats(d.ctx(), &st, &sq, 1, (uint32_t) (2 * n), u.data()), "nori_hip_pcg32_floats"); }
        if (type == 7) {
            std::vector<float> wis(3 * n);
            for (size_t k = 0; k < n; ++k) { wis[3 * k] = wi.x(); wis[3 * k + 1] = wi.y(); wis[3 * k + 2] = wi.z(); }
            bsdf->sampleBatch(wis.data(), u.data(), n, pts.data(), w.data(), nullptr, nullptr);
        } else {
            Warp::warpBatch((nori_warp_type) type, param, u.data(), n, pts.data());
        }
        std::vector<double> obs(res, 0.0), expd(res, 0.0);
        for (size_t i = 0; i < n; ++i) {
            if (w[3 * i] == 0) continue;
            float x, y;
            const float *s = &pts[3 * i];
            if (type == 0) { x = s[0]; y = s[1]; }
            else if (planar) { x = s[0] * 0.5f + 0.5f; y = s[1] * 0.5f + 0.5f; }
            else { x = std::atan2(s[1], s[0]) * INV_TWOPI; if (x < 0) x += 1; y = s[2] * 0.5f + 0.5f; }
            int xbin = std::min(xres - 1, std::max(0, (int) std::floor(x * xres)));
            int ybin = std::min(yres - 1, std::max(0, (int) std::floor(y * yres)));
            obs[ybin * xres + xbin] += 1;
        }

        /* 2. expected counts: integrate the pdf over each bin (composite Simpson,
              refined per bin until it settles; all evaluations batched on the device) */
        double scale = (double) n * (type == 0 ? 1.0 : (planar ? 4.0 : 4.0 * (double) M_PI));
        std::vector<double> prev(res, 0.0);
        std::vector<int> active(res);
        for (int i = 0; i < res; ++i) active[i] = i;
        for (int panels = 4; panels <= 128 && !active.empty(); panels *= 2) {
            const int p1 = panels + 1;
            const size_t per = (size_t) p1 * p1, m = per * active.size();
            std::vector<float> q(3 * m), pdf(m), wis;
            for (size_t a = 0; a < active.size(); ++a) {
                const int bin = active[a], by = bin / xres, bx = bin % xres;
                for (int iy = 0; iy < p1; ++iy)
                    for (int ix = 0; ix < p1; ++ix) {
                        double x = (bx + (double) ix / panels) / xres, y = (by + (double) iy / panels) / yres;
                        float *o = &q[3 * (a * per + (size_t) iy * p1 + ix)];
                        if (type == 0) { o[0] = (float) x; o[1] = (float) y; o[2] = 0; }
                        else if (planar) { o[0] = (float) (x * 2 - 1); o[1] = (float) (y * 2 - 1); o[2] = 0; }
                        else {
                            x *= 2 * (double) M_PI; y = y * 2 - 1;
                            double st = std::sqrt(std::max(0.0, 1 - y * y));
                            o[0] = (float) (st * std::cos(x)); o[1] = (float) (st * std::sin(x)); o[2] = (float) y;
                        }
                    }
            }
            if (type == 7) {
                wis.resize(3 * m);
                for (size_t k = 0; k < m; ++k) { wis[3 * k] = wi.x(); wis[3 * k + 1] = wi.y(); wis[3 * k + 2] = wi.z(); }
                bsdf->pdfBatch(wis.data(), q.data(), m, pdf.data());
            } else {
                Warp::pdfBatch((nori_warp_type) type, param, q.data(), m, pdf.data());
            }
            std::vector<int> still;
            for (size_t a = 0; a < active.size(); ++a) {
                double sum = 0;
                for (int iy = 0; iy < p1; ++iy) {
                    const double wy = (iy == 0 || iy == panels) ? 1 : ((iy & 1) ? 4 : 2);
                    for (int ix = 0; ix < p1; ++ix) {
                        const double wx = (ix == 0 || ix == panels) ? 1 : ((ix & 1) ? 4 : 2);
                        const float v = pdf[a * per + (size_t) iy * p1 + ix];
                        if (v < 0) throw NoriException("The Pdf() function returned negative values!");
                        sum += wx * wy * (double) v;
                    }
                }
                const double h = (1.0 / xres / panels) * (1.0 / yres / panels);
                const double integral = sum * h / 9.0 * scale;
                const int bin = active[a];
                expd[bin] = integral;
                if (panels > 4 && std::fabs(integral - prev[bin]) <= 1e-4 + 1e-5 * std::fabs(integral)) continue;
                prev[bin] = integral;
                still.push_back(bin);
            }
            active.swap(still);
        }
        auto result = hypothesis::chi2_test(res, obs.data(), expd.data(), (double) n, 5, 0.01f, 1);
        cout << result.second << endl;
        if (result.first) return 0;
        cout << format("warptest failed: %s", result.second) << endl;
        return 1;
    } catch (const std::exception &e) {
        cerr << e.what() << endl;
        return -1;
    }
}
