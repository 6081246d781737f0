/*
 * tests.cpp -- <test type="ttest"> and <test type="chi2test">: the reference's
 * statistical test objects (src/ttest.cpp:45-206, src/chi2test.cpp:28-213),
 * running the BSDF / Li code that is under test ON THE DEVICE through the
 * batched twins of the plugin virtuals.  Parameters, defaults, pass/fail
 * logic, console output shape and the "Some tests failed :(" exception follow
 * the reference; the statistics come from nori/hypothesis.h.
 *
 * As in the reference the tests execute inside activate(), i.e. while the XML
 * is being parsed.  TestBase::s_defer = true turns that off so a test file can
 * be loaded and inspected without a GPU (used by the C API for Python).
 */
#include <nori/hypothesis.h>
#include <nori/plugins.h>
#include <nori/testobjects.h>

NORI_NAMESPACE_BEGIN

bool TestBase::s_defer = false;

static void welford(const float *rgb, size_t n, double &mean, double &variance) {
    /* src/ttest.cpp:118-126: Knuth's online variance on the luminance */
    mean = 0; variance = 0;
    for (size_t k = 0; k < n; ++k) {
        double result = (double) Color3f(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]).getLuminance();
        double delta = result - mean;
        mean += delta / (double) (k + 1);
        variance += delta * (result - mean);
    }
    variance /= (double) n - 1;
}

/* pcg32 floats from the device; the reference's test objects use a
   default-constructed pcg32; here stream pcg32.seed(42, 54 + k) (the PCG demo
   seed), successive draws using successive streams */
static std::vector<float> deviceFloats(uint64_t stream, size_t count) {
    std::vector<float> out(count);
    Device &d = Device::shared();
    const size_t kChunk = 1u << 20;
    uint64_t st = 42u;
    for (size_t done = 0, part = 0; done < count; done += kChunk, ++part) {
        uint64_t sq = 54u + (stream << 20) + part;
        size_t n = std::min(kChunk, count - done);
        d.check(nori_hip_pcg32_floats(d.ctx(), &st, &sq, 1, (uint32_t) n, out.data() + done), "nori_hip_pcg32_floats");
    }
    return out;
}

/* ================================================================ t-test */
StudentsTTest::StudentsTTest(const PropertyList &propList) {
    m_significanceLevel = propList.getFloat("significanceLevel", 0.01f);
    for (auto a : tokenize(propList.getString("angles", ""))) m_angles.push_back(toFloat(a));
    for (auto r : tokenize(propList.getString("references", ""))) m_references.push_back(toFloat(r));
    m_sampleCount = propList.getInteger("sampleCount", 100000);
}

StudentsTTest::~StudentsTTest() {
    for (auto b : m_bsdfs) delete b;
    for (auto s : m_scenes) delete s;
}

void StudentsTTest::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
    case EBSDF: m_bsdfs.push_back(static_cast<BSDF *>(obj)); break;
    case EScene: m_scenes.push_back(static_cast<Scene *>(obj)); break;
    default: throw NoriException("StudentsTTest::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()));
    }
}

void StudentsTTest::activate() { if (!s_defer) run(); }

void StudentsTTest::run() {
    int total = 0, passed = 0;
    uint64_t stream = 0;
    const size_t n = (size_t) m_sampleCount;
    if (!m_bsdfs.empty()) {
        if (m_references.size() * m_bsdfs.size() != m_angles.size())
            throw NoriException("Specified a different number of angles and reference values!");
        if (!m_scenes.empty()) throw NoriException("Cannot test BSDFs and scenes at the same time!");
        int ctr = 0;
        for (auto bsdf : m_bsdfs) {
            for (size_t i = 0; i < m_references.size(); ++i) {
                float angle = m_angles[i], reference = m_references[ctr++];
                cout << "------------------------------------------------------" << endl;
                cout << "Testing (angle=" << angle << "): " << bsdf->toString() << endl;
                ++total;
                Vector3f wi = sphericalDirection(degToRad(angle), 0);
                cout << "Drawing " << m_sampleCount << " samples .. " << endl;
                std::vector<float> wis(3 * n), wo(3 * n), weight(3 * n);
                for (size_t k = 0; k < n; ++k) { wis[3 * k] = wi.x(); wis[3 * k + 1] = wi.y(); wis[3 * k + 2] = wi.z(); }
                std::vector<float> samples = deviceFloats(stream++, 2 * n);
                bsdf->sampleBatch(wis.data(), samples.data(), n, wo.data(), weight.data(), nullptr, nullptr);
                double mean, variance;
                welford(weight.data(), n, mean, variance);
                auto result = hypothesis::students_t_test(mean, variance, reference, m_sampleCount, m_significanceLevel, (int) m_references.size());
                if (result.first) ++passed;
                cout << result.second << endl;
            }
        }
    } else {
        if (m_references.size() != m_scenes.size())
            throw NoriException("Specified a different number of scenes and reference values!");
        int ctr = 0;
        for (auto scene : m_scenes) {
            const Integrator *integrator = scene->getIntegrator();
            const Camera *camera = scene->getCamera();
            float reference = m_references[ctr++];
            cout << "------------------------------------------------------" << endl;
            cout << "Testing scene: " << scene->toString() << endl;
            ++total;
            cout << "Generating " << m_sampleCount << " paths.. " << endl;
            /* src/ttest.cpp:155-164: pixelSample = next2D * outputSize, aperture next2D, sampleRay, Li */
            std::vector<float> u = deviceFloats(stream++, 2 * n);
            const Vector2i size = camera->getOutputSize();
            for (size_t k = 0; k < n; ++k) { u[2 * k] *= (float) size.x(); u[2 * k + 1] *= (float) size.y(); }
            std::vector<nori_ray> rays(n);
            Device &d = scene->device();
            d.check(nori_hip_sample_rays(d.ctx(), u.data(), n, rays.data()), "nori_hip_sample_rays");
            std::vector<uint64_t> st(n, 42u + stream), sq(n);
            for (size_t k = 0; k < n; ++k) sq[k] = 54u + k;
            std::vector<float> rgb(3 * n);
            integrator->LiBatch(scene, rays.data(), n, st.data(), sq.data(), rgb.data());
            double mean, variance;
            welford(rgb.data(), n, mean, variance);
            auto result = hypothesis::students_t_test(mean, variance, reference, m_sampleCount, m_significanceLevel, (int) m_references.size());
            if (result.first) ++passed;
            cout << result.second << endl;
        }
    }
    cout << "Passed " << passed << "/" << total << " tests." << endl;
    if (passed < total) throw std::runtime_error("Some tests failed :(");
}

std::string StudentsTTest::toString() const {
    return format("StudentsTTest[\n  significanceLevel = %f,\n  sampleCount= %i\n]", m_significanceLevel, m_sampleCount);
}
NORI_REGISTER_CLASS(StudentsTTest, "ttest");

/* ============================================================== chi^2 test */
ChiSquareTest::ChiSquareTest(const PropertyList &propList) {
    m_significanceLevel = propList.getFloat("significanceLevel", 0.01f);
    m_cosThetaResolution = propList.getInteger("resolution", 10);
    m_minExpFrequency = propList.getInteger("minExpFrequency", 5);
    m_sampleCount = propList.getInteger("sampleCount", -1);
    m_testCount = propList.getInteger("testCount", 5);
    m_phiResolution = 2 * m_cosThetaResolution;
    if (m_sampleCount < 0) m_sampleCount = m_cosThetaResolution * m_phiResolution * 5000;
}

ChiSquareTest::~ChiSquareTest() { for (auto b : m_bsdfs) delete b; }

void ChiSquareTest::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
    case EBSDF: m_bsdfs.push_back(static_cast<BSDF *>(obj)); break;
    default: throw NoriException("ChiSquareTest::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()));
    }
}

void ChiSquareTest::activate() { if (!s_defer) run(); }

/* Integral of pdf(wi, .) over every (cosTheta, phi) bin: composite Simpson per
   bin with the panel count doubled until the bin's estimate settles (stands in
   for hypothesis::adaptiveSimpson2D, src/chi2test.cpp:152-160); all pdf
   evaluations of one refinement level go to the device as one batch. */
void integratePdfOverBins(const BSDF *bsdf, const Vector3f &wi, int thetaRes, int phiRes, std::vector<double> &out) {
    const int nBins = thetaRes * phiRes;
    out.assign(nBins, 0.0);
    std::vector<double> prev(nBins, 0.0);
    std::vector<int> active(nBins);
    for (int i = 0; i < nBins; ++i) active[i] = i;
    for (int panels = 8; panels <= 256 && !active.empty(); panels *= 2) {
        const int pts = panels + 1;
        const size_t per = (size_t) pts * pts, n = per * active.size();
        std::vector<float> wis(3 * n), wos(3 * n), pdf(n);
        for (size_t a = 0; a < active.size(); ++a) {
            const int bin = active[a], i = bin / phiRes, j = bin % phiRes;
            const double c0 = -1.0 + i * 2.0 / thetaRes, c1 = -1.0 + (i + 1) * 2.0 / thetaRes;
            const double p0 = j * 2 * (double) M_PI / phiRes, p1 = (j + 1) * 2 * (double) M_PI / phiRes;
            for (int y = 0; y < pts; ++y)
                for (int x = 0; x < pts; ++x) {
                    const double cosTheta = c0 + (c1 - c0) * y / panels, phi = p0 + (p1 - p0) * x / panels;
                    const double sinTheta = std::sqrt(std::max(0.0, 1 - cosTheta * cosTheta));
                    const size_t k = a * per + (size_t) y * pts + x;
                    wos[3 * k] = (float) (sinTheta * std::cos(phi)); wos[3 * k + 1] = (float) (sinTheta * std::sin(phi)); wos[3 * k + 2] = (float) cosTheta;
                    wis[3 * k] = wi.x(); wis[3 * k + 1] = wi.y(); wis[3 * k + 2] = wi.z();
                }
        }
        bsdf->pdfBatch(wis.data(), wos.data(), n, pdf.data());
        std::vector<int> still;
        for (size_t a = 0; a < active.size(); ++a) {
            const int bin = active[a];
            double sum = 0;
            for (int y = 0; y < pts; ++y) {
                const double wy = (y == 0 || y == panels) ? 1 : ((y & 1) ? 4 : 2);
                for (int x = 0; x < pts; ++x) {
                    const double wx = (x == 0 || x == panels) ? 1 : ((x & 1) ? 4 : 2);
                    sum += wx * wy * (double) pdf[a * per + (size_t) y * pts + x];
                }
            }
            const double hx = (2 * (double) M_PI / phiRes) / panels, hy = (2.0 / thetaRes) / panels;
            const double integral = sum * hx * hy / 9.0;
            out[bin] = integral;
            if (panels > 8 && std::fabs(integral - prev[bin]) <= 1e-7 + 1e-5 * std::fabs(integral)) continue;
            prev[bin] = integral;
            still.push_back(bin);
        }
        active.swap(still);
    }
}

void ChiSquareTest::run() {
    int passed = 0, total = 0, res = m_cosThetaResolution * m_phiResolution;
    uint64_t stream = 1000;
    std::vector<double> obsFrequencies(res), expFrequencies(res);
    const size_t n = (size_t) m_sampleCount;
    for (auto bsdf : m_bsdfs) {
        for (int l = 0; l < m_testCount; ++l) {
            std::fill(obsFrequencies.begin(), obsFrequencies.end(), 0.0);
            cout << "------------------------------------------------------" << endl;
            cout << "Testing: " << bsdf->toString() << endl;
            ++total;
            std::vector<float> u = deviceFloats(stream++, 2);
            float cosTheta = u[0];
            float sinTheta = std::sqrt(std::max((float) 0, 1 - cosTheta * cosTheta));
            float sinPhi, cosPhi;
            sincosf(2.0f * M_PI * u[1], &sinPhi, &cosPhi);
            Vector3f wi(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
            cout << "Accumulating " << m_sampleCount << " samples into a " << m_cosThetaResolution << "x" << m_phiResolution
                 << " contingency table .. ";
            cout.flush();
            std::vector<float> wis(3 * n), wo(3 * n), weight(3 * n);
            for (size_t k = 0; k < n; ++k) { wis[3 * k] = wi.x(); wis[3 * k + 1] = wi.y(); wis[3 * k + 2] = wi.z(); }
            std::vector<float> samples = deviceFloats(stream++, 2 * n);
            bsdf->sampleBatch(wis.data(), samples.data(), n, wo.data(), weight.data(), nullptr, nullptr);
            for (size_t i = 0; i < n; ++i) {
                if (weight[3 * i] == 0 && weight[3 * i + 1] == 0 && weight[3 * i + 2] == 0) continue;
                int cosThetaBin = std::min(std::max(0, (int) std::floor((wo[3 * i + 2] * 0.5f + 0.5f) * m_cosThetaResolution)), m_cosThetaResolution - 1);
                float scaledPhi = std::atan2(wo[3 * i + 1], wo[3 * i]) * INV_TWOPI;
                if (scaledPhi < 0) scaledPhi += 1;
                int phiBin = std::min(std::max(0, (int) std::floor(scaledPhi * m_phiResolution)), m_phiResolution - 1);
                obsFrequencies[cosThetaBin * m_phiResolution + phiBin] += 1;
            }
            cout << "done." << endl;
            cout << "Integrating expected frequencies .. ";
            cout.flush();
            integratePdfOverBins(bsdf, wi, m_cosThetaResolution, m_phiResolution, expFrequencies);
            for (auto &e : expFrequencies) e *= m_sampleCount;
            cout << "done." << endl;
            auto result = hypothesis::chi2_test(res, obsFrequencies.data(), expFrequencies.data(), m_sampleCount, m_minExpFrequency,
                                                m_significanceLevel, m_testCount * (int) m_bsdfs.size());
            if (result.first) ++passed;
            cout << result.second << endl;
        }
    }
    cout << "Passed " << passed << "/" << total << " tests." << endl;
    if (passed < total) throw std::runtime_error("Some tests failed :(");
}

std::string ChiSquareTest::toString() const {
    return format("ChiSquareTest[\n  thetaResolution = %i,\n  phiResolution = %i,\n  minExpFrequency = %i,\n  sampleCount = %i,\n  testCount = %i,\n  significanceLevel = %f\n]",
                  m_cosThetaResolution, m_phiResolution, m_minExpFrequency, m_sampleCount, m_testCount, m_significanceLevel);
}
NORI_REGISTER_CLASS(ChiSquareTest, "chi2test");

NORI_NAMESPACE_END
