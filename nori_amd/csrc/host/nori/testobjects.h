/*
 * nori/testobjects.h -- the two <test> plugin classes (src/ttest.cpp:45-206,
 * src/chi2test.cpp:28-213), exposed so that the C API can enumerate their
 * children and parameters.
 */
#pragma once
#include <nori/plugins.h>

NORI_NAMESPACE_BEGIN

class TestBase : public NoriObject {
public:
    EClassType getClassType() const { return ETest; }
    /* true: activate() does not run the test (inspection without a GPU) */
    static bool s_defer;
    virtual void run() = 0;
};

class StudentsTTest : public TestBase {
public:
    StudentsTTest(const PropertyList &propList);
    virtual ~StudentsTTest();
    void addChild(NoriObject *obj);
    void activate();
    void run();
    std::string toString() const;
    std::vector<BSDF *> m_bsdfs;
    std::vector<Scene *> m_scenes;
    std::vector<float> m_angles, m_references;
    float m_significanceLevel;
    int m_sampleCount;
};

class ChiSquareTest : public TestBase {
public:
    ChiSquareTest(const PropertyList &propList);
    virtual ~ChiSquareTest();
    void addChild(NoriObject *obj);
    void activate();
    void run();
    std::string toString() const;
    int m_cosThetaResolution, m_phiResolution, m_minExpFrequency, m_sampleCount, m_testCount;
    float m_significanceLevel;
    std::vector<BSDF *> m_bsdfs;
};

void integratePdfOverBins(const BSDF *bsdf, const Vector3f &wi, int thetaRes, int phiRes, std::vector<double> &out);

NORI_NAMESPACE_END
