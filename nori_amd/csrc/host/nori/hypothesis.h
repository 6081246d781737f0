/*
 * nori/hypothesis.h -- the statistics the reference takes from the un-vendored
 * ext/hypothesis (wjakob/hypothesis, .gitmodules:13-15; call sites
 * src/ttest.cpp:127,175, src/chi2test.cpp:157-173, src/warptest.cpp:198-212):
 * Student-t test of a mean, Pearson chi^2 goodness-of-fit test with pooling of
 * low-expectation cells, both with a Sidak correction for `testCount`
 * independent tests.  Restated from the published definitions; the CDFs are
 * cross-checked against scipy.stats in tests/test_host_statistics.py.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

namespace hypothesis {

/* regularized lower incomplete gamma P(a, x) */
inline double gammaP(double a, double x) {
    if (x <= 0) return 0.0;
    const double gln = std::lgamma(a);
    if (x < a + 1.0) {                          /* series */
        double ap = a, sum = 1.0 / a, del = sum;
        for (int n = 0; n < 10000; ++n) {
            ap += 1.0; del *= x / ap; sum += del;
            if (std::fabs(del) < std::fabs(sum) * 1e-16) break;
        }
        return sum * std::exp(-x + a * std::log(x) - gln);
    }
    double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;   /* Lentz continued fraction for Q */
    for (int i = 1; i < 10000; ++i) {
        const double an = -i * (i - a);
        b += 2.0;
        d = an * d + b; if (std::fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c; if (std::fabs(c) < 1e-300) c = 1e-300;
        d = 1.0 / d;
        const double del = d * c; h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return 1.0 - std::exp(-x + a * std::log(x) - gln) * h;
}

inline double chi2_cdf(double x, int dof) { return gammaP(0.5 * dof, 0.5 * x); }

/* regularized incomplete beta I_x(a, b) */
inline double betaI(double a, double b, double x) {
    if (x <= 0) return 0.0;
    if (x >= 1) return 1.0;
    const double bt = std::exp(std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log(1.0 - x));
    auto cf = [](double a_, double b_, double x_) {
        const double qab = a_ + b_, qap = a_ + 1.0, qam = a_ - 1.0;
        double c = 1.0, d = 1.0 - qab * x_ / qap;
        if (std::fabs(d) < 1e-300) d = 1e-300;
        d = 1.0 / d;
        double h = d;
        for (int m = 1; m < 10000; ++m) {
            const int m2 = 2 * m;
            double aa = m * (b_ - m) * x_ / ((qam + m2) * (a_ + m2));
            d = 1.0 + aa * d; if (std::fabs(d) < 1e-300) d = 1e-300;
            c = 1.0 + aa / c; if (std::fabs(c) < 1e-300) c = 1e-300;
            d = 1.0 / d; h *= d * c;
            aa = -(a_ + m) * (qab + m) * x_ / ((a_ + m2) * (qap + m2));
            d = 1.0 + aa * d; if (std::fabs(d) < 1e-300) d = 1e-300;
            c = 1.0 + aa / c; if (std::fabs(c) < 1e-300) c = 1e-300;
            d = 1.0 / d;
            const double del = d * c; h *= del;
            if (std::fabs(del - 1.0) < 1e-16) break;
        }
        return h;
    };
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * cf(a, b, x) / a;
    return 1.0 - bt * cf(b, a, 1.0 - x) / b;
}

inline double students_t_cdf(double t, int dof) {
    const double x = dof / (dof + t * t);
    const double tail = 0.5 * betaI(0.5 * dof, 0.5, x);
    return t > 0 ? 1.0 - tail : tail;
}

inline std::string fmt(const char *f, double a = 0, double b = 0, double c = 0, double d = 0) {
    char buf[512]; snprintf(buf, sizeof(buf), f, a, b, c, d); return buf;
}

/* Two-sided one-sample t-test of H0: E[X] == reference. */
inline std::pair<bool, std::string> students_t_test(double mean, double variance, double reference, int sampleCount,
                                                     double significanceLevel, int numTests) {
    if (sampleCount < 5) return {false, "t-test failed: too few samples"};
    const double t = std::fabs(mean - reference) * std::sqrt(sampleCount / std::max(variance, 1e-5));
    const int dof = sampleCount - 1;
    const double pval = 2.0 * students_t_cdf(-t, dof);
    const double alpha = 1.0 - std::pow(1.0 - significanceLevel, 1.0 / numTests);      /* Sidak */
    const bool ok = pval > alpha;
    std::string msg = fmt("Sample mean = %g (reference value = %g)\nSample variance = %g\n", mean, reference, variance) +
                      fmt("t-statistic = %g (d.o.f. = %g)\np-value = %g (corrected significance level = %g)\n", t, dof, pval, alpha) +
                      (ok ? "Accepted the null hypothesis" : "***** Rejected ***** the null hypothesis");
    return {ok, msg};
}

/* Pearson chi^2 test; cells with expectation < minExpFrequency are pooled. */
inline std::pair<bool, std::string> chi2_test(int nCells, const double *obs, const double *exp, double sampleCount,
                                              double minExpFrequency, double significanceLevel, int numTests) {
    std::vector<int> order(nCells);
    for (int i = 0; i < nCells; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return exp[a] < exp[b]; });
    double pooledObs = 0, pooledExp = 0, chsq = 0;
    int pooledCells = 0, dof = 0;
    for (int k = 0; k < nCells; ++k) {
        const int c = order[k];
        if (exp[c] == 0) {
            if (obs[c] > sampleCount * 1e-5)
                return {false, fmt("Encountered %g samples in a cell with expected frequency 0. Rejecting the null hypothesis!", obs[c])};
        } else if (exp[c] < minExpFrequency) {
            pooledObs += obs[c]; pooledExp += exp[c]; ++pooledCells;
        } else if (pooledExp > 0 && pooledExp < minExpFrequency) {
            pooledObs += obs[c]; pooledExp += exp[c]; ++pooledCells;     /* keep pooling until the pool is large enough */
        } else {
            const double diff = obs[c] - exp[c];
            chsq += diff * diff / exp[c]; ++dof;
        }
    }
    std::string msg;
    if (pooledExp > 0 || pooledObs > 0) {
        msg += fmt("Pooled %g to ensure sufficiently high expected cell frequencies (>%g)\n", pooledCells, minExpFrequency);
        const double diff = pooledObs - pooledExp;
        chsq += diff * diff / pooledExp; ++dof;
    }
    dof -= 1;
    if (dof <= 0) return {false, msg + fmt("The number of degrees of freedom (%g) is too low!", dof)};
    const double pval = 1.0 - chi2_cdf(chsq, dof);
    const double alpha = 1.0 - std::pow(1.0 - significanceLevel, 1.0 / numTests);
    msg += fmt("Chi^2 statistic = %g (d.o.f. = %g)\n", chsq, dof);
    if (pval < alpha || !std::isfinite(pval))
        return {false, msg + fmt("***** Rejected ***** the null hypothesis (p-value = %g, significance level = %g)", pval, alpha)};
    return {true, msg + fmt("Accepted the null hypothesis (p-value = %g, significance level = %g)", pval, alpha)};
}

} // namespace hypothesis
