// Host side of the class-level drop-in: rBergomi parameter estimation from a price history.
// This is row a2 of SURVEY.md section 8 -- O(len(hist)) work that runs once per call and stays on
// the CPU; it exists so that RoughVolatility::GenerateStockPricePaths(hist, steps, paths) behaves
// like the reference when handed the same history.
//
// Behaviour follows /root/reference/src/models/RoughVolatility.cpp:20-169 and :321-331:
//   rets = log(p[i]/p[i-1]);  xi = var(rets)/dt (sample variance, n-1);  H = DFA slope;
//   eta = 2*stdev(rets);  rho = corr(rets, rets^2), replaced by -0.3 when positive;  S0 = last price;
//   dt = 1/252 and r = 0.04 are the reference's literals.
// Summation orders are kept (plain left-to-right accumulation) so results match the reference
// bit for bit when built without FMA contraction (this TU is compiled -ffp-contract=off).
#include <cmath>
#include <cstddef>
#include <numeric>
#include <vector>

#include "../csrc/mcg_internal.hpp"

namespace {

using vec = std::vector<double>;

struct Moments {
    static double avg(const vec& v) {
        if (v.empty()) return 0.0;
        return std::accumulate(v.begin(), v.end(), 0.0) / v.size();
    }
    // unbiased second central moment of one series, or cross moment of two
    static double cross(const vec& a, const vec& b) {
        if (a.size() != b.size() || a.size() < 2) return 0.0;
        const double ma = avg(a), mb = avg(b);
        double acc = 0.0;
        for (size_t i = 0; i < a.size(); ++i) acc += (a[i] - ma) * (b[i] - mb);
        return acc / (a.size() - 1);
    }
    static double var(const vec& a) {
        if (a.size() < 2) return 0.0;
        const double ma = avg(a);
        double acc = 0.0;
        for (double x : a) {
            const double d = x - ma;
            acc += d * d;
        }
        return acc / (a.size() - 1);
    }
};

// Detrended fluctuation analysis (RoughVolatility.cpp:72-122): integrate the centred series, and
// for dyadic window lengths 4, 8, ... <= n/4 take the mean RMS residual of a per-window linear
// fit (abscissa 1..w, :44-70); H is the OLS slope of log F(w) on log w.
class Dfa {
public:
    explicit Dfa(const vec& series) : profile_(series) {
        const double m = Moments::avg(profile_);
        for (double& x : profile_) x -= m;
        std::partial_sum(profile_.begin(), profile_.end(), profile_.begin());
    }

    double slope() const {
        const size_t n = profile_.size();
        vec lw, lf, rms;
        rms.reserve(n / 4 + 1);
        for (size_t w = 4; w <= n / 4; w *= 2) {
            rms.clear();
            for (size_t s = 0; s + w <= n; s += w) rms.push_back(window_rms(s, w));
            const double f = Moments::avg(rms);
            if (f > 0.0) {
                lw.push_back(std::log(static_cast<double>(w)));
                lf.push_back(std::log(f));
            }
        }
        const size_t k = lw.size();
        if (k < 2) return 0.5;
        double sx = 0, sy = 0, sxx = 0, sxy = 0;
        for (size_t i = 0; i < k; ++i) {
            sx += lw[i];
            sy += lf[i];
            sxx += lw[i] * lw[i];
            sxy += lw[i] * lf[i];
        }
        return (k * sxy - sx * sy) / (k * sxx - sx * sx);
    }

private:
    // One window's RMS residual of its linear fit.  The same operations on the same values in the same order as the
    // reference's detrendSegment (:44-70) -- the abscissa 1..w and the window's values are read where they lie instead of being
    // copied into two fresh vectors per window (a 1 826-price history has ~900 windows: that was 1 800 allocations and half of
    // this function's 115 us; the driver calls it once per option row).  Bit for bit the same H (tests/golden/estimators.npz).
    double window_rms(size_t start, size_t w) const {
        const double* y = profile_.data() + start;
        double b = 0.0, a = 0.0;
        bool fitted = false;
        if (w >= 2) {
            double tsum = 0.0, ysum = 0.0;
            for (size_t i = 0; i < w; ++i) tsum += static_cast<double>(i + 1);
            for (size_t i = 0; i < w; ++i) ysum += y[i];
            const double tm = tsum / w, ym = ysum / w;
            double num = 0.0, den = 0.0;
            for (size_t i = 0; i < w; ++i) {
                const double ti = static_cast<double>(i + 1);
                num += (ti - tm) * (y[i] - ym);
                den += (ti - tm) * (ti - tm);
            }
            if (!(std::abs(den) < 1e-14)) {
                b = num / den;
                a = ym - b * tm;
                fitted = true;
            }
        }
        double ss = 0.0;
        for (size_t i = 0; i < w; ++i) {
            const double q = fitted ? y[i] - (b * static_cast<double>(i + 1) + a) : y[i];
            ss += q * q;
        }
        return std::sqrt(ss / w);
    }

    vec profile_;
};

}  // namespace

namespace mcg {

int host_estimate_params(const double* hist, size_t n, double out5[5]) {
    if (!hist || n < 2) return fail(MCG_ERR_HISTORY_TOO_SMALL, "Historical prices vector too small.");
    vec rets;
    rets.reserve(n - 1);
    for (size_t i = 1; i < n; ++i) rets.push_back(std::log(hist[i] / hist[i - 1]));

    const double dt = 1.0 / 252.0;
    const double var_r = Moments::var(rets);
    out5[0] = var_r / dt;                                              // xi
    out5[1] = rets.size() < 2 ? 0.5 : Dfa(rets).slope();               // H
    out5[2] = std::sqrt(var_r) * 2.0;                                  // eta

    vec sq(rets.size());
    for (size_t i = 0; i < rets.size(); ++i) sq[i] = rets[i] * rets[i];
    double rho = Moments::cross(rets, sq) / (std::sqrt(Moments::var(rets) * Moments::var(sq)));
    if (rho > 0.0) rho = -0.3;
    out5[3] = rho;
    out5[4] = hist[n - 1];                                             // S0
    return MCG_OK;
}

}  // namespace mcg
