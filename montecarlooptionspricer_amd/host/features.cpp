// Host side of the batched driver rows: what the reference's production caller computes per option
// row BEFORE it calls the pricers -- /root/reference/src/core/PredictionGen.cpp:313-347 (compute20DayVolAndMomentum: two of
// the six CSV feature columns, and the `sigma` of AsymptoticAnalysis) and :612-620, :664-719 (the row's contract terms from
// its CSV fields).  Plain host arithmetic in the reference's order of operations: the two features are pinned bit for bit
// to the compiled reference (tests/golden/features.npz).
#include <cmath>
#include <vector>

#include "../csrc/mcg_internal.hpp"

namespace mcg {

// PredictionGen.cpp:313-347.  The last 21 prices give 20 log returns (a return over a non-positive price, or a
// non-finite one, counts as 0); vol = sqrt(max(0, E[lr^2] - E[lr]^2)) * sqrt(252), momentum = their sum.
void host_row_features(const double* hist, size_t n, double* vol, double* momentum) {
    *vol = 0.0;
    *momentum = 0.0;
    if (n < 21) return;
    const double* w = hist + (n - 21);
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < 20; ++k) {
        double lr = 0.0;
        if (w[k] > 0.0 && w[k + 1] > 0.0) {
            lr = std::log(w[k + 1] / w[k]);
            if (!std::isfinite(lr)) lr = 0.0;
        }
        s1 += lr;
        s2 += lr * lr;
    }
    const double mean = s1 / 20.0;
    double var = (s2 / 20.0) - (mean * mean);
    if (var < 0.0) var = 0.0;
    *vol = std::sqrt(var) * std::sqrt(252.0);
    *momentum = s1;
}

int host_row_build(const double* hist, size_t n, double underlying_last, double dte, double strike_dist_pct, int option_type,
                   double dividend, mcg_row* row, double features2[2]) {
    *row = mcg_row{};
    features2[0] = features2[1] = 0.0;
    // :612-620 -- inputs the driver rejects; :664 -- no spot history
    if (!std::isfinite(underlying_last) || !std::isfinite(dte) || !std::isfinite(strike_dist_pct) || underlying_last <= 0.0 ||
        dte <= 0.0 || strike_dist_pct < -1.0 || strike_dist_pct > 1.0 || n == 0 || !hist)
        return MCG_OK;
    std::vector<double> h(hist, hist + n);
    if (h.size() < 2) h.push_back(underlying_last);  // :671-673
    for (double s : h)
        if (!std::isfinite(s)) return MCG_OK;  // :675-693
    double vol, mom;
    host_row_features(h.data(), h.size(), &vol, &mom);
    const double maturity = dte / 365.0;                           // :702
    const int n_steps = (int)std::floor(maturity * 252.0);         // :718
    if (n_steps < 1) return MCG_OK;                                // :721-731: ",0,0,0,0,0,0"
    double p5[5];
    int rc = host_estimate_params(h.data(), h.size(), p5);         // RoughVolatility.cpp:324-331, inside GenerateStockPricePaths
    if (rc) return rc;
    row->xi = p5[0];
    row->H = p5[1];
    row->eta = p5[2];
    row->rho = p5[3];
    row->S0 = p5[4];
    row->strike = underlying_last * (1.0 - strike_dist_pct);       // :705
    row->maturity = maturity;
    row->sigma = vol;                                              // :706
    row->dividend = dividend;                                      // :707-716 (the caller parses the field; 0.08 if it cannot)
    row->n_steps = n_steps;
    row->is_call = option_type == 1 ? 1 : 0;                       // :704
    features2[0] = vol;
    features2[1] = mom;
    return MCG_OK;
}

}  // namespace mcg
