// Host precompute for the AsymptoticAnalysis scan: the exercise boundary and the discount factor of
// every date that the reference visits (t_j = j*dt <= maturity, in grid order; the reference breaks
// at the first t > maturity, /root/reference/src/models/AsymptoticAnalysisPricer.cpp:71-72).
// Boundary (:8-36): eps = T - t; K when eps < 1e-10; otherwise K +/- 0.5 sigma sqrt(eps ln(1/eps)),
// shifted by the carry term -/+ 0.5 (r - D) eps when eps < 0.01.  For eps > 1 the square root is of a
// negative number: the boundary is NaN and no price is "beyond" it -- kept as is.
// Built with -ffp-contract=off so the values equal the reference's bit for bit.
#include <cmath>
#include <vector>

#include "../csrc/mcg_internal.hpp"

namespace mcg {

void host_asymptotic_tables(int n_cols, double r, double K, double maturity, double dt, int is_call, double sigma,
                            double dividend, std::vector<double>& bnd, std::vector<double>& disc) {
    bnd.clear();
    disc.clear();
    for (int j = 0; j < n_cols; ++j) {
        const double t = j * dt;
        if (t > maturity) break;
        const double eps = maturity - t;
        double b = K;
        if (!(eps < 1e-10)) {
            const double half_width = 0.5 * sigma * std::sqrt(eps * std::log(1.0 / eps));
            if (is_call) {
                b = K - half_width;
                if (eps < 0.01) b += 0.5 * (dividend - r) * eps;
            } else {
                b = K + half_width;
                if (eps < 0.01) b -= 0.5 * (r - dividend) * eps;
            }
        }
        bnd.push_back(b);
        disc.push_back(std::exp(-r * t));
    }
}

}  // namespace mcg
