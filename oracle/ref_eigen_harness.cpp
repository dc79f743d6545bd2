// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// C-callable harness around the two pricers of the reference whose least-squares solve is Eigen's
// (`A.bdcSvd(ComputeThinU | ComputeThinV).solve(b)`, src/models/LSMPricer.cpp:76 and
// src/models/MartingaleOptimizationPricer.cpp:166).  oracle/Makefile builds it TOGETHER WITH those two translation units,
// compiled in place from /root/reference (nothing copied), into oracle/_ref/libmcref_eigen.so -- but only where an Eigen3
// exists ($(EIGEN_INC)/Eigen/Dense).  This image has none, so here the rule is dormant and the LSM / MartingaleOptimization
// half of the oracle stays "parity unpinned" (DESIGN.md section 2); on an image with Eigen,
//     make -C oracle && python oracle/gen_golden.py --eigen
// captures tests/golden/{lsm,martingale}.npz from the compiled reference and tests/test_oracle_golden.py pins the restatement
// (oracle/mcg_oracle.cpp: orc_lsm_price, orc_martingale_price) to them -- without a line of new code.
// No stand-in for Eigen is written or vendored anywhere.
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "models/LSMPricer.h"
#include "models/MartingaleOptimizationPricer.h"

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>  // only for the version the fixtures were captured with
#endif

namespace {
std::vector<std::vector<double>> rows_of(const double* row_major, long n, int m) {
    std::vector<std::vector<double>> paths((size_t)std::max<long>(n, 0));
    for (long i = 0; i < n; ++i) paths[i].assign(row_major + (size_t)i * m, row_major + (size_t)(i + 1) * m);
    return paths;
}
int report(const std::exception& e, char* err, size_t errlen) {
    if (err && errlen) {
        std::strncpy(err, e.what(), errlen - 1);
        err[errlen - 1] = 0;
    }
    return 1;
}
}  // namespace

extern "C" {

// {world, major, minor} of the Eigen the library was built against (0, 0, 0 if its version macros are not visible).
void ref_eigen_version(int* out3) {
#if defined(EIGEN_WORLD_VERSION) && defined(EIGEN_MAJOR_VERSION) && defined(EIGEN_MINOR_VERSION)
    out3[0] = EIGEN_WORLD_VERSION;
    out3[1] = EIGEN_MAJOR_VERSION;
    out3[2] = EIGEN_MINOR_VERSION;
#else
    out3[0] = out3[1] = out3[2] = 0;
#endif
}

// LSM::PredictOptionPrice (LSMPricer.cpp:19-102) on a [n][m] path-major matrix (the reference's pricePaths layout).
int ref_lsm_price(const double* row_major, long n, int m, double r, double strike, double maturity, double dt, int is_call,
                  int poly_order, double* price, char* err, size_t errlen) {
    try {
        const auto paths = rows_of(row_major, n, m);
        LSM lsm;
        *price = lsm.PredictOptionPrice(paths, r, strike, maturity, dt, is_call != 0, poly_order);
        return 0;
    } catch (const std::exception& e) {
        return report(e, err, errlen);
    }
}

// MartingaleOptimization::PredictOptionPrice (MartingaleOptimizationPricer.cpp:21-189), same layout.
int ref_martingale_price(const double* row_major, long n, int m, double r, double strike, double maturity, double dt,
                         int is_call, int poly_order, int max_iterations, double* price, char* err, size_t errlen) {
    try {
        const auto paths = rows_of(row_major, n, m);
        MartingaleOptimization mo;
        *price = mo.PredictOptionPrice(paths, r, strike, maturity, dt, is_call != 0, poly_order, max_iterations);
        return 0;
    } catch (const std::exception& e) {
        return report(e, err, errlen);
    }
}

}  // extern "C"
