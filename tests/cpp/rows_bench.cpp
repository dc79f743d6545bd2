// Rows-per-second of the reference driver's per-row work (src/core/PredictionGen.cpp:566-791: 250 rBergomi
// paths, then AsymptoticAnalysis, BranchingProcesses(10), LSM(2), MartingaleOptimization(2)) through the
// drop-in classes, under the driver's own OpenMP row parallelism.  Dev tool: prints one line per thread count.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "models/AsymptoticAnalysisPricer.h"
#include "models/BranchingProcessPricer.h"
#include "models/LSMPricer.h"
#include "models/MartingaleOptimizationPricer.h"
#include "models/RoughVolatility.h"

int main(int argc, char** argv) {
    const int n_rows = argc > 1 ? std::atoi(argv[1]) : 512;
    std::vector<double> hist(1001);
    double s = 100.0;
    for (size_t i = 0; i < hist.size(); ++i) {
        s *= std::exp(0.0002 + 0.012 * std::sin(0.37 * (double)i) * std::cos(0.11 * (double)i * i));
        hist[i] = s;
    }
    for (int threads : {1, 4, 16, 64}) {
#ifdef _OPENMP
        omp_set_num_threads(threads);
#endif
        double checksum = 0.0;
        for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms every thread's context
            const auto t0 = std::chrono::steady_clock::now();
            checksum = 0.0;
#pragma omp parallel for schedule(dynamic) reduction(+ : checksum)
            for (int row = 0; row < n_rows; ++row) {
                RoughVolatility rv;
                LSM lsm;
                AsymptoticAnalysis aa;
                BranchingProcesses bp;
                MartingaleOptimization mo;
                const int steps = 20 + (row % 5) * 20;  // 20..100 trading days to expiry
                const double r = 0.04, dt = 1.0 / 252.0, maturity = steps / 252.0, K = hist.back();
                auto paths = rv.GenerateStockPricePaths(hist, steps, 250);
                std::vector<int> ex(steps);
                for (int i = 0; i < steps; ++i) ex[i] = i;
                checksum += aa.PredictOptionPrice(paths, r, K, maturity, dt, false, 0.2, 0.08);
                checksum += bp.PredictOptionPrice(paths, r, K, maturity, dt, false, 10, ex);
                checksum += lsm.PredictOptionPrice(paths, r, K, maturity, dt, false, 2);
                checksum += mo.PredictOptionPrice(paths, r, K, maturity, dt, false, 2);
            }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (pass == 1)
                std::printf("threads %3d: %d rows in %.3f s = %.0f rows/s (%.1f us/row/thread)  checksum %.6f\n", threads,
                            n_rows, sec, n_rows / sec, 1e6 * sec * threads / n_rows, checksum);
        }
    }
    return 0;
}
