// Drop-in check of the C++ boundary: this file is written against the REFERENCE's public API only
// (include/models/RoughVolatility.h, include/models/LSMPricer.h, include/core/common.h -- same include
// paths as bcosm/MonteCarloOptionsPricer) and uses it the way the reference's driver does
// (src/core/PredictionGen.cpp:542-570, :736-737, :790): pricer objects default-constructed per row
// inside an OpenMP parallel-for, 250 paths per row, LSM with polyOrder 2, exceptions caught per row.
// It is compiled with plain g++ and linked against libmcgpu.so; nothing here knows about HIP.
//
// Output: one line per row "row <i> steps <n> lsm <price> european_put <price>", then "OK rows=<n>".
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "core/common.h"
#include "models/AsymptoticAnalysisPricer.h"
#include "models/BranchingProcessPricer.h"
#include "models/LSMPricer.h"
#include "models/MartingaleOptimizationPricer.h"
#include "models/RoughVolatility.h"

extern "C" int mcg_compat_set_seed(unsigned long long seed, int enabled);

int main(int argc, char** argv) {
    const int n_rows = argc > 1 ? std::atoi(argv[1]) : 16;
    // synthetic spot history (deterministic), like fetchSpotHistory would return
    std::vector<double> hist(400);
    double s = 100.0;
    for (size_t i = 0; i < hist.size(); ++i) {
        s *= std::exp(0.0002 + 0.012 * std::sin(0.37 * (double)i) * std::cos(0.11 * (double)i * i));
        hist[i] = s;
    }
    mcg_compat_set_seed(1234, 1);
    std::vector<std::string> out(n_rows);
    int failures = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : failures)
    for (int row = 0; row < n_rows; ++row) {
        LSM lsm;                  // per row, per thread (PredictionGen.cpp:566-570)
        AsymptoticAnalysis aa;
        MartingaleOptimization mo;
        BranchingProcesses bp;
        RoughVolatility roughVol;
        try {
            const int steps = 10 + 5 * (row % 7);
            const double r = 0.04, dt = 1.0 / 252.0, maturity = steps * dt;   // :700-702
            const double strike = hist.back();
            auto paths = roughVol.GenerateStockPricePaths(hist, steps, 250);   // :736-737, 250 = :719
            if (paths.size() != 250 || paths[0].size() != (size_t)steps + 1) throw std::runtime_error("bad shape");
            for (auto& p : paths)
                for (double px : p)
                    if (!std::isfinite(px)) throw std::runtime_error("non-finite path");   // :753-766
            const double v = lsm.PredictOptionPrice(paths, r, strike, maturity, dt, false, 2);   // :790
            const double asym = aa.PredictOptionPrice(paths, r, strike, maturity, dt, false, 0.2, 0.08);   // :788
            if (!(asym >= 0.0) || !(asym < strike)) throw std::runtime_error("asymptotic price out of range");
            const double mart = mo.PredictOptionPrice(paths, r, strike, maturity, dt, false, 2);            // :791
            std::vector<int> exerciseTimes(steps);                                                          // :780-783
            for (int i = 0; i < steps; ++i) exerciseTimes[i] = i;
            const double branch = bp.PredictOptionPrice(paths, r, strike, maturity, dt, false, 10, exerciseTimes);   // :789
            if (!(branch >= 0.0) || !(branch < strike)) throw std::runtime_error("branching price out of range");
            if (!(mart >= 0.0) || !(mart < strike)) throw std::runtime_error("martingale price out of range");
            double eu = 0.0;
            for (auto& p : paths) eu += PayoffFunction(false, p.back(), strike);
            eu = std::exp(-r * maturity) * eu / paths.size();
            char buf[160];
            std::snprintf(buf, sizeof buf, "row %d steps %d lsm %.10f european_put %.10f", row, steps, v, eu);
            out[row] = buf;
            if (!(v >= eu - 1e-9) || !(v < strike)) ++failures;   // American >= European on the same paths
        } catch (const std::exception& e) {
            out[row] = std::string("row ") + std::to_string(row) + " EXCEPTION " + e.what();
            ++failures;
        }
    }
    for (auto& l : out) std::puts(l.c_str());
    // the reference's error paths
    try {
        RoughVolatility().GenerateStockPricePaths({100.0}, 5, 5);
        ++failures;
    } catch (const std::runtime_error& e) {
        if (std::string(e.what()) != "Historical prices vector too small.") ++failures;
    }
    try {
        LSM().PredictOptionPrice({}, 0.04, 100.0, 1.0, 1.0 / 252.0, false, 2);
        ++failures;
    } catch (const std::runtime_error& e) {
        if (std::string(e.what()) != "LSM::PredictOptionPrice: Empty pricePaths.") ++failures;
    }
    if (failures) {
        std::printf("FAILED %d\n", failures);
        return 1;
    }
    std::printf("OK rows=%d\n", n_rows);
    return 0;
}
