// The reference driver's row loop, UNCHANGED in shape, as a measured workload: written against the
// reference's public class API only (include/models/*.h, include/core/common.h -- the include paths of
// bcosm/MonteCarloOptionsPricer), compiled with plain g++, linked against libmcgpu.so; nothing here knows about HIP.
//
//   src/core/PredictionGen.cpp:542-546   #pragma omp parallel / omp for schedule(dynamic) over option rows
//                             :566-570   bp, mo, lsm, aa, roughVol default-constructed per row, per thread
//                             :700-719   r = 0.04, maturity = dte / 365, dt = 1 / 252, steps = floor(maturity * 252), 250 paths
//                             :736-737   roughVol.GenerateStockPricePaths(spotHist, steps, 250)
//                             :753-777   the scan of the whole matrix for inf / nan
//                             :780-783   exerciseTimes = 0 .. steps - 1
//                             :788-791   aa, bp (10 branches), lsm (order 2), mo (order 2)
//                             :792-805   std::exception caught per row -> the row's six columns are zeros
// What is NOT here is the driver's CSV and date plumbing (out of scope): rows are synthetic -- days to expiry
// 8 .. 183 (5 .. 126 steps), strikes 0.8 .. 1.2 of spot, calls and puts, three spot histories of 400 .. 1826 prices -- and a
// few rows are built to throw the way the reference's classes throw (a one-price history; sigma = 0).
//
//   unchanged_driver <n_rows> <coalesce 0|1|2> [prices_out.txt] [seed] [max arena slots]      (0: per-thread contexts, 1: coalesced + prefetch, 2: coalesced)
// prints one JSON line: rows, threads, seconds, rows_per_s, priced, threw, checksum.  With a seed (default 20251031) the
// prices of a row do not depend on the thread that priced it or on what else was in flight: prices_out.txt of two runs compare equal.
#include <omp.h>
#include <sys/resource.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
extern "C" int mcg_compat_set_coalescing(int enabled);
extern "C" int mcg_debug_coalesce_slots(int max_slots);   // include/mcgpu_debug.h (a test hook: argv[5])
extern "C" int mcg_stats(long long* out, int reset);   // (mcg_stats_t is a block of int64 counters, include/mcgpu.h)

namespace {

std::vector<double> history(size_t n, double s0, double drift, double wobble) {
    std::vector<double> h(n);
    double s = s0;
    for (size_t i = 0; i < n; ++i) {
        s *= std::exp(drift + wobble * std::sin(0.37 * (double)i) * std::cos(0.11 * (double)i * (double)(i % 97)));
        h[i] = s;
    }
    return h;
}

struct Row {
    int hist, dte, is_call;
    double strike_dist, sigma;
};

}  // namespace

int main(int argc, char** argv) {
    const int n_rows = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int coalesce = argc > 2 ? std::atoi(argv[2]) : 1;
    const char* out_path = argc > 3 ? argv[3] : nullptr;
    const unsigned long long seed = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 20251031ull;
    if (argc > 5) mcg_debug_coalesce_slots(std::atoi(argv[5]));   // fewer arena slots than threads: the rest price on contexts of their own
    const std::vector<std::vector<double>> hists = {history(400, 100.0, 0.0002, 0.012), history(1001, 166.5, 0.0001, 0.009),
                                                    history(1826, 42.0, -0.0001, 0.015), std::vector<double>{100.0}};
    std::vector<Row> rows((size_t)n_rows);
    unsigned lcg = 12345u;
    auto next = [&]() { return lcg = lcg * 1664525u + 1013904223u; };
    for (int i = 0; i < n_rows; ++i) {
        Row& r = rows[(size_t)i];
        r.hist = (int)(next() >> 8) % 3;
        r.dte = 8 + (int)((next() >> 8) % 176);          // 8 .. 183 days -> 5 .. 126 steps
        r.is_call = (int)((next() >> 12) & 1u);
        r.strike_dist = ((double)((next() >> 8) % 4001) - 2000.0) / 10000.0;   // -0.2 .. 0.2
        r.sigma = 0.12 + 0.0001 * (double)((next() >> 8) % 2000);
        if (i % 211 == 17) r.hist = 3;      // one price: "Historical prices vector too small." (RoughVolatility.cpp:317-319)
        if (i % 257 == 29) r.sigma = 0.0;   // "AsymptoticAnalysis: Volatility must be positive." (AsymptoticAnalysisPricer.cpp:50-52)
    }
    mcg_compat_set_seed(seed, 1);
    mcg_compat_set_coalescing(coalesce);
    std::vector<double> out((size_t)n_rows * 4, 0.0);
    std::vector<char> threw((size_t)n_rows, 0);
    {   // one row before the clock: contexts, the arena, the first pinned buffers
        RoughVolatility rv;
        LSM lsm;
        auto p = rv.GenerateStockPricePaths(hists[0], 10, 250);
        (void)lsm.PredictOptionPrice(p, 0.04, hists[0].back(), 10 / 252.0, 1 / 252.0, false, 2);
    }
    auto cpu_seconds = [] {
        rusage u;
        getrusage(RUSAGE_SELF, &u);
        return (double)u.ru_utime.tv_sec + 1e-6 * u.ru_utime.tv_usec + (double)u.ru_stime.tv_sec + 1e-6 * u.ru_stime.tv_usec;
    };
    const double cpu0 = cpu_seconds();
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic)
    for (int row = 0; row < n_rows; ++row) {
        BranchingProcesses bp;       // :566-570
        MartingaleOptimization mo;
        LSM lsm;
        AsymptoticAnalysis aa;
        RoughVolatility roughVol;
        const Row& q = rows[(size_t)row];
        try {
            const std::vector<double>& spotHist = hists[(size_t)q.hist];
            const double S = spotHist.back();
            const double r = 0.04, maturity = q.dte / 365.0, dt = 1.0 / 252.0;   // :700-704
            const double strike = S * (1.0 - q.strike_dist);                      // :709
            const int steps = (int)std::floor(maturity * 252.0);                  // :718
            if (steps <= 0) throw std::runtime_error("No time steps");
            auto pricePaths = roughVol.GenerateStockPricePaths(spotHist, steps, 250);   // :736-737
            if (pricePaths.empty() || pricePaths[0].empty()) throw std::runtime_error("Empty pricePaths");   // :739-749
            bool valid = true;
            for (const auto& path : pricePaths) {                                 // :753-766
                if (path.size() != (size_t)steps + 1) valid = false;
                for (double px : path)
                    if (std::isnan(px) || std::isinf(px)) valid = false;
            }
            if (!valid) throw std::runtime_error("Invalid pricePaths");
            std::vector<int> exerciseTimes((size_t)steps);                        // :780-783
            for (int i = 0; i < steps; ++i) exerciseTimes[(size_t)i] = i;
            const bool isCall = q.is_call != 0;
            const double aPrice = aa.PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, q.sigma, 0.08);      // :788
            const double bPrice = bp.PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, 10, exerciseTimes);   // :789
            const double lPrice = lsm.PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, 2);                  // :790
            const double mPrice = mo.PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, 2);                   // :791
            double* o = &out[(size_t)row * 4];
            o[0] = aPrice;
            o[1] = bPrice;
            o[2] = lPrice;
            o[3] = mPrice;
        } catch (const std::exception&) {   // :792-805, :825-847: the row is written as zeros
            threw[(size_t)row] = 1;
        }
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double cpu = cpu_seconds() - cpu0;
    double checksum = 0.0;
    int n_threw = 0, n_priced = 0;
    for (int i = 0; i < n_rows; ++i) {
        n_threw += threw[(size_t)i];
        bool finite = true;
        for (int c = 0; c < 4; ++c) finite = finite && std::isfinite(out[(size_t)i * 4 + c]);
        if (!threw[(size_t)i] && finite) {
            ++n_priced;
            for (int c = 0; c < 4; ++c) checksum += out[(size_t)i * 4 + c];
        }
    }
    if (out_path) {
        if (FILE* f = std::fopen(out_path, "w")) {
            for (int i = 0; i < n_rows; ++i)
                std::fprintf(f, "%d %d %.17g %.17g %.17g %.17g\n", i, (int)threw[(size_t)i], out[(size_t)i * 4], out[(size_t)i * 4 + 1],
                             out[(size_t)i * 4 + 2], out[(size_t)i * 4 + 3]);
            std::fclose(f);
        }
    }
    long long st[64] = {0};
    mcg_stats(st, 0);   // counters 15.. = coalesced rounds, calls, peak calls per round, fall-backs, round us, device-wait us, wake us, prefetched, hits
    std::printf("{\"rows\": %d, \"threads\": %d, \"coalescing\": %d, \"seconds\": %.6f, \"rows_per_s\": %.1f, \"priced\": %d, \"threw\": %d, "
                "\"checksum\": %.10f, \"rounds\": %lld, \"calls\": %lld, \"peak_calls_per_round\": %lld, \"own_context_calls\": %lld, "
                "\"round_us\": %lld, \"device_wait_us\": %lld, \"wake_us\": %lld, \"prefetched\": %lld, \"prefetch_hits\": %lld, \"cpu_seconds\": %.3f}\n",
                n_rows, omp_get_max_threads(), coalesce, sec, n_rows / sec, n_priced, n_threw, checksum, st[15], st[16], st[17], st[18], st[19],
                st[20], st[21], st[22], st[23], cpu);
    return 0;
}
