// The batched driver boundary from C++ (the reference's language), against include/mcgpu.h only: the reference driver's
// row loop (src/core/PredictionGen.cpp:542-823) rewritten the way INTEGRATION.md section 1 shows -- rows built on the host
// threads by mcg_row_build (the driver's own checks and contract terms, :612-620, :664-719), ONE call of
// mcg_batch_price_rows6 for all lines, six columns per line (:471-477, :809-816).
// Compiled with plain g++ and linked against libmcgpu.so; run by tests/test_gpu_cpp_dropin.py.
//
// Output: "row <i> <six columns>" per line, then "OK rows=<n>".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mcgpu.h"

int main(int argc, char** argv) {
    const long n_lines = argc > 1 ? std::atol(argv[1]) : 64;
    // per line: a spot history (what fetchSpotHistory returns, :657-658) and the CSV fields the driver parses (:590-716)
    std::vector<std::vector<double>> hist(n_lines);
    std::vector<double> last(n_lines), dte(n_lines), dist(n_lines), dividend(n_lines);
    std::vector<int> type(n_lines);
    for (long i = 0; i < n_lines; ++i) {
        const size_t len = i % 11 == 3 ? 15 : 60 + (size_t)(i * 37 % 400);   // every 11th line: fewer than 21 prices -> sigma 0 -> six zeros
        hist[i].resize(len);
        double s = 40.0 + 3.0 * (double)(i % 50);
        for (size_t k = 0; k < len; ++k) {
            s *= std::exp(0.0003 + 0.015 * std::sin(0.31 * (double)k + (double)i) * std::cos(0.07 * (double)(k * k % 97)));
            hist[i][k] = s;
        }
        last[i] = s;
        dte[i] = i % 13 == 5 ? 1.0 : 10.0 + (double)(i * 7 % 170);            // every 13th line: no time step -> six zeros (:721-731)
        dist[i] = -0.08 + 0.01 * (double)(i % 17);
        type[i] = (int)(i % 2);
        dividend[i] = 0.01 * (double)(i % 5);
    }
    std::vector<mcg_row> rows(n_lines);
    std::vector<double> feat(2 * n_lines), six(6 * n_lines), again(6 * n_lines), four(4 * n_lines);
    int failures = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : failures)
    for (long i = 0; i < n_lines; ++i) {
        if (mcg_row_build(hist[i].data(), hist[i].size(), last[i], dte[i], dist[i], type[i], dividend[i], &rows[i], &feat[2 * i]) != MCG_OK)
            ++failures;
        double vol = -1.0, mom = -1.0;
        if (mcg_row_features(hist[i].data(), hist[i].size(), &vol, &mom) != MCG_OK) ++failures;
        const bool skipped = rows[i].n_steps < 1;
        if (!skipped && (feat[2 * i] != vol || feat[2 * i + 1] != mom || rows[i].sigma != vol)) ++failures;   // sigma = twentyDayVol (:706)
        if (skipped && (feat[2 * i] != 0.0 || feat[2 * i + 1] != 0.0)) ++failures;
    }
    mcg_ctx* ctx = nullptr;
    if (mcg_init(&ctx, 0) != MCG_OK) {
        std::printf("FAILED init: %s\n", mcg_last_error());
        return 1;
    }
    const int rc = mcg_batch_price_rows6(ctx, rows.data(), feat.data(), n_lines, 250, 0.04, 1.0 / 252.0, 10, 2, 5, 2024, six.data());
    if (rc != MCG_OK) {
        std::printf("FAILED batch: %s\n", mcg_last_error());
        return 1;
    }
    if (mcg_batch_price_rows6(ctx, rows.data(), feat.data(), n_lines, 250, 0.04, 1.0 / 252.0, 10, 2, 5, 2024, again.data()) != MCG_OK) ++failures;
    if (mcg_batch_price_rows(ctx, rows.data(), n_lines, 250, 0.04, 1.0 / 252.0, 10, 2, 5, 2024, four.data()) != MCG_OK) ++failures;
    long priced = 0;
    for (long i = 0; i < n_lines; ++i) {
        const double* c = &six[6 * i];
        std::printf("row %ld %.10g %.10g %.10g %.10g %.10g %.10g\n", i, c[0], c[1], c[2], c[3], c[4], c[5]);
        for (int k = 0; k < 6; ++k)
            if (c[k] != again[6 * i + k] || !std::isfinite(c[k])) ++failures;                // a call is reproducible
        const bool zeros = rows[i].n_steps < 1 || !(rows[i].sigma > 0.0);                    // the driver's ",0,0,0,0,0,0" lines
        if (zeros) {
            for (int k = 0; k < 6; ++k)
                if (c[k] != 0.0) ++failures;
        } else {
            ++priced;
            for (int k = 0; k < 4; ++k)
                if (c[k] != four[4 * i + k] || !(c[k] >= 0.0) || !(c[k] < 3.0 * rows[i].S0)) ++failures;
            if (c[4] != feat[2 * i] || c[5] != feat[2 * i + 1] || !(c[2] > 0.0)) ++failures;
        }
    }
    mcg_stats_t st;
    if (mcg_stats(&st, 0) != MCG_OK || st.batch_calls < 3 || st.batch_rows < 3 * priced) ++failures;
    mcg_finalize(ctx);
    if (failures || priced < n_lines / 2) {
        std::printf("FAILED %d (priced %ld)\n", failures, priced);
        return 1;
    }
    std::printf("OK rows=%ld priced=%ld\n", n_lines, priced);
    return 0;
}
