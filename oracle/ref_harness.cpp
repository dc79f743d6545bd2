// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// C-callable harness around the *compiled reference* path engine.  It is built by
// oracle/Makefile from the reference sources where they lie (/root/reference/src/models/
// RoughVolatility.cpp, compiled in place, nothing copied) into oracle/_ref/libmcref.so.
// Purpose: (1) generate the golden vectors under tests/golden/ (oracle/gen_golden.py),
// (2) validate oracle/mcg_oracle.cpp (our CPU restatement), (3) serve as the
// cpu_baseline ("kind": "reference") that bench.py times on the GPU box's host cores.
//
// The reference keeps its numeric helpers private (include/models/RoughVolatility.h:21-53);
// the harness reaches them by pre-including the std headers the class header needs and then
// re-reading the class header with `private` spelled `public`.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define private public
#include "models/RoughVolatility.h"
#undef private
#include "core/common.h"
#include "models/AsymptoticAnalysisPricer.h"
#define private public
#include "models/BranchingProcessPricer.h"
#undef private

namespace {
using cvec = std::vector<std::complex<double>>;

cvec unpack(const double* reim, size_t n) {
    cvec v(n);
    for (size_t i = 0; i < n; ++i) v[i] = std::complex<double>(reim[2 * i], reim[2 * i + 1]);
    return v;
}
void pack(const cvec& v, double* reim) {
    for (size_t i = 0; i < v.size(); ++i) {
        reim[2 * i] = v[i].real();
        reim[2 * i + 1] = v[i].imag();
    }
}
}  // namespace

extern "C" {

// RoughVolatility.cpp:126-169, :324-331.  out = {xi, H, eta, rho, S0}; rets gets n-1 values.
int ref_estimators(const double* hist, size_t n, double* rets_out, double* out5) {
    if (n < 2) return 1;
    RoughVolatility rv;
    std::vector<double> h(hist, hist + n);
    std::vector<double> rets = rv.logReturns(h);
    if (rets_out) std::copy(rets.begin(), rets.end(), rets_out);
    const double dt = 1.0 / 252.0;
    out5[0] = rv.estimateXi(rets, dt);
    out5[1] = rv.estimateH(rets);
    out5[2] = rv.estimateEta(rets, out5[1]);
    out5[3] = rv.estimateRho(rets);
    out5[4] = h.back();
    return 0;
}

size_t ref_next_pow2(size_t n) { return RoughVolatility::nextPowerOfTwo(n); }

// RoughVolatility.cpp:171-202; interleaved re/im, in place.
void ref_fft(double* reim, size_t n, int inv) {
    RoughVolatility rv;
    cvec a = unpack(reim, n);
    rv.fft(a, inv);
    pack(a, reim);
}

// RoughVolatility.cpp:227-236 on the grid of :337-340 (t_i = i*dt, i = 0..steps).
void ref_lambda(int steps, double H, double dt, double* lam_out) {
    RoughVolatility rv;
    std::vector<double> grid(steps + 1);
    for (size_t i = 0; i <= (size_t)steps; ++i) grid[i] = i * dt;
    std::vector<double> lam = rv.rbergomiLambda(grid, H);
    std::copy(lam.begin(), lam.end(), lam_out);
}

// RoughVolatility.cpp:212-225.  Returns M_phi; phi_out needs 2*nextpow2(n) doubles.
size_t ref_phi(const double* lam, size_t n, double H, double* phi_out) {
    RoughVolatility rv;
    cvec phi = rv.rbergomiPhi(std::vector<double>(lam, lam + n), H);
    pack(phi, phi_out);
    return phi.size();
}

// RoughVolatility.cpp:264-292.  Z has `steps` complex entries, X_out `steps` doubles.
void ref_fractional_gaussian(const double* phi, size_t mphi, const double* Z, size_t steps,
                             double H, double eta, double* X_out) {
    RoughVolatility rv;
    std::vector<double> X = rv.fractionalGaussian(unpack(phi, mphi), unpack(Z, steps), H, eta);
    std::copy(X.begin(), X.end(), X_out);
}

// RoughVolatility.cpp:294-309.
void ref_forward_variance(const double* X, size_t steps, double dt, double xi, double H, double eta,
                          double* v_out) {
    RoughVolatility rv;
    std::vector<double> grid(steps + 1);
    for (size_t i = 0; i <= steps; ++i) grid[i] = i * dt;
    std::vector<double> v = rv.forwardVariance(std::vector<double>(X, X + steps), grid, xi, H, eta);
    std::copy(v.begin(), v.end(), v_out);
}

double ref_payoff(int is_call, double s, double k) { return PayoffFunction(is_call != 0, s, k); }

// RoughVolatility.cpp:312-368 (unseeded: std::random_device).  out is path-major
// [paths][steps+1], exactly the reference's return layout.  Returns 0, or 1 when the reference
// throws ("Historical prices vector too small.", :317-319) with the message in err.
int ref_generate_paths(const double* hist, size_t n, int steps, int paths, double* out, char* err,
                       size_t errlen) {
    try {
        RoughVolatility rv;
        auto m = rv.GenerateStockPricePaths(std::vector<double>(hist, hist + n), steps, paths);
        for (size_t i = 0; i < m.size(); ++i)
            std::memcpy(out + i * (size_t)(steps + 1), m[i].data(), sizeof(double) * (steps + 1));
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen) {
            std::strncpy(err, e.what(), errlen - 1);
            err[errlen - 1] = 0;
        }
        return 1;
    }
}

// CPU baseline: the reference generator run the way the reference parallelises it --
// independent GenerateStockPricePaths calls under `omp parallel for schedule(dynamic)`
// (src/core/PredictionGen.cpp:542-546), one chunk of `chunk` paths per call.  Only the terminal
// column is kept (sum of S_T for a sanity check), so the baseline does not need paths*steps of
// host memory.  Returns the number of threads used; *sum_ST gets sum of S_T over all paths.
int ref_generate_paths_omp(const double* hist, size_t n, int steps, long total_paths, int chunk,
                           double* sum_ST) {
    std::vector<double> h(hist, hist + n);
    long n_chunks = (total_paths + chunk - 1) / chunk;
    double acc = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : acc)
    for (long c = 0; c < n_chunks; ++c) {
        long lo = c * (long)chunk;
        int cnt = (int)std::min<long>(chunk, total_paths - lo);
        RoughVolatility rv;
        auto m = rv.GenerateStockPricePaths(h, steps, cnt);
        for (auto& p : m) acc += p.back();
    }
    *sum_ST = acc;
    return threads;
}

// Same run, additionally pricing a European option on the reference's own sample: out4 = {sum S_T,
// sum payoff, sum payoff^2, paths}, payoff = PayoffFunction(is_call, S_T, strike) (include/core/common.h:8-14),
// undiscounted.  bench.py sets the engine's price on the same (S0, xi, H, eta, steps) beside it.
int ref_generate_paths_omp_payoff(const double* hist, size_t n, int steps, long total_paths, int chunk, double strike,
                                  int is_call, double* out4) {
    std::vector<double> h(hist, hist + n);
    long n_chunks = (total_paths + chunk - 1) / chunk;
    double acc = 0.0, pay = 0.0, pay2 = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : acc, pay, pay2)
    for (long c = 0; c < n_chunks; ++c) {
        long lo = c * (long)chunk;
        int cnt = (int)std::min<long>(chunk, total_paths - lo);
        RoughVolatility rv;
        auto m = rv.GenerateStockPricePaths(h, steps, cnt);
        for (auto& p : m) {
            const double f = PayoffFunction(is_call != 0, p.back(), strike);
            acc += p.back();
            pay += f;
            pay2 += f * f;
        }
    }
    out4[0] = acc;
    out4[1] = pay;
    out4[2] = pay2;
    out4[3] = (double)total_paths;
    return threads;
}

// Rough-regime sample of the reference with EXPLICIT parameters.  Through its public entry point the reference
// only ever simulates history-estimated parameters (eta = 2 stdev(returns), :151-155: vol-of-vol ~0.03, H ~0.55),
// so C4/C5's regime (H = 0.1, eta = 1.9) is reached here by calling the reference's OWN private members per path --
// genComplexGaussians, fractionalGaussian, forwardVariance, gaussians, rbergomiLambda, rbergomiPhi, exactly as
// GenerateStockPricePaths chains them (:337-352) -- followed by the ten-line stepping loop of :354-364 restated
// below (it is inline in GenerateStockPricePaths and cannot be called separately).  Per path ten statistics are
// formed and their sums and sums of squares over all paths returned in out20 = {sum s_0..s_9, sum s_0^2..s_9^2}:
//   s0 S_T   s1 call payoff   s2 put payoff   (PayoffFunction at `strike`, undiscounted)
//   s3 realised variance sum_j r_j^2, r_j = ln(S_j / S_{j-1})
//   s4, s5, s6  mean_j r_j^2 r_{j+L}^2 for L = 1, 8, 64 (clustering of squared returns: the Volterra structure as
//               it shows in the price matrix itself, so the same statistic can be taken from GPU paths)
//   s7 integrated variance sum_j v_j dt   s8 mean_n X_n^2   s9 mean_n X_n X_{n+1}
// Returns the number of threads used.
int ref_explicit_stats_omp(double S0, double r, double xi, double H, double eta, double rho, int steps, long total_paths,
                           double strike, double* out20) {
    const double dt = 1.0 / 252.0;
    double acc[20];
    for (double& a : acc) a = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
#pragma omp parallel
    {
        RoughVolatility rv;
        std::vector<double> grid((size_t)steps + 1);
        for (size_t i = 0; i <= (size_t)steps; ++i) grid[i] = i * dt;          // :337-340
        const std::vector<double> lambda = rv.rbergomiLambda(grid, H);          // :342
        const cvec phi = rv.rbergomiPhi(lambda, H);                             // :343
        double loc[20];
        for (double& a : loc) a = 0.0;
        std::vector<double> S((size_t)steps + 1), ret((size_t)steps);
#pragma omp for schedule(dynamic, 64)
        for (long i = 0; i < total_paths; ++i) {
            const cvec Z = rv.genComplexGaussians((size_t)steps);               // :347
            const std::vector<double> X = rv.fractionalGaussian(phi, Z, H, eta);  // :348
            const std::vector<double> v = rv.forwardVariance(X, grid, xi, H, eta);  // :349
            const std::vector<double> W1 = rv.gaussians((size_t)steps);         // :351
            const std::vector<double> W2 = rv.gaussians((size_t)steps);         // :352
            S[0] = S0;                                                          // :354
            for (size_t j = 1; j <= (size_t)steps; ++j) {                       // :355-364
                const double dw1 = std::sqrt(dt) * W1[j - 1];
                const double dw2 = std::sqrt(dt) * W2[j - 1];
                const double dW = rho * dw1 + std::sqrt(1.0 - rho * rho) * dw2;
                const double vt = v[j - 1];
                const double drift = (r - 0.5 * vt) * dt;
                const double diff = std::sqrt(std::max(0.0, vt)) * dW;
                S[j] = S[j - 1] * std::exp(drift + diff);
            }
            double st[10];
            st[0] = S[(size_t)steps];
            st[1] = PayoffFunction(true, st[0], strike);
            st[2] = PayoffFunction(false, st[0], strike);
            double rvar = 0.0, iv = 0.0, x2 = 0.0, x1 = 0.0;
            for (int j = 0; j < steps; ++j) {
                ret[(size_t)j] = std::log(S[(size_t)j + 1] / S[(size_t)j]);
                rvar += ret[(size_t)j] * ret[(size_t)j];
                iv += v[(size_t)j] * dt;
                x2 += X[(size_t)j] * X[(size_t)j];
                if (j + 1 < steps) x1 += X[(size_t)j] * X[(size_t)j + 1];
            }
            st[3] = rvar;
            const int lags[3] = {1, 8, 64};
            for (int q = 0; q < 3; ++q) {
                const int L = lags[q];
                double c = 0.0;
                for (int j = 0; j + L < steps; ++j) c += ret[(size_t)j] * ret[(size_t)j] * ret[(size_t)(j + L)] * ret[(size_t)(j + L)];
                st[4 + q] = steps > L ? c / (steps - L) : 0.0;
            }
            st[7] = iv;
            st[8] = x2 / steps;
            st[9] = steps > 1 ? x1 / (steps - 1) : 0.0;
            for (int q = 0; q < 10; ++q) {
                loc[q] += st[q];
                loc[10 + q] += st[q] * st[q];
            }
        }
#pragma omp critical
        for (int q = 0; q < 20; ++q) acc[q] += loc[q];
    }
    for (int q = 0; q < 20; ++q) out20[q] = acc[q];
    return threads;
}

// AsymptoticAnalysis::PredictOptionPrice (src/models/AsymptoticAnalysisPricer.cpp:38-113) on a
// row-major [n][m] matrix.  Returns 0, or 1 when the reference throws (message in err).
int ref_asymptotic_price(const double* row_major, long n, int m, double r, double strike, double maturity,
                         double dt, int is_call, double sigma, double dividend, double* price, char* err,
                         size_t errlen) {
    try {
        std::vector<std::vector<double>> paths((size_t)std::max<long>(n, 0));
        for (long i = 0; i < n; ++i) paths[i].assign(row_major + (size_t)i * m, row_major + (size_t)(i + 1) * m);
        AsymptoticAnalysis aa;
        *price = aa.PredictOptionPrice(paths, r, strike, maturity, dt, is_call != 0, sigma, dividend);
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen) {
            std::strncpy(err, e.what(), errlen - 1);
            err[errlen - 1] = 0;
        }
        return 1;
    }
}

// BranchingProcesses (src/models/BranchingProcessPricer.cpp:12-134) on a row-major [n][m] matrix.
// out3 = {price, lower bound, upper bound}; the upper bound resamples with an UNSEEDED mt19937 (:83-85).
// The bounds are private members, reached like RoughVolatility's helpers.  Returns 1 on a throw.
int ref_branching_price(const double* row_major, long n, int m, double r, double strike, double maturity, double dt,
                        int is_call, int num_branches, const int* ex, int n_ex, double* out3, char* err, size_t errlen) {
    try {
        std::vector<std::vector<double>> paths((size_t)std::max<long>(n, 0));
        for (long i = 0; i < n; ++i) paths[i].assign(row_major + (size_t)i * m, row_major + (size_t)(i + 1) * m);
        std::vector<int> times(ex, ex + std::max(n_ex, 0));
        BranchingProcesses bp;
#ifdef _OPENMP
        // In the reference's driver the pricers run inside an outer `omp parallel` (PredictionGen.cpp:542), so
        // their own `omp parallel for` loops are nested regions and execute on ONE thread.  Called from a
        // serial context they would share one mt19937 between threads (:83-93, a data race that biases the
        // upper bound: 14.30 vs 14.05 on the fixture).  Reproduce the driver's conditions.
        const int saved_threads = omp_get_max_threads();
        omp_set_num_threads(1);
#endif
        out3[0] = bp.PredictOptionPrice(paths, r, strike, maturity, dt, is_call != 0, num_branches, times);
        out3[1] = bp.ComputeLowerBound(paths, r, strike, dt, is_call != 0, times);   // maturity_ was set by the call above
        out3[2] = bp.ComputeUpperBound(paths, r, strike, dt, is_call != 0, num_branches, times);
#ifdef _OPENMP
        omp_set_num_threads(saved_threads);
#endif
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen) {
            std::strncpy(err, e.what(), errlen - 1);
            err[errlen - 1] = 0;
        }
        return 1;
    }
}

// ---- CPU baselines of bench.py's widened rows, timed the way the reference's driver runs its pricers:
// `#pragma omp parallel for schedule(dynamic)` over option rows, 250 paths per row (src/core/PredictionGen.cpp:542-546,
// :719), every pricer a fresh object per row (:566-570), the pricers' own parallel loops nested and therefore serial.

// One pricer of the compiled reference over a resident sample: the [n_total][m] matrix is cut into rows of `chunk` paths
// (built as vector<vector<double>> BEFORE the clock starts: in the driver they come out of the generator like that), each
// priced by one PredictOptionPrice call.  which = 0: AsymptoticAnalysis (AsymptoticAnalysisPricer.cpp:38-113), 1:
// BranchingProcesses (BranchingProcessPricer.cpp:12-134; exercise times 0..m-2 as :780-783).  seconds = wall time of the
// parallel loop, checksum = sum of the prices.  Returns the number of threads.
int ref_pricer_chunks_omp(int which, const double* row_major, long n_total, int m, int chunk, double r, double strike,
                          double maturity, double dt, int is_call, double sigma, double dividend, int num_branches,
                          double* seconds, double* checksum) {
    const long n_chunks = n_total / chunk;
    std::vector<std::vector<std::vector<double>>> rows((size_t)n_chunks);
    for (long c = 0; c < n_chunks; ++c) {
        rows[c].resize((size_t)chunk);
        for (int i = 0; i < chunk; ++i) {
            const double* src = row_major + ((size_t)c * chunk + i) * m;
            rows[c][i].assign(src, src + m);
        }
    }
    std::vector<int> times((size_t)std::max(m - 1, 0));
    for (int i = 0; i + 1 < m; ++i) times[i] = i;
    double acc = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
    const double t0 = omp_get_wtime();
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : acc)
    for (long c = 0; c < n_chunks; ++c) {
        try {
            if (which == 0) {
                AsymptoticAnalysis aa;
                acc += aa.PredictOptionPrice(rows[c], r, strike, maturity, dt, is_call != 0, sigma, dividend);
            } else {
                BranchingProcesses bp;
                acc += bp.PredictOptionPrice(rows[c], r, strike, maturity, dt, is_call != 0, num_branches, times);
            }
        } catch (const std::exception&) {
        }
    }
#ifdef _OPENMP
    *seconds = omp_get_wtime() - t0;
#else
    *seconds = 0.0;
#endif
    *checksum = acc;
    return threads;
}

// Whole driver rows (PredictionGen.cpp:719-791): GenerateStockPricePaths(hist, steps, paths_per_row), the finiteness scan,
// then the four pricers.  AsymptoticAnalysis and BranchingProcesses are the compiled reference; LSMPricer.cpp and
// MartingaleOptimizationPricer.cpp need Eigen and cannot be built here, so those two calls go through `lsm` / `mart`:
// pointers to the repo's CPU restatement (orc_lsm_price / orc_martingale_price in oracle/libmcgoracle.so), fed the row's
// matrix flattened.  seconds = wall time of the parallel loop, sums4 = sum of each pricer's price over the rows.
typedef int (*lsm_fn_t)(const double*, size_t, size_t, long, int, double, double, double, double, int, int, double*, double*);
typedef int (*mart_fn_t)(const double*, size_t, size_t, long, int, double, double, double, double, int, int, int, double*, double*, double*);
int ref_driver_rows_omp(const double* hist, size_t n, const int* steps, const double* strike, const int* is_call, long n_rows,
                        int paths_per_row, double sigma, double dividend, void* lsm, void* mart, double* seconds, double* sums4) {
    std::vector<double> h(hist, hist + n);
    const lsm_fn_t lsm_price = (lsm_fn_t)lsm;
    const mart_fn_t mart_price = (mart_fn_t)mart;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
    const double t0 = omp_get_wtime();
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : a0, a1, a2, a3)
    for (long row = 0; row < n_rows; ++row) {
        const int st = steps[row];
        const double r = 0.04, dt = 1.0 / 252.0, maturity = st / 252.0;   // (:700-702)
        try {
            RoughVolatility rv;
            AsymptoticAnalysis aa;
            BranchingProcesses bp;
            auto paths = rv.GenerateStockPricePaths(h, st, paths_per_row);
            bool valid = !paths.empty();
            for (const auto& p : paths)
                for (double px : p) valid = valid && std::isfinite(px);
            if (!valid) continue;
            std::vector<int> times((size_t)st);
            for (int i = 0; i < st; ++i) times[i] = i;
            a0 += aa.PredictOptionPrice(paths, r, strike[row], maturity, dt, is_call[row] != 0, sigma, dividend);
            a1 += bp.PredictOptionPrice(paths, r, strike[row], maturity, dt, is_call[row] != 0, 10, times);
            if (lsm_price && mart_price) {
                std::vector<double> flat((size_t)paths_per_row * (st + 1));
                for (int i = 0; i < paths_per_row; ++i) std::copy(paths[i].begin(), paths[i].end(), flat.begin() + (size_t)i * (st + 1));
                double v = 0.0, lo = 0.0, up = 0.0;
                if (lsm_price(flat.data(), (size_t)(st + 1), 1, paths_per_row, st + 1, r, strike[row], maturity, dt, is_call[row], 2, &v, nullptr) == 0) a2 += v;
                if (mart_price(flat.data(), (size_t)(st + 1), 1, paths_per_row, st + 1, r, strike[row], maturity, dt, is_call[row], 2, 5, &v, &lo, &up) == 0) a3 += v;
            }
        } catch (const std::exception&) {
        }
    }
#ifdef _OPENMP
    *seconds = omp_get_wtime() - t0;
#else
    *seconds = 0.0;
#endif
    sums4[0] = a0;
    sums4[1] = a1;
    sums4[2] = a2;
    sums4[3] = a3;
    return threads;
}

}  // extern "C"
