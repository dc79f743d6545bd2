// Batched driver rows: the per-row work of the reference's production caller
// (/root/reference/src/core/PredictionGen.cpp:700-791 -- 250 rBergomi paths, then AsymptoticAnalysis,
// BranchingProcesses(10 branches, exercise dates 0..steps-1), LSM(polyOrder 2), MartingaleOptimization(2))
// for MANY option rows in six launches instead of ~15 launches and ~8 host synchronisations per row:
//   k_batch_weights    one workgroup per row: lambda -> |phi_k|^2 -> spectral amplitudes a_k and compensator
//                      (the host/volterra.cpp math, done with a direct DFT against an LDS table of roots of unity)
//   k_batch_paths      a few workgroups per row: the FFT generator of rbergomi_device.hpp
//   k_batch_asym / _branching / _lsm / _martingale    one workgroup per row (n_paths <= 256: one path per
//                      thread), all reductions inside the workgroup, regression solves on thread 0
// Row c uses Philox path ids (c << 32) + p, so its four prices equal those of the single-contract entry
// points called with path_begin = c << 32 (up to the ~1e-13 difference between the device DFT and the host FFT
// in the amplitudes).  Matrix layout: step-major, row c owns columns [256 c, 256 c + n_paths).
#include <cmath>

#include "lsm_device.hpp"
#include "mcg_internal.hpp"
#include "rbergomi_device.hpp"

namespace mcg {

// steps of a row the row kernels serve (their transforms go up to Mz = 1024, their per-row LDS tables with them); longer
// rows take the single-contract entry points (run_batch_rows)
constexpr int BATCH_MAX_STEPS = 1020;

struct BatchRow {  // device image of one option row
    double S0, logS0, xi, H, eta, strike, maturity, sigma, dividend;
    int n_steps, M, is_call, valid;
};

struct BatchArgs {
    const BatchRow* rows;
    int64_t n_rows;
    int n_paths, max_steps, m_max;
    double r, dt, sqdt, disc;  // disc = exp(-r dt)
    uint32_t k0, k1;
    double* amp;  // [n_rows][m_max] spectral amplitudes a_k of each row
    double* comp;   // [n_rows][max_steps]
    double* S;      // [(max_steps+1)][ld]
    int64_t ld;     // n_rows * 256
    const double2* log_tab;
    double* out;    // [n_rows][4]: asymptotic, branching, lsm, martingale
    int num_branches, max_iterations;
};

__device__ __forceinline__ int next_pow2_dev(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ---- weights ---------------------------------------------------------------------------------
// LDS: ct/st[Mphi] roots of unity, lam[steps+1], P[M] (then amp in place)
__global__ __launch_bounds__(256) void k_batch_weights(BatchArgs a) {
    extern __shared__ double sm[];
    const BatchRow row = a.rows[blockIdx.x];
    if (!row.valid) return;
    const int steps = row.n_steps, M = row.M, Mphi = next_pow2_dev(steps + 1);
    double* ct = sm;
    double* st = ct + Mphi;
    double* lam = st + Mphi;
    double* P = lam + (steps + 1);
    for (int q = threadIdx.x; q < Mphi; q += 256) {
        double s, c;
        sincospi(2.0 * (double)q / (double)Mphi, &s, &c);
        ct[q] = c;
        st[q] = s;
    }
    for (int i = threadIdx.x; i <= steps; i += 256) lam[i] = 0.5 * pow((double)i * a.dt, 2.0 * row.H);
    __syncthreads();
    for (int k = threadIdx.x; k < M; k += 256) {
        double p = 0.0;
        if (k < steps) {  // |phi_k|^2, phi = sum_n lam_n e^{+2 pi i k n / Mphi}  (RoughVolatility.cpp:212-225)
            double re = 0.0, im = 0.0;
            for (int n = 0; n <= steps; ++n) {
                const int q = (k * n) & (Mphi - 1);
                re = fma(lam[n], ct[q], re);
                im = fma(lam[n], st[q], im);
            }
            p = re * re + im * im;
        }
        P[k] = p;
    }
    __syncthreads();
    // a_k = eta sqrt(2H)/M * sqrt((P_k + P_{M-k})/2): the symmetric spectral amplitudes of host/volterra.cpp
    const double scale = row.eta * sqrt(2.0 * row.H) / (double)M;
    double* amp = a.amp + (int64_t)blockIdx.x * a.m_max;
    for (int k = threadIdx.x; k < M; k += 256) amp[k] = scale * sqrt(0.5 * (P[k] + P[(M - k) & (M - 1)]));
    double* cmp = a.comp + (int64_t)blockIdx.x * a.max_steps;
    for (int n = threadIdx.x; n < steps; n += 256) cmp[n] = -0.5 * row.eta * row.eta * pow((double)n * a.dt, 2.0 * row.H);
}

// ---- paths -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_batch_paths(BatchArgs a, int blocks_per_row) {
    extern __shared__ double smem[];
    __shared__ fm::Tables tabs;
    const int64_t r_idx = blockIdx.x / blocks_per_row;
    const int sub = (int)(blockIdx.x % blocks_per_row);
    const BatchRow row = a.rows[r_idx];
    if (!row.valid) return;
    const int n_pairs = (a.n_paths + 1) / 2;
    if ((int64_t)sub * rb_pairs_per_block(row.M) >= n_pairs) return;  // this row needs fewer workgroups than the widest
    RbArgs g;
    g.out = a.S + r_idx * 256;
    g.ld = a.ld;
    g.n_paths = a.n_paths;
    g.n_steps = row.n_steps;
    g.M = row.M;
    g.path_begin = (uint64_t)r_idx << 32;
    g.k0 = a.k0;
    g.k1 = a.k1;
    g.S0 = row.S0;
    g.logS0 = row.logS0;
    g.r = a.r;
    g.xi = row.xi;
    g.dt = a.dt;
    g.sqdt = a.sqdt;
    g.amp = a.amp + r_idx * a.m_max;
    g.comp = a.comp + r_idx * a.max_steps;
    g.log_tab = a.log_tab;
    g.K = 0.0;
    g.is_call = 0;
    g.partials = nullptr;
    g.n_blocks = 0;
    g.ticket = nullptr;
    double la, lb;
    bool va, vb, lead;
    switch (row.M) {  // wave-uniform
        case 32: rb_generate_fft<0 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        case 64: rb_generate_fft<1 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        case 128: rb_generate_fft<2 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        case 256: rb_generate_fft<3 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        case 512: rb_generate_fft<4 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        case 1024: rb_generate_fft<5 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        default: rb_generate_small(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
    }
}

// ---- AsymptoticAnalysis (AsymptoticAnalysisPricer.cpp:38-113), one workgroup per row ------------
__global__ __launch_bounds__(256) void k_batch_asym(BatchArgs a) {
    extern __shared__ double sm[];  // bnd[n_cols], disc[n_cols]
    __shared__ double red[2 * 4];
    const BatchRow row = a.rows[blockIdx.x];
    if (!row.valid) return;
    const int n_cols = row.n_steps + 1;
    double* bnd = sm;
    double* dsc = sm + (a.max_steps + 1);
    const bool call = row.is_call != 0;
    for (int j = threadIdx.x; j < n_cols; j += 256) {
        const double t = j * a.dt;
        const double eps = row.maturity - t;
        double b = row.strike;
        if (!(eps < 1e-10)) {
            const double hw = 0.5 * row.sigma * sqrt(eps * log(1.0 / eps));
            if (call) {
                b = row.strike - hw;
                if (eps < 0.01) b += 0.5 * (row.dividend - a.r) * eps;
            } else {
                b = row.strike + hw;
                if (eps < 0.01) b -= 0.5 * (a.r - row.dividend) * eps;
            }
        }
        bnd[j] = b;
        dsc[j] = exp(-a.r * t);
    }
    __syncthreads();
    double v[2] = {0.0, 0.0};
    const int p = threadIdx.x;
    if (p < a.n_paths) {
        const double* col = a.S + (int64_t)blockIdx.x * 256 + p;
        double best = 0.0;
        for (int j = 0; j < n_cols; ++j) {
            if (j * a.dt > row.maturity) break;
            const double S = col[(int64_t)j * a.ld];
            if (isnan(S) || isinf(S)) continue;
            const bool in = call ? (S > bnd[j]) : (S < bnd[j]);
            if (in) {
                const double d = dsc[j] * payoff_of(call, S, row.strike);
                if (d > best) best = d;
            }
        }
        if (!isnan(best) && !isinf(best)) {
            v[0] = best;
            v[1] = 1.0;
        }
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) a.out[4 * (int64_t)blockIdx.x + 0] = v[1] > 0.0 ? v[0] / v[1] : 0.0;
}

// ---- BranchingProcesses (BranchingProcessPricer.cpp:12-134), exercise dates 0..steps-1 ----------
// One workgroup per row, one path per thread.  The reference's continuation at date e averages, over `numBranches`
// uniformly resampled paths, the best discounted payoff over all LATER columns (:104-121) = the suffix maximum
// F[e+1][rp].  The dates are walked BACKWARDS: every thread carries its own path's running suffix maximum in a register,
// publishes it for the current date in one of two alternating LDS rows (one barrier per date) and gathers its resampled
// paths' values from there -- the suffix-maximum matrix never exists in memory (until round 3 it was written to and
// gathered from global memory: 3.8 ms of the batch's 9 ms at 20 000 rows).  The upper bound is a maximum over the dates
// and the lower bound the FIRST date with a positive payoff, i.e. the one found last on the way back: the order of the
// walk does not matter.  Same Philox blocks (counter = date, branch quad) as the single-contract kernel.
__global__ __launch_bounds__(256) void k_batch_branching(BatchArgs a) {
    extern __shared__ double sm[];  // disc[n_cols]
    __shared__ double red[2 * 4];
    __shared__ double frow[2][256];
    const BatchRow row = a.rows[blockIdx.x];
    if (!row.valid) return;
    const int n_cols = row.n_steps + 1;
    const bool call = row.is_call != 0;
    double* dsc = sm;
    for (int j = threadIdx.x; j < n_cols; j += 256) dsc[j] = exp(-a.r * (j * a.dt));
    __syncthreads();
    int n_dates = 0;
    while (n_dates < n_cols && !(n_dates * a.dt > row.maturity)) ++n_dates;
    const int p = threadIdx.x;
    const bool live = p < a.n_paths;
    const double* col = a.S + (int64_t)blockIdx.x * 256 + (live ? p : 0);
    const uint64_t id = ((uint64_t)blockIdx.x << 32) + (uint64_t)p;
    const PhiloxLane lane_rng = philox_lane_setup(id, 2u, a.k1);
    const int quads = (a.num_branches + 3) >> 2;
    const double inv_b = a.num_branches > 0 ? 1.0 / (double)a.num_branches : 0.0;
    const int ex_last = row.n_steps - 1;
    // run = F[j][p] = max_{k >= j, k < n_dates} disc_k payoff_k, floored at 0; start at j = n_cols - 1 (the last column,
    // which is no exercise date of the driver's list but counts as a later column)
    double run = 0.0;
    {
        const int j = n_cols - 1;
        if (j < n_dates) run = fmax(run, dsc[j] * payoff_of(call, col[(int64_t)j * a.ld], row.strike));
    }
    double lower = 0.0, upper = 0.0;
    for (int e = row.n_steps - 1; e >= 0; --e) {  // exercise date index == column index; run == F[e+1][p] here
        const double now = dsc[e] * payoff_of(call, col[(int64_t)e * a.ld], row.strike);
        const bool is_date = !(e * a.dt > row.maturity);  // (:94-96: the reference stops at the first date beyond maturity)
        double* mine = frow[e & 1];
        mine[p] = run;
        __syncthreads();  // (two rows alternate: the gathers of date e+1 are over before anybody writes that row again at e-1)
        if (is_date) {
            if (now > 0.0) lower = now;  // walking back: the earliest such date wins
            double better = now;
            if (e < ex_last && a.num_branches > 0) {
                double sum = 0.0;
                for (int q = 0; q < quads; ++q) {
                    const Philox4 w = philox4x32_10_lane(lane_rng, (uint32_t)(e * quads + q), a.k0, a.k1);
                    const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (4 * q + s < a.num_branches) sum += mine[__umulhi(ws[s], (uint32_t)a.n_paths)];
                }
                const double cont = sum * inv_b;
                if (cont > better) better = cont;
            }
            if (better > upper) upper = better;
        }
        if (e < n_dates && now > run) run = now;
    }
    double v[2] = {live ? lower : 0.0, live ? upper : 0.0};
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) a.out[4 * (int64_t)blockIdx.x + 1] = 0.5 * (v[0] + v[1]) / (double)a.n_paths;
}

// ---- LSM (LSMPricer.cpp:19-102) ------------------------------------------------------------------
// One wavefront per row, four rows per workgroup (lsm_wave_body).
template <int NB>
__global__ __launch_bounds__(256) void k_batch_lsm(BatchArgs a) {
    __shared__ double ws[4][lsm_ws_doubles(NB) + LSM_COEF_DOUBLES];  // per wave: workspace of a refined date's solve
    const int64_t r_idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r_idx >= a.n_rows) return;
    const BatchRow row = a.rows[r_idx];
    if (!row.valid) return;
    double sum_v, sum_v2;
    lsm_wave_body<NB>(a.S + r_idx * 256, a.ld, a.n_paths, row.n_steps + 1, row.strike, row.maturity, a.dt, a.disc, row.is_call,
                      ws[threadIdx.x >> 6], sum_v, sum_v2);
    if ((threadIdx.x & 63) == 0) a.out[4 * r_idx + 2] = sum_v / (double)a.n_paths;
}

// ---- MartingaleOptimization (MartingaleOptimizationPricer.cpp:21-189) ----------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_batch_martingale(BatchArgs a) {
    constexpr int NM = 3 * NB - 1;
    extern __shared__ double sm[];  // disc[n_cols] with the maturity clamp
    __shared__ double red[(NM + 1) * 4];
    __shared__ double sm_mom[32];
    __shared__ double sm_coef[LSM_COEF_STRIDE];
    __shared__ double sm_off, sm_primal;
    __shared__ double sm_ws[lsm_ws_doubles(NB)];
    const BatchRow row = a.rows[blockIdx.x];
    if (!row.valid) return;
    const int n_cols = row.n_steps + 1;
    const bool call = row.is_call != 0;
    const double invK = 1.0 / row.strike;
    double* dsc = sm;
    for (int j = threadIdx.x; j < n_cols; j += 256) {
        double t = j * a.dt;
        if (t > row.maturity) t = row.maturity;
        dsc[j] = exp(-a.r * t);
    }
    __syncthreads();
    int n_dates = 0;
    while (n_dates < n_cols && !(n_dates * a.dt > row.maturity)) ++n_dates;
    const int p = threadIdx.x;
    const bool live = p < a.n_paths;
    const double* col = a.S + (int64_t)blockIdx.x * 256 + p;
    double m[NM + 1];
    double xs[2] = {0.0, 0.0}, ys[2] = {0.0, 0.0};  // this path's two regression samples (kept for a re-fit)
#pragma unroll
    for (int q = 0; q <= NM; ++q) m[q] = 0.0;
    if (live) {
        double best = 0.0;
        int stop = 0;
        for (int j = 0; j < n_dates; ++j) {
            const double d = payoff_of(call, col[(int64_t)j * a.ld], row.strike) * dsc[j];
            if (d > best) {
                best = d;
                stop = j;
            }
        }
        m[NM] = best;
        const int other = (stop + n_cols / 2) % n_cols;
        xs[0] = col[(int64_t)stop * a.ld];
        xs[1] = col[(int64_t)other * a.ld];
        ys[0] = 0.5 * (payoff_of(call, xs[0], row.strike) * dsc[stop]);
        ys[1] = 0.2 * (payoff_of(call, xs[1], row.strike) * dsc[other]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const double x = fma(xs[s], invK, -1.0);
            double pw = 1.0;
#pragma unroll
            for (int q = 0; q < 2 * NB - 1; ++q) {
                m[q] += pw;
                if (q < NB) m[2 * NB - 1 + q] = fma(pw, ys[s], m[2 * NB - 1 + q]);
                pw *= x;
            }
        }
    }
    block_sum<NM + 1, 4>(m, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q <= NM; ++q) sm_mom[q] = m[q];
        sm_primal = m[NM];
        lsm_solve_nb<NB>(sm_mom, (double)NB, row.strike, sm_coef);
    }
    __syncthreads();
    if (sm_coef[LSM_C_REFINE] != 0.0) {  // workgroup-uniform: re-fit about the samples' mean (lsm_solve_nb; MartingaleOptimizationPricer.cpp:166)
        const double mu = sm_coef[LSM_C_HINT];
        double mc[NM];
#pragma unroll
        for (int q = 0; q < NM; ++q) mc[q] = 0.0;
        if (live) {
#pragma unroll
            for (int s = 0; s < 2; ++s) lsm_accumulate_centered<NB>(mc, true, xs[s], ys[s], invK, mu, 1.0);
        }
        block_sum<NM, 4>(mc, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NM; ++q) sm_mom[q] = mc[q];
            lsm_solve_centered(sm_mom, NB, mu, row.strike, sm_coef, sm_ws);
        }
        __syncthreads();
    }
    const double primal = sm_primal / (double)a.n_paths;
    double c[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) c[q] = sm_coef[q];
    const double center = sm_coef[LSM_C_CENTER];
    auto poly = [&](double S) { return lsm_continuation<NB>(c, center, fma(S, invK, -1.0)); };
    double o[2] = {live ? poly(col[0]) : 0.0, 0.0};
    __syncthreads();  // red is reused
    block_sum<2, 4>(o, red);
    if (threadIdx.x == 0) sm_off = o[0] / (double)a.n_paths;
    __syncthreads();
    const double offset = sm_off;
    double d2[2] = {0.0, 0.0};
    if (live) {
        double best = 0.0;
        for (int j = 0; j < n_dates; ++j) {
            const double S = col[(int64_t)j * a.ld];
            const double cand = payoff_of(call, S, row.strike) * dsc[j] - (poly(S) - offset);
            if (cand > best) best = cand;
        }
        d2[0] = best;
    }
    __syncthreads();
    block_sum<2, 4>(d2, red);
    if (threadIdx.x == 0) {
        const double dual = a.max_iterations >= 2 ? d2[0] / (double)a.n_paths : primal;
        a.out[4 * (int64_t)blockIdx.x + 3] = 0.5 * (primal + dual);
    }
}

template <int NB>
static void launch_row_regressions(mcg_ctx* ctx, const BatchArgs& a, size_t smem_cols) {
    hipLaunchKernelGGL(k_batch_lsm<NB>, dim3((unsigned)((a.n_rows + 3) / 4)), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_batch_martingale<NB>, dim3((unsigned)a.n_rows), dim3(256), smem_cols, ctx->stream, a);
}

int run_batch_rows(mcg_ctx* ctx, const mcg_row* rows, int64_t n_rows, int n_paths, double r, double dt, int num_branches,
                   int poly_order, int max_iterations, uint64_t seed, double* out) {
    std::vector<BatchRow> h((size_t)n_rows);
    std::vector<int64_t> long_rows;
    int max_steps = 1, m_max = 1;
    for (int64_t i = 0; i < n_rows; ++i) {
        const mcg_row& s = rows[i];
        BatchRow& d = h[(size_t)i];
        d.S0 = s.S0;
        d.logS0 = s.S0 > 0.0 ? std::log(s.S0) : 0.0;
        d.xi = s.xi;
        d.H = s.H;
        d.eta = s.eta;
        d.strike = s.strike;
        d.maturity = s.maturity;
        d.sigma = s.sigma;
        d.dividend = s.dividend;
        d.n_steps = s.n_steps;
        d.is_call = s.is_call;
        // a row the reference's driver would answer with zeros (no steps, non-finite paths, a throwing pricer)
        d.valid = s.n_steps >= 1 && s.S0 > 0.0 && std::isfinite(s.S0) && s.xi >= 0.0 &&
                  std::isfinite(s.xi) && s.H >= 0.0 && std::isfinite(s.H) && std::isfinite(s.eta) &&
                  std::fabs(s.rho) <= 1.0 && s.strike > 0.0 && std::isfinite(s.strike) && s.sigma > 0.0 &&
                  std::isfinite(s.maturity);
        // A row longer than the row kernels' LDS tables reach (more than four years of trading days), or any row of a call
        // with more than 256 paths per row or an order above 4, is priced after the batch through the single-contract
        // entry points, on the same Philox path ids: never refused, never answered with zeros.
        if (d.valid && (s.n_steps > BATCH_MAX_STEPS || n_paths > 256 || poly_order > 4)) {
            long_rows.push_back(i);
            d.valid = 0;
        }
        d.M = 1;
        if (d.valid) {
            while (d.M < d.n_steps) d.M <<= 1;
            max_steps = std::max(max_steps, d.n_steps);
            m_max = std::max(m_max, d.M);
        }
    }
    const int64_t ld = n_rows * 256;
    const size_t mat_bytes = (size_t)ld * (size_t)(max_steps + 1) * sizeof(double);
    const size_t small_doubles = (size_t)n_rows * ((size_t)m_max + (size_t)max_steps + 4) + (sizeof(BatchRow) * (size_t)n_rows + 7) / 8;
    void *S = nullptr, *small = nullptr;
    int rc = pool_alloc(ctx, mat_bytes, &S);
    if (rc) return rc;
    rc = pool_alloc(ctx, small_doubles * sizeof(double), &small);
    if (rc) {
        pool_release(ctx, S, mat_bytes);
        return rc;
    }
    auto release_all = [&] {
        pool_release(ctx, S, mat_bytes);
        pool_release(ctx, small, small_doubles * sizeof(double));
    };
    BatchArgs a;
    double* sd = (double*)small;
    a.out = sd;
    a.amp = sd + 4 * n_rows;
    a.comp = a.amp + (size_t)n_rows * m_max;
    a.rows = reinterpret_cast<const BatchRow*>(a.comp + (size_t)n_rows * max_steps);
    a.n_rows = n_rows;
    a.n_paths = n_paths;
    a.max_steps = max_steps;
    a.m_max = m_max;
    a.r = r;
    a.dt = dt;
    a.sqdt = std::sqrt(dt);
    a.disc = std::exp(-r * dt);
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.S = (double*)S;
    a.ld = ld;
    a.log_tab = (const double2*)ctx->log_tab;
    a.num_branches = num_branches;
    a.max_iterations = max_iterations;

    hipError_t e = hipMemcpyAsync((void*)a.rows, h.data(), sizeof(BatchRow) * (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(a.out, 0, 4 * sizeof(double) * (size_t)n_rows, ctx->stream);
    if (e != hipSuccess) {
        release_all();
        return fail(MCG_ERR_HIP, "batch upload failed: %s", hipGetErrorString(e));
    }
    const int mphi_max = 2 * m_max >= 2 ? 2 * m_max : 2;
    const size_t smem_w = ((size_t)2 * mphi_max + (size_t)max_steps + 1 + (size_t)m_max) * sizeof(double);
    size_t smem_p = 0;  // largest over the transform sizes present (the staging part does not grow with Mz)
    for (int m = 1; m <= m_max; m <<= 1) smem_p = std::max(smem_p, rb_smem_bytes(m, std::min(m, max_steps)));
    const size_t smem_c = ((size_t)max_steps + 1) * sizeof(double);
    // workgroups per row: enough for the row with the fewest pairs per workgroup (the largest Mz)
    const int n_pairs = (n_paths + 1) / 2;
    int bpr = 1;
    for (int m = 32; m <= m_max; m <<= 1) bpr = std::max(bpr, (n_pairs + rb_pairs_per_block(m) - 1) / rb_pairs_per_block(m));
    {
        TimedLaunch t(ctx, MCG_K_BATCH);
        hipLaunchKernelGGL(k_batch_weights, dim3((unsigned)n_rows), dim3(256), smem_w, ctx->stream, a);
        if (smem_p > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)k_batch_paths, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_p);
        hipLaunchKernelGGL(k_batch_paths, dim3((unsigned)(n_rows * bpr)), dim3(256), smem_p, ctx->stream, a, bpr);
        hipLaunchKernelGGL(k_batch_asym, dim3((unsigned)n_rows), dim3(256), 2 * smem_c, ctx->stream, a);
        hipLaunchKernelGGL(k_batch_branching, dim3((unsigned)n_rows), dim3(256), smem_c, ctx->stream, a);
        switch (poly_order + 1) {
            case 1: launch_row_regressions<1>(ctx, a, smem_c); break;
            case 2: launch_row_regressions<2>(ctx, a, smem_c); break;
            case 3: launch_row_regressions<3>(ctx, a, smem_c); break;
            case 4: launch_row_regressions<4>(ctx, a, smem_c); break;
            default: launch_row_regressions<5>(ctx, a, smem_c); break;
        }
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, a.out, 4 * sizeof(double) * (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    // the host vector `h` must outlive the upload: it does (synchronised above)
    release_all();
    if (e != hipSuccess) return fail(MCG_ERR_HIP, "batch run failed: %s", hipGetErrorString(e));
    for (int64_t i : long_rows) {  // PredictionGen.cpp:718-816 for one row, through the single-contract entry points
        const mcg_row& s = rows[i];
        double* o = out + 4 * i;
        o[0] = o[1] = o[2] = o[3] = 0.0;
        mcg_paths* P = nullptr;
        if (mcg_paths_rbergomi(ctx, seed, s.S0, r, s.xi, s.H, s.eta, s.rho, dt, s.n_steps, (uint64_t)i << 32, n_paths, &P) != MCG_OK)
            continue;  // (the driver logs a failing row and writes zeros, :792-805)
        std::vector<int> ex((size_t)s.n_steps);
        for (int t = 0; t < s.n_steps; ++t) ex[(size_t)t] = t;  // :780-783
        double v = 0.0;
        if (mcg_price_asymptotic(ctx, P, r, s.strike, s.maturity, dt, s.is_call, s.sigma, s.dividend, &v) == MCG_OK) o[0] = v;
        if (mcg_price_branching(ctx, P, r, s.strike, s.maturity, dt, s.is_call, num_branches, ex.data(), s.n_steps, seed, &v,
                                nullptr, nullptr) == MCG_OK)
            o[1] = v;
        if (mcg_price_lsm(ctx, P, r, s.strike, s.maturity, dt, s.is_call, poly_order, &v, nullptr) == MCG_OK) o[2] = v;
        if (mcg_price_martingale(ctx, P, r, s.strike, s.maturity, dt, s.is_call, poly_order, max_iterations, &v, nullptr,
                                 nullptr) == MCG_OK)
            o[3] = v;
        mcg_paths_free(P);
    }
    return MCG_OK;
}

}  // namespace mcg
