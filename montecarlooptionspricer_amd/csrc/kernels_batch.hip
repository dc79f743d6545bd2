// Batched driver rows: the per-row work of the reference's production caller
// (/root/reference/src/core/PredictionGen.cpp:700-791 -- 250 rBergomi paths, then AsymptoticAnalysis,
// BranchingProcesses(10 branches, exercise dates 0..steps-1), LSM(polyOrder 2), MartingaleOptimization(2))
// for MANY option rows in six launches (per chunk of rows, run_batch_rows) instead of ~15 launches and ~8 host
// synchronisations per row:
//   k_batch_weights    one workgroup per row: lambda -> |phi_k|^2 -> spectral amplitudes a_k and compensator
//                      (the host/volterra.cpp math, done with a direct DFT against an LDS table of roots of unity)
//   k_batch_paths      a few workgroups per row: the FFT generator of rbergomi_device.hpp
//   k_batch_asym / _branching / _lsm / _martingale    one workgroup per row (n_paths <= 256: one path per
//                      thread), all reductions inside the workgroup, regression solves on thread 0
// Row c uses Philox path ids (c << 32) + p, so its four prices equal those of the single-contract entry
// points called with path_begin = c << 32 (up to the ~1e-13 difference between the device DFT and the host FFT
// in the amplitudes).  Matrix layout: every row owns ONE contiguous block of (n_steps + 1) x 256 doubles, step-major
// inside it (element (j, p) at off + 256 j + p) -- nothing is padded to the longest row, consecutive steps of a row are
// 2 KiB apart instead of n_rows x 2 KiB -- and its amplitudes and compensator sit at an offset of their own.  Any number
// of rows: the call works through them in chunks under a memory budget (run_batch_rows); a row's Philox ids, and
// therefore its prices, do not depend on which chunk it falls into.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include <cstring>

#include "coalesce.hpp"
#include "lsm_device.hpp"
#include "mcg_internal.hpp"
#include "rbergomi_device.hpp"

namespace mcg {

// steps of a row the row kernels serve (their transforms go up to Mz = 1024, their per-row LDS tables with them); longer
// rows take the single-contract entry points (run_batch_rows)
constexpr int BATCH_MAX_STEPS = 1020;

struct BatchRow {  // device image of one option row
    double S0, logS0, xi, H, eta, strike, maturity, sigma, dividend;
    int64_t off;   // its block of the path matrix: S[off + 256 j + p]
    int64_t woff;  // its amplitudes w[woff .. woff + M), compensator w[woff + M .. woff + M + n_steps)
    uint64_t id;   // its index in the caller's array: Philox path ids (id << 32) + p
    uint32_t k0, k1;  // its Philox key: the call's seed (mcg_batch_price_rows) or the row's own (coalesced class-API calls, coalesce.hpp)
    int n_steps, M, is_call, valid;
};
constexpr int64_t BATCH_LD = 256;  // columns of a row's block (n_paths <= 256)

struct BatchArgs {
    const BatchRow* rows;
    int64_t n_rows;
    int n_paths, max_steps, m_max;
    double r, dt, sqdt, disc;  // disc = exp(-r dt)
    uint32_t k0, k1;
    double* w;      // amplitudes and compensators, row by row (BatchRow::woff)
    double* S;      // the rows' matrix blocks (BatchRow::off)
    const double2* log_tab;
    double* out;    // [n_rows][4]: asymptotic, branching, lsm, martingale (chunk-local row order)
    const uint32_t* wg_map;  // k_batch_paths: workgroup -> (chunk-local row << 6) | share of the row's pairs
    double* bad;    // [n_rows]: != 0 when the row's path block holds a non-finite price (the driver zeroes such a row, PredictionGen.cpp:752-777)
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
    double* amp = a.w + row.woff;
    for (int k = threadIdx.x; k < M; k += 256) amp[k] = scale * sqrt(0.5 * (P[k] + P[(M - k) & (M - 1)]));
    double* cmp = amp + M;
    for (int n = threadIdx.x; n < steps; n += 256) cmp[n] = -0.5 * row.eta * row.eta * pow((double)n * a.dt, 2.0 * row.H);
}

// ---- paths -----------------------------------------------------------------------------------
#ifndef MCG_BATCH_PATHS_WAVES
#define MCG_BATCH_PATHS_WAVES 2
#endif
// CLS = the LDS class of the launch's rows (lds_class: Mz <= 128 / 256 / 512 / 1024): a launch holds rows of ONE class, and its
// kernel only the transforms that class can need -- the register count of the Mz = 1024 variant is not what the many short
// rows of the driver (Mz <= 128) should pay for.
template <int CLS>
__global__ __launch_bounds__(256, MCG_BATCH_PATHS_WAVES) void k_batch_paths(BatchArgs a) {
    extern __shared__ double smem[];
    __shared__ fm::Tables tabs;
    // One workgroup per (row, share) that exists (run_batch_chunk's map).  Until round 5 the grid was rows x the widest row's
    // shares and the others left at once -- but a workgroup that leaves at once has still been given its 71 KB of LDS and 185
    // registers: 45 % of the 80 000 workgroups of a 20 000-row call queued for the two slots of a CU only to return.
    const uint32_t m = a.wg_map[blockIdx.x];
    const int64_t r_idx = m >> 6;
    const int sub = (int)(m & 63u);
    const BatchRow row = a.rows[r_idx];
    if (!row.valid) return;
    RbArgs g;
    g.out = a.S + row.off;
    g.ld = BATCH_LD;
    g.n_paths = a.n_paths;
    g.n_steps = row.n_steps;
    g.M = row.M;
    g.path_begin = row.id << 32;
    g.k0 = row.k0;
    g.k1 = row.k1;
    g.S0 = row.S0;
    g.logS0 = row.logS0;
    g.r = a.r;
    g.xi = row.xi;
    g.dt = a.dt;
    g.sqdt = a.sqdt;
    g.amp = a.w + row.woff;
    g.comp = g.amp + row.M;
    g.log_tab = a.log_tab;
    g.K = 0.0;
    g.is_call = 0;
    g.partials = nullptr;
    g.n_blocks = 0;
    g.ticket = nullptr;
    double la, lb;
    bool va, vb, lead;
    if constexpr (CLS == 0) {
        switch (row.M) {  // wave-uniform
            case 32: rb_generate_fft<0 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
            case 64: rb_generate_fft<1 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
            case 128: rb_generate_fft<2 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
            default: rb_generate_small(g, sub, smem, &tabs, la, lb, va, vb, lead); break;
        }
    } else if constexpr (CLS == 1) {
        rb_generate_fft<3 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead);
    } else if constexpr (CLS == 2) {
        rb_generate_fft<4 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead);
    } else {
        rb_generate_fft<5 + 3 - RB_LT_DEFAULT, RB_LT_DEFAULT>(g, sub, smem, &tabs, la, lb, va, vb, lead);
    }
}

#ifndef MCG_ASYM_DEPTH
#define MCG_ASYM_DEPTH 4   // loads of a thread's scan in flight (A/B builds: 8)
#endif
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
        const double* col = a.S + row.off + p;
        double best = 0.0;
        int n_scan = 0;  // the reference scans until the first date beyond the maturity (:62-64)
        while (n_scan < n_cols && !(n_scan * a.dt > row.maturity)) ++n_scan;
        // a thread's scan is a chain of dependent-looking loads (one row of the block per date): four at a time are in flight
        for (int j0 = 0; j0 < n_scan; j0 += MCG_ASYM_DEPTH) {
            double S4[MCG_ASYM_DEPTH];
#pragma unroll
            for (int u = 0; u < MCG_ASYM_DEPTH; ++u) S4[u] = col[(int64_t)min(j0 + u, n_scan - 1) * BATCH_LD];
#pragma unroll
            for (int u = 0; u < MCG_ASYM_DEPTH; ++u) {
                const int j = j0 + u;
                const double S = S4[u];
                if (j >= n_scan || isnan(S) || isinf(S)) continue;
                const bool in = call ? (S > bnd[j]) : (S < bnd[j]);
                if (in) {
                    const double d = dsc[j] * payoff_of(call, S, row.strike);
                    if (d > best) best = d;
                }
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
    const double* col = a.S + row.off + (live ? p : 0);
    const uint64_t id = (row.id << 32) + (uint64_t)p;
    const PhiloxLane lane_rng = philox_lane_setup(id, 2u, row.k1);
    const int quads = (a.num_branches + 3) >> 2;
    const double inv_b = a.num_branches > 0 ? 1.0 / (double)a.num_branches : 0.0;
    const int ex_last = row.n_steps - 1;
    // run = F[j][p] = max_{k >= j, k < n_dates} disc_k payoff_k, floored at 0; start at j = n_cols - 1 (the last column,
    // which is no exercise date of the driver's list but counts as a later column)
    double run = 0.0;
    // This walk reads EVERY column of the row's block once: it also answers the driver's scan for inf / nan in the paths
    // (PredictionGen.cpp:752-777), which sends a row to ",0,0,0,0,0,0" before any pricer sees it.
    bool finite = true;
    {
        const int j = n_cols - 1;
        const double s_last = col[(int64_t)j * BATCH_LD];
        finite = isfinite(s_last);
        if (j < n_dates) run = fmax(run, dsc[j] * payoff_of(call, s_last, row.strike));
    }
    double lower = 0.0, upper = 0.0;
    for (int e = row.n_steps - 1; e >= 0; --e) {  // exercise date index == column index; run == F[e+1][p] here
        const double s_e = col[(int64_t)e * BATCH_LD];
        finite = finite && isfinite(s_e);
        const double now = dsc[e] * payoff_of(call, s_e, row.strike);
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
                    const Philox4 w = philox4x32_10_lane(lane_rng, (uint32_t)(e * quads + q), row.k0, row.k1);
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
    if (live && !finite) a.bad[blockIdx.x] = 1.0;  // (any number of threads may say so)
    double v[2] = {live ? lower : 0.0, live ? upper : 0.0};
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) a.out[4 * (int64_t)blockIdx.x + 1] = 0.5 * (v[0] + v[1]) / (double)a.n_paths;
}

// ---- LSM (LSMPricer.cpp:19-102) ------------------------------------------------------------------
// One wavefront per row (lsm_wave_body) and -- round 5 -- one wavefront per WORKGROUP: with four rows to a workgroup the
// workgroup lived as long as its longest row (the driver's rows run from 5 to 126 steps: the longest of four is ~100 on
// average, the mean 66), its other three wave slots idle meanwhile; the scheduler packs single waves as they come.
template <int NB>
__global__ __launch_bounds__(64) void k_batch_lsm(BatchArgs a) {
    __shared__ double ws[lsm_ws_doubles(NB) + LSM_COEF_DOUBLES];  // workspace of a refined date's solve
    const int64_t r_idx = (int64_t)blockIdx.x;
    const BatchRow row = a.rows[r_idx];
    if (!row.valid) return;
    double sum_v, sum_v2;
    lsm_wave_body<NB>(a.S + row.off, BATCH_LD, a.n_paths, row.n_steps + 1, row.strike, row.maturity, a.dt, a.disc, row.is_call, ws,
                      sum_v, sum_v2);
    if (threadIdx.x == 0) a.out[4 * r_idx + 2] = sum_v / (double)a.n_paths;
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
    const double* col = a.S + row.off + (live ? p : 0);
    double m[NM + 1];
    double xs[2] = {0.0, 0.0}, ys[2] = {0.0, 0.0};  // this path's two regression samples (kept for a re-fit)
#pragma unroll
    for (int q = 0; q <= NM; ++q) m[q] = 0.0;
    if (live) {
        double best = 0.0;
        int stop = 0;
        for (int j = 0; j < n_dates; ++j) {
            const double d = payoff_of(call, col[(int64_t)j * BATCH_LD], row.strike) * dsc[j];
            if (d > best) {
                best = d;
                stop = j;
            }
        }
        m[NM] = best;
        const int other = (stop + n_cols / 2) % n_cols;
        xs[0] = col[(int64_t)stop * BATCH_LD];
        xs[1] = col[(int64_t)other * BATCH_LD];
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
            const double S = col[(int64_t)j * BATCH_LD];
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

// LSM on `lsm_stream`, MartingaleOptimization on `mo_stream` (run_batch_chunk: the same stream, or the auxiliary one)
template <int NB>
static void launch_row_regressions(const BatchArgs& a, size_t smem_cols, hipStream_t lsm_stream, hipStream_t mo_stream) {
    hipLaunchKernelGGL(k_batch_lsm<NB>, dim3((unsigned)a.n_rows), dim3(64), 0, lsm_stream, a);
    hipLaunchKernelGGL(k_batch_martingale<NB>, dim3((unsigned)a.n_rows), dim3(256), smem_cols, mo_stream, a);
}

namespace {

// Device workspace one row needs in a chunk: its matrix block, its amplitudes + compensator, its image and its four prices.
size_t row_workspace_bytes(int n_steps, int M) {
    // (+ the generator's workgroup map: at most 32 shares of 4 bytes per row at 256 paths)
    return ((size_t)BATCH_LD * (size_t)(n_steps + 1) + (size_t)M + (size_t)n_steps + 5) * sizeof(double) + sizeof(BatchRow) + 32 * sizeof(uint32_t);
}

// LDS class of a row: the row kernels' dynamic LDS is sized by the longest row of a LAUNCH, so the rare long rows (Mz >= 256:
// more than half a year of trading days) are launched apart from the many short ones, whose occupancy they would cost.
int lds_class(int M) { return M <= 128 ? 0 : M == 256 ? 1 : M == 512 ? 2 : 3; }
constexpr int N_LDS_CLASSES = 4;

}  // namespace

// The six launches for ONE chunk of rows (h: their device images, already with offsets); prices into out[4 * h[k].id ...].
static int run_batch_chunk(mcg_ctx* ctx, std::vector<BatchRow>& h, BatchArgs a, double* d_S, double* d_small, int poly_order, double* out,
                           unsigned char* priced) {
    const int64_t n = (int64_t)h.size();
    int max_steps = 1, m_max = 1;
    int64_t off = 0, woff = 0;
    for (BatchRow& d : h) {
        d.off = off;
        d.woff = woff;
        off += BATCH_LD * (int64_t)(d.n_steps + 1);
        woff += (int64_t)d.M + d.n_steps;
        max_steps = std::max(max_steps, d.n_steps);
        m_max = std::max(m_max, d.M);
    }
    a.S = d_S;
    a.out = d_small;
    a.bad = d_small + 4 * n;
    a.w = d_small + 5 * n;
    a.rows = reinterpret_cast<const BatchRow*>(a.w + woff + (woff & 1));
    a.n_rows = n;
    a.max_steps = max_steps;
    a.m_max = m_max;
    hipError_t e = hipMemcpyAsync((void*)a.rows, h.data(), sizeof(BatchRow) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(a.out, 0, 5 * sizeof(double) * (size_t)n, ctx->stream);  // (prices and the rows' flags)
    if (e != hipSuccess) return fail(MCG_ERR_HIP, "batch upload failed: %s", hipGetErrorString(e));
    const int mphi_max = 2 * m_max >= 2 ? 2 * m_max : 2;
    const size_t smem_w = ((size_t)2 * mphi_max + (size_t)max_steps + 1 + (size_t)m_max) * sizeof(double);
    size_t smem_p = 0;  // largest over the transform sizes present (the staging part does not grow with Mz)
    for (int m = 1; m <= m_max; m <<= 1) smem_p = std::max(smem_p, rb_smem_bytes(m, std::min(m, max_steps)));
    smem_p += (size_t)study_switch("MCG_BATCH_PATHS_EXTRA_LDS_KB", 0) << 10;  // (A/B builds: fewer workgroups per CU -- how much does the generator live off its occupancy?)
    const size_t smem_c = ((size_t)max_steps + 1) * sizeof(double);
    // k_batch_paths: one workgroup per share of rb_pairs_per_block(Mz) pairs of a row -- exactly those that exist
    const int n_pairs = (a.n_paths + 1) / 2;
    std::vector<uint32_t> wg_map;
    wg_map.reserve((size_t)n * 2);
    for (int64_t k = 0; k < n; ++k) {
        const int shares = (n_pairs + rb_pairs_per_block(h[(size_t)k].M) - 1) / rb_pairs_per_block(h[(size_t)k].M);  // <= 63 (n_paths <= 256)
        for (int sub = 0; sub < shares; ++sub) wg_map.push_back((uint32_t)(k << 6) | (uint32_t)sub);
    }
    std::vector<double> four((size_t)n * 5);   // (host memory first: a bad_alloc below this line would leave a pool buffer out)
    void* d_map = nullptr;
    const size_t map_bytes = wg_map.size() * sizeof(uint32_t);
    int rc_map = pool_alloc(ctx, map_bytes, &d_map);
    if (rc_map) return rc_map;
    struct PoolGuard {   // the map goes back to the pool on every way out
        mcg_ctx* c;
        void* p;
        size_t b;
        ~PoolGuard() { pool_release(c, p, b); }
    } map_guard{ctx, d_map, map_bytes};
    if (hipMemcpyAsync(d_map, wg_map.data(), map_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(MCG_ERR_HIP, "batch upload failed (workgroup map)");
    a.wg_map = (const uint32_t*)d_map;
    {
        TimedLaunch t(ctx, MCG_K_BATCH);
        hipLaunchKernelGGL(k_batch_weights, dim3((unsigned)n), dim3(256), smem_w, ctx->stream, a);
        typedef void (*PathsKernel)(BatchArgs);
        static const PathsKernel paths_kernel[N_LDS_CLASSES] = {k_batch_paths<0>, k_batch_paths<1>, k_batch_paths<2>, k_batch_paths<3>};
        const PathsKernel pk = paths_kernel[lds_class(m_max)];  // (a chunk holds rows of one class, run_batch_rows)
        if (smem_p > 48 * 1024) (void)hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_p);
        hipLaunchKernelGGL(pk, dim3((unsigned)wg_map.size()), dim3(256), smem_p, ctx->stream, a);
        // The four pricers only read the row blocks and write columns of their own, so two of them run on a second stream beside the
        // other two, forked and joined by events: LSM + MartingaleOptimization (one wavefront per row walking its dates; chains of
        // dependent loads: 0.79 / 0.42 VALU busy at 20 000 rows) beside AsymptoticAnalysis + BranchingProcesses (0.53 / 0.85) -- one
        // pair's latencies hide behind the other's issue.  20 000 rows, same board, three alternations: device span 6.7 -> 6.2-6.3 ms,
        // 2.63 -> 2.80 M rows/s through the entry point (the other split -- Asymptotic + Martingale aside -- 6.4 ms);
        // gpurun_out/r6z2_batch_aux.log.  (A/B builds: MCG_BATCH_AUX_STREAM = 0 one stream, 1 the other split.)
        hipStream_t aux = ctx->stream;
        const int aux_mode = study_switch("MCG_BATCH_AUX_STREAM", 2);
        if (aux_mode != 0) {
            if (!ctx->batch_aux) {
                if (hipStreamCreateWithFlags(&ctx->batch_aux, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&ctx->batch_fork, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ctx->batch_join, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    if (ctx->batch_aux) (void)hipStreamDestroy(ctx->batch_aux);
                    ctx->batch_aux = nullptr;
                }
            }
            if (ctx->batch_aux && ctx->batch_fork && ctx->batch_join && hipEventRecord(ctx->batch_fork, ctx->stream) == hipSuccess &&
                hipStreamWaitEvent(ctx->batch_aux, ctx->batch_fork, 0) == hipSuccess)
                aux = ctx->batch_aux;
        }
        const hipStream_t s_asym = aux_mode == 2 ? ctx->stream : aux, s_lsm = aux_mode == 2 ? aux : ctx->stream;
        hipLaunchKernelGGL(k_batch_asym, dim3((unsigned)n), dim3(256), 2 * smem_c, s_asym, a);
        hipLaunchKernelGGL(k_batch_branching, dim3((unsigned)n), dim3(256), smem_c, ctx->stream, a);
        switch (poly_order + 1) {
            case 1: launch_row_regressions<1>(a, smem_c, s_lsm, aux); break;
            case 2: launch_row_regressions<2>(a, smem_c, s_lsm, aux); break;
            case 3: launch_row_regressions<3>(a, smem_c, s_lsm, aux); break;
            case 4: launch_row_regressions<4>(a, smem_c, s_lsm, aux); break;
            default: launch_row_regressions<5>(a, smem_c, s_lsm, aux); break;
        }
        if (aux != ctx->stream) {   // join: nothing behind this point on the main stream starts before the auxiliary stream's kernels are done
            if (hipEventRecord(ctx->batch_join, aux) != hipSuccess || hipStreamWaitEvent(ctx->stream, ctx->batch_join, 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(aux);
            }
        }
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(four.data(), a.out, 5 * sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (also: the host vectors `h` and `wg_map` have outlived their uploads)
    else (void)hipStreamSynchronize(ctx->stream);                // (nothing of this chunk may still read the map when it goes back to the pool)
    if (e != hipSuccess) return fail(MCG_ERR_HIP, "batch run failed: %s", hipGetErrorString(e));
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = (int64_t)h[(size_t)k].id;
        const bool bad = four[(size_t)(4 * n + k)] != 0.0;  // inf / nan among the row's paths: zeros, like the driver (:752-777)
        for (int c = 0; c < 4; ++c) out[4 * i + c] = bad ? 0.0 : four[(size_t)(4 * k + c)];
        if (bad && priced) priced[i] = 0;
    }
    return MCG_OK;
}

// inf / nan anywhere among the n_paths columns of a step-major matrix (the driver's scan, PredictionGen.cpp:752-777) for the rows
// that are priced singly: *flag != 0 afterwards (any number of threads may say so)
__global__ __launch_bounds__(256) void k_scan_finite(const double* data, int64_t ld, int64_t n_paths, int n_rows, unsigned* flag) {
    bool bad = false;
    for (int64_t j = blockIdx.y; j < n_rows; j += gridDim.y)
        for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_paths; p += (int64_t)gridDim.x * 256) bad = bad || !isfinite(data[j * ld + p]);
    if (bad) *flag = 1u;
}

// true: the matrix holds a non-finite price (or the scan itself failed: such a row is zeroed, like one whose generation failed)
static bool paths_hold_non_finite(mcg_ctx* ctx, const mcg_paths* P) {
    unsigned* flag = reinterpret_cast<unsigned*>(ctx->scalars + SC_BARRIER) + 1;   // (the second word of the hand-shake's slot: free between sweeps)
    unsigned* h_flag = reinterpret_cast<unsigned*>(ctx->h_scalars + SC_BARRIER) + 1;
    if (hipMemsetAsync(flag, 0, sizeof(unsigned), ctx->stream) != hipSuccess) return true;
    const unsigned gx = (unsigned)std::min<int64_t>((P->n_paths + 255) / 256, 1024), gy = (unsigned)std::min(P->n_steps + 1, 1024);
    hipLaunchKernelGGL(k_scan_finite, dim3(gx, gy), dim3(256), 0, ctx->stream, (const double*)P->data, P->ld, P->n_paths, P->n_steps + 1, flag);
    if (hipMemcpyAsync(h_flag, flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return true;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return true;
    return *h_flag != 0u;
}

#ifdef MCG_BATCH_TRACE   // (A/B builds: where the host time of a call goes, printed to stderr)
#define BT_NOW() std::chrono::steady_clock::now()
#define BT_US(a, b) (std::chrono::duration<double, std::micro>((b) - (a)).count())
#endif

int run_batch_rows(mcg_ctx* ctx, const mcg_row* rows, int64_t n_rows, int n_paths, double r, double dt, int num_branches,
                   int poly_order, int max_iterations, uint64_t seed, double* out, unsigned char* priced) {
    g_stats.batch_calls.fetch_add(1, std::memory_order_relaxed);
#ifdef MCG_BATCH_TRACE
    const auto bt0 = BT_NOW();
#endif
    std::vector<int64_t> cls[N_LDS_CLASSES], long_rows;
    for (int64_t i = 0; i < n_rows; ++i) {
        const mcg_row& s = rows[i];
        out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = out[4 * i + 3] = 0.0;
        // a row the reference's driver would answer with zeros (no steps, non-finite paths, a throwing pricer)
        const bool valid = s.n_steps >= 1 && s.S0 > 0.0 && std::isfinite(s.S0) && s.xi >= 0.0 && std::isfinite(s.xi) && s.H >= 0.0 &&
                           std::isfinite(s.H) && std::isfinite(s.eta) && std::fabs(s.rho) <= 1.0 && s.strike > 0.0 &&
                           std::isfinite(s.strike) && s.sigma > 0.0 && std::isfinite(s.maturity);
        if (priced) priced[i] = valid ? 1 : 0;
        if (!valid) continue;
        // A row longer than the row kernels' LDS tables reach (more than four years of trading days), or any row of a call
        // with more than 256 paths per row or an order above 4, is priced after the batch through the single-contract
        // entry points, on the same Philox path ids: never refused, never answered with zeros.
        if (s.n_steps > BATCH_MAX_STEPS || n_paths > 256 || poly_order > 4) {
            long_rows.push_back(i);
            continue;
        }
        int M = 1;
        while (M < s.n_steps) M <<= 1;
        cls[lds_class(M)].push_back(i);
    }
#ifdef MCG_BATCH_TRACE
    const auto bt1 = BT_NOW();
#endif
    // Memory budget of one chunk: a quarter of what is free now (mcg_debug_batch_budget overrides), never less than one row.
    size_t budget = ctx->batch_budget;
    if (budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            free_b = (size_t)8 << 30;
        }
        budget = std::max<size_t>(free_b / 4, (size_t)256 << 20);
    }
    // Plan: per LDS class, consecutive rows (the caller's order) while their workspace fits the budget.
    struct Chunk {
        int c;
        size_t begin, end, bytes_S, doubles_small;
    };
    std::vector<Chunk> plan;
    size_t max_S = 0, max_small = 0;
    for (int c = 0; c < N_LDS_CLASSES; ++c) {
        size_t k = 0;
        while (k < cls[c].size()) {
            Chunk ch{c, k, k, 0, 0};
            size_t used = 0, w = 0;
            while (ch.end < cls[c].size()) {
                const mcg_row& s = rows[cls[c][ch.end]];
                int M = 1;
                while (M < s.n_steps) M <<= 1;
                const size_t need = row_workspace_bytes(s.n_steps, M);
                if (ch.end > ch.begin && (used + need > budget || ch.end - ch.begin >= ((size_t)1 << 24))) break;  // (k_batch_paths' workgroup map holds the chunk-local row in 26 bits)
                used += need;
                ch.bytes_S += (size_t)BATCH_LD * (size_t)(s.n_steps + 1) * sizeof(double);
                w += (size_t)M + (size_t)s.n_steps;
                ++ch.end;
            }
            const size_t nr = ch.end - ch.begin;
            ch.doubles_small = 5 * nr + w + 2 + (sizeof(BatchRow) * nr + 7) / 8;
            max_S = std::max(max_S, ch.bytes_S);
            max_small = std::max(max_small, ch.doubles_small);
            plan.push_back(ch);
            k = ch.end;
        }
    }
#ifdef MCG_BATCH_TRACE
    const auto bt2 = BT_NOW();
    double bt_build = 0.0, bt_chunk = 0.0;
#endif
    int rc = MCG_OK;
    if (!plan.empty()) {
        void *S = nullptr, *small = nullptr;  // ONE pair of buffers for all chunks
        rc = pool_alloc(ctx, max_S, &S);
        if (rc) return rc;
        rc = pool_alloc(ctx, max_small * sizeof(double), &small);
        if (rc) {
            pool_release(ctx, S, max_S);
            return rc;
        }
        struct Release {   // both buffers go back to the pool on every way out (a bad_alloc in a chunk included)
            mcg_ctx* c;
            void *S, *small;
            size_t bS, bsmall;
            ~Release() {
                pool_release(c, S, bS);
                pool_release(c, small, bsmall);
            }
        } release{ctx, S, small, max_S, max_small * sizeof(double)};
        int64_t peak = (int64_t)(max_S + max_small * sizeof(double)), seen = g_stats.batch_peak_workspace_bytes.load(std::memory_order_relaxed);
        while (peak > seen && !g_stats.batch_peak_workspace_bytes.compare_exchange_weak(seen, peak, std::memory_order_relaxed)) {
        }
        BatchArgs a{};
        a.n_paths = n_paths;
        a.r = r;
        a.dt = dt;
        a.sqdt = std::sqrt(dt);
        a.disc = std::exp(-r * dt);
        a.k0 = (uint32_t)seed;
        a.k1 = (uint32_t)(seed >> 32);
        a.log_tab = (const double2*)ctx->log_tab;
        a.num_branches = num_branches;
        a.max_iterations = max_iterations;
        std::vector<BatchRow> h;
        for (const Chunk& ch : plan) {
#ifdef MCG_BATCH_TRACE
            const auto bc0 = BT_NOW();
#endif
            h.clear();
            h.reserve(ch.end - ch.begin);
            for (size_t k = ch.begin; k < ch.end; ++k) {
                const int64_t i = cls[ch.c][k];
                const mcg_row& s = rows[i];
                BatchRow d{};
                d.S0 = s.S0;
                d.logS0 = std::log(s.S0);
                d.xi = s.xi;
                d.H = s.H;
                d.eta = s.eta;
                d.strike = s.strike;
                d.maturity = s.maturity;
                d.sigma = s.sigma;
                d.dividend = s.dividend;
                d.id = (uint64_t)i;
                d.k0 = a.k0;
                d.k1 = a.k1;
                d.n_steps = s.n_steps;
                d.is_call = s.is_call;
                d.valid = 1;
                d.M = 1;
                while (d.M < d.n_steps) d.M <<= 1;
                h.push_back(d);
            }
#ifdef MCG_BATCH_TRACE
            const auto bc1 = BT_NOW();
#endif
            rc = run_batch_chunk(ctx, h, a, (double*)S, (double*)small, poly_order, out, priced);
#ifdef MCG_BATCH_TRACE
            bt_build += BT_US(bc0, bc1);
            bt_chunk += BT_US(bc1, BT_NOW());
#endif
            if (rc) break;
            g_stats.batch_chunks.fetch_add(1, std::memory_order_relaxed);
            g_stats.batch_rows.fetch_add((int64_t)h.size(), std::memory_order_relaxed);
        }
        if (rc) return rc;
    }
#ifdef MCG_BATCH_TRACE
    std::fprintf(stderr, "batch trace: classify %.0f us, budget+plan %.0f us, build images %.0f us, chunks (upload, launches, download, scatter) %.0f us, total %.0f us\n",
                 BT_US(bt0, bt1), BT_US(bt1, bt2), bt_build, bt_chunk, BT_US(bt0, BT_NOW()));
#endif
    g_stats.batch_rows_singly.fetch_add((int64_t)long_rows.size(), std::memory_order_relaxed);
    for (int64_t i : long_rows) {  // PredictionGen.cpp:718-816 for one row, through the single-contract entry points
        const mcg_row& s = rows[i];
        double* o = out + 4 * i;
        mcg_paths* P = nullptr;
        // A row whose generation fails, or whose paths hold an inf / nan, is the driver's ",0,0,0,0,0,0" (PredictionGen.cpp:739-777):
        // zeros AND priced[i] = 0, so that mcg_batch_price_rows6 zeroes its feature columns too -- exactly like a row the row
        // kernels flag (this route used to keep priced[i] = 1 and hand NaN prices through).
        if (mcg_paths_rbergomi(ctx, seed, s.S0, r, s.xi, s.H, s.eta, s.rho, dt, s.n_steps, (uint64_t)i << 32, n_paths, &P) != MCG_OK) {
            if (priced) priced[i] = 0;
            continue;
        }
        if (paths_hold_non_finite(ctx, P)) {
            if (priced) priced[i] = 0;
            mcg_paths_free(P);
            continue;
        }
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


// =================================================================================================================
// Coalesced class-API calls (coalesce.hpp): one ROUND = the calls of many host threads, grouped by kind, through the
// row kernels above.  The requests' matrices live in per-thread slots of one arena (a slot = one row block of this
// file's layout: step-major, 256 columns); generated paths and prices return through device-visible pinned host memory,
// written by the kernels themselves -- a round costs ONE upload (its descriptors), its launches and ONE synchronisation.
// =================================================================================================================
namespace co {

struct Xfer {  // one matrix between a slot and its owner's pinned buffer
    int64_t slot_off;
    double* host;  // [n_paths][n_cols] path-major, device-visible
    int n_paths, n_cols;
};
constexpr int CO_TILE = 16;  // columns (dates) of one transposed tile: a path's piece of it is 128 contiguous bytes on the host side

// slot (step-major, 256 columns) -> the owner's host buffer (path-major: the reference's vector<vector<double>> order)
__global__ __launch_bounds__(256) void k_co_gather(const Xfer* x, const double* S, int tiles_max) {
    __shared__ double tile[CO_TILE][BATCH_LD + 1];
    const Xfer d = x[blockIdx.x / tiles_max];
    const int j0 = (int)(blockIdx.x % tiles_max) * CO_TILE;
    if (j0 >= d.n_cols) return;
    const int nj = min(CO_TILE, d.n_cols - j0);
    const double* blk = S + d.slot_off;
    const int p = threadIdx.x;
    for (int jj = 0; jj < nj; ++jj) tile[jj][p] = blk[(int64_t)(j0 + jj) * BATCH_LD + p];
    __syncthreads();
    const int jj = p & (CO_TILE - 1);
    if (jj < nj)
        for (int q = p / CO_TILE; q < d.n_paths; q += 256 / CO_TILE) d.host[(int64_t)q * d.n_cols + j0 + jj] = tile[jj][q];
}

// the owner's host buffer -> its slot (a pricer called with a matrix the slot does not hold)
__global__ __launch_bounds__(256) void k_co_scatter(const Xfer* x, double* S, int tiles_max) {
    __shared__ double tile[CO_TILE][BATCH_LD + 1];
    const Xfer d = x[blockIdx.x / tiles_max];
    const int j0 = (int)(blockIdx.x % tiles_max) * CO_TILE;
    if (j0 >= d.n_cols) return;
    const int nj = min(CO_TILE, d.n_cols - j0);
    double* blk = S + d.slot_off;
    const int p = threadIdx.x;
    const int jj = p & (CO_TILE - 1);
    if (jj < nj)
        for (int q = p / CO_TILE; q < d.n_paths; q += 256 / CO_TILE) tile[jj][q] = d.host[(int64_t)q * d.n_cols + j0 + jj];
    __syncthreads();
    for (int k = 0; k < nj; ++k) blk[(int64_t)(j0 + k) * BATCH_LD + p] = p < d.n_paths ? tile[k][p] : 0.0;
}

namespace {

struct Key {  // what the requests of one launch must share
    int kind, n_paths, cls, a, b;
    double r, dt;
    bool operator==(const Key& o) const {
        return kind == o.kind && n_paths == o.n_paths && cls == o.cls && a == o.a && b == o.b && r == o.r && dt == o.dt;
    }
    bool operator<(const Key& o) const {
        if (kind != o.kind) return kind < o.kind;
        if (n_paths != o.n_paths) return n_paths < o.n_paths;
        if (cls != o.cls) return cls < o.cls;
        if (a != o.a) return a < o.a;
        if (b != o.b) return b < o.b;
        if (r != o.r) return r < o.r;
        return dt < o.dt;
    }
};

Key key_of(const Request& q) {
    Key k{q.kind, q.n_paths, 0, 0, 0, q.r, q.dt};
    switch (q.kind) {
        case GEN: k.cls = lds_class(q.M); k.r = 0.04; k.dt = 1.0 / 252.0; break;  // RoughVolatility.cpp:321-326
        case BRANCH: k.a = q.num_branches; break;
        case LSM: k.a = q.poly_order; break;
        case MART: k.a = q.poly_order; k.b = q.max_iterations >= 2 ? 2 : 1; break;  // (k_batch_martingale: iteration 1 has M = 0, every later one the fitted M)
        default: break;
    }
    return k;
}

int grow(RoundBuffers& rb, size_t bytes, size_t n_req) {
    if (bytes > rb.cap) {
        const size_t cap = std::max<size_t>(bytes * 2, (size_t)1 << 20);
        if (rb.h) (void)hipHostFree(rb.h);
        if (rb.d) (void)hipFree(rb.d);
        rb.h = rb.d = nullptr;
        rb.cap = 0;
        MCG_HIP(hipHostMalloc((void**)&rb.h, cap, hipHostMallocDefault));
        MCG_HIP(hipMalloc((void**)&rb.d, cap));
        rb.cap = cap;
    }
    if (n_req > rb.out_cap) {
        const size_t cap = std::max<size_t>(n_req * 2, 1024);
        if (rb.h_out) (void)hipHostFree(rb.h_out);
        if (rb.d_scratch) (void)hipFree(rb.d_scratch);
        rb.h_out = rb.d_scratch = nullptr;
        rb.out_cap = 0;
        MCG_HIP(hipHostMalloc((void**)&rb.h_out, cap * 4 * sizeof(double), hipHostMallocDefault));
        MCG_HIP(hipHostGetDevicePointer((void**)&rb.d_out, rb.h_out, 0));
        MCG_HIP(hipMalloc((void**)&rb.d_scratch, cap * sizeof(double)));
        rb.out_cap = cap;
    }
    return MCG_OK;
}

template <int NB>
void launch_lsm_rows(mcg_ctx* ctx, const BatchArgs& a) {
    hipLaunchKernelGGL(k_batch_lsm<NB>, dim3((unsigned)a.n_rows), dim3(64), 0, ctx->stream, a);
}
template <int NB>
void launch_martingale_rows(mcg_ctx* ctx, const BatchArgs& a, size_t smem_c) {
    hipLaunchKernelGGL(k_batch_martingale<NB>, dim3((unsigned)a.n_rows), dim3(256), smem_c, ctx->stream, a);
}

int run_round(mcg_ctx* ctx, RoundBuffers& rb, double* arena_base, Request** reqs, int n) {
    const auto t_begin = std::chrono::steady_clock::now();
    MCG_HIP(hipSetDevice(ctx->device));
    // ---- order the requests by launch group -------------------------------------------------------------------
    std::vector<int> order((size_t)n);
    std::vector<Key> keys((size_t)n);
    for (int i = 0; i < n; ++i) {
        order[(size_t)i] = i;
        keys[(size_t)i] = key_of(*reqs[i]);
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return keys[(size_t)x] < keys[(size_t)y]; });
    // ---- layout of the round's descriptors: [BatchRow x n][w][wg_map][Xfer x (generated + uploaded)] ------------
    size_t w_doubles = 0, n_map = 0, n_gen = 0, n_up = 0;
    for (int i = 0; i < n; ++i) {
        const Request& q = *reqs[i];
        if (q.kind == GEN) {
            w_doubles += (size_t)q.M + (size_t)q.n_steps;
            const int ppb = rb_pairs_per_block(q.M);
            n_map += (size_t)(((q.n_paths + 1) / 2 + ppb - 1) / ppb);
            ++n_gen;
        } else if (q.upload) {
            ++n_up;
        }
    }
    const size_t off_rows = 0;
    const size_t off_w = off_rows + sizeof(BatchRow) * (size_t)n;
    const size_t off_map = off_w + sizeof(double) * (w_doubles + (w_doubles & 1));
    const size_t off_x = (off_map + sizeof(uint32_t) * n_map + 15) & ~(size_t)15;
    const size_t total = off_x + sizeof(Xfer) * (n_gen + n_up);
    int rc = grow(rb, total, (size_t)n);
    if (rc) return rc;
    BatchRow* h_rows = reinterpret_cast<BatchRow*>(rb.h + off_rows);
    double* h_w = reinterpret_cast<double*>(rb.h + off_w);
    uint32_t* h_map = reinterpret_cast<uint32_t*>(rb.h + off_map);
    Xfer* h_x = reinterpret_cast<Xfer*>(rb.h + off_x);
    const BatchRow* d_rows = reinterpret_cast<const BatchRow*>(rb.d + off_rows);
    double* d_w = reinterpret_cast<double*>(rb.d + off_w);
    const uint32_t* d_map = reinterpret_cast<const uint32_t*>(rb.d + off_map);
    const Xfer* d_x = reinterpret_cast<const Xfer*>(rb.d + off_x);
    double* out_dev = rb.d_out;

    struct Group {
        Key key;
        int begin, end;        // positions in `order`
        size_t map_begin, map_n;
        int max_steps, m_max;
    };
    std::vector<Group> groups;
    size_t woff = 0, mpos = 0, xg = 0, xu = n_gen;  // generated matrices' transfers first, then the uploads
    int up_tiles = 1, gen_tiles = 1;
    for (int pos = 0; pos < n; ++pos) {
        const int i = order[(size_t)pos];
        const Request& q = *reqs[i];
        const Key& k = keys[(size_t)i];
        if (groups.empty() || !(groups.back().key == k)) groups.push_back(Group{k, pos, pos, mpos, 0, 1, 1});
        Group& g = groups.back();
        g.end = pos + 1;
        BatchRow d{};
        d.S0 = q.S0;
        d.logS0 = q.kind == GEN ? std::log(q.S0) : 0.0;
        d.xi = q.xi;
        d.H = q.H;
        d.eta = q.eta;
        d.strike = q.strike;
        d.maturity = q.maturity;
        d.sigma = q.sigma;
        d.dividend = q.dividend;
        d.off = q.slot_off;
        d.id = 0;  // Philox path ids 0 .. n_paths - 1: what mcg_paths_rbergomi / mcg_price_branching use for a matrix that begins at path 0
        d.k0 = (uint32_t)q.seed;
        d.k1 = (uint32_t)(q.seed >> 32);
        d.n_steps = q.n_steps;
        d.is_call = q.is_call;
        d.valid = 1;
        d.M = 1;
        while (d.M < d.n_steps) d.M <<= 1;
        g.max_steps = std::max(g.max_steps, d.n_steps);
        g.m_max = std::max(g.m_max, d.M);
        const int n_cols = q.n_steps + 1;
        if (q.kind == GEN) {
            d.woff = (int64_t)woff;
            std::memcpy(h_w + woff, q.amp, sizeof(double) * (size_t)q.M);
            std::memcpy(h_w + woff + q.M, q.comp, sizeof(double) * (size_t)q.n_steps);
            woff += (size_t)q.M + (size_t)q.n_steps;
            const int ppb = rb_pairs_per_block(q.M);
            const int shares = ((q.n_paths + 1) / 2 + ppb - 1) / ppb;
            for (int sub = 0; sub < shares; ++sub) h_map[mpos++] = ((uint32_t)(pos - g.begin) << 6) | (uint32_t)sub;
            g.map_n += (size_t)shares;
            h_x[xg++] = Xfer{q.slot_off, q.host_dev, q.n_paths, n_cols};
            gen_tiles = std::max(gen_tiles, (n_cols + CO_TILE - 1) / CO_TILE);
        } else if (q.upload) {
            h_x[xu++] = Xfer{q.slot_off, q.host_dev, q.n_paths, n_cols};
            up_tiles = std::max(up_tiles, (n_cols + CO_TILE - 1) / CO_TILE);
        }
        h_rows[pos] = d;
    }
    MCG_HIP(hipMemcpyAsync(rb.d, rb.h, total, hipMemcpyHostToDevice, ctx->stream));
    // ---- launches ---------------------------------------------------------------------------------------------
    if (n_up) hipLaunchKernelGGL(k_co_scatter, dim3((unsigned)(n_up * (size_t)up_tiles)), dim3(256), 0, ctx->stream, d_x + n_gen, arena_base, up_tiles);
    for (const Group& g : groups) {
        BatchArgs a{};
        a.rows = d_rows + g.begin;
        a.n_rows = g.end - g.begin;
        a.n_paths = g.key.n_paths;
        a.max_steps = g.max_steps;
        a.m_max = g.m_max;
        a.r = g.key.r;
        a.dt = g.key.dt;
        a.sqdt = std::sqrt(a.dt);
        a.disc = std::exp(-a.r * a.dt);
        a.w = d_w;
        a.S = arena_base;
        a.log_tab = (const double2*)ctx->log_tab;
        a.out = out_dev + 4 * (size_t)g.begin;
        a.bad = rb.d_scratch + g.begin;
        a.wg_map = d_map + g.map_begin;
        a.num_branches = g.key.a;
        a.max_iterations = g.key.b == 1 ? 1 : 5;
        const size_t smem_c = ((size_t)g.max_steps + 1) * sizeof(double);
        switch (g.key.kind) {
            case GEN: {
                size_t smem_p = 0;
                for (int m = 1; m <= g.m_max; m <<= 1) smem_p = std::max(smem_p, rb_smem_bytes(m, std::min(m, g.max_steps)));
                typedef void (*PathsKernel)(BatchArgs);
                static const PathsKernel paths_kernel[N_LDS_CLASSES] = {k_batch_paths<0>, k_batch_paths<1>, k_batch_paths<2>, k_batch_paths<3>};
                const PathsKernel pk = paths_kernel[g.key.cls];
                if (smem_p > 48 * 1024) (void)hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_p);
                hipLaunchKernelGGL(pk, dim3((unsigned)g.map_n), dim3(256), smem_p, ctx->stream, a);
                break;
            }
            case ASYM: hipLaunchKernelGGL(k_batch_asym, dim3((unsigned)a.n_rows), dim3(256), 2 * smem_c, ctx->stream, a); break;
            case BRANCH: hipLaunchKernelGGL(k_batch_branching, dim3((unsigned)a.n_rows), dim3(256), smem_c, ctx->stream, a); break;
            case LSM:
                switch (g.key.a + 1) {
                    case 1: launch_lsm_rows<1>(ctx, a); break;
                    case 2: launch_lsm_rows<2>(ctx, a); break;
                    case 3: launch_lsm_rows<3>(ctx, a); break;
                    case 4: launch_lsm_rows<4>(ctx, a); break;
                    default: launch_lsm_rows<5>(ctx, a); break;
                }
                break;
            default:
                switch (g.key.a + 1) {
                    case 1: launch_martingale_rows<1>(ctx, a, smem_c); break;
                    case 2: launch_martingale_rows<2>(ctx, a, smem_c); break;
                    case 3: launch_martingale_rows<3>(ctx, a, smem_c); break;
                    case 4: launch_martingale_rows<4>(ctx, a, smem_c); break;
                    default: launch_martingale_rows<5>(ctx, a, smem_c); break;
                }
                break;
        }
    }
    if (n_gen) hipLaunchKernelGGL(k_co_gather, dim3((unsigned)(n_gen * (size_t)gen_tiles)), dim3(256), 0, ctx->stream, d_x, (const double*)arena_base, gen_tiles);
    MCG_HIP(hipGetLastError());
    const auto t_sync = std::chrono::steady_clock::now();
    // (a blocking event instead of this spinning wait, 100 / 1000 spins of the callers before they sleep and 200 / 4000 of an idle
    // lane were measured on the 16-CPU GPU box at 128 and 16 threads: no difference beyond the run-to-run noise of +-10 %,
    // gpurun_out/r6h_co_sweep.log, r6s_co_sweep.log)
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    const auto t_end = std::chrono::steady_clock::now();
    g_stats.coalesced_round_us.fetch_add((int64_t)std::chrono::duration<double, std::micro>(t_end - t_begin).count(), std::memory_order_relaxed);
    g_stats.coalesced_device_wait_us.fetch_add((int64_t)std::chrono::duration<double, std::micro>(t_end - t_sync).count(), std::memory_order_relaxed);
    static const int column[N_KINDS] = {0, 0, 1, 2, 3};
    for (int pos = 0; pos < n; ++pos) {
        Request& q = *reqs[order[(size_t)pos]];
        q.status = MCG_OK;
        if (q.kind != GEN) q.price = rb.h_out[4 * (size_t)pos + column[q.kind]];
    }
    g_stats.coalesced_rounds.fetch_add(1, std::memory_order_relaxed);
    g_stats.coalesced_calls.fetch_add(n, std::memory_order_relaxed);
    int64_t seen = g_stats.coalesced_peak_calls_per_round.load(std::memory_order_relaxed);
    while (n > seen && !g_stats.coalesced_peak_calls_per_round.compare_exchange_weak(seen, n, std::memory_order_relaxed)) {
    }
    return MCG_OK;
}

}  // namespace

int execute_round(mcg_ctx* ctx, RoundBuffers& rb, double* arena_base, Request** reqs, int n) {
    int rc;
    try {
        rc = run_round(ctx, rb, arena_base, reqs, n);
    } catch (const std::exception& e) {
        rc = fail(MCG_ERR_OOM, "coalesced round: %s", e.what());
    }
    if (rc != MCG_OK) {
        (void)hipStreamSynchronize(ctx->stream);  // nothing of this round may still be writing into the requesters' buffers
        const char* msg = mcg_last_error();
        for (int i = 0; i < n; ++i) {
            reqs[i]->status = rc;
            std::snprintf(reqs[i]->err, sizeof reqs[i]->err, "%s", msg ? msg : "coalesced round failed");
        }
    }
    return rc;
}

}  // namespace co

}  // namespace mcg
