// Device-side pieces of the Longstaff-Schwartz sweep shared by kernels_lsm.hip and the batched
// driver-row kernels (kernels_batch.hip): the small normal-equation solve and the one-workgroup sweep.
#pragma once
#include "devmath.hpp"

namespace mcg {

// Solve the normal equations of the scaled basis from the (all-reduced) moments; one thread.
// G[a][b] = m[a+b], rhs[a] = m[2p+1+a], equilibrated to unit diagonal.
// Fast path: LDL^T without pivoting when every pivot stays above 1e-10 (the usual, well-conditioned
// date: ~1 us).  Otherwise (one ITM path, all paths equal at j = 0, ...): cyclic Jacobi
// eigen-decomposition and a pseudo-inverse with relative eigenvalue cut 1e-12, which yields the
// projection the reference's min-norm SVD solve gives at the data points.
// NB is a template parameter so that every loop unrolls and G, Q live in registers: with a run-time size the
// arrays go to scratch memory and the (serial, one-thread) solve takes ~20 us instead of ~2.
template <int NB>
__device__ __forceinline__ void lsm_solve_nb(const double* moments, double min_count, double* coef) {
    constexpr int nb = NB;
    double G[NB][NB], Q[NB][NB], rhs[NB], d[NB], sol[NB];
    const double count = moments[0];
    coef[9] = count;
    for (int a = 0; a < 9; ++a) coef[a] = 0.0;
    if (!(count >= min_count) || !(count > 0.0)) return;  // too few samples: coefficients stay 0
    for (int a = 0; a < nb; ++a) {
        const double g = moments[2 * a];
        d[a] = g > 0.0 ? 1.0 / sqrt(g) : 0.0;
    }
    for (int a = 0; a < nb; ++a) {
        rhs[a] = moments[2 * nb - 1 + a] * d[a];
        for (int b = 0; b < nb; ++b) G[a][b] = moments[a + b] * d[a] * d[b];
    }
    // ---- fast path: LDL^T in Q (L below the diagonal, D on it) ----
    bool ok = true;
    for (int j = 0; j < nb && ok; ++j) {
        double dj = G[j][j];
        for (int k = 0; k < j; ++k) dj -= Q[j][k] * Q[j][k] * Q[k][k];
        if (!(dj > 1e-10)) {
            ok = false;
            break;
        }
        Q[j][j] = dj;
        const double inv = 1.0 / dj;
        for (int i = j + 1; i < nb; ++i) {
            double v = G[i][j];
            for (int k = 0; k < j; ++k) v -= Q[i][k] * Q[j][k] * Q[k][k];
            Q[i][j] = v * inv;
        }
    }
    if (ok) {
        for (int i = 0; i < nb; ++i) {  // L y = rhs
            double v = rhs[i];
            for (int k = 0; k < i; ++k) v -= Q[i][k] * sol[k];
            sol[i] = v;
        }
        for (int i = 0; i < nb; ++i) sol[i] /= Q[i][i];
        for (int i = nb - 1; i >= 0; --i) {  // L^T x = y
            double v = sol[i];
            for (int k = i + 1; k < nb; ++k) v -= Q[k][i] * sol[k];
            sol[i] = v;
        }
        for (int a = 0; a < nb; ++a) coef[a] = sol[a] * d[a];
        return;
    }
    // ---- rank-deficient / ill-conditioned date: Jacobi pseudo-inverse ----
    for (int a = 0; a < nb; ++a)
        for (int b = 0; b < nb; ++b) Q[a][b] = a == b ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < nb; ++p)
            for (int q = p + 1; q < nb; ++q) off += G[p][q] * G[p][q];
        if (off < 1e-60) break;
        for (int p = 0; p < nb - 1; ++p) {
            for (int q = p + 1; q < nb; ++q) {
                const double apq = G[p][q];
                if (apq == 0.0) continue;
                const double theta = (G[q][q] - G[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < nb; ++k) {
                    const double gkp = G[k][p], gkq = G[k][q];
                    G[k][p] = cs * gkp - sn * gkq;
                    G[k][q] = sn * gkp + cs * gkq;
                }
                for (int k = 0; k < nb; ++k) {
                    const double gpk = G[p][k], gqk = G[q][k];
                    G[p][k] = cs * gpk - sn * gqk;
                    G[q][k] = sn * gpk + cs * gqk;
                }
                for (int k = 0; k < nb; ++k) {
                    const double qkp = Q[k][p], qkq = Q[k][q];
                    Q[k][p] = cs * qkp - sn * qkq;
                    Q[k][q] = sn * qkp + cs * qkq;
                }
            }
        }
    }
    double lmax = 0.0;
    for (int a = 0; a < nb; ++a) lmax = fmax(lmax, G[a][a]);
    const double cut = lmax * 1e-12;
    for (int a = 0; a < nb; ++a) sol[a] = 0.0;
    for (int e = 0; e < nb; ++e) {
        const double lam = G[e][e];
        if (!(lam > cut)) continue;
        double proj = 0.0;
        for (int a = 0; a < nb; ++a) proj += Q[a][e] * rhs[a];
        const double w = proj / lam;
        for (int a = 0; a < nb; ++a) sol[a] += w * Q[a][e];
    }
    for (int a = 0; a < nb; ++a) coef[a] = sol[a] * d[a];
}

// run-time order -> the unrolled instance
__device__ inline void lsm_solve_one(const double* moments, int nb, double min_count, double* coef) {
    switch (nb) {
        case 1: lsm_solve_nb<1>(moments, min_count, coef); break;
        case 2: lsm_solve_nb<2>(moments, min_count, coef); break;
        case 3: lsm_solve_nb<3>(moments, min_count, coef); break;
        case 4: lsm_solve_nb<4>(moments, min_count, coef); break;
        case 5: lsm_solve_nb<5>(moments, min_count, coef); break;
        case 6: lsm_solve_nb<6>(moments, min_count, coef); break;
        case 7: lsm_solve_nb<7>(moments, min_count, coef); break;
        case 8: lsm_solve_nb<8>(moments, min_count, coef); break;
        default: lsm_solve_nb<9>(moments, min_count, coef); break;
    }
}

// The whole backward sweep in ONE launch for a small path count (n <= 1024: the reference's production
// calls price 250 paths per option row, src/core/PredictionGen.cpp:719): one 256-thread block, up to four
// paths per thread with V in registers, per date a block reduction of the regression moments, the solve on
// thread 0, the coefficients handed over through LDS (PPT = paths per thread: 1 up to 256 paths, else 4).  Same arithmetic as k_lsm_sweep / lsm_solve_one;
// only the summation order of the moments differs.  out3 = {sum V, sum V^2, n}.
template <int NB, int PPT>
__device__ __forceinline__ void lsm_small_body(const double* data, int64_t ld, int n, int n_cols, double K,
                                               double maturity, double dt, double disc, int is_call, double* out3) {
    constexpr int NM = 3 * NB - 1;
    __shared__ double red[NM * 4];
    __shared__ double sm_mom[32];
    __shared__ double sm_coef[16];
    const bool call = is_call != 0;
    const double invK = 1.0 / K;
    double V[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int p = threadIdx.x + 256 * q;
        V[q] = p < n ? payoff_of(call, data[(int64_t)(n_cols - 1) * ld + p], K) : 0.0;
    }
    for (int j = n_cols - 2; j >= 0; --j) {
        const double this_time = j * dt;
        if (this_time > maturity) {  // LSMPricer.cpp:43-49 (wave-uniform)
#pragma unroll
            for (int q = 0; q < PPT; ++q) V[q] = V[q] * disc;
            continue;
        }
        const double* row = data + (int64_t)j * ld;
        double s_j[PPT], pay_j[PPT];
        double m[NM];
#pragma unroll
        for (int q = 0; q < NM; ++q) m[q] = 0.0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int p = threadIdx.x + 256 * q;
            s_j[q] = p < n ? row[p] : 0.0;
            pay_j[q] = payoff_of(call, s_j[q], K);
            if (p < n && pay_j[q] > 1e-14) {
                const double x = fma(s_j[q], invK, -1.0);
                const double y = V[q] * disc;
                double pw = 1.0;
#pragma unroll
                for (int t = 0; t < 2 * NB - 1; ++t) {
                    m[t] += pw;
                    if (t < NB) m[2 * NB - 1 + t] = fma(pw, y, m[2 * NB - 1 + t]);
                    pw *= x;
                }
            }
        }
        block_sum<NM, 4>(m, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int t = 0; t < NM; ++t) sm_mom[t] = m[t];
            lsm_solve_nb<NB>(sm_mom, 1.0, sm_coef);  // writes sm_coef[0..9], [9] = ITM count
        }
        __syncthreads();
        double c[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) c[t] = sm_coef[t];
        const double n_itm = sm_coef[9];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double vn = V[q] * disc;
            double v;
            if (pay_j[q] > 1e-14 && n_itm > 0.0) {
                const double x = fma(s_j[q], invK, -1.0);
                double cont = c[NB - 1];
#pragma unroll
                for (int t = NB - 2; t >= 0; --t) cont = fma(cont, x, c[t]);
                v = fmax(pay_j[q], cont);
            } else if (pay_j[q] < 1e-14) {
                v = vn;
            } else {
                v = 0.0;
            }
            V[q] = v;
        }
        __syncthreads();  // sm_coef / red are rewritten on the next date
    }
    double f[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (threadIdx.x + 256 * q < n) {
            f[0] += V[q];
            f[1] += V[q] * V[q];
        }
    }
    __shared__ double red2[2 * 4];
    block_sum<2, 4>(f, red2);
    if (threadIdx.x == 0) {
        out3[0] = f[0];
        out3[1] = f[1];
        out3[2] = (double)n;
    }
}

// The same sweep for one row of at most 256 paths on ONE wavefront (four paths per lane): the moments are reduced
// with an xor butterfly, which leaves the totals in every lane, so every lane runs the identical solve and no hand-over
// through LDS and no workgroup barrier is needed.  A 256-thread workgroup prices four rows at once; the per-date
// critical path (reduce -> solve -> update) is latency-bound, so four independent rows per workgroup give close to
// four times the throughput of lsm_small_body on the batched driver rows.  sum_v, sum_v2: sums of V and V^2 (all lanes).
template <int NB>
__device__ __forceinline__ void lsm_wave_body(const double* data, int64_t ld, int n, int n_cols, double K, double maturity,
                                              double dt, double disc, int is_call, double& sum_v, double& sum_v2) {
    constexpr int NM = 3 * NB - 1;
    const int lane = threadIdx.x & 63;
    const bool call = is_call != 0;
    const double invK = 1.0 / K;
    double V[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = lane + 64 * q;
        V[q] = p < n ? payoff_of(call, data[(int64_t)(n_cols - 1) * ld + p], K) : 0.0;
    }
    for (int j = n_cols - 2; j >= 0; --j) {
        const double this_time = j * dt;
        if (this_time > maturity) {  // LSMPricer.cpp:43-49 (wave-uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) V[q] *= disc;
            continue;
        }
        const double* row = data + (int64_t)j * ld;
        double s_j[4], m[NM];
#pragma unroll
        for (int t = 0; t < NM; ++t) m[t] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = lane + 64 * q;
            s_j[q] = p < n ? row[p] : 0.0;
            if (p < n && payoff_of(call, s_j[q], K) > 1e-14) {
                const double x = fma(s_j[q], invK, -1.0);
                const double y = V[q] * disc;
                double pw = 1.0;
#pragma unroll
                for (int t = 0; t < 2 * NB - 1; ++t) {
                    m[t] += pw;
                    if (t < NB) m[2 * NB - 1 + t] = fma(pw, y, m[2 * NB - 1 + t]);
                    pw *= x;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NM; ++t) m[t] = wave_sum(m[t]);
        double coef[10];
        lsm_solve_nb<NB>(m, 1.0, coef);  // identical in every lane
        const double n_itm = coef[9];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double pay = payoff_of(call, s_j[q], K);
            const double vn = V[q] * disc;
            double v;
            if (pay > 1e-14 && n_itm > 0.0) {
                const double x = fma(s_j[q], invK, -1.0);
                double cont = coef[NB - 1];
#pragma unroll
                for (int t = NB - 2; t >= 0; --t) cont = fma(cont, x, coef[t]);
                v = fmax(pay, cont);
            } else if (pay < 1e-14) {
                v = vn;
            } else {
                v = 0.0;
            }
            V[q] = lane + 64 * q < n ? v : 0.0;
        }
    }
    double f = 0.0, f2 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f += V[q];
        f2 = fma(V[q], V[q], f2);
    }
    sum_v = wave_sum(f);
    sum_v2 = wave_sum(f2);
}

}  // namespace mcg
