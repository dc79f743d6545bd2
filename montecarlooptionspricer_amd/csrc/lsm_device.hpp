// Device-side pieces of the Longstaff-Schwartz sweep shared by kernels_lsm.hip and the batched
// driver-row kernels (kernels_batch.hip): the small normal-equation solve and the one-workgroup sweep.
#pragma once
#include "devmath.hpp"

namespace mcg {

// Layout of the coefficient block every solve writes (doubles): the fitted continuation value at price S is
//   sum_t coef[t] y^t,  y = (S/K - 1) - coef[LSM_C_CENTER].
constexpr int LSM_MAX_NB = 16;     // basis functions a coefficient block has room for (poly_order <= 15)
constexpr int LSM_C_COUNT = 16;    // number of regression rows (in-the-money paths; global when sharded)
constexpr int LSM_C_CENTER = 17;   // centre of the regressor (0 unless the date was refined)
constexpr int LSM_C_REFINE = 18;   // 1: the caller should re-accumulate the moments about coef[LSM_C_HINT] and call
constexpr int LSM_C_HINT = 19;     //    lsm_solve_centered; the coefficients already written are provisional
constexpr int LSM_COEF_DOUBLES = 20;
constexpr int LSM_COEF_STRIDE = 24;  // doubles between two coefficient blocks in memory / size of one in LDS

// LDS workspace of lsm_solve_centered for basis size nb (doubles)
__host__ __device__ constexpr int lsm_ws_doubles(int nb) { return 7 * nb * nb + 8 * nb; }

// First-pass solve from the moments of x = S/K - 1 (power sums m[0..2p], cross sums m[2p+1..3p+1]); one thread.
// G[a][b] = m[a+b], rhs[a] = m[2p+1+a], equilibrated to unit diagonal.
//
// What the reference computes (LSMPricer.cpp:61-76) is Eigen's bdcSvd().solve(b) on the RAW monomials 1, S, .., S^p:
// the minimum-norm least-squares solution with singular values below min(rows, cols) eps sigma_max treated as zero.
// Where that matrix has full numerical rank and the data are well spread, the fitted values are those of the unique
// least-squares polynomial and any accurate method yields them: here LDL^T on the equilibrated normal equations of
// the scaled regressor (cond ~1e2 instead of ~1e8; ~1 us).  The fast path is taken only when both hold:
//   * every pivot > 1e-6 (normal equations then lose at most ~1e-10 of relative accuracy), and
//   * est = (std(S) / max(mean(S), 1)^2)^p -- the size of sigma_min / sigma_max of the raw-monomial matrix for data
//     of that spread -- exceeds Eigen's threshold by four orders of magnitude, so Eigen sees full rank as well.
// Otherwise (few or nearly coincident in-the-money prices, all paths equal at j = 0, high orders whose raw monomials
// Eigen itself truncates) the date is REFINED: coef[LSM_C_REFINE] = 1 asks the caller to re-accumulate the moments
// about the mean coef[LSM_C_HINT] and to call lsm_solve_centered, which reproduces Eigen's truncated solve; every
// caller does (the per-date kernels by spending a second launch on the date).  The coefficients written here are
// then provisional (the LDL^T solution when it exists, else zeros).
// K <= 0 switches the refinement request off (no in-the-money path can exist for a put then; a call degenerates).
// NB is a template parameter so that every loop unrolls and G, Q live in registers: with a run-time size the
// arrays go to scratch memory and the (serial, one-thread) solve takes ~20 us instead of ~2.
// 1/x and x^(-1/2) for positive normal x from the hardware seeds (v_rcp_f64, v_rsq_f64) and two Newton steps (one
// second-order step), <= 1 ulp: the solve runs on ONE thread between two grid-wide hand-shakes of the one-launch
// sweeps, where the ~35 dependent instructions of a correctly rounded fp64 division (a dozen of them per date) were a
// quarter of the date's 9 us.
__device__ __forceinline__ double lsm_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double lsm_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);  // 1 - x y^2
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

template <int NB>
__device__ __forceinline__ void lsm_solve_nb(const double* moments, double min_count, double K, double* coef) {
    constexpr int nb = NB;
    double G[NB][NB], Q[NB][NB], rhs[NB], d[NB], sol[NB], inv_piv[NB];
    const double count = moments[0];
    coef[LSM_C_COUNT] = count;
    for (int a = 0; a < LSM_MAX_NB; ++a) coef[a] = 0.0;
    coef[LSM_C_CENTER] = 0.0;
    coef[LSM_C_REFINE] = 0.0;
    coef[LSM_C_HINT] = 0.0;
    if (!(count >= min_count) || !(count > 0.0)) return;  // too few samples: coefficients stay 0
#pragma unroll
    for (int a = 0; a < nb; ++a) {
        const double g = moments[2 * a];
        d[a] = g > 0.0 ? lsm_rsqrt(g) : 0.0;
    }
#pragma unroll
    for (int a = 0; a < nb; ++a) {
        rhs[a] = moments[2 * nb - 1 + a] * d[a];
#pragma unroll
        for (int b = 0; b < nb; ++b) G[a][b] = moments[a + b] * d[a] * d[b];
    }
    // ---- fast path: LDL^T in Q (L below the diagonal, D on it) ----
    // No early exit: a loop that can break is not unrolled, and its G / Q then live in scratch memory behind run-time
    // indices (that was most of the 2 us this solve took between two grid-wide hand-shakes).  After a failed pivot the
    // remaining columns are computed on a stand-in pivot of 1 and never used.
    bool ok = true;
    double piv_min = 1.0;
#pragma unroll
    for (int j = 0; j < nb; ++j) {
        double dj = G[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= Q[j][k] * Q[j][k] * Q[k][k];
        const bool good = dj > 1e-10;
        if (ok && good) piv_min = fmin(piv_min, dj);
        ok = ok && good;
        Q[j][j] = dj;
        const double inv = lsm_rcp(good ? dj : 1.0);
        inv_piv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < nb; ++i) {
            double v = G[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= Q[i][k] * Q[j][k] * Q[k][k];
            Q[i][j] = v * inv;
        }
    }
    bool trusted = ok;  // fast path accurate AND Eigen certainly at full rank
    if (nb > 1 && K > 0.0) {
        const double inv_count = lsm_rcp(count);
        const double mean_x = moments[1] * inv_count;
        const double var_x = fmax(fma(moments[2], inv_count, -(mean_x * mean_x)), 0.0);
        const double mean_s = K * (1.0 + mean_x);
        const double scale = fmax(fabs(mean_s), 1.0);
        // (an estimate compared with a threshold four orders of magnitude away: sqrt as x * x^(-1/2))
        const double ratio = var_x > 0.0 ? K * (var_x * lsm_rsqrt(var_x)) * lsm_rcp(scale * scale) : 0.0;
        double est = 1.0;
        for (int t = 1; t < nb; ++t) est *= ratio;
        trusted = ok && piv_min > 1e-6 && est > 1e4 * nb * 2.220446049250313e-16;
        if (!trusted) {
            coef[LSM_C_REFINE] = 1.0;
            coef[LSM_C_HINT] = mean_x;
        }
    }
    if (ok) {
#pragma unroll
        for (int i = 0; i < nb; ++i) {  // L y = rhs
            double v = rhs[i];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= Q[i][k] * sol[k];
            sol[i] = v;
        }
#pragma unroll
        for (int i = 0; i < nb; ++i) sol[i] *= inv_piv[i];
#pragma unroll
        for (int i = nb - 1; i >= 0; --i) {  // L^T x = y
            double v = sol[i];
#pragma unroll
            for (int k = i + 1; k < nb; ++k) v -= Q[k][i] * sol[k];
            sol[i] = v;
        }
#pragma unroll
        for (int a = 0; a < nb; ++a) coef[a] = sol[a] * d[a];
        return;
    }
    // ---- no LDL^T factorisation (fewer distinct in-the-money prices than basis functions): the coefficients stay 0 and
    // the refinement request above (always set here: !ok implies !trusted) has the caller re-fit the date with
    // lsm_solve_centered, which handles rank deficiency by the reference's own rule.  (Until round 3 a Jacobi
    // pseudo-inverse stood here as a fall-back for callers that could not refine; every caller refines now, and its
    // 720 bytes of private arrays were the scratch memory of every kernel that inlines this solve.)
}

// The refined solve: Eigen's bdcSvd().solve(b) on the raw monomials (LSMPricer.cpp:76), reproduced from the moments
// of the CENTRED regressor y = (S/K - 1) - mu (mu = mean over the regression rows, so the centred design matrix
// A_y = [1, y, .., y^p] is as well conditioned as the data allow).  One thread; all matrices in the LDS workspace ws
// (lsm_ws_doubles(nb) doubles): a rare path, kept out of the registers of the kernels that inline it.
//   1. G_y = A_y^T A_y from the moments; equilibrate; cyclic Jacobi: G_y = D^-1 Q L Q^T D^-1.  Eigenvalues below
//      1e-13 of the largest are exact duplicates in the data (columns of A_y that are numerically dependent
//      whatever the basis) and are dropped.  F = L^1/2 Q^T D^-1 has F^T F = G_y, i.e. A_y = U_y F with orthonormal
//      U_y, and z = F^-T A_y^T b = U_y^T b.
//   2. The raw monomials are A_raw = A_y T, S^k = (K (1 + mu) + K y)^k = sum_i C(k,i) (K (1 + mu))^(k-i) K^i y^i,
//      so A_raw = U_y B with B = F T (at most nb x nb): the singular values of B ARE Eigen's, and its left singular
//      vectors W give Eigen's in U_y coordinates.  B = diag(row scales) x (well-conditioned), the case in which
//      one-sided Jacobi on the rows delivers even the smallest singular values to high relative accuracy
//      (Demmel-Veselic); checked against 60-digit arithmetic for spreads 1e-2 .. 1e-8 in tests/.
//   3. Eigen's rule: keep sigma_i > min(rows, cols) eps sigma_max.  Fitted values = A_raw V_k S_k^-1 U_k^T b
//      = A_y F^-1 P_k z with P_k the projector on the kept left singular vectors: the coefficients in y are
//      F^-1 P_k z -- no raw-monomial coefficient (huge, cancelling) is ever formed.
__device__ __noinline__ void lsm_solve_centered(const double* mc, int nb, double mu, double K, double* coef, double* ws) {
    double* G = ws;                 // [nb][nb]
    double* Q = G + nb * nb;        // [nb][nb]
    double* F = Q + nb * nb;        // [nb][nb]  rows: kept eigen-directions
    double* Fi = F + nb * nb;       // [nb][nb]  F^-1: Fi[a][e]
    double* T = Fi + nb * nb;       // [nb][nb]
    double* B = T + nb * nb;        // [nb][nb]  rows of B are rotated in place
    double* W = B + nb * nb;        // [nb][nb]
    double* d = W + nb * nb;        // [nb]
    double* z = d + nb;             // [nb]
    double* sig = z + nb;           // [nb]
    double* pz = sig + nb;          // [nb]
    const double count = mc[0];
    for (int a = 0; a < LSM_MAX_NB; ++a) coef[a] = 0.0;
    coef[LSM_C_COUNT] = count;
    coef[LSM_C_CENTER] = mu;
    coef[LSM_C_REFINE] = 0.0;
    coef[LSM_C_HINT] = 0.0;
    if (!(count > 0.0)) return;
    for (int a = 0; a < nb; ++a) {
        const double g = mc[2 * a];
        d[a] = g > 0.0 ? 1.0 / sqrt(g) : 0.0;
    }
    for (int a = 0; a < nb; ++a)
        for (int b = 0; b < nb; ++b) {
            G[a * nb + b] = mc[a + b] * d[a] * d[b];
            Q[a * nb + b] = a == b ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {  // cyclic Jacobi on the symmetric G
        double off = 0.0;
        for (int p = 0; p < nb; ++p)
            for (int q = p + 1; q < nb; ++q) off += G[p * nb + q] * G[p * nb + q];
        if (off < 1e-60) break;
        for (int p = 0; p < nb - 1; ++p) {
            for (int q = p + 1; q < nb; ++q) {
                const double apq = G[p * nb + q];
                if (apq == 0.0) continue;
                const double theta = (G[q * nb + q] - G[p * nb + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < nb; ++k) {
                    const double gkp = G[k * nb + p], gkq = G[k * nb + q];
                    G[k * nb + p] = cs * gkp - sn * gkq;
                    G[k * nb + q] = sn * gkp + cs * gkq;
                }
                for (int k = 0; k < nb; ++k) {
                    const double gpk = G[p * nb + k], gqk = G[q * nb + k];
                    G[p * nb + k] = cs * gpk - sn * gqk;
                    G[q * nb + k] = sn * gpk + cs * gqk;
                }
                for (int k = 0; k < nb; ++k) {
                    const double qkp = Q[k * nb + p], qkq = Q[k * nb + q];
                    Q[k * nb + p] = cs * qkp - sn * qkq;
                    Q[k * nb + q] = sn * qkp + cs * qkq;
                }
            }
        }
    }
    double lmax = 0.0;
    for (int a = 0; a < nb; ++a) lmax = fmax(lmax, G[a * nb + a]);
    int kk = 0;  // kept eigen-directions
    for (int e = 0; e < nb; ++e) {
        const double lam = G[e * nb + e];
        if (!(lam > 1e-13 * lmax)) continue;
        const double sq = sqrt(lam);
        double ze = 0.0;
        for (int a = 0; a < nb; ++a) {
            F[kk * nb + a] = d[a] > 0.0 ? sq * Q[a * nb + e] / d[a] : 0.0;
            Fi[a * nb + kk] = d[a] * Q[a * nb + e] / sq;
            ze += Fi[a * nb + kk] * mc[2 * nb - 1 + a];
        }
        z[kk] = ze;
        ++kk;
    }
    if (kk == 0) return;
    const double c0 = K * (1.0 + mu);
    for (int k = 0; k < nb; ++k) {  // T[i][k] = C(k,i) c0^(k-i) K^i
        double binom = 1.0;
        for (int i = 0; i <= k; ++i) {
            double v = binom;
            for (int t = 0; t < k - i; ++t) v *= c0;
            for (int t = 0; t < i; ++t) v *= K;
            T[i * nb + k] = v;
            binom = binom * (double)(k - i) / (double)(i + 1);
        }
        for (int i = k + 1; i < nb; ++i) T[i * nb + k] = 0.0;
    }
    for (int e = 0; e < kk; ++e)
        for (int k = 0; k < nb; ++k) {
            double v = 0.0;
            for (int i = 0; i < nb; ++i) v += F[e * nb + i] * T[i * nb + k];
            B[e * nb + k] = v;
        }
    for (int a = 0; a < kk; ++a)
        for (int b = 0; b < kk; ++b) W[a * kk + b] = a == b ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {  // one-sided Jacobi: make the rows of B orthogonal, W collects the rotations
        double worst = 0.0;
        for (int p = 0; p < kk - 1; ++p) {
            for (int q = p + 1; q < kk; ++q) {
                double a2 = 0.0, c2 = 0.0, g = 0.0;
                for (int k = 0; k < nb; ++k) {
                    a2 += B[p * nb + k] * B[p * nb + k];
                    c2 += B[q * nb + k] * B[q * nb + k];
                    g += B[p * nb + k] * B[q * nb + k];
                }
                const double lim = sqrt(a2 * c2);
                if (!(fabs(g) > 1e-19 * lim)) continue;
                worst = fmax(worst, fabs(g) / lim);
                const double zeta = (c2 - a2) / (2.0 * g);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int k = 0; k < nb; ++k) {
                    const double bp = B[p * nb + k], bq = B[q * nb + k];
                    B[p * nb + k] = cs * bp - sn * bq;
                    B[q * nb + k] = sn * bp + cs * bq;
                }
                for (int k = 0; k < kk; ++k) {
                    const double wp = W[k * kk + p], wq = W[k * kk + q];
                    W[k * kk + p] = cs * wp - sn * wq;
                    W[k * kk + q] = sn * wp + cs * wq;
                }
            }
        }
        if (worst < 1e-15) break;
    }
    double smax = 0.0;
    for (int e = 0; e < kk; ++e) {
        double v = 0.0;
        for (int k = 0; k < nb; ++k) v += B[e * nb + k] * B[e * nb + k];
        sig[e] = sqrt(v);
        smax = fmax(smax, sig[e]);
    }
    const double thr = fmin(count, (double)nb) * 2.220446049250313e-16 * smax;  // Eigen: min(rows, cols) eps sigma_max
    for (int a = 0; a < kk; ++a) pz[a] = 0.0;
    for (int e = 0; e < kk; ++e) {
        if (!(sig[e] > thr)) continue;
        double proj = 0.0;
        for (int a = 0; a < kk; ++a) proj += W[a * kk + e] * z[a];
        for (int a = 0; a < kk; ++a) pz[a] += proj * W[a * kk + e];
    }
    for (int a = 0; a < nb; ++a) {
        double v = 0.0;
        for (int e = 0; e < kk; ++e) v += Fi[a * nb + e] * pz[e];
        coef[a] = v;
    }
}

// Regression inputs about a centre (the refinement pass): the same sums as every kernel's first pass with
// y = (S/K - 1) - mu in place of x.
template <int NB>
__device__ __forceinline__ void lsm_accumulate_centered(double (&m)[3 * NB - 1], bool itm, double s, double v, double invK, double mu,
                                                        double disc) {
    if (itm) {
        const double yv = fma(s, invK, -1.0) - mu;
        const double b = v * disc;
        double pw = 1.0;
#pragma unroll
        for (int t = 0; t < 2 * NB - 1; ++t) {
            m[t] += pw;
            if (t < NB) m[2 * NB - 1 + t] = fma(pw, b, m[2 * NB - 1 + t]);
            pw *= yv;
        }
    }
}

// Continuation value from a coefficient block: Horner in y = x - centre (centre = 0 exactly unless refined).
template <int NB>
__device__ __forceinline__ double lsm_continuation(const double (&c)[NB], double center, double x) {
    const double y = x - center;
    double cont = c[NB - 1];
#pragma unroll
    for (int t = NB - 2; t >= 0; --t) cont = fma(cont, y, c[t]);
    return cont;
}

// run-time order -> the unrolled instance
__device__ inline void lsm_solve_one(const double* moments, int nb, double min_count, double K, double* coef) {
    switch (nb) {
        case 1: lsm_solve_nb<1>(moments, min_count, K, coef); break;
        case 2: lsm_solve_nb<2>(moments, min_count, K, coef); break;
        case 3: lsm_solve_nb<3>(moments, min_count, K, coef); break;
        case 4: lsm_solve_nb<4>(moments, min_count, K, coef); break;
        case 5: lsm_solve_nb<5>(moments, min_count, K, coef); break;
        case 6: lsm_solve_nb<6>(moments, min_count, K, coef); break;
        case 7: lsm_solve_nb<7>(moments, min_count, K, coef); break;
        case 8: lsm_solve_nb<8>(moments, min_count, K, coef); break;
        default: lsm_solve_nb<9>(moments, min_count, K, coef); break;
    }
}

// The whole backward sweep in ONE launch for a small path count (n <= 1024: the reference's production
// calls price 250 paths per option row, src/core/PredictionGen.cpp:719): one 256-thread block, up to four
// paths per thread with V in registers, per date a block reduction of the regression moments, the solve on
// thread 0, the coefficients handed over through LDS (PPT = paths per thread: 1 up to 256 paths, else 4).  Same arithmetic as k_lsm_date / lsm_solve_one;
// only the summation order of the moments differs.  out3 = {sum V, sum V^2, n}.
template <int NB, int PPT>
__device__ __forceinline__ void lsm_small_body(const double* data, int64_t ld, int n, int n_cols, double K,
                                               double maturity, double dt, double disc, int is_call, double* out3) {
    constexpr int NM = 3 * NB - 1;
    __shared__ double red[NM * 4];
    __shared__ double sm_mom[32];
    __shared__ double sm_coef[LSM_COEF_STRIDE];
    __shared__ double sm_ws[lsm_ws_doubles(NB)];
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
            lsm_solve_nb<NB>(sm_mom, 1.0, K, sm_coef);  // writes the coefficient block (lsm_device.hpp: LSM_C_*)
        }
        __syncthreads();
        if (sm_coef[LSM_C_REFINE] != 0.0) {  // uniform: the date is re-fitted about the mean (see lsm_solve_nb)
            const double mu = sm_coef[LSM_C_HINT];
#pragma unroll
            for (int q = 0; q < NM; ++q) m[q] = 0.0;
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                lsm_accumulate_centered<NB>(m, threadIdx.x + 256 * q < n && pay_j[q] > 1e-14, s_j[q], V[q], invK, mu, disc);
            block_sum<NM, 4>(m, red);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int t = 0; t < NM; ++t) sm_mom[t] = m[t];
                lsm_solve_centered(sm_mom, NB, mu, K, sm_coef, sm_ws);
            }
            __syncthreads();
        }
        double c[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) c[t] = sm_coef[t];
        const double n_itm = sm_coef[LSM_C_COUNT], center = sm_coef[LSM_C_CENTER];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double vn = V[q] * disc;
            double v;
            if (pay_j[q] > 1e-14 && n_itm > 0.0) {
                v = fmax(pay_j[q], lsm_continuation<NB>(c, center, fma(s_j[q], invK, -1.0)));
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

// The same sweep for one row of at most 256 paths on ONE wavefront (four paths per lane): the moments are reduced by the
// folded butterfly (wave_sum_all: the totals come back wave-uniform), so every lane runs the identical solve and no
// hand-over through LDS and no workgroup barrier is needed.  The per-date critical path (reduce -> solve -> update) is
// latency-bound, so independent rows on the waves of a CU give close to four times the throughput of lsm_small_body on the
// batched driver rows (k_batch_lsm: one wavefront per workgroup).  sum_v, sum_v2: sums of V and V^2 (all lanes).
template <int NB>
__device__ __forceinline__ void lsm_wave_body(const double* data, int64_t ld, int n, int n_cols, double K, double maturity,
                                              double dt, double disc, int is_call, double* ws /* LDS, this wave's own:
                                              lsm_ws_doubles(NB) + LSM_COEF_DOUBLES doubles */,
                                              double& sum_v, double& sum_v2) {
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
    // A date's critical path is load -> moments -> eight wave reductions -> solve -> update, and the load does not depend
    // on anything before it: row j-1 is requested before date j's moments are formed (round 4; the batched rows' LSM
    // kernel: 2.13 -> 1.88 ms at 20 000 rows; the asymptotic scan with four loads in flight 0.62 -> 0.46; the same
    // treatment made MartingaleOptimization's scans slower, 0.80 -> 1.10, and left BranchingProcesses where it was: not taken there).
    double s_nxt[4] = {0.0, 0.0, 0.0, 0.0};
    auto load_row = [&](int jj, double (&dst)[4]) {
        const double* row = data + (int64_t)jj * ld;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = lane + 64 * q;
            dst[q] = p < n ? row[p] : 0.0;
        }
    };
    if (n_cols >= 2) load_row(n_cols - 2, s_nxt);
    for (int j = n_cols - 2; j >= 0; --j) {
        const double this_time = j * dt;
        double s_j[4], m[NM];
#pragma unroll
        for (int q = 0; q < 4; ++q) s_j[q] = s_nxt[q];
        if (j >= 1) load_row(j - 1, s_nxt);
        if (this_time > maturity) {  // LSMPricer.cpp:43-49 (wave-uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) V[q] *= disc;
            continue;
        }
#pragma unroll
        for (int t = 0; t < NM; ++t) m[t] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = lane + 64 * q;
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
        wave_sum_all<NM>(m);   // folded: 42 instead of 176 instructions for the eight moments of order 2, bit for bit the same sums
        double coef[LSM_COEF_DOUBLES];
        lsm_solve_nb<NB>(m, 1.0, K, coef);  // identical in every lane
        if (coef[LSM_C_REFINE] != 0.0) {    // wave-uniform: re-fit about the mean (see lsm_solve_nb)
            const double mu = coef[LSM_C_HINT];
            double mc[NM];
#pragma unroll
            for (int t = 0; t < NM; ++t) mc[t] = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                lsm_accumulate_centered<NB>(mc, lane + 64 * q < n && payoff_of(call, s_j[q], K) > 1e-14, s_j[q], V[q], invK, mu,
                                            disc);
            wave_sum_all<NM>(mc);
            double* ws_coef = ws + lsm_ws_doubles(NB);
            if (lane == 0) lsm_solve_centered(mc, NB, mu, K, ws_coef, ws);
            // one wave: its LDS operations execute in program order, so the reads below see lane 0's result
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < LSM_COEF_DOUBLES; ++t) coef[t] = ws_coef[t];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read out before the next refined date overwrites it
        }
        const double n_itm = coef[LSM_C_COUNT], center = coef[LSM_C_CENTER];
        double c[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) c[t] = coef[t];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double pay = payoff_of(call, s_j[q], K);
            const double vn = V[q] * disc;
            double v;
            if (pay > 1e-14 && n_itm > 0.0) {
                v = fmax(pay, lsm_continuation<NB>(c, center, fma(s_j[q], invK, -1.0)));
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
