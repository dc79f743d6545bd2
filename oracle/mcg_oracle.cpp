// TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a CPU restatement of the reference's
// hot path (bcosm/MonteCarloOptionsPricer).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it -- and only as the checker.  The product (libmcgpu.so) never links,
// loads or calls anything in oracle/; it fails loudly without a GPU.
//
// Pinning status (see DESIGN.md "Oracle"):
//   * path-engine numerics (estimators, FFT, lambda, phi, fractionalGaussian, forwardVariance,
//     payoff), AsymptoticAnalysis and the BranchingProcesses lower bound: PINNED bit-for-bit against the
//     compiled reference (oracle/_ref/libmcref.so, built in place from /root/reference/src/models/
//     {RoughVolatility,AsymptoticAnalysisPricer,BranchingProcessPricer}.cpp) via tests/golden/*.npz.
//   * path generation: the reference is unseeded (std::random_device per call), so parity is
//     statistical; "mt" mode below reproduces the reference's RNG consumption order with an explicit
//     seed, "philox" mode mirrors the device algorithm draw-for-draw.
//     In the rough regime (H = 0.1, eta = 1.9), which the reference's public entry point cannot reach, both modes
//     are compared with a committed sample drawn through the compiled reference's own private members
//     (ref_harness.cpp: ref_explicit_stats_omp; tests/golden/rough_regime_reference.json).
//   * LSM and MartingaleOptimization: PARITY UNPINNED at the Eigen boundary -- Eigen3 is not in this image and
//     the reference has no tests; `orc_lsm_price` / `orc_martingale_price` restate the two pricers with an
//     independent one-sided Jacobi SVD (min-norm least squares, Eigen's rank threshold) and are cross-checked
//     against LAPACK gelsd (numpy.linalg.lstsq) and, on near-degenerate dates, against Eigen's rule evaluated in
//     60-digit arithmetic (mpmath) in tests/.
//   * BranchingProcesses upper bound: the reference resamples with an unseeded mt19937; compared statistically.
//
// Every function cites the reference lines it follows (paths relative to /root/reference).
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// src/models/RoughVolatility.cpp:20-42 -- mean / sample variance / sample covariance
// ---------------------------------------------------------------------------------------------
double mean_of(const double* v, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += v[i];
    return n == 0 ? 0.0 : s / n;
}

double variance_of(const double* v, size_t n) {
    if (n < 2) return 0.0;
    const double m = mean_of(v, n);
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = v[i] - m;
        acc += d * d;
    }
    return acc / (n - 1);
}

double covariance_of(const double* x, const double* y, size_t n) {
    if (n < 2) return 0.0;
    const double mx = mean_of(x, n), my = mean_of(y, n);
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) acc += (x[i] - mx) * (y[i] - my);
    return acc / (n - 1);
}

// RoughVolatility.cpp:44-70 -- remove the least-squares line over t = 1..n from a window.
void detrend_window(std::vector<double>& w) {
    const size_t n = w.size();
    if (n < 2) return;
    std::vector<double> t(n);
    for (size_t i = 0; i < n; ++i) t[i] = static_cast<double>(i + 1);
    const double tm = mean_of(t.data(), n);
    const double ym = mean_of(w.data(), n);
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < n; ++i) {
        num += (t[i] - tm) * (w[i] - ym);
        den += (t[i] - tm) * (t[i] - tm);
    }
    if (std::abs(den) < 1e-14) return;
    const double slope = num / den;
    const double icpt = ym - slope * tm;
    for (size_t i = 0; i < n; ++i) w[i] -= (slope * t[i] + icpt);
}

// RoughVolatility.cpp:72-122 -- detrended fluctuation analysis slope.
double hurst_dfa(const double* in, size_t n) {
    if (n < 2) return 0.5;
    std::vector<double> prof(in, in + n);
    const double m = mean_of(prof.data(), n);
    for (size_t i = 0; i < n; ++i) prof[i] -= m;
    for (size_t i = 1; i < n; ++i) prof[i] += prof[i - 1];

    std::vector<double> lx, ly;
    const size_t wmax = n / 4;
    for (size_t w = 4; w <= wmax; w *= 2) {
        std::vector<double> fl;
        for (size_t s = 0; s + w <= n; s += w) {
            std::vector<double> seg(prof.begin() + s, prof.begin() + s + w);
            detrend_window(seg);
            double ss = 0.0;
            for (double q : seg) ss += q * q;
            fl.push_back(std::sqrt(ss / w));
        }
        const double mf = mean_of(fl.data(), fl.size());
        if (mf > 0.0) {
            lx.push_back(std::log((double)w));
            ly.push_back(std::log(mf));
        }
    }
    const size_t k = lx.size();
    if (k < 2) return 0.5;
    double sx = 0.0, sy = 0.0, sxx = 0.0, sxy = 0.0;
    for (size_t i = 0; i < k; ++i) {
        sx += lx[i];
        sy += ly[i];
        sxx += lx[i] * lx[i];
        sxy += lx[i] * ly[i];
    }
    return (k * sxy - sx * sy) / (k * sxx - sx * sx);
}

// RoughVolatility.cpp:171-202 -- iterative radix-2 FFT, twiddle by repeated multiplication;
// `inv >= 0` uses e^{+i...} and no scaling, `inv < 0` uses e^{-i...} and divides by n.
// Complex products are written out the way libstdc++'s std::complex<double> evaluates them
// (ac-bd, ad+bc; no FMA -- this TU is built with -ffp-contract=off).
void fft_inplace(double* re, double* im, size_t n, int inv) {
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = 2 * M_PI / len * (inv < 0 ? -1 : 1);
        const double wlr = std::cos(ang), wli = std::sin(ang);
        for (size_t i = 0; i < n; i += len) {
            double wr = 1.0, wi = 0.0;
            for (size_t j = 0; j < len / 2; ++j) {
                const size_t a = i + j, b = i + j + len / 2;
                const double ur = re[a], ui = im[a];
                const double vr = re[b] * wr - im[b] * wi;
                const double vi = re[b] * wi + im[b] * wr;
                re[a] = ur + vr;
                im[a] = ui + vi;
                re[b] = ur - vr;
                im[b] = ui - vi;
                const double nr = wr * wlr - wi * wli;
                const double ni = wr * wli + wi * wlr;
                wr = nr;
                wi = ni;
            }
        }
    }
    if (inv < 0) {
        const double dn = static_cast<double>(n);
        for (size_t i = 0; i < n; ++i) {
            re[i] /= dn;
            im[i] /= dn;
        }
    }
}

size_t next_pow2(size_t n) {  // RoughVolatility.cpp:204-210
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; Random123 constants).  Stream definition shared with the
// device code (DESIGN.md "RNG contract"): key = (seed_lo, seed_hi),
// counter = (path_lo, path_hi, block, stream).
// ---------------------------------------------------------------------------------------------
inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}

void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        if (r) {
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        philox_round(c, k);
    }
    std::memcpy(out, c, sizeof(c));
}

// One Philox block -> FOUR standard normals: two Box-Muller pairs of 64 bits each.  For a pair
// (wa, wb): radius uniform u = ((wb & 0xFF)*2^32 + wa + 1/2) * 2^-40, angle fraction
// f = ((wb >> 8) + 1/2) * 2^-24, z_even = sqrt(-2 ln u) cos(2 pi f), z_odd = ... sin(2 pi f).
// Pair A = (w0, w1), pair B = (w2, w3).  Element e of block b is draw 4b + e of its (path, stream).
void normal_quad(uint64_t seed, uint64_t path, uint32_t block, uint32_t stream, double z[4]) {
    const uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), block, stream};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    philox4x32_10(ctr, key, x);
    for (int h = 0; h < 2; ++h) {
        const uint32_t wa = x[2 * h], wb = x[2 * h + 1];
        const uint64_t a40 = ((uint64_t)(wb & 0xFFu) << 32) | wa;
        const double u = ((double)a40 + 0.5) * 0x1p-40;
        const double f = ((double)(wb >> 8) + 0.5) * 0x1p-24;
        const double rad = std::sqrt(-2.0 * std::log(u));
        const double th = 2.0 * M_PI * f;
        z[2 * h] = rad * std::cos(th);
        z[2 * h + 1] = rad * std::sin(th);
    }
}

enum : uint32_t { STREAM_PRICE = 0, STREAM_VOL = 1 };

inline double payoff_of(bool is_call, double s, double k) {  // include/core/common.h:8-14
    return is_call ? std::max(0.0, s - k) : std::max(0.0, k - s);
}

// One-sided (Hestenes) Jacobi SVD of a tall column-major matrix A (rows x cols, cols <= 16),
// then the minimum-norm least-squares solution of A c = b with Eigen's rank rule
// (singular values <= min(rows,cols) * eps * sigma_max count as zero).  Stands in for
// `A.bdcSvd(ComputeThinU|ComputeThinV).solve(b)` at src/models/LSMPricer.cpp:76.
void minnorm_lstsq(std::vector<double>& A, size_t rows, int cols, const std::vector<double>& b,
                   double* c_out) {
    std::vector<double> V((size_t)cols * cols, 0.0);
    for (int i = 0; i < cols; ++i) V[(size_t)i * cols + i] = 1.0;
    auto col = [&](int j) { return A.data() + (size_t)j * rows; };
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < cols - 1; ++p) {
            for (int q = p + 1; q < cols; ++q) {
                double app = 0.0, aqq = 0.0, apq = 0.0;
                const double *cp = col(p), *cq = col(q);
                for (size_t i = 0; i < rows; ++i) {
                    app += cp[i] * cp[i];
                    aqq += cq[i] * cq[i];
                    apq += cp[i] * cq[i];
                }
                if (apq == 0.0 || std::abs(apq) <= 1e-17 * std::sqrt(app * aqq)) continue;
                rotated = true;
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::abs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                double *wp = col(p), *wq = col(q);
                for (size_t i = 0; i < rows; ++i) {
                    const double x = wp[i], y = wq[i];
                    wp[i] = cs * x - sn * y;
                    wq[i] = sn * x + cs * y;
                }
                for (int i = 0; i < cols; ++i) {
                    const double x = V[(size_t)p * cols + i], y = V[(size_t)q * cols + i];
                    V[(size_t)p * cols + i] = cs * x - sn * y;
                    V[(size_t)q * cols + i] = sn * x + cs * y;
                }
            }
        }
        if (!rotated) break;
    }
    // Columns of A are now U*Sigma; V holds right singular vectors column j in V[j*cols + :].
    double sig[16], smax = 0.0;
    for (int j = 0; j < cols; ++j) {
        double s = 0.0;
        const double* cj = col(j);
        for (size_t i = 0; i < rows; ++i) s += cj[i] * cj[i];
        sig[j] = std::sqrt(s);
        smax = std::max(smax, sig[j]);
    }
    const double diag = (double)std::min<size_t>(rows, (size_t)cols);
    const double thr = std::max(smax * diag * 2.220446049250313e-16, 2.2250738585072014e-308);
    for (int i = 0; i < cols; ++i) c_out[i] = 0.0;
    for (int j = 0; j < cols; ++j) {
        if (!(sig[j] > thr)) continue;
        double utb = 0.0;  // (u_j . b) = (A_j . b) / sigma_j
        const double* cj = col(j);
        for (size_t i = 0; i < rows; ++i) utb += cj[i] * b[i];
        const double w = utb / (sig[j] * sig[j]);
        for (int i = 0; i < cols; ++i) c_out[i] += w * V[(size_t)j * cols + i];
    }
}

}  // namespace

extern "C" {

// ---- estimators --------------------------------------------------------------------------------
// RoughVolatility.cpp:126-133.  Writes n-1 log returns.
void orc_log_returns(const double* prices, size_t n, double* out) {
    for (size_t i = 1; i < n; ++i) out[i - 1] = std::log(prices[i] / prices[i - 1]);
}

// RoughVolatility.cpp:321-331 with :141-169.  out5 = {xi, H, eta, rho, S0}.  Returns 1 when the
// history is too short (the reference throws at :317-319).
int orc_estimate_params(const double* hist, size_t n, double* out5) {
    if (n < 2) return 1;
    std::vector<double> rets(n - 1);
    orc_log_returns(hist, n, rets.data());
    const size_t m = rets.size();
    const double dt = 1.0 / 252.0;
    out5[0] = variance_of(rets.data(), m) / dt;            // estimateXi :141-145
    out5[1] = hurst_dfa(rets.data(), m);                   // estimateH  :147-149
    out5[2] = std::sqrt(variance_of(rets.data(), m)) * 2.0;  // estimateEta :151-155
    std::vector<double> sq(m);
    for (size_t i = 0; i < m; ++i) sq[i] = rets[i] * rets[i];
    const double c = covariance_of(rets.data(), sq.data(), m);
    double rho = c / (std::sqrt(variance_of(rets.data(), m) * variance_of(sq.data(), m)));
    if (rho > 0.0) rho = -0.3;  // :165-167
    out5[3] = rho;
    out5[4] = hist[n - 1];
    return 0;
}

// ---- spectral pieces ---------------------------------------------------------------------------
size_t orc_next_pow2(size_t n) { return next_pow2(n); }

void orc_fft(double* reim, size_t n, int inv) {  // interleaved re/im, in place
    std::vector<double> re(n), im(n);
    for (size_t i = 0; i < n; ++i) {
        re[i] = reim[2 * i];
        im[i] = reim[2 * i + 1];
    }
    fft_inplace(re.data(), im.data(), n, inv);
    for (size_t i = 0; i < n; ++i) {
        reim[2 * i] = re[i];
        reim[2 * i + 1] = im[i];
    }
}

// RoughVolatility.cpp:337-342, :227-236.  lam[i] = 0.5 * (i*dt)^(2H), i = 0..steps.
void orc_lambda(int steps, double H, double dt, double* lam) {
    for (size_t i = 0; i <= (size_t)steps; ++i) lam[i] = 0.5 * (std::pow(i * dt, 2 * H));
}

// RoughVolatility.cpp:212-225.  phi = FFT(+)(lam zero-padded to M_phi = nextpow2(n)).
size_t orc_phi(const double* lam, size_t n, double* phi_reim) {
    const size_t M = next_pow2(n);
    std::vector<double> re(M, 0.0), im(M, 0.0);
    for (size_t i = 0; i < n; ++i) re[i] = lam[i];
    fft_inplace(re.data(), im.data(), M, 1);
    for (size_t i = 0; i < M; ++i) {
        phi_reim[2 * i] = re[i];
        phi_reim[2 * i + 1] = im[i];
    }
    return M;
}

// RoughVolatility.cpp:264-292.  A_k = phi_k * Z_k (k < steps), zero-pad to M_z = nextpow2(steps),
// FFT(-)/M_z, X_n = sqrt(2H)*eta * Re A_n for n < steps.
void orc_fractional_gaussian(const double* phi_reim, const double* Z_reim, size_t steps, double H,
                             double eta, double* X) {
    const size_t M = next_pow2(steps);
    std::vector<double> re(M, 0.0), im(M, 0.0);
    for (size_t k = 0; k < steps; ++k) {
        const double pr = phi_reim[2 * k], pi = phi_reim[2 * k + 1];
        const double zr = Z_reim[2 * k], zi = Z_reim[2 * k + 1];
        re[k] = pr * zr - pi * zi;
        im[k] = pr * zi + pi * zr;
    }
    fft_inplace(re.data(), im.data(), M, -1);
    const double scale = std::sqrt(2 * H) * eta;
    for (size_t n = 0; n < steps; ++n) X[n] = scale * re[n];
}

// RoughVolatility.cpp:294-309.  v_n = xi * exp(X_n - 0.5*eta^2 * t_n^(2H)), t_n = n*dt.
void orc_forward_variance(const double* X, size_t steps, double dt, double xi, double H, double eta,
                          double* v) {
    for (size_t n = 0; n < steps; ++n) {
        const double t = n * dt;
        const double ma = -0.5 * eta * eta * std::pow(t, 2 * H);
        v[n] = xi * std::exp(X[n] + ma);
    }
}

// RoughVolatility.cpp:354-364.  S has steps+1 entries.
void orc_step_prices(double S0, double r, double dt, double rho, const double* v, const double* W1,
                     const double* W2, size_t steps, double* S) {
    S[0] = S0;
    for (size_t j = 1; j <= steps; ++j) {
        const double dw1 = std::sqrt(dt) * W1[j - 1];
        const double dw2 = std::sqrt(dt) * W2[j - 1];
        const double dW = rho * dw1 + std::sqrt(1.0 - rho * rho) * dw2;
        const double vt = v[j - 1];
        const double drift = (r - 0.5 * vt) * dt;
        const double diff = std::sqrt(std::max(0.0, vt)) * dW;
        S[j] = S[j - 1] * std::exp(drift + diff);
    }
}

double orc_payoff(int is_call, double s, double k) { return payoff_of(is_call != 0, s, k); }

// ---- reference-faithful generator ("mt" mode) --------------------------------------------------
// RoughVolatility.cpp:312-368 with explicit parameters instead of the history, and with the
// per-call `std::random_device` seeds (:239-240, :253-254) replaced by draws from a seeded
// mt19937 so a run is reproducible.  RNG consumption order per path is the reference's:
// one engine for Z (re, im interleaved), one for W1, one for W2; libstdc++'s
// std::normal_distribution (polar method, caches the second deviate) as in the reference.
// out is path-major [paths][steps+1].  If v_mean_out != nullptr it receives, per path,
// sum_j v_j*dt (the integrated variance used by the mixing-formula check).
int orc_generate_paths_mt(double S0, double r, double xi, double H, double eta, double rho, int steps,
                          long paths, uint64_t seed, double* out, double* intvar_out) {
    if (steps < 1 || paths < 0) return 1;
    const double dt = 1.0 / 252.0;
    std::vector<double> lam(steps + 1);
    orc_lambda(steps, H, dt, lam.data());
    std::vector<double> phi(2 * next_pow2((size_t)steps + 1));
    orc_phi(lam.data(), (size_t)steps + 1, phi.data());

    std::mt19937 seeder((uint32_t)(seed ^ (seed >> 32)));
    std::vector<double> Z(2 * (size_t)steps), X(steps), v(steps), W1(steps), W2(steps);
    for (long p = 0; p < paths; ++p) {
        {
            std::mt19937 g(seeder());
            std::normal_distribution<double> d(0.0, 1.0);
            for (int k = 0; k < steps; ++k) {
                Z[2 * k] = d(g);
                Z[2 * k + 1] = d(g);
            }
        }
        orc_fractional_gaussian(phi.data(), Z.data(), (size_t)steps, H, eta, X.data());
        orc_forward_variance(X.data(), (size_t)steps, dt, xi, H, eta, v.data());
        {
            std::mt19937 g(seeder());
            std::normal_distribution<double> d(0.0, 1.0);
            for (int k = 0; k < steps; ++k) W1[k] = d(g);
        }
        {
            std::mt19937 g(seeder());
            std::normal_distribution<double> d(0.0, 1.0);
            for (int k = 0; k < steps; ++k) W2[k] = d(g);
        }
        orc_step_prices(S0, r, dt, rho, v.data(), W1.data(), W2.data(), (size_t)steps,
                        out + (size_t)p * (steps + 1));
        if (intvar_out) {
            double s = 0.0;
            for (int k = 0; k < steps; ++k) s += v[k] * dt;
            intvar_out[p] = s;
        }
    }
    return 0;
}

// History-driven form of the above: the class-level entry point (:312-331).
int orc_generate_paths_mt_hist(const double* hist, size_t n, int steps, long paths, uint64_t seed,
                               double* out) {
    double p[5];
    if (orc_estimate_params(hist, n, p)) return 1;
    return orc_generate_paths_mt(p[4], 0.04, p[0], p[1], p[2], p[3], steps, paths, seed, out, nullptr);
}

// CPU baseline ("kind": "port"): the reference-faithful generator parallelised the way the
// reference's driver does it (src/core/PredictionGen.cpp:542-546) -- independent generator calls
// under omp parallel for schedule(dynamic), one chunk per call.  Returns threads used.
int orc_generate_paths_mt_omp(double S0, double r, double xi, double H, double eta, double rho,
                              int steps, long total_paths, int chunk, uint64_t seed,
                              double* sum_ST) {
    const long n_chunks = (total_paths + chunk - 1) / chunk;
    double acc = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : acc)
    for (long c = 0; c < n_chunks; ++c) {
        const long lo = c * (long)chunk;
        const long cnt = std::min<long>(chunk, total_paths - lo);
        std::vector<double> buf((size_t)cnt * (steps + 1));
        orc_generate_paths_mt(S0, r, xi, H, eta, rho, steps, cnt, seed + 0x9E3779B97F4A7C15ull * (uint64_t)(c + 1),
                              buf.data(), nullptr);
        for (long p = 0; p < cnt; ++p) acc += buf[(size_t)p * (steps + 1) + steps];
    }
    *sum_ST = acc;
    return threads;
}

// ---- Philox mode: mirrors the device algorithm draw-for-draw ----------------------------------
void orc_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
    philox4x32_10(ctr4, key2, out4);
}

void orc_normal_quad(uint64_t seed, uint64_t path, uint32_t block, uint32_t stream, double* z4) {
    normal_quad(seed, path, block, stream, z4);
}

// GBM = the stepping loop of RoughVolatility.cpp:354-364 with v == sigma^2.  One normal per step,
// step n = element n&3 of Philox block n>>2
// (the reference's rho-mix of two independent normals is itself N(0,1)).
// out is step-major: out[j*ld + p], j = 0..steps, p = 0..n_paths-1 (global id path_begin + p).
int orc_paths_gbm(uint64_t seed, double S0, double r, double sigma, double dt, int steps,
                  uint64_t path_begin, long n_paths, double* out, size_t ld) {
    if (steps < 1 || n_paths < 0) return 1;
    const double drift = (r - 0.5 * sigma * sigma) * dt;
    const double vol = sigma * std::sqrt(dt);
    for (long p = 0; p < n_paths; ++p) {
        double S = S0;
        out[p] = S;
        double z[4] = {0, 0, 0, 0};
        for (int n = 0; n < steps; ++n) {
            if ((n & 3) == 0) normal_quad(seed, path_begin + p, (uint32_t)(n >> 2), STREAM_PRICE, z);
            S = S * std::exp(std::fma(vol, z[n & 3], drift));
            out[(size_t)(n + 1) * ld + p] = S;
        }
    }
    return 0;
}

// Spectral amplitudes of the reference's X (RoughVolatility.cpp:264-292).  With P_k = |phi_k|^2 for k < steps
// (0 for steps <= k < Mz), a_k = eta*sqrt(2H)/Mz * sqrt((P_k + P_{(Mz-k) mod Mz}) / 2) and
// Y_k = a_k (g_k + i h_k), g, h ~ iid N(0,1),
//   x_n = sum_{k<Mz} Y_k e^{+2 pi i k n / Mz}
// is a complex stationary circular Gaussian sequence with
//   Cov(Re x_n, Re x_{n+d}) = Cov(Im x_n, Im x_{n+d}) = sum_k a_k^2 cos(2 pi k d / Mz)
//                           = (2H eta^2/Mz^2) sum_{k<steps} |phi_k|^2 cos(2 pi k d/Mz),
// exactly the covariance of the reference's X, and -- because a_k is symmetric in
// k <-> Mz-k -- Cov(Re x_n, Im x_{n+d}) = sum_k a_k^2 sin(2 pi k d/Mz) = 0 for every lag: the real and imaginary
// parts are two INDEPENDENT copies of X.  One transform therefore serves a PAIR of paths:
// Re x -> path 2q, Im x -> path 2q + 1.  amp gets Mz entries; comp_n = -0.5 eta^2 (n dt)^(2H) (:305).  Returns Mz.
size_t orc_rbergomi_spectrum(double H, double eta, double dt, int steps, double* amp, double* comp) {
    std::vector<double> lam(steps + 1);
    orc_lambda(steps, H, dt, lam.data());
    std::vector<double> phi(2 * next_pow2((size_t)steps + 1));
    orc_phi(lam.data(), (size_t)steps + 1, phi.data());
    const size_t M = next_pow2((size_t)steps);
    std::vector<double> P(M, 0.0);
    for (size_t k = 0; k < (size_t)steps && k < M; ++k) P[k] = phi[2 * k] * phi[2 * k] + phi[2 * k + 1] * phi[2 * k + 1];
    const double scale = eta * std::sqrt(2.0 * H) / (double)M;
    for (size_t k = 0; k < M; ++k) amp[k] = scale * std::sqrt(0.5 * (P[k] + P[(M - k) % M]));
    for (int n = 0; n < steps; ++n) comp[n] = -0.5 * eta * eta * std::pow(n * dt, 2 * H);
    return M;
}

// rBergomi paths, device algorithm (DESIGN.md "rBergomi kernel").  Step-major output as above.
// Volatility driver of the pair q = (path id) >> 1: Philox stream 1 with the PAIR id in the path field;
// block b holds (g_{2b}, h_{2b}, g_{2b+1}, h_{2b+1}).  Price driver per path: stream 0 as in GBM.
// If X_out != nullptr it receives X[p*steps + n] (for covariance tests).  path_begin must be even.
int orc_paths_rbergomi(uint64_t seed, double S0, double r, double xi, double H, double eta, double rho,
                       double dt, int steps, uint64_t path_begin, long n_paths, double* out, size_t ld,
                       double* X_out) {
    (void)rho;  // inert in the reference (W1, W2 independent of Z); kept for interface parity
    if (steps < 1 || n_paths < 0 || (path_begin & 1)) return 1;
    const size_t M = next_pow2((size_t)steps);
    std::vector<double> amp(M), comp(steps), yr(M + 2), yi(M + 2), ct(M), st(M);
    orc_rbergomi_spectrum(H, eta, dt, steps, amp.data(), comp.data());
    for (size_t q = 0; q < M; ++q) {
        ct[q] = std::cos(2.0 * M_PI * (double)q / (double)M);
        st[q] = std::sin(2.0 * M_PI * (double)q / (double)M);
    }
    const double sqdt = std::sqrt(dt);
    std::vector<double> XA(steps), XB(steps);
    for (long p0 = 0; p0 < n_paths; p0 += 2) {
        const uint64_t pair = (path_begin + (uint64_t)p0) >> 1;
        for (size_t k = 0; k < M; k += 2) {
            double z[4];
            normal_quad(seed, pair, (uint32_t)(k >> 1), STREAM_VOL, z);
            yr[k] = amp[k] * z[0];
            yi[k] = amp[k] * z[1];
            if (k + 1 < M) {
                yr[k + 1] = amp[k + 1] * z[2];
                yi[k + 1] = amp[k + 1] * z[3];
            }
        }
        for (int n = 0; n < steps; ++n) {
            double re = 0.0, im = 0.0;
            for (size_t k = 0; k < M; ++k) {
                const size_t qi = (k * (size_t)n) & (M - 1);
                re += yr[k] * ct[qi] - yi[k] * st[qi];
                im += yr[k] * st[qi] + yi[k] * ct[qi];
            }
            XA[n] = re;
            XB[n] = im;
        }
        for (int h = 0; h < 2 && p0 + h < n_paths; ++h) {
            const long p = p0 + h;
            const uint64_t id = path_begin + (uint64_t)p;
            const std::vector<double>& X = h ? XB : XA;
            double S = S0;
            out[p] = S;
            double z[4] = {0, 0, 0, 0};
            for (int n = 0; n < steps; ++n) {
                if (X_out) X_out[(size_t)p * steps + n] = X[n];
                const double v = xi * std::exp(X[n] + comp[n]);
                if ((n & 3) == 0) normal_quad(seed, id, (uint32_t)(n >> 2), STREAM_PRICE, z);
                const double drift = (r - 0.5 * v) * dt;
                const double sd = std::sqrt(std::max(0.0, v)) * sqdt;
                S = S * std::exp(std::fma(sd, z[n & 3], drift));
                out[(size_t)(n + 1) * ld + p] = S;
            }
        }
    }
    return 0;
}

// ---- pricing -----------------------------------------------------------------------------------
// European price = e^{-rT} mean(Payoff(S_T)) (include/core/common.h:8-14 at the last column);
// the reference has no European pricer, the discounted mean and its std-err are what the build adds.
// Generic strides: element (path p, step j) at paths[p*path_stride + j*step_stride].
int orc_price_european(const double* paths, size_t path_stride, size_t step_stride, long n_paths,
                       int steps, double K, double r, double T, int is_call, double* mean,
                       double* stderr_out) {
    if (n_paths < 1) return 1;
    double s = 0.0, s2 = 0.0;
    for (long p = 0; p < n_paths; ++p) {
        const double pay = payoff_of(is_call != 0, paths[(size_t)p * path_stride + (size_t)steps * step_stride], K);
        s += pay;
        s2 += pay * pay;
    }
    const double disc = std::exp(-r * T);
    const double m = s / n_paths;
    const double var = n_paths > 1 ? std::max(0.0, (s2 - n_paths * m * m) / (n_paths - 1)) : 0.0;
    *mean = disc * m;
    *stderr_out = disc * std::sqrt(var / n_paths);
    return 0;
}

// src/models/LSMPricer.cpp:19-102, statement for statement, except that only the two live columns
// of `Values` are kept (SURVEY.md section 3.3 note b).  Returns 1 for empty input (the reference
// throws "LSM::PredictOptionPrice: Empty pricePaths." at :28-30).
// If v0_out != nullptr it receives Values[:,0] (n_paths doubles) for std-err computations.
int orc_lsm_price(const double* paths, size_t path_stride, size_t step_stride, long n_paths, int n_cols,
                  double r, double K, double maturity, double dt, int is_call, int poly_order,
                  double* price, double* v0_out) {
    if (n_paths < 1 || n_cols < 1) return 1;
    if (poly_order < 0 || poly_order > 15) return 2;
    const bool call = is_call != 0;
    const long N = n_paths;
    const int M = n_cols;
    auto S = [&](long i, int j) { return paths[(size_t)i * path_stride + (size_t)j * step_stride]; };
    std::vector<double> V(N), Vn(N);
    for (long i = 0; i < N; ++i) V[i] = payoff_of(call, S(i, M - 1), K);  // :37-40

    const int nb = poly_order + 1;
    std::vector<long> itm;
    itm.reserve(N);
    for (int j = M - 2; j >= 0; --j) {  // :42
        const double this_time = j * dt;
        if (this_time > maturity) {  // :43-49
            for (long i = 0; i < N; ++i) V[i] = V[i] * std::exp(-r * dt);
            continue;
        }
        std::fill(Vn.begin(), Vn.end(), 0.0);  // Values[.][j] starts at 0 (:35)
        itm.clear();
        for (long i = 0; i < N; ++i)  // :51-58
            if (payoff_of(call, S(i, j), K) > 1e-14) itm.push_back(i);
        if (!itm.empty()) {  // :60-87
            const size_t rows = itm.size();
            std::vector<double> A(rows * nb), b(rows);
            for (size_t k = 0; k < rows; ++k) {
                const long i = itm[k];
                b[k] = V[i] * std::exp(-r * dt);  // :69-70
                double pw = 1.0;                  // PolynomialBasis :9-17
                const double s = S(i, j);
                for (int q = 0; q < nb; ++q) {
                    A[(size_t)q * rows + k] = pw;
                    pw = pw * s;
                }
            }
            double c[16];
            std::vector<double> Awork = A;
            minnorm_lstsq(Awork, rows, nb, b, c);  // :76
            for (size_t k = 0; k < rows; ++k) {   // :78-86
                const long i = itm[k];
                const double s = S(i, j);
                const double immediate = payoff_of(call, s, K);
                double pw = 1.0, cont = 0.0;
                for (int q = 0; q < nb; ++q) {
                    cont += pw * c[q];
                    pw = pw * s;
                }
                Vn[i] = std::max(immediate, cont);
            }
        }
        for (long i = 0; i < N; ++i)  // :89-94
            if (payoff_of(call, S(i, j), K) < 1e-14) Vn[i] = V[i] * std::exp(-r * dt);
        V.swap(Vn);
    }
    double sum = 0.0;  // :97-101
    for (long i = 0; i < N; ++i) sum += V[i];
    *price = sum / N;
    if (v0_out) std::memcpy(v0_out, V.data(), sizeof(double) * N);
    return 0;
}

// src/models/AsymptoticAnalysisPricer.cpp:8-36 -- short-time exercise boundary K +/- 0.5 sigma sqrt(eps ln(1/eps)),
// eps = T - t, with the carry correction for eps < 0.01; K itself when eps < 1e-10.
static double asym_boundary(bool call, double t, double T, double K, double r, double D, double sigma) {
    const double eps = T - t;
    if (eps < 1e-10) return K;
    const double c0 = 0.5 * sigma * std::sqrt(eps * std::log(1.0 / eps));
    double b = call ? K - c0 : K + c0;
    if (eps < 0.01) {
        if (call) b += 0.5 * (D - r) * eps;
        else b -= 0.5 * (r - D) * eps;
    }
    return b;
}

// AsymptoticAnalysis::PredictOptionPrice (src/models/AsymptoticAnalysisPricer.cpp:38-113): per path the
// best discounted payoff over the dates (t <= maturity) at which S is beyond the boundary; mean over
// paths.  Returns 0 (price may be 0.0 for empty input, :47-49) or 1 when sigma <= 0 (the reference
// throws "AsymptoticAnalysis: Volatility must be positive.", :50-52).
int orc_asymptotic_price(const double* paths, size_t path_stride, size_t step_stride, long n_paths, int n_cols,
                         double r, double K, double maturity, double dt, int is_call, double sigma, double dividend,
                         double* price) {
    *price = 0.0;
    if (n_paths < 1 || n_cols < 1) return 0;
    if (sigma <= 0.0) return 1;
    const bool call = is_call != 0;
    double sum = 0.0;
    long valid = 0;
    for (long i = 0; i < n_paths; ++i) {
        double best = 0.0;
        for (int j = 0; j < n_cols; ++j) {
            const double t = j * dt;
            if (t > maturity) break;
            const double S = paths[(size_t)i * path_stride + (size_t)j * step_stride];
            if (std::isnan(S) || std::isinf(S)) continue;
            const double b = asym_boundary(call, t, maturity, K, r, dividend, sigma);
            const bool in = call ? (S > b) : (S < b);
            if (in) {
                const double pay = payoff_of(call, S, K);
                if (std::isnan(pay) || std::isinf(pay)) continue;
                const double d = std::exp(-r * t) * pay;
                if (d > best) best = d;
            }
        }
        if (!std::isnan(best) && !std::isinf(best)) {
            sum += best;
            ++valid;
        }
    }
    *price = valid > 0 ? sum / valid : 0.0;
    return 0;
}

// MartingaleOptimization::PredictOptionPrice (src/models/MartingaleOptimizationPricer.cpp:21-189), statement
// for statement: maxIterations x { DoIteration (:69-124), UpdateMartingale (:126-176 + offset :178-183) }.
// PARITY UNPINNED at the Eigen boundary (:166), like LSM: the least squares is our Jacobi-SVD min-norm solve.
// Returns 0; 1 for empty input; 2 for maxIterations <= 0 (the reference throws in both cases).
int orc_martingale_price(const double* paths, size_t path_stride, size_t step_stride, long n_paths, int n_cols,
                         double r, double K, double maturity, double dt, int is_call, int poly_order,
                         int max_iterations, double* price, double* lower, double* upper) {
    if (n_paths < 1 || n_cols < 1) return 1;
    if (max_iterations <= 0) return 2;
    if (poly_order < 0 || poly_order > 15) return 3;
    const bool call = is_call != 0;
    const long N = n_paths;
    const int M = n_cols, nb = poly_order + 1;
    auto S = [&](long i, int j) { return paths[(size_t)i * path_stride + (size_t)j * step_stride]; };
    auto dfac = [&](int j) {  // PathDiscountFactor, header :46-51
        double t = j * dt;
        if (t > maturity) t = maturity;
        return std::exp(-r * t);
    };
    std::vector<double> coef(nb, 0.0);
    double offset = 0.0;
    auto evalM = [&](double s) {  // :178-187
        double val = 0.0, power = 1.0;
        for (int k = 0; k < nb; ++k) {
            val += coef[k] * power;
            power *= s;
        }
        return val;
    };
    std::vector<int> stop(N, 0);
    double fin_lo = 0.0, fin_up = 0.0;
    for (int iter = 1; iter <= max_iterations; ++iter) {
        double sum_p = 0.0;
        for (long i = 0; i < N; ++i) {  // :77-98
            double best = 0.0;
            int idx = 0;
            for (int j = 0; j < M; ++j) {
                if (j * dt > maturity) break;
                const double d = payoff_of(call, S(i, j), K) * dfac(j);
                if (d > best) {
                    best = d;
                    idx = j;
                }
            }
            stop[i] = idx;
            sum_p += best;
        }
        double sum_d = 0.0;
        for (long i = 0; i < N; ++i) {  // :100-121
            double best = 0.0;
            for (int j = 0; j < M; ++j) {
                if (j * dt > maturity) break;
                const double s = S(i, j);
                const double cand = payoff_of(call, s, K) * dfac(j) - (evalM(s) - offset);
                if (cand > best) best = cand;
            }
            sum_d += best;
        }
        fin_lo = sum_p / N;
        fin_up = sum_d / N;
        // UpdateMartingale :126-176
        const size_t rows = 2 * (size_t)N;
        if ((long)rows >= nb) {
            std::vector<double> A(rows * nb), b(rows);
            for (long i = 0; i < N; ++i) {
                const int js = stop[i], jo = (js + M / 2) % M;
                const double xs[2] = {S(i, js), S(i, jo)};
                const double ys[2] = {0.5 * (payoff_of(call, xs[0], K) * dfac(js)), 0.2 * (payoff_of(call, xs[1], K) * dfac(jo))};
                for (int s2 = 0; s2 < 2; ++s2) {
                    const size_t row = 2 * (size_t)i + s2;
                    double pw = 1.0;
                    for (int q = 0; q < nb; ++q) {
                        A[(size_t)q * rows + row] = pw;
                        pw = pw * xs[s2];
                    }
                    b[row] = ys[s2];
                }
            }
            double c[16];
            minnorm_lstsq(A, rows, nb, b, c);
            for (int k = 0; k < nb; ++k) coef[k] = c[k];
            double sum0 = 0.0;
            for (long i = 0; i < N; ++i) sum0 += evalM(S(i, 0));
            offset = sum0 / N;
        }
    }
    *price = 0.5 * (fin_lo + fin_up);
    if (lower) *lower = fin_lo;
    if (upper) *upper = fin_up;
    return 0;
}

// BranchingProcesses::PredictOptionPrice (src/models/BranchingProcessPricer.cpp:12-134).
//   mode 0 ("mt"): statement for statement, the unseeded std::random_device seed (:83-84) replaced by `seed`;
//                  one generator, consumed in (path, date, branch) order like the reference's serial loop.
//   mode 1 ("philox"): the device algorithm -- suffix-maximum matrix F[j][p] = max_{k>=j, t_k<=T} e^{-r t_k} pay,
//                  continuation = mean_b F[t+1][rp_b], rp_b = mulhi(word, N) of Philox stream 2, counter
//                  (path, date_index * ceil(B/4) + b/4) -- draw for draw.
// out3 = {price, lower, upper}.  Returns 0; 1 empty paths; 2 no exercise times; 3 strike <= 0 (the reference
// throws "BranchingProcesses: Empty pricePaths." / "No exercise times." / "Strike must be positive.").
int orc_branching_price(const double* paths, size_t path_stride, size_t step_stride, long n_paths, int n_cols, double r,
                        double K, double maturity, double dt, int is_call, int num_branches, const int* ex, int n_ex,
                        uint64_t seed, uint64_t path_begin, int mode, double* out3) {
    if (n_paths < 1 || n_cols < 1) return 1;
    if (n_ex < 1) return 2;
    if (K <= 0.0) return 3;
    const bool call = is_call != 0;
    const long N = n_paths;
    const int M = n_cols;
    auto S = [&](long i, int j) { return paths[(size_t)i * path_stride + (size_t)j * step_stride]; };
    // lower bound :41-72
    double sum_lo = 0.0;
    for (long i = 0; i < N; ++i) {
        double best = 0.0;
        for (int e = 0; e < n_ex; ++e) {
            const double t = ex[e] * dt;
            if (t > maturity) break;
            const double d = std::exp(-r * t) * payoff_of(call, S(i, ex[e]), K);
            if (d > best) {
                best = d;
                break;
            }
        }
        sum_lo += best;
    }
    double sum_up = 0.0;
    if (mode == 0) {  // :74-134
        std::mt19937 gen((uint32_t)(seed ^ (seed >> 32)));
        std::uniform_int_distribution<> pathDist(0, (int)N - 1);
        for (long i = 0; i < N; ++i) {
            double best = 0.0;
            for (int e = 0; e < n_ex; ++e) {
                const int tI = ex[e];
                const double t = tI * dt;
                if (t > maturity) break;
                const double now = std::exp(-r * t) * payoff_of(call, S(i, tI), K);
                double cont = 0.0;
                if (tI < ex[n_ex - 1]) {
                    double sumF = 0.0;
                    for (int b = 0; b < num_branches; ++b) {
                        const int rp = pathDist(gen);
                        double bf = 0.0;
                        for (int k = tI + 1; k < M; ++k) {
                            const double tk = k * dt;
                            if (tk > maturity) break;
                            const double d = std::exp(-r * (tk - t)) * payoff_of(call, S(rp, k), K);
                            if (d > bf) bf = d;
                        }
                        sumF += bf;
                    }
                    cont = (sumF / num_branches) * std::exp(-r * t);
                }
                const double better = std::max(now, cont);
                if (better > best) best = better;
            }
            sum_up += best;
        }
    } else {
        std::vector<double> disc(M), F((size_t)M * N);
        int n_dates = 0;
        for (int j = 0; j < M; ++j) {
            const double t = j * dt;
            if (!(t > maturity) && n_dates == j) n_dates = j + 1;
            disc[j] = std::exp(-r * t);
        }
        for (long p = 0; p < N; ++p) {
            double run = 0.0;
            for (int j = M - 1; j >= 0; --j) {
                if (j < n_dates) {
                    const double d = disc[j] * payoff_of(call, S(p, j), K);
                    if (d > run) run = d;
                }
                F[(size_t)j * N + p] = run;
            }
        }
        const int quads = (num_branches + 3) >> 2;
        const double inv_b = num_branches > 0 ? 1.0 / (double)num_branches : 0.0;
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        for (long p = 0; p < N; ++p) {
            const uint64_t id = path_begin + (uint64_t)p;
            double upper = 0.0;
            int e_idx = 0;
            for (int e = 0; e < n_ex; ++e, ++e_idx) {
                const int tI = ex[e];
                if (tI * dt > maturity) break;
                const double now = disc[tI] * payoff_of(call, S(p, tI), K);
                double better = now;
                // (tI + 1 < M: for the last column the reference's `k` loop, :110, is empty -> continuation 0)
                if (tI < ex[n_ex - 1] && tI + 1 < M && num_branches > 0) {
                    double sum = 0.0;
                    for (int q = 0; q < quads; ++q) {
                        const uint32_t ctr[4] = {(uint32_t)id, (uint32_t)(id >> 32), (uint32_t)(e_idx * quads + q), 2u};
                        uint32_t w[4];
                        philox4x32_10(ctr, key, w);
                        for (int s2 = 0; s2 < 4; ++s2)
                            if (4 * q + s2 < num_branches) {
                                const uint32_t rp = (uint32_t)(((uint64_t)w[s2] * (uint64_t)(uint32_t)N) >> 32);
                                sum += F[(size_t)(tI + 1) * N + rp];
                            }
                    }
                    const double cont = sum * inv_b;
                    if (cont > better) better = cont;
                }
                if (better > upper) upper = better;
            }
            sum_up += upper;
        }
    }
    out3[1] = sum_lo / N;
    out3[2] = sum_up / N;
    out3[0] = 0.5 * (out3[1] + out3[2]);
    return 0;
}

// compute20DayVolAndMomentum, /root/reference/src/core/PredictionGen.cpp:313-347 (two of the driver's six feature columns;
// its first value is also the sigma handed to AsymptoticAnalysis, :706).  out2 = {twenty_day_vol, twenty_day_momentum}.
void orc_row_features(const double* hist, size_t n, double* out2) {
    out2[0] = 0.0;
    out2[1] = 0.0;
    if (n < 21) return;                                   // :315-317
    std::vector<double> slice(hist + (n - 21), hist + n);  // :319
    std::vector<double> logRets;                           // :320-333
    for (int i = 0; i < 20; ++i) {
        const double p0 = slice[i], p1 = slice[i + 1];
        if (p0 <= 0.0 || p1 <= 0.0) {
            logRets.push_back(0.0);
        } else {
            double lr = std::log(p1 / p0);
            if (!std::isfinite(lr)) lr = 0.0;
            logRets.push_back(lr);
        }
    }
    double sum = 0.0, sum2 = 0.0;                          // :334-338
    for (double lr : logRets) {
        sum += lr;
        sum2 += lr * lr;
    }
    const double mean = sum / 20.0;                        // :339-341
    double var = (sum2 / 20.0) - (mean * mean);
    if (var < 0.0) var = 0.0;
    out2[0] = std::sqrt(var) * std::sqrt(252.0);           // :343
    out2[1] = sum;                                         // :344
}

// CPU baseline of bench.py's LSM (C3) and MartingaleOptimization rows, "kind": "port": this restatement over a resident
// sample cut into the driver's rows of `chunk` paths (250, src/core/PredictionGen.cpp:719), one call per row under
// `omp parallel for schedule(dynamic)` (:542-546).  which = 0: orc_lsm_price (LSMPricer.cpp:19-102), 1:
// orc_martingale_price (MartingaleOptimizationPricer.cpp:21-189, 5 iterations).  paths: row-major [n_total][m].
int orc_pricer_chunks_omp(int which, const double* row_major, long n_total, int m, int chunk, double r, double K,
                          double maturity, double dt, int is_call, int poly_order, double* seconds, double* checksum) {
    const long n_chunks = n_total / chunk;
    double acc = 0.0;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
    const double t0 = omp_get_wtime();
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : acc)
    for (long c = 0; c < n_chunks; ++c) {
        const double* p = row_major + (size_t)c * chunk * m;
        double v = 0.0, lo = 0.0, up = 0.0;
        const int rc = which == 0 ? orc_lsm_price(p, (size_t)m, 1, chunk, m, r, K, maturity, dt, is_call, poly_order, &v, nullptr)
                                  : orc_martingale_price(p, (size_t)m, 1, chunk, m, r, K, maturity, dt, is_call, poly_order, 5, &v, &lo, &up);
        if (rc == 0) acc += v;
    }
#ifdef _OPENMP
    *seconds = omp_get_wtime() - t0;
#else
    *seconds = 0.0;
#endif
    *checksum = acc;
    return threads;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
