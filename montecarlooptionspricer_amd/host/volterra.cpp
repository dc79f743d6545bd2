// Host precompute for the rBergomi kernels (row a3 of SURVEY.md section 8: rbergomiLambda / rbergomiPhi /
// fft / nextPowerOfTwo, once per call): the spectral amplitudes a_k and the compensator table.
//
// What the reference does per path (/root/reference/src/models/RoughVolatility.cpp:264-292):
//   A_k = phi_k * Z_k (k < steps, Z complex standard normal), zero-pad to Mz = nextpow2(steps),
//   X = sqrt(2H)*eta * Re( FFT^-(A) / Mz ),   phi = FFT^+(lambda zero-padded to nextpow2(steps+1)),
//   lambda_i = 0.5 * (i*dt)^(2H)                                             (:212-236, :337-343).
// X is therefore a zero-mean stationary *circular* Gaussian sequence of period Mz with
//   Cov(X_n, X_{n+d}) = (2H eta^2 / Mz^2) * sum_{k<steps} |phi_k|^2 cos(2 pi k d / Mz).
// A Gaussian vector is fixed by its covariance.  With P_k = |phi_k|^2 (k < steps, else 0) and the SYMMETRIC
// amplitudes a_k = eta*sqrt(2H)/Mz * sqrt((P_k + P_{Mz-k})/2), the complex sequence
//   x_n = sum_{k<Mz} a_k (g_k + i h_k) e^{+2 pi i k n/Mz},  g, h ~ iid N(0,1),
// has Re x and Im x each with exactly that covariance and, by the symmetry of a_k, zero cross-covariance at
// every lag: one transform yields two independent copies of the reference's X (rbergomi_device.hpp).
// The M_phi != M_z quirk at power-of-two step counts (:217 vs :270) is inherited through phi.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <vector>

#include "../csrc/mcg_internal.hpp"

namespace {

using cd = std::complex<double>;
constexpr double kPi = 3.14159265358979323846;

size_t pow2_at_least(size_t n) {
    size_t p = 1;
    while (p < n) p *= 2;
    return p;
}

// |phi_k|^2 for k < count, phi_k = sum_{n < N} lam_n e^{+2 pi i k n / N}, lam REAL, N a power of two >= 2 (the
// reference's phi = FFT+ of the zero-padded lambda, RoughVolatility.cpp:212-225; only its modulus is consumed here).
// A real sequence needs half a transform: z_m = lam_{2m} + i lam_{2m+1} goes through ONE complex transform of length
// N/2 -- Stockham's autosort form (decimation in frequency, ping-pong between two buffers, no bit-reversal pass), the
// roots of unity read from a table filled by cos / sin of exact rational angles -- and the even / odd halves are
// separated by the Hermitian symmetry of a real signal's spectrum: with Z = DFT(z),
//   E_k = (Z_k + conj Z_{N/2-k}) / 2,  O_k = (Z_k - conj Z_{N/2-k}) / (2i),  phi_k = E_k + w^k O_k  (k <= N/2),
//   phi_{N-k} = conj phi_k.
std::vector<double> real_power_spectrum(const std::vector<double>& lam, size_t N, size_t count) {
    const size_t L = N / 2;
    std::vector<cd> root(L > 0 ? L : 1);  // w^j = e^{+2 pi i j / N}, j < N/2
    for (size_t j = 0; j < L; ++j) {
        const double ang = 2.0 * kPi * (double)j / (double)N;
        root[j] = cd(std::cos(ang), std::sin(ang));
    }
    std::vector<cd> x(L), y(L);
    for (size_t m = 0; m < L; ++m) x[m] = cd(2 * m < lam.size() ? lam[2 * m] : 0.0, 2 * m + 1 < lam.size() ? lam[2 * m + 1] : 0.0);
    for (size_t n = L, s = 1; n > 1; n /= 2, s *= 2) {  // n: length of the sub-transforms left, s: how many of them interleave
        const size_t m = n / 2;
        for (size_t p = 0; p < m; ++p) {
            const cd w = root[p * (N / n)];  // e^{+2 pi i p / n}
            for (size_t q = 0; q < s; ++q) {
                const cd a = x[q + s * p], b = x[q + s * (p + m)];
                y[q + s * 2 * p] = a + b;
                y[q + s * (2 * p + 1)] = (a - b) * w;
            }
        }
        x.swap(y);
    }
    std::vector<double> P(count, 0.0);
    for (size_t k = 0; k < count; ++k) {
        const size_t kk = k <= L ? k : N - k;  // |phi_{N-k}| = |phi_k|
        cd phi;
        if (kk == L) {
            phi = cd(x[0].real() - x[0].imag(), 0.0);  // E_0 - O_0
        } else {
            const cd zk = x[kk], zc = std::conj(x[(L - kk) % L]);
            const cd E = 0.5 * (zk + zc), O = cd(0.0, -0.5) * (zk - zc);
            phi = E + root[kk] * O;
        }
        P[k] = std::norm(phi);
    }
    return P;
}

}  // namespace

namespace mcg {

int host_rbergomi_spectrum(double H, double eta, double dt, int n_steps, std::vector<double>& amp,
                           std::vector<double>& comp) {
    if (n_steps < 1) return fail(MCG_ERR_INVALID, "n_steps must be >= 1");
    const size_t steps = (size_t)n_steps;
    // lambda on the grid t_i = i*dt, i = 0..steps; P_k = |phi_k|^2 with M_phi = nextpow2(steps+1) points, k < steps
    std::vector<double> lam(steps + 1);
    for (size_t i = 0; i <= steps; ++i) lam[i] = 0.5 * (std::pow(i * dt, 2 * H));
    const size_t M = pow2_at_least(steps);  // M_z
    std::vector<double> P = real_power_spectrum(lam, pow2_at_least(steps + 1), std::min(steps, M));
    P.resize(M, 0.0);
    const double scale = eta * std::sqrt(2.0 * H) / (double)M;
    amp.resize(M);
    for (size_t k = 0; k < M; ++k) amp[k] = scale * std::sqrt(0.5 * (P[k] + P[(M - k) % M]));
    // compensator of RoughVolatility.cpp:305 on t_n = n*dt
    comp.resize(steps);
    for (size_t n = 0; n < steps; ++n) comp[n] = -0.5 * eta * eta * std::pow(n * dt, 2 * H);
    return MCG_OK;
}

}  // namespace mcg
