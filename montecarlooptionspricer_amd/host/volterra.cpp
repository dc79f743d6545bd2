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

// Decimation-in-time radix-2 transform, sign = +1 or -1 in the exponent, unnormalised;
// twiddles advance by repeated multiplication as in RoughVolatility.cpp:183-196.
void dit_fft(std::vector<cd>& a, int sign) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        while (j & bit) {
            j ^= bit;
            bit >>= 1;
        }
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t span = 2; span <= n; span *= 2) {
        const double ang = 2 * kPi / span * (sign < 0 ? -1 : 1);
        const cd step(std::cos(ang), std::sin(ang));
        for (size_t base = 0; base < n; base += span) {
            cd w(1.0, 0.0);
            for (size_t j = 0; j < span / 2; ++j) {
                const cd u = a[base + j], v = a[base + j + span / 2] * w;
                a[base + j] = u + v;
                a[base + j + span / 2] = u - v;
                w *= step;
            }
        }
    }
}

}  // namespace

namespace mcg {

int host_rbergomi_spectrum(double H, double eta, double dt, int n_steps, std::vector<double>& amp,
                           std::vector<double>& comp) {
    if (n_steps < 1) return fail(MCG_ERR_INVALID, "n_steps must be >= 1");
    const size_t steps = (size_t)n_steps;
    // lambda on the grid t_i = i*dt, i = 0..steps, then phi (M_phi = nextpow2(steps+1))
    std::vector<cd> phi(pow2_at_least(steps + 1), cd(0.0, 0.0));
    for (size_t i = 0; i <= steps; ++i) phi[i] = cd(0.5 * (std::pow(i * dt, 2 * H)), 0.0);
    dit_fft(phi, +1);

    const size_t M = pow2_at_least(steps);  // M_z
    std::vector<double> P(M, 0.0);
    for (size_t k = 0; k < steps && k < M; ++k) P[k] = std::norm(phi[k]);
    const double scale = eta * std::sqrt(2.0 * H) / (double)M;
    amp.resize(M);
    for (size_t k = 0; k < M; ++k) amp[k] = scale * std::sqrt(0.5 * (P[k] + P[(M - k) % M]));
    // compensator of RoughVolatility.cpp:305 on t_n = n*dt
    comp.resize(steps);
    for (size_t n = 0; n < steps; ++n) comp[n] = -0.5 * eta * eta * std::pow(n * dt, 2 * H);
    return MCG_OK;
}

}  // namespace mcg
