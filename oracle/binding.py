"""ctypes bindings for the parity oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (montecarlooptionspricer_amd) never does.

Two libraries:
  * libmcgoracle.so  -- our CPU restatement (oracle/mcg_oracle.cpp), always buildable.
  * _ref/libmcref.so -- the reference's own RoughVolatility.cpp compiled in place by
                        oracle/Makefile (present when built in the dev container; travels to the
                        GPU box as a prebuilt file, the reference sources do not).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libmcgoracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmcref.so")
REF_DRIVER_SO = os.path.join(HERE, "_ref", "libmcref_driver.so")
REF_EIGEN_SO = os.path.join(HERE, "_ref", "libmcref_eigen.so")   # only where an Eigen3 exists (oracle/Makefile: EIGEN_INC)

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)


def _p(a):
    return a.ctypes.data_as(_dp)


def build(quiet: bool = True) -> None:
    """(Re)build the oracle; builds _ref too when /root/reference is present."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class Oracle:
    """Our restatement (mcg_oracle.cpp)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        self.L = L
        L.orc_log_returns.argtypes = [_dp, C.c_size_t, _dp]
        L.orc_estimate_params.argtypes = [_dp, C.c_size_t, _dp]
        L.orc_estimate_params.restype = C.c_int
        L.orc_next_pow2.argtypes = [C.c_size_t]
        L.orc_next_pow2.restype = C.c_size_t
        L.orc_fft.argtypes = [_dp, C.c_size_t, C.c_int]
        L.orc_lambda.argtypes = [C.c_int, C.c_double, C.c_double, _dp]
        L.orc_phi.argtypes = [_dp, C.c_size_t, _dp]
        L.orc_phi.restype = C.c_size_t
        L.orc_fractional_gaussian.argtypes = [_dp, _dp, C.c_size_t, C.c_double, C.c_double, _dp]
        L.orc_forward_variance.argtypes = [_dp, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                           C.c_double, _dp]
        L.orc_step_prices.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp,
                                      C.c_size_t, _dp]
        L.orc_payoff.argtypes = [C.c_int, C.c_double, C.c_double]
        L.orc_payoff.restype = C.c_double
        L.orc_generate_paths_mt.argtypes = [C.c_double] * 6 + [C.c_int, C.c_long, C.c_uint64, _dp, _dp]
        L.orc_generate_paths_mt.restype = C.c_int
        L.orc_generate_paths_mt_hist.argtypes = [_dp, C.c_size_t, C.c_int, C.c_long, C.c_uint64, _dp]
        L.orc_generate_paths_mt_hist.restype = C.c_int
        L.orc_generate_paths_mt_omp.argtypes = [C.c_double] * 6 + [C.c_int, C.c_long, C.c_int,
                                                                  C.c_uint64, _dp]
        L.orc_generate_paths_mt_omp.restype = C.c_int
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_normal_quad.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _dp]
        L.orc_paths_gbm.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                    C.c_uint64, C.c_long, _dp, C.c_size_t]
        L.orc_paths_gbm.restype = C.c_int
        L.orc_rbergomi_spectrum.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.orc_rbergomi_spectrum.restype = C.c_size_t
        L.orc_paths_rbergomi.argtypes = [C.c_uint64] + [C.c_double] * 7 + [C.c_int, C.c_uint64, C.c_long,
                                                                          _dp, C.c_size_t, _dp]
        L.orc_paths_rbergomi.restype = C.c_int
        L.orc_price_european.argtypes = [_dp, C.c_size_t, C.c_size_t, C.c_long, C.c_int, C.c_double,
                                         C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.orc_price_european.restype = C.c_int
        L.orc_lsm_price.argtypes = [_dp, C.c_size_t, C.c_size_t, C.c_long, C.c_int, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp]
        L.orc_lsm_price.restype = C.c_int
        L.orc_asymptotic_price.argtypes = [_dp, C.c_size_t, C.c_size_t, C.c_long, C.c_int] + [C.c_double] * 4 + \
            [C.c_int, C.c_double, C.c_double, _dp]
        L.orc_asymptotic_price.restype = C.c_int
        L.orc_martingale_price.argtypes = [_dp, C.c_size_t, C.c_size_t, C.c_long, C.c_int] + [C.c_double] * 4 + \
            [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_martingale_price.restype = C.c_int
        L.orc_branching_price.argtypes = [_dp, C.c_size_t, C.c_size_t, C.c_long, C.c_int] + [C.c_double] * 4 + \
            [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_uint64, C.c_int, _dp]
        L.orc_branching_price.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        L.orc_row_features.argtypes = [_dp, C.c_size_t, _dp]

    def row_features(self, hist):
        """(twenty_day_vol, twenty_day_momentum): PredictionGen.cpp:313-347 restated."""
        h = np.ascontiguousarray(hist, dtype=np.float64)
        out = np.zeros(2)
        self.L.orc_row_features(_p(h), len(h), _p(out))
        return out[0], out[1]

    # -- estimators / spectral ------------------------------------------------------------------
    def log_returns(self, prices):
        p = np.ascontiguousarray(prices, dtype=np.float64)
        out = np.empty(max(len(p) - 1, 0))
        self.L.orc_log_returns(_p(p), len(p), _p(out))
        return out

    def estimate_params(self, hist):
        h = np.ascontiguousarray(hist, dtype=np.float64)
        out = np.empty(5)
        rc = self.L.orc_estimate_params(_p(h), len(h), _p(out))
        if rc:
            raise RuntimeError("Historical prices vector too small.")
        return dict(xi=out[0], H=out[1], eta=out[2], rho=out[3], S0=out[4])

    def next_pow2(self, n):
        return int(self.L.orc_next_pow2(n))

    def fft(self, z, inv):
        z = np.asarray(z, dtype=np.complex128)
        buf = np.ascontiguousarray(np.stack([z.real, z.imag], axis=1))
        self.L.orc_fft(_p(buf), len(z), inv)
        return buf[:, 0] + 1j * buf[:, 1]

    def lam(self, steps, H, dt=1.0 / 252.0):
        out = np.empty(steps + 1)
        self.L.orc_lambda(steps, H, dt, _p(out))
        return out

    def phi(self, lam):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        M = self.next_pow2(len(lam))
        buf = np.empty((M, 2))
        self.L.orc_phi(_p(lam), len(lam), _p(buf))
        return buf[:, 0] + 1j * buf[:, 1]

    def fractional_gaussian(self, phi, Z, H, eta):
        phi = np.asarray(phi, dtype=np.complex128)
        Z = np.asarray(Z, dtype=np.complex128)
        pb = np.ascontiguousarray(np.stack([phi.real, phi.imag], axis=1))
        zb = np.ascontiguousarray(np.stack([Z.real, Z.imag], axis=1))
        X = np.empty(len(Z))
        self.L.orc_fractional_gaussian(_p(pb), _p(zb), len(Z), H, eta, _p(X))
        return X

    def forward_variance(self, X, xi, H, eta, dt=1.0 / 252.0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        v = np.empty(len(X))
        self.L.orc_forward_variance(_p(X), len(X), dt, xi, H, eta, _p(v))
        return v

    def step_prices(self, S0, r, dt, rho, v, W1, W2):
        v = np.ascontiguousarray(v, dtype=np.float64)
        W1 = np.ascontiguousarray(W1, dtype=np.float64)
        W2 = np.ascontiguousarray(W2, dtype=np.float64)
        S = np.empty(len(v) + 1)
        self.L.orc_step_prices(S0, r, dt, rho, _p(v), _p(W1), _p(W2), len(v), _p(S))
        return S

    def payoff(self, is_call, s, k):
        return float(self.L.orc_payoff(int(bool(is_call)), s, k))

    # -- generators -----------------------------------------------------------------------------
    def generate_paths_mt(self, S0, r, xi, H, eta, rho, steps, paths, seed, want_intvar=False):
        """Reference-faithful generator with explicit seed.  Returns path-major [paths][steps+1]."""
        out = np.empty((paths, steps + 1))
        iv = np.empty(paths) if want_intvar else None
        rc = self.L.orc_generate_paths_mt(S0, r, xi, H, eta, rho, steps, paths, seed, _p(out),
                                          _p(iv) if want_intvar else None)
        if rc:
            raise RuntimeError("orc_generate_paths_mt failed")
        return (out, iv) if want_intvar else out

    def generate_paths_mt_hist(self, hist, steps, paths, seed):
        h = np.ascontiguousarray(hist, dtype=np.float64)
        out = np.empty((paths, steps + 1))
        rc = self.L.orc_generate_paths_mt_hist(_p(h), len(h), steps, paths, seed, _p(out))
        if rc:
            raise RuntimeError("Historical prices vector too small.")
        return out

    def generate_paths_mt_omp(self, S0, r, xi, H, eta, rho, steps, total_paths, chunk, seed):
        s = C.c_double(0.0)
        th = self.L.orc_generate_paths_mt_omp(S0, r, xi, H, eta, rho, steps, total_paths, chunk, seed,
                                              C.byref(s))
        return th, s.value

    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.L.orc_philox4x32_10(c, k, o)
        return [int(x) for x in o]

    def normal_quad(self, seed, path, block, stream):
        z = np.empty(4)
        self.L.orc_normal_quad(seed, path, block, stream, _p(z))
        return z

    def paths_gbm(self, seed, S0, r, sigma, dt, steps, path_begin, n_paths):
        """Philox-mode GBM; returns step-major [steps+1][n_paths]."""
        out = np.empty((steps + 1, n_paths))
        rc = self.L.orc_paths_gbm(seed, S0, r, sigma, dt, steps, path_begin, n_paths, _p(out), n_paths)
        if rc:
            raise RuntimeError("orc_paths_gbm failed")
        return out

    def rbergomi_spectrum(self, H, eta, dt, steps):
        """(amp[Mz], comp[steps]): spectral amplitudes a_k and the compensator."""
        M = self.next_pow2(steps)
        amp = np.empty(M)
        comp = np.empty(steps)
        self.L.orc_rbergomi_spectrum(H, eta, dt, steps, _p(amp), _p(comp))
        return amp, comp

    def paths_rbergomi(self, seed, S0, r, xi, H, eta, rho, dt, steps, path_begin, n_paths, want_X=False):
        out = np.empty((steps + 1, n_paths))
        X = np.empty((n_paths, steps)) if want_X else None
        rc = self.L.orc_paths_rbergomi(seed, S0, r, xi, H, eta, rho, dt, steps, path_begin, n_paths,
                                       _p(out), n_paths, _p(X) if want_X else None)
        if rc:
            raise RuntimeError("orc_paths_rbergomi failed")
        return (out, X) if want_X else out

    # -- pricing --------------------------------------------------------------------------------
    def price_european(self, paths, K, r, T, is_call, step_major=True):
        a = np.ascontiguousarray(paths, dtype=np.float64)
        if step_major:
            n_cols, n_paths = a.shape
            ps, ss = 1, n_paths
        else:
            n_paths, n_cols = a.shape
            ps, ss = n_cols, 1
        m, se = C.c_double(), C.c_double()
        rc = self.L.orc_price_european(_p(a), ps, ss, n_paths, n_cols - 1, K, r, T, int(bool(is_call)),
                                       C.byref(m), C.byref(se))
        if rc:
            raise RuntimeError("orc_price_european failed")
        return m.value, se.value

    def lsm_price(self, paths, r, K, maturity, dt, is_call, poly_order, step_major=True, want_v0=False):
        a = np.ascontiguousarray(paths, dtype=np.float64)
        if a.size == 0:
            raise RuntimeError("LSM::PredictOptionPrice: Empty pricePaths.")
        if step_major:
            n_cols, n_paths = a.shape
            ps, ss = 1, n_paths
        else:
            n_paths, n_cols = a.shape
            ps, ss = n_cols, 1
        price = C.c_double()
        v0 = np.empty(n_paths) if want_v0 else None
        rc = self.L.orc_lsm_price(_p(a), ps, ss, n_paths, n_cols, r, K, maturity, dt, int(bool(is_call)),
                                  poly_order, C.byref(price), _p(v0) if want_v0 else None)
        if rc == 1:
            raise RuntimeError("LSM::PredictOptionPrice: Empty pricePaths.")
        if rc:
            raise RuntimeError("orc_lsm_price failed rc=%d" % rc)
        return (price.value, v0) if want_v0 else price.value

    def asymptotic_price(self, paths, r, K, maturity, dt, is_call, sigma, dividend, step_major=True):
        a = np.ascontiguousarray(paths, dtype=np.float64)
        if a.ndim != 2 or a.size == 0:
            return 0.0
        if step_major:
            n_cols, n_paths = a.shape
            ps, ss = 1, n_paths
        else:
            n_paths, n_cols = a.shape
            ps, ss = n_cols, 1
        price = C.c_double()
        rc = self.L.orc_asymptotic_price(_p(a), ps, ss, n_paths, n_cols, r, K, maturity, dt, int(bool(is_call)),
                                         sigma, dividend, C.byref(price))
        if rc:
            raise RuntimeError("AsymptoticAnalysis: Volatility must be positive.")
        return price.value

    def martingale_price(self, paths, r, K, maturity, dt, is_call, poly_order, max_iterations=5, step_major=True):
        """(price, lower, upper)."""
        a = np.ascontiguousarray(paths, dtype=np.float64)
        if a.ndim != 2 or a.size == 0:
            raise RuntimeError("MartingaleOptimization: Empty pricePaths.")
        if step_major:
            n_cols, n_paths = a.shape
            ps, ss = 1, n_paths
        else:
            n_paths, n_cols = a.shape
            ps, ss = n_cols, 1
        p, lo, up = C.c_double(), C.c_double(), C.c_double()
        rc = self.L.orc_martingale_price(_p(a), ps, ss, n_paths, n_cols, r, K, maturity, dt, int(bool(is_call)),
                                         poly_order, max_iterations, C.byref(p), C.byref(lo), C.byref(up))
        if rc == 1:
            raise RuntimeError("MartingaleOptimization: Empty pricePaths.")
        if rc == 2:
            raise RuntimeError("MartingaleOptimization: maxIterations must be positive.")
        if rc:
            raise RuntimeError("orc_martingale_price failed rc=%d" % rc)
        return p.value, lo.value, up.value

    def branching_price(self, paths, r, K, maturity, dt, is_call, num_branches, exercise_times, seed, mode="philox",
                        path_begin=0, step_major=True):
        """(price, lower, upper); mode "mt" = the reference's algorithm with an explicit seed, "philox" = device."""
        a = np.ascontiguousarray(paths, dtype=np.float64)
        if a.ndim != 2 or a.size == 0:
            raise RuntimeError("BranchingProcesses: Empty pricePaths.")
        if step_major:
            n_cols, n_paths = a.shape
            ps, ss = 1, n_paths
        else:
            n_paths, n_cols = a.shape
            ps, ss = n_cols, 1
        ex = np.ascontiguousarray(exercise_times, dtype=np.int32)
        out = np.empty(3)
        rc = self.L.orc_branching_price(_p(a), ps, ss, n_paths, n_cols, r, K, maturity, dt, int(bool(is_call)),
                                        num_branches, ex.ctypes.data_as(C.POINTER(C.c_int)), len(ex), seed, path_begin,
                                        0 if mode == "mt" else 1, _p(out))
        msgs = {1: "BranchingProcesses: Empty pricePaths.", 2: "BranchingProcesses: No exercise times.",
                3: "BranchingProcesses: Strike must be positive."}
        if rc:
            raise RuntimeError(msgs.get(rc, "orc_branching_price failed"))
        return tuple(out)

    def num_threads(self):
        return int(self.L.orc_num_threads())

    def pricer_chunks_omp(self, which, row_major, chunk, r, K, maturity, dt, is_call, poly_order=2):
        """CPU baseline ("port"): which = "lsm" | "martingale" over a resident [n][m] sample in the driver's rows of `chunk`
        paths under omp dynamic.  -> (threads, seconds, sum of prices)."""
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        self.L.orc_pricer_chunks_omp.argtypes = [C.c_int, _dp, C.c_long, C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, _dp, _dp]
        self.L.orc_pricer_chunks_omp.restype = C.c_int
        sec, chk = C.c_double(), C.c_double()
        th = self.L.orc_pricer_chunks_omp({"lsm": 0, "martingale": 1}[which], _p(a), a.shape[0], a.shape[1], chunk, r, K, maturity, dt,
                                          int(bool(is_call)), poly_order, C.byref(sec), C.byref(chk))
        return th, sec.value, chk.value


class Reference:
    """The reference's own compiled path engine (oracle/_ref/libmcref.so)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " missing (built only where /root/reference exists)")
        L = C.CDLL(REF_SO)
        self.L = L
        L.ref_estimators.argtypes = [_dp, C.c_size_t, _dp, _dp]
        L.ref_estimators.restype = C.c_int
        L.ref_next_pow2.argtypes = [C.c_size_t]
        L.ref_next_pow2.restype = C.c_size_t
        L.ref_fft.argtypes = [_dp, C.c_size_t, C.c_int]
        L.ref_lambda.argtypes = [C.c_int, C.c_double, C.c_double, _dp]
        L.ref_phi.argtypes = [_dp, C.c_size_t, C.c_double, _dp]
        L.ref_phi.restype = C.c_size_t
        L.ref_fractional_gaussian.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, C.c_double, C.c_double, _dp]
        L.ref_forward_variance.argtypes = [_dp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.ref_payoff.argtypes = [C.c_int, C.c_double, C.c_double]
        L.ref_payoff.restype = C.c_double
        L.ref_generate_paths.argtypes = [_dp, C.c_size_t, C.c_int, C.c_int, _dp, C.c_char_p, C.c_size_t]
        L.ref_generate_paths.restype = C.c_int
        L.ref_generate_paths_omp.argtypes = [_dp, C.c_size_t, C.c_int, C.c_long, C.c_int, _dp]
        L.ref_generate_paths_omp.restype = C.c_int
        if hasattr(L, "ref_generate_paths_omp_payoff"):
            L.ref_generate_paths_omp_payoff.argtypes = [_dp, C.c_size_t, C.c_int, C.c_long, C.c_int, C.c_double, C.c_int, _dp]
            L.ref_generate_paths_omp_payoff.restype = C.c_int
        if hasattr(L, "ref_branching_price"):
            L.ref_branching_price.argtypes = [_dp, C.c_long, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, C.POINTER(C.c_int),
                                                                                   C.c_int, _dp, C.c_char_p, C.c_size_t]
            L.ref_branching_price.restype = C.c_int
        if hasattr(L, "ref_asymptotic_price"):
            L.ref_asymptotic_price.argtypes = [_dp, C.c_long, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_double, C.c_double,
                                                                                    _dp, C.c_char_p, C.c_size_t]
            L.ref_asymptotic_price.restype = C.c_int

        if hasattr(L, "ref_explicit_stats_omp"):
            L.ref_explicit_stats_omp.argtypes = [C.c_double] * 6 + [C.c_int, C.c_long, C.c_double, _dp]
            L.ref_explicit_stats_omp.restype = C.c_int

    def explicit_stats(self, S0, r, xi, H, eta, rho, steps, total_paths, strike):
        """Rough-regime sample through the reference's own private members (ref_harness.cpp: ref_explicit_stats_omp).
        Returns (threads, sums[10], sums_of_squares[10]) of the per-path statistics STAT_NAMES."""
        out = np.zeros(20)
        th = self.L.ref_explicit_stats_omp(S0, r, xi, H, eta, rho, int(steps), int(total_paths), strike, _p(out))
        return th, out[:10].copy(), out[10:].copy()

    def estimators(self, hist):
        h = np.ascontiguousarray(hist, dtype=np.float64)
        rets = np.empty(max(len(h) - 1, 0))
        out = np.empty(5)
        rc = self.L.ref_estimators(_p(h), len(h), _p(rets), _p(out))
        if rc:
            raise RuntimeError("Historical prices vector too small.")
        return rets, dict(xi=out[0], H=out[1], eta=out[2], rho=out[3], S0=out[4])

    def next_pow2(self, n):
        return int(self.L.ref_next_pow2(n))

    def fft(self, z, inv):
        z = np.asarray(z, dtype=np.complex128)
        buf = np.ascontiguousarray(np.stack([z.real, z.imag], axis=1))
        self.L.ref_fft(_p(buf), len(z), inv)
        return buf[:, 0] + 1j * buf[:, 1]

    def lam(self, steps, H, dt=1.0 / 252.0):
        out = np.empty(steps + 1)
        self.L.ref_lambda(steps, H, dt, _p(out))
        return out

    def phi(self, lam, H):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        M = self.next_pow2(len(lam))
        buf = np.empty((M, 2))
        self.L.ref_phi(_p(lam), len(lam), H, _p(buf))
        return buf[:, 0] + 1j * buf[:, 1]

    def fractional_gaussian(self, phi, Z, H, eta):
        phi = np.asarray(phi, dtype=np.complex128)
        Z = np.asarray(Z, dtype=np.complex128)
        pb = np.ascontiguousarray(np.stack([phi.real, phi.imag], axis=1))
        zb = np.ascontiguousarray(np.stack([Z.real, Z.imag], axis=1))
        X = np.empty(len(Z))
        self.L.ref_fractional_gaussian(_p(pb), len(phi), _p(zb), len(Z), H, eta, _p(X))
        return X

    def forward_variance(self, X, xi, H, eta, dt=1.0 / 252.0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        v = np.empty(len(X))
        self.L.ref_forward_variance(_p(X), len(X), dt, xi, H, eta, _p(v))
        return v

    def payoff(self, is_call, s, k):
        return float(self.L.ref_payoff(int(bool(is_call)), s, k))

    def generate_paths(self, hist, steps, paths):
        h = np.ascontiguousarray(hist, dtype=np.float64)
        out = np.empty((paths, steps + 1))
        err = C.create_string_buffer(256)
        rc = self.L.ref_generate_paths(_p(h), len(h), steps, paths, _p(out), err, 256)
        if rc:
            raise RuntimeError(err.value.decode())
        return out

    def asymptotic_price(self, row_major, r, K, maturity, dt, is_call, sigma, dividend):
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        n, m = (a.shape if a.ndim == 2 else (0, 0))
        price = C.c_double()
        err = C.create_string_buffer(256)
        rc = self.L.ref_asymptotic_price(_p(a) if a.size else None, n, m, r, K, maturity, dt, int(bool(is_call)), sigma,
                                         dividend, C.byref(price), err, 256)
        if rc:
            raise RuntimeError(err.value.decode())
        return price.value

    def branching_price(self, row_major, r, K, maturity, dt, is_call, num_branches, exercise_times):
        """(price, lower, upper) from the compiled reference (upper bound unseeded)."""
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        n, m = (a.shape if a.ndim == 2 else (0, 0))
        ex = np.ascontiguousarray(exercise_times, dtype=np.int32)
        out = np.empty(3)
        err = C.create_string_buffer(256)
        rc = self.L.ref_branching_price(_p(a) if a.size else None, n, m, r, K, maturity, dt, int(bool(is_call)),
                                        num_branches, ex.ctypes.data_as(C.POINTER(C.c_int)), len(ex), _p(out), err, 256)
        if rc:
            raise RuntimeError(err.value.decode())
        return tuple(out)

    def row_features(self, hist):
        """compute20DayVolAndMomentum of the reference's driver TU, compiled in place (oracle/_ref/libmcref_driver.so)."""
        if not os.path.exists(REF_DRIVER_SO):
            raise FileNotFoundError(REF_DRIVER_SO + " missing (built only where /root/reference exists)")
        D = C.CDLL(REF_DRIVER_SO)
        D.ref_row_features.argtypes = [_dp, C.c_size_t, _dp]
        h = np.ascontiguousarray(hist, dtype=np.float64)
        out = np.zeros(2)
        D.ref_row_features(_p(h), len(h), _p(out))
        return out[0], out[1]

    def pricer_chunks_omp(self, which, row_major, chunk, r, K, maturity, dt, is_call, sigma=0.2, dividend=0.0, num_branches=10):
        """CPU baseline ("reference"): which = "asymptotic" | "branching" of the compiled reference over a resident [n][m]
        sample in the driver's rows of `chunk` paths under omp dynamic.  -> (threads, seconds, sum of prices)."""
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        f = self.L.ref_pricer_chunks_omp
        f.argtypes = [C.c_int, _dp, C.c_long, C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp]
        f.restype = C.c_int
        sec, chk = C.c_double(), C.c_double()
        th = f({"asymptotic": 0, "branching": 1}[which], _p(a), a.shape[0], a.shape[1], chunk, r, K, maturity, dt, int(bool(is_call)),
               sigma, dividend, num_branches, C.byref(sec), C.byref(chk))
        return th, sec.value, chk.value

    def driver_rows_omp(self, hist, steps, strikes, is_call, paths_per_row, sigma, dividend, oracle=None):
        """CPU baseline of whole driver rows (generation + four pricers per row, PredictionGen.cpp:719-791) under omp
        dynamic; LSM and MartingaleOptimization run through `oracle` (an Oracle: the restatement; the reference's need
        Eigen).  -> (threads, seconds, sums of the four prices)."""
        h = np.ascontiguousarray(hist, dtype=np.float64)
        st = np.ascontiguousarray(steps, dtype=np.int32)
        k = np.ascontiguousarray(strikes, dtype=np.float64)
        ic = np.ascontiguousarray(is_call, dtype=np.int32)
        f = self.L.ref_driver_rows_omp
        ip = C.POINTER(C.c_int)
        f.argtypes = [_dp, C.c_size_t, ip, _dp, ip, C.c_long, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, _dp, _dp]
        f.restype = C.c_int
        lsm = C.cast(oracle.L.orc_lsm_price, C.c_void_p) if oracle is not None else None
        mart = C.cast(oracle.L.orc_martingale_price, C.c_void_p) if oracle is not None else None
        sec, sums = C.c_double(), np.zeros(4)
        th = f(_p(h), len(h), st.ctypes.data_as(ip), _p(k), ic.ctypes.data_as(ip), len(st), paths_per_row, sigma, dividend, lsm, mart,
               C.byref(sec), _p(sums))
        return th, sec.value, sums

    def generate_paths_omp(self, hist, steps, total_paths, chunk):
        h = np.ascontiguousarray(hist, dtype=np.float64)
        s = C.c_double(0.0)
        th = self.L.ref_generate_paths_omp(_p(h), len(h), steps, total_paths, chunk, C.byref(s))
        return th, s.value


    def generate_paths_omp_payoff(self, hist, steps, total_paths, chunk, strike, is_call):
        """(threads, {sum S_T, sum payoff, sum payoff^2, n}) over the reference's own sample."""
        h = np.ascontiguousarray(hist, dtype=np.float64)
        out = np.zeros(4)
        th = self.L.ref_generate_paths_omp_payoff(_p(h), len(h), steps, total_paths, chunk, strike, int(is_call), _p(out))
        return th, out


STAT_NAMES = ("S_T", "call_payoff", "put_payoff", "realised_var", "sq_ret_lag1", "sq_ret_lag8", "sq_ret_lag64",
              "integrated_var", "mean_X2", "mean_X_lag1")


def have_ref_eigen() -> bool:
    return os.path.exists(REF_EIGEN_SO)


class ReferenceEigen:
    """The reference's LSMPricer.cpp and MartingaleOptimizationPricer.cpp compiled in place against an Eigen3
    (oracle/_ref/libmcref_eigen.so, oracle/ref_eigen_harness.cpp).  Exists only where the image has Eigen; this one does not."""

    def __init__(self):
        if not have_ref_eigen():
            raise FileNotFoundError(REF_EIGEN_SO + " missing (built only where $(EIGEN_INC)/Eigen/Dense exists)")
        L = C.CDLL(REF_EIGEN_SO)
        self.L = L
        L.ref_eigen_version.argtypes = [C.POINTER(C.c_int)]
        L.ref_lsm_price.argtypes = [_dp, C.c_long, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, _dp, C.c_char_p, C.c_size_t]
        L.ref_lsm_price.restype = C.c_int
        L.ref_martingale_price.argtypes = [_dp, C.c_long, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_int, _dp, C.c_char_p,
                                                                                      C.c_size_t]
        L.ref_martingale_price.restype = C.c_int

    def eigen_version(self):
        v = (C.c_int * 3)()
        self.L.ref_eigen_version(v)
        return tuple(v)

    def lsm_price(self, row_major, r, K, maturity, dt, is_call, poly_order):
        """LSM::PredictOptionPrice (LSMPricer.cpp:19-102) on [n_paths][n_steps + 1]."""
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        n, m = (a.shape if a.ndim == 2 else (0, 0))
        out, err = C.c_double(), C.create_string_buffer(256)
        if self.L.ref_lsm_price(_p(a), n, m, r, K, maturity, dt, int(bool(is_call)), int(poly_order), C.byref(out), err, 256):
            raise RuntimeError(err.value.decode())
        return out.value

    def martingale_price(self, row_major, r, K, maturity, dt, is_call, poly_order, max_iterations=5):
        """MartingaleOptimization::PredictOptionPrice (MartingaleOptimizationPricer.cpp:21-189)."""
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        n, m = (a.shape if a.ndim == 2 else (0, 0))
        out, err = C.c_double(), C.create_string_buffer(256)
        if self.L.ref_martingale_price(_p(a), n, m, r, K, maturity, dt, int(bool(is_call)), int(poly_order), int(max_iterations),
                                       C.byref(out), err, 256):
            raise RuntimeError(err.value.decode())
        return out.value


def path_stats(step_major: np.ndarray, strike: float):
    """The first seven statistics of ref_explicit_stats_omp (those that can be read off a price matrix), per path,
    from a step-major matrix [steps+1][paths]: returns (sums[7], sums_of_squares[7], n_paths)."""
    S = np.asarray(step_major, dtype=np.float64)
    steps = S.shape[0] - 1
    r2 = np.log(S[1:] / S[:-1]) ** 2
    st = [S[-1], np.maximum(0.0, S[-1] - strike), np.maximum(0.0, strike - S[-1]), r2.sum(axis=0)]
    for L in (1, 8, 64):
        st.append((r2[:-L] * r2[L:]).sum(axis=0) / (steps - L) if steps > L else np.zeros(S.shape[1]))
    st = np.array(st)
    return st.sum(axis=1), (st * st).sum(axis=1), S.shape[1]


def mean_and_se(sums, sums2, n):
    m = np.asarray(sums) / n
    var = np.maximum(0.0, (np.asarray(sums2) - n * m * m) / (n - 1))
    return m, np.sqrt(var / n)


def synthetic_history(n: int, seed: int = 42, s0: float = 100.0, mu: float = 0.05,
                      sigma: float = 0.2) -> np.ndarray:
    """Deterministic GBM price history for fixtures (numpy MT19937 via RandomState -- stable)."""
    rs = np.random.RandomState(seed)
    dt = 1.0 / 252.0
    z = rs.standard_normal(max(n - 1, 0))
    lr = (mu - 0.5 * sigma * sigma) * dt + sigma * np.sqrt(dt) * z
    return s0 * np.exp(np.concatenate([[0.0], np.cumsum(lr)]))
