#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libmcref.so).

TEST INFRASTRUCTURE ONLY.  Run in the dev container (where /root/reference exists):

    make -C oracle && python oracle/gen_golden.py
    make -C oracle && python oracle/gen_golden.py --eigen     # only where an Eigen3 exists: the LSM / MartingaleOptimization pin

The reference has no tests and no golden vectors of its own, so every
deterministic function on the hot path is pinned by calling the reference's own compiled code on
fixed inputs and storing inputs + outputs bit-exactly (float64 .npz).  The fixtures are data only:
no reference source text is stored.  tests/test_oracle_golden.py then requires our restatement
(oracle/mcg_oracle.cpp) to reproduce them.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.binding import Reference, synthetic_history  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def capture_eigen() -> None:
    """--eigen: tests/golden/{lsm,martingale}.npz from the reference's LSMPricer.cpp / MartingaleOptimizationPricer.cpp compiled in
    place against an Eigen3 (oracle/_ref/libmcref_eigen.so; `make -C oracle` builds it where $(EIGEN_INC)/Eigen/Dense exists --
    not in the image this repo was developed in).  Every case of oracle/eigen_fixtures.py is stored as inputs + the reference's
    price + the tolerance it will be held to; `eigen_version` records what the prices were captured with (the reference does not
    pin one: CMakeLists.txt:17)."""
    from oracle.binding import ReferenceEigen, have_ref_eigen
    from oracle.eigen_fixtures import lsm_cases, martingale_cases
    if not have_ref_eigen():
        raise SystemExit("oracle/_ref/libmcref_eigen.so is missing: no Eigen3 was found when `make -C oracle` ran "
                         "(set EIGEN_INC=<directory holding Eigen/Dense>); nothing captured, LSM / MartingaleOptimization stay unpinned")
    os.makedirs(OUT, exist_ok=True)
    ref = ReferenceEigen()
    ver = np.array(ref.eigen_version())
    out = {"eigen_version": ver, "names": np.array(sorted(lsm_cases()))}
    for name, c in lsm_cases().items():
        out[f"{name}_paths"] = c["paths"]
        out[f"{name}_args"] = np.array([c["r"], c["K"], c["maturity"], c["dt"], c["is_call"], c["poly"], c["tol"]])
        out[f"{name}_price"] = np.array(ref.lsm_price(c["paths"], c["r"], c["K"], c["maturity"], c["dt"], bool(c["is_call"]), c["poly"]))
    np.savez(os.path.join(OUT, "lsm.npz"), **out)
    out = {"eigen_version": ver, "names": np.array(sorted(martingale_cases()))}
    for name, c in martingale_cases().items():
        out[f"{name}_paths"] = c["paths"]
        out[f"{name}_args"] = np.array([c["r"], c["K"], c["maturity"], c["dt"], c["is_call"], c["poly"], c["iters"], c["tol"]])
        out[f"{name}_price"] = np.array(ref.martingale_price(c["paths"], c["r"], c["K"], c["maturity"], c["dt"], bool(c["is_call"]),
                                                             c["poly"], c["iters"]))
    np.savez(os.path.join(OUT, "martingale.npz"), **out)
    print("Eigen", ".".join(map(str, ver)), "-> tests/golden/lsm.npz, tests/golden/martingale.npz")


def main() -> None:
    if "--eigen" in sys.argv[1:]:
        capture_eigen()
        return
    os.makedirs(OUT, exist_ok=True)
    ref = Reference()

    # (1) estimators: RoughVolatility.cpp:126-169 on three synthetic histories.
    est = {}
    for n in (2, 3, 64, 1001):
        h = synthetic_history(n, seed=42, s0=100.0, mu=0.05, sigma=0.2)
        rets, p = ref.estimators(h)
        est[f"hist_{n}"] = h
        est[f"rets_{n}"] = rets
        est[f"params_{n}"] = np.array([p["xi"], p["H"], p["eta"], p["rho"], p["S0"]])
    # a history with a strong leverage pattern so rho < 0 comes out of the estimator itself
    h = synthetic_history(400, seed=7, s0=50.0, mu=0.0, sigma=0.35)
    h[1::7] *= 0.97
    rets, p = ref.estimators(h)
    est["hist_lev"] = h
    est["rets_lev"] = rets
    est["params_lev"] = np.array([p["xi"], p["H"], p["eta"], p["rho"], p["S0"]])
    np.savez(os.path.join(OUT, "estimators.npz"), **est)

    # (2)-(4) lambda, phi, fractionalGaussian, forwardVariance: RoughVolatility.cpp:212-309.
    spec = {}
    shapes = [(252, 0.1), (512, 0.1), (7, 0.57), (252, 0.57), (50, 0.3), (1, 0.25), (64, 0.05)]
    for steps, H in shapes:
        tag = f"s{steps}_H{str(H).replace('.', 'p')}"
        lam = ref.lam(steps, H)
        phi = ref.phi(lam, H)
        k = np.arange(steps, dtype=np.float64)
        Z = np.cos(k) + 1j * np.sin(2.0 * k)           # hand-made, deterministic "Gaussians"
        eta, xi = 1.9, 0.04
        X = ref.fractional_gaussian(phi, Z, H, eta)
        v = ref.forward_variance(X, xi, H, eta)
        spec[f"{tag}_lam"] = lam
        spec[f"{tag}_phi"] = phi
        spec[f"{tag}_Z"] = Z
        spec[f"{tag}_X"] = X
        spec[f"{tag}_v"] = v
    spec["shapes"] = np.array(shapes)
    spec["eta_xi"] = np.array([1.9, 0.04])
    np.savez(os.path.join(OUT, "spectral.npz"), **spec)

    # (5) the FFT itself, both directions (RoughVolatility.cpp:171-202), plus nextPowerOfTwo.
    fft = {}
    rs = np.random.RandomState(123)
    for n in (1, 2, 8, 64, 256):
        z = rs.standard_normal(n) + 1j * rs.standard_normal(n)
        fft[f"in_{n}"] = z
        fft[f"fwd_{n}"] = ref.fft(z, 1)
        fft[f"inv_{n}"] = ref.fft(z, -1)
    ns = np.array([0, 1, 2, 3, 4, 5, 7, 8, 9, 252, 253, 256, 257, 512, 513, 1000])
    fft["np2_in"] = ns
    fft["np2_out"] = np.array([ref.next_pow2(int(n)) for n in ns])
    np.savez(os.path.join(OUT, "fft.npz"), **fft)

    # (6) PayoffFunction table (include/core/common.h:8-14).
    s = np.array([0.0, 50.0, 99.999999, 100.0, 100.000001, 150.0, 1e-300, 1e300])
    k = np.array([100.0, 100.0, 100.0, 100.0, 100.0, 100.0, 1.0, 1.0])
    pay = {"S": s, "K": k,
           "call": np.array([ref.payoff(True, a, b) for a, b in zip(s, k)]),
           "put": np.array([ref.payoff(False, a, b) for a, b in zip(s, k)])}
    np.savez(os.path.join(OUT, "payoff.npz"), **pay)
    # (7) AsymptoticAnalysis::PredictOptionPrice (src/models/AsymptoticAnalysisPricer.cpp:38-113):
    # deterministic given the path matrix.  Inputs are stored with the outputs.
    rs = np.random.RandomState(11)
    base = 100.0 * np.exp(np.cumsum(0.02 * rs.standard_normal((300, 41)), axis=1))
    base[:, 0] = 100.0
    dirty = base.copy()
    dirty[5, 7] = np.nan
    dirty[9, 30] = np.inf
    asy = {"paths": base, "paths_dirty": dirty}
    cases = []
    for which, is_call, maturity, dt, sigma, div, K in [
            (0, 0, 40 / 252.0, 1 / 252.0, 0.2, 0.08, 100.0), (0, 1, 40 / 252.0, 1 / 252.0, 0.2, 0.08, 100.0),
            (0, 0, 15 / 252.0, 1 / 252.0, 0.35, 0.0, 105.0), (0, 1, 0.5, 0.02, 0.15, 0.03, 95.0),
            (1, 0, 40 / 252.0, 1 / 252.0, 0.2, 0.08, 100.0), (1, 1, 2.0, 0.05, 0.5, 0.01, 90.0),
            (0, 0, 1e-12, 1 / 252.0, 0.2, 0.08, 100.0)]:
        m = dirty if which else base
        cases.append([which, is_call, maturity, dt, sigma, div, K, 0.04,
                      ref.asymptotic_price(m, 0.04, K, maturity, dt, bool(is_call), sigma, div)])
    asy["cases"] = np.array(cases)
    np.savez(os.path.join(OUT, "asymptotic.npz"), **asy)
    # (8) BranchingProcesses lower bound (src/models/BranchingProcessPricer.cpp:41-72) -- deterministic; the upper
    # bound resamples with an unseeded generator and is compared statistically in the tests instead.
    br = []
    ex_all = np.arange(40, dtype=np.int32)                  # the driver's 0..steps-1 (PredictionGen.cpp:780-783)
    for is_call, maturity, K, ex in [(0, 40 / 252.0, 100.0, ex_all), (1, 40 / 252.0, 100.0, ex_all),
                                     (0, 20.5 / 252.0, 103.0, ex_all), (0, 40 / 252.0, 100.0, ex_all[::5])]:
        _, lo, _ = ref.branching_price(base, 0.04, K, maturity, 1 / 252.0, bool(is_call), 10, ex)
        br.append([is_call, maturity, K, len(ex), lo])
    np.savez(os.path.join(OUT, "branching.npz"), cases=np.array(br), ex_all=ex_all)
    # (9) compute20DayVolAndMomentum (src/core/PredictionGen.cpp:313-347): the driver's two remaining feature columns.
    feat = {}
    rs = np.random.RandomState(2024)
    hs = {"short20": synthetic_history(20, seed=3), "exact21": synthetic_history(21, seed=4), "long1001": synthetic_history(1001, seed=42),
          "flat": np.full(40, 123.25), "noisy": 50.0 * np.exp(np.cumsum(0.05 * rs.standard_normal(300)))}
    z = synthetic_history(60, seed=9)
    z[-5] = 0.0             # a non-positive price inside the window: both returns around it count as 0
    hs["zero_inside"] = z
    n = synthetic_history(60, seed=10)
    n[-3] = -4.0
    hs["negative_inside"] = n
    big = synthetic_history(30, seed=11)
    big[-2] = 1e308         # a return that overflows to inf is replaced by 0
    big[-1] = 1e-308
    hs["overflow"] = big
    hs["empty"] = np.zeros(0)
    for k, h in hs.items():
        feat[f"hist_{k}"] = h
        feat[f"out_{k}"] = np.array(ref.row_features(h))
    np.savez(os.path.join(OUT, "features.npz"), **feat)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  ", f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
