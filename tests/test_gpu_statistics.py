"""Statistical parity at the stated bar, over MANY seeds (run with -m gpu on an MI355X).

The north-star statistic is |price - ref| <= 2 Monte Carlo standard errors.  One seed either sits inside 2 sigma or not
(BASELINE's own seed puts the C1 put at z = 2.6: one of the twenty draws that do), so the tests here price K independent
Philox streams each and judge what K draws of a standard normal must look like:

  * the POOLED estimate (all K streams together -- the statistic of the K-fold sample) within 2 of its standard errors;
  * the z-scores' mean square inside the chi-square band of K degrees of freedom, their largest inside the band of the
    maximum of K normals, both at a false-alarm level of 1e-3 computed by scipy -- no hand-picked literal;
  * for Longstaff-Schwartz prices the error bar is the SPREAD of the price across the K streams (the per-path standard
    error mcg_price_lsm returns ignores the noise of the shared regression coefficients), against reference samples of
    the same path count, so the finite-sample bias of the regression is the same on both sides.
"""
import json
import math
import os

import numpy as np
import pytest
from scipy import stats

import montecarlooptionspricer_amd as mc
from oracle.binding import STAT_NAMES, Oracle, path_stats

pytestmark = pytest.mark.gpu

SEED, DT = 20251031, 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)
ALPHA = 1e-3   # false-alarm level of every band below


@pytest.fixture(scope="module")
def eng():
    e = mc.PathEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def bs_price(S0, K, r, sigma, T, call=True):
    d1 = (math.log(S0 / K) + (r + 0.5 * sigma * sigma) * T) / (sigma * math.sqrt(T))
    d2 = d1 - sigma * math.sqrt(T)
    N = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))  # noqa: E731
    return S0 * N(d1) - K * math.exp(-r * T) * N(d2) if call else K * math.exp(-r * T) * N(-d2) - S0 * N(-d1)


def normal_sample_bands(k):
    """What K independent N(0,1) z-scores must satisfy at level ALPHA: (lo, hi) of their mean square, bound of their
    largest absolute value."""
    lo, hi = stats.chi2.ppf(ALPHA / 2, k) / k, stats.chi2.ppf(1 - ALPHA / 2, k) / k
    zmax = stats.norm.isf(ALPHA / (2 * k))
    return lo, hi, zmax


def check_z_scores(z, what):
    z = np.asarray(z)
    k = len(z)
    lo, hi, zmax = normal_sample_bands(k)
    pooled = z.sum() / math.sqrt(k)     # the z-score of the K-fold sample when the streams have equal standard errors
    assert abs(pooled) <= 2.0, (what, "pooled", pooled, z)
    assert lo <= (z ** 2).mean() <= hi, (what, "mean square", (z ** 2).mean(), lo, hi)
    assert np.abs(z).max() <= zmax, (what, "max", np.abs(z).max(), zmax)


def test_gbm_european_prices_over_seeds_vs_black_scholes(eng):
    """C1's shape (100k x 252, S0 = K = 100, r = 0.04, sigma = 0.2) on 24 Philox streams, call (fused sums) and put
    (pricing pass over the stored matrix) against Black-Scholes."""
    k, n = 24, 100_000
    ref_c, ref_p = bs_price(100.0, 100.0, 0.04, 0.2, 1.0), bs_price(100.0, 100.0, 0.04, 0.2, 1.0, call=False)
    zc, zp, zf, pc, sc = [], [], [], [], []
    for s in range(k):
        P = eng.gbm(SEED + 1000 * s, 100.0, 0.04, 0.2, DT, 252, n, payoff=(100.0, True))
        c, cse = eng.price_european(P, 100.0, 0.04, 1.0, True)
        p, pse = eng.price_european(P, 100.0, 0.04, 1.0, False)
        host_T = P.to_host_step_major()[-1]
        P.free()
        zc.append((c - ref_c) / cse)
        zp.append((p - ref_p) / pse)
        zf.append((host_T.mean() - 100.0 * math.exp(0.04)) / (host_T.std(ddof=1) / math.sqrt(n)))   # martingale: E[S_T] = S0 e^{rT}
        pc.append(c)
        sc.append(cse)
    check_z_scores(zc, "call")
    check_z_scores(zp, "put")
    check_z_scores(zf, "forward")
    # and the north-star statistic itself on the pooled 2.4M paths
    pooled, pooled_se = float(np.mean(pc)), math.sqrt(float(np.sum(np.square(sc)))) / k
    assert abs(pooled - ref_c) <= 2.0 * pooled_se, (pooled, pooled_se, ref_c)


@pytest.mark.parametrize("steps", [252, 512])
def test_rbergomi_over_seeds_vs_compiled_reference_sample(eng, steps):
    """C4 / C5 parameters (H = 0.1, eta = 1.9) on 16 Philox streams of 100k paths against the committed sample of the
    COMPILED REFERENCE (tests/golden/rough_regime_reference.json): the seven statistics a price matrix shows -- E[S_T],
    call, put, realised variance, clustering of squared returns at lags 1, 8, 64.  Each statistic's error bar is the
    spread of its 16 per-stream means combined with the reference sample's own standard error; the prices (call, put)
    must sit within 2 of it -- the north-star bar --, the seven together within the Bonferroni bound of seven
    comparisons at level ALPHA (they are comparisons against ONE reference sample, hence not independent draws)."""
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rough_regime_reference.json")))
    p, fix = fx["params"], fx["samples"][str(steps)]
    fm, fse = np.array(fix["mean"])[:7], np.array(fix["std_err"])[:7]
    k, n = 16, 100_000
    means = []
    for s in range(k):
        P = eng.rbergomi(SEED + 77 * s, p["S0"], p["r"], p["xi"], p["H"], p["eta"], p["rho"], DT, steps, n)
        a, _, cnt = path_stats(P.to_host_step_major(), p["strike"])
        P.free()
        means.append(a / cnt)
    means = np.array(means)
    m, se = means.mean(axis=0), means.std(axis=0, ddof=1) / math.sqrt(k)
    z = (m - fm) / np.hypot(se, fse)
    names = dict(zip(STAT_NAMES, np.round(z, 2)))
    assert abs(z[1]) <= 2.0 and abs(z[2]) <= 2.0, (steps, names)          # call, put: |price - ref| <= 2 std-errs
    assert np.abs(z).max() <= stats.norm.isf(ALPHA / (2 * 7)), (steps, names)


def spread(prices):
    a = np.asarray(prices, dtype=float)
    return float(a.mean()), float(a.std(ddof=1)) / math.sqrt(len(a))


def test_lsm_prices_over_seeds_vs_independent_reference_samples(eng, orc):
    """SURVEY 8(d)'s parity statistic for the LSM configs, z = |price_gpu - price_ref| / std-err <= 2, with BOTH error
    bars taken from spreads: sixteen GPU streams against eight samples of the CPU-restated reference LSM on independent
    draws (another generator), the same path count per sample on both sides.  C3-shaped: GBM, 50 dates, 1e5 paths per
    sample.  C5-shaped: rBergomi H = 0.1, 64 dates, 2e4 paths per sample, the reference side through the
    reference-faithful "mt" generator (fresh mt19937 streams, one complex FFT per path)."""
    ref, ref_se = spread([orc.lsm_price(orc.paths_gbm(1000 + q, 100.0, 0.04, 0.2, 0.02, 50, 0, 100_000), 0.04, 100.0, 1.0, 0.02,
                                        False, 2) for q in range(8)])
    got = []
    for s in range(16):
        P = eng.gbm(SEED + 31 * s, 100.0, 0.04, 0.2, 0.02, 50, 100_000)
        got.append(eng.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)[0])
        P.free()
    am, ase = spread(got)
    z = abs(am - ref) / math.hypot(ase, ref_se)
    assert z <= 2.0, ("C3-shaped", am, ase, ref, ref_se, z)

    steps, T = 64, 64 * DT
    ref, ref_se = spread([orc.lsm_price(orc.generate_paths_mt(RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], steps,
                                                              20_000, 2024 + q), RB["r"], 100.0, T, DT, False, 2, step_major=False)
                          for q in range(8)])
    got = []
    for s in range(16):
        R = eng.rbergomi(SEED + 31 * s, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, 20_000)
        got.append(eng.price_lsm(R, RB["r"], 100.0, T, DT, False, 2)[0])
        R.free()
    am, ase = spread(got)
    z = abs(am - ref) / math.hypot(ase, ref_se)
    assert z <= 2.0, ("C5-shaped", am, ase, ref, ref_se, z)
