"""The oracle (oracle/mcg_oracle.cpp) against the golden vectors captured from the COMPILED
reference (oracle/gen_golden.py -> tests/golden/*.npz).  Bar: bit-exact -- same compiler, same
libm, the restatement follows the reference operation for operation.

Reference lines: src/models/RoughVolatility.cpp:20-309, include/core/common.h:8-14.
"""
import os

import numpy as np
import pytest

from oracle.binding import Oracle, Reference, have_ref

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _same(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        return _same(a.real, b.real) and _same(a.imag, b.imag)
    ok = (a == b) | (np.isnan(a) & np.isnan(b))
    assert ok.all(), f"max abs diff {np.nanmax(np.abs(a - b))}"
    return True


@pytest.mark.parametrize("tag", ["3", "64", "1001", "lev"])
def test_estimators_bit_exact(orc, tag):
    d = np.load(os.path.join(G, "estimators.npz"))
    h = d[f"hist_{tag}"]
    _same(orc.log_returns(h), d[f"rets_{tag}"])
    p = orc.estimate_params(h)
    _same(np.array([p["xi"], p["H"], p["eta"], p["rho"], p["S0"]]), d[f"params_{tag}"])


def test_estimators_length_two(orc):
    """n=2: one return, variance 0 -> xi=0, H=0.5, eta=0, rho=NaN (0/0) -- reference behaviour."""
    d = np.load(os.path.join(G, "estimators.npz"))
    p = orc.estimate_params(d["hist_2"])
    _same(np.array([p["xi"], p["H"], p["eta"], p["rho"], p["S0"]]), d["params_2"])
    assert np.isnan(p["rho"]) and p["H"] == 0.5


def test_history_too_small(orc):
    """RoughVolatility.cpp:317-319 throws "Historical prices vector too small."."""
    with pytest.raises(RuntimeError, match="Historical prices vector too small."):
        orc.estimate_params(np.array([100.0]))
    with pytest.raises(RuntimeError, match="Historical prices vector too small."):
        orc.estimate_params(np.array([]))


def test_fft_both_directions(orc):
    d = np.load(os.path.join(G, "fft.npz"))
    for n in (1, 2, 8, 64, 256):
        _same(orc.fft(d[f"in_{n}"], 1), d[f"fwd_{n}"])
        _same(orc.fft(d[f"in_{n}"], -1), d[f"inv_{n}"])
    for n, m in zip(d["np2_in"], d["np2_out"]):
        assert orc.next_pow2(int(n)) == int(m)


def test_spectral_chain(orc):
    """lambda -> phi -> fractionalGaussian -> forwardVariance, incl. the M_phi != M_z quirk at 512."""
    d = np.load(os.path.join(G, "spectral.npz"))
    eta, xi = d["eta_xi"]
    for steps, H in d["shapes"]:
        steps = int(steps)
        tag = f"s{steps}_H{str(float(H)).replace('.', 'p')}"
        lam = orc.lam(steps, float(H))
        _same(lam, d[f"{tag}_lam"])
        phi = orc.phi(lam)
        _same(phi, d[f"{tag}_phi"])
        assert len(phi) == orc.next_pow2(steps + 1)
        X = orc.fractional_gaussian(phi, d[f"{tag}_Z"], float(H), float(eta))
        _same(X, d[f"{tag}_X"])
        v = orc.forward_variance(X, float(xi), float(H), float(eta))
        _same(v, d[f"{tag}_v"])
    # the quirk itself: 512 steps -> phi has 1024 bins, the inverse transform 512
    assert len(d["s512_H0p1_phi"]) == 1024 and orc.next_pow2(512) == 512


def test_payoff_table(orc):
    d = np.load(os.path.join(G, "payoff.npz"))
    for s, k, c, p in zip(d["S"], d["K"], d["call"], d["put"]):
        assert orc.payoff(True, s, k) == c
        assert orc.payoff(False, s, k) == p


@pytest.mark.skipif(not have_ref(), reason="compiled reference not present")
def test_live_reference_random_inputs(orc):
    """Beyond the committed vectors: random inputs through both, live (dev container / prebuilt)."""
    ref = Reference()
    rs = np.random.RandomState(5)
    for steps, H, eta in [(33, 0.2, 0.7), (128, 0.45, 2.5), (300, 0.07, 1.1)]:
        lam_r = ref.lam(steps, H)
        _same(orc.lam(steps, H), lam_r)
        phi_r = ref.phi(lam_r, H)
        _same(orc.phi(lam_r), phi_r)
        Z = rs.standard_normal(steps) + 1j * rs.standard_normal(steps)
        X_r = ref.fractional_gaussian(phi_r, Z, H, eta)
        _same(orc.fractional_gaussian(phi_r, Z, H, eta), X_r)
        _same(orc.forward_variance(X_r, 0.09, H, eta), ref.forward_variance(X_r, 0.09, H, eta))
    h = 80.0 * np.exp(np.cumsum(0.02 * rs.standard_normal(777)))
    rets, p = ref.estimators(h)
    q = orc.estimate_params(h)
    for k in ("xi", "H", "eta", "rho", "S0"):
        assert p[k] == q[k]


def test_asymptotic_pricer_bit_exact(orc):
    """AsymptoticAnalysis::PredictOptionPrice (src/models/AsymptoticAnalysisPricer.cpp:38-113) against the
    compiled reference on stored path matrices, incl. NaN/inf entries and a maturity shorter than the grid."""
    d = np.load(os.path.join(G, "asymptotic.npz"))
    for which, is_call, maturity, dt, sigma, div, K, r, want in d["cases"]:
        m = d["paths_dirty"] if which else d["paths"]
        got = orc.asymptotic_price(m, r, K, maturity, dt, bool(is_call), sigma, div, step_major=False)
        assert got == want, (got, want)
        # layout-independent
        assert orc.asymptotic_price(np.ascontiguousarray(m.T), r, K, maturity, dt, bool(is_call), sigma, div) == want
    assert orc.asymptotic_price(np.zeros((0, 0)), 0.04, 100.0, 1.0, 0.1, False, 0.2, 0.0) == 0.0   # :47-49
    with pytest.raises(RuntimeError, match="AsymptoticAnalysis: Volatility must be positive."):       # :50-52
        orc.asymptotic_price(d["paths"], 0.04, 100.0, 1.0, 0.1, False, 0.0, 0.0, step_major=False)


def test_row_features_bit_exact(orc):
    """compute20DayVolAndMomentum (src/core/PredictionGen.cpp:313-347: the driver's twenty_day_vol / twenty_day_momentum
    columns, and the sigma of AsymptoticAnalysis) against the reference's own driver TU compiled in place
    (oracle/ref_driver_harness.cpp): short and empty histories, exactly 21 prices, non-positive prices inside the window."""
    d = np.load(os.path.join(G, "features.npz"))
    tags = [k[5:] for k in d.files if k.startswith("hist_")]
    assert len(tags) >= 8
    for t in tags:
        got = np.array(orc.row_features(d[f"hist_{t}"]))
        assert (got == d[f"out_{t}"]).all(), (t, got, d[f"out_{t}"])
    if have_ref():   # build container: fresh inputs through the compiled reference as well
        ref = Reference()
        rs = np.random.RandomState(77)
        for n in (21, 22, 100, 1826):
            h = 30.0 * np.exp(np.cumsum(0.03 * rs.standard_normal(n)))
            assert orc.row_features(h) == ref.row_features(h)
