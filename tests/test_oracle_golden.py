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


# ---- the Eigen-backed pin of LSM and MartingaleOptimization (LSMPricer.cpp:76, MartingaleOptimizationPricer.cpp:166) -------------
# The fixtures exist only where `make -C oracle && python oracle/gen_golden.py --eigen` has run on an image with an Eigen3.  The
# image this repo was developed in has none: the two comparisons below are skipped there, and DESIGN.md section 2 says "parity
# unpinned" for that half of the oracle.  The recipe itself is exercised by test_eigen_recipe_is_dormant_without_eigen_and_fires_with_it.
_NO_EIGEN = ("tests/golden/{}.npz absent: captured only where an Eigen3 exists (oracle/Makefile EIGEN_INC, oracle/gen_golden.py --eigen); "
             "this image has none -- LSM / MartingaleOptimization parity is unpinned at the bdcSvd().solve boundary")


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "lsm.npz")), reason=_NO_EIGEN.format("lsm"))
def test_lsm_restatement_pinned_to_eigen(orc):
    """orc_lsm_price (LSMPricer.cpp:19-102 restated with a one-sided Jacobi SVD) against the reference compiled with Eigen, case by
    case at the tolerance stored with the case: 1e-9 where every regression has full numerical rank, 1e-6 .. 2e-5 where Eigen's
    rank threshold decides the fit (single in-the-money path, S0 in the money at j = 0, near-coincident prices), 5e-6 at order 5."""
    d = np.load(os.path.join(G, "lsm.npz"))
    assert len(d["names"]) >= 15
    for name in d["names"]:
        r, K, maturity, dt, is_call, poly, tol = d[f"{name}_args"]
        got = orc.lsm_price(d[f"{name}_paths"], r, K, maturity, dt, bool(is_call), int(poly), step_major=False)
        want = float(d[f"{name}_price"])
        assert abs(got - want) <= tol * abs(want), (str(name), got, want, tuple(d["eigen_version"]))


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "martingale.npz")), reason=_NO_EIGEN.format("martingale"))
def test_martingale_restatement_pinned_to_eigen(orc):
    """orc_martingale_price (MartingaleOptimizationPricer.cpp:21-189) against the reference compiled with Eigen."""
    d = np.load(os.path.join(G, "martingale.npz"))
    assert len(d["names"]) >= 8
    for name in d["names"]:
        r, K, maturity, dt, is_call, poly, iters, tol = d[f"{name}_args"]
        got = orc.martingale_price(d[f"{name}_paths"], r, K, maturity, dt, bool(is_call), int(poly), int(iters), step_major=False)[0]
        want = float(d[f"{name}_price"])
        assert abs(got - want) <= tol * abs(want), (str(name), got, want, tuple(d["eigen_version"]))


def test_eigen_recipe_is_dormant_without_eigen_and_fires_with_it(tmp_path):
    """oracle/Makefile's conditional rule, by dry run (`make -n`: nothing is compiled, no stand-in header is ever read): with no
    Eigen/Dense under EIGEN_INC the build is what it always was; with one, the reference's LSMPricer.cpp and
    MartingaleOptimizationPricer.cpp are compiled IN PLACE together with oracle/ref_eigen_harness.cpp into oracle/_ref/.  And the
    fixture list the capture would run is well formed and goes through the restatement."""
    import subprocess
    from oracle.eigen_fixtures import lsm_cases, martingale_cases
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if not os.path.exists("/root/reference/src/models/LSMPricer.cpp"):
        pytest.skip("the reference tree is not on this machine (GPU box): the rule has nothing to compile")
    off = subprocess.run(["make", "-n", "-B", "-C", here, "EIGEN_INC=/nonexistent"], capture_output=True, text=True, check=True).stdout
    assert "LSMPricer.cpp" not in off and "libmcref.so" in off
    (tmp_path / "Eigen").mkdir()
    (tmp_path / "Eigen" / "Dense").write_text("")          # an empty marker for the wildcard; -n never opens it
    on = subprocess.run(["make", "-n", "-B", "-C", here, f"EIGEN_INC={tmp_path}"], capture_output=True, text=True, check=True).stdout
    line = [ln for ln in on.replace("\\\n", " ").splitlines() if "libmcref_eigen.so" in ln]
    assert line and "/root/reference/src/models/LSMPricer.cpp" in on and "/root/reference/src/models/MartingaleOptimizationPricer.cpp" in on
    assert f"-I{tmp_path}" in on and "ref_eigen_harness.cpp" in on
    # the harness is valid C++ against the reference's headers (it needs Eigen only for the version macros)
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I/root/reference/include", os.path.join(here, "ref_eigen_harness.cpp")], check=True)
    o = Oracle()
    for c in lsm_cases().values():
        assert np.isfinite(o.lsm_price(c["paths"], c["r"], c["K"], c["maturity"], c["dt"], bool(c["is_call"]), c["poly"], step_major=False))
    for c in martingale_cases().values():
        assert np.isfinite(o.martingale_price(c["paths"], c["r"], c["K"], c["maturity"], c["dt"], bool(c["is_call"]), c["poly"], c["iters"],
                                              step_major=False)[0])
