"""CPU tests of the oracle beyond the golden vectors: RNG contract, the law-preserving Volterra
form, the reference-faithful generator against the compiled reference, and the LSM restatement
against an independent LAPACK (gelsd) least squares.  Everything here runs without a GPU."""
import math

import numpy as np
import pytest

from oracle.binding import Oracle, Reference, have_ref, synthetic_history

DT = 1.0 / 252.0


@pytest.fixture(scope="module")
def orc():
    return Oracle()


# ---- RNG contract ------------------------------------------------------------------------------
def test_philox_known_answers(orc):
    """Random123 kat_vectors for philox4x32-10."""
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_normal_quad_definition(orc):
    """One block -> 4 normals: pair (wa, wb): u = ((wb & 0xFF)<<32 | wa) + 1/2) 2^-40,
    f = ((wb >> 8) + 1/2) 2^-24, z = sqrt(-2 ln u) (cos, sin)(2 pi f); pairs (w0,w1), (w2,w3)."""
    seed, path, block, stream = 20251031, (5 << 32) + 17, 3, 1
    w = orc.philox([path & 0xFFFFFFFF, path >> 32, block, stream], [seed & 0xFFFFFFFF, seed >> 32])
    z = orc.normal_quad(seed, path, block, stream)
    for h in range(2):
        wa, wb = w[2 * h], w[2 * h + 1]
        u = ((((wb & 0xFF) << 32) | wa) + 0.5) * 2.0 ** -40
        f = ((wb >> 8) + 0.5) * 2.0 ** -24
        r = math.sqrt(-2.0 * math.log(u))
        assert z[2 * h] == r * math.cos(2.0 * math.pi * f) and z[2 * h + 1] == r * math.sin(2.0 * math.pi * f)


def test_normal_quad_moments(orc):
    z = np.array([orc.normal_quad(1, p, 0, 0) for p in range(20000)])
    flat = z.ravel()
    assert abs(flat.mean()) < 4 / math.sqrt(len(flat))
    assert abs(flat.var() - 1.0) < 0.02
    assert abs((flat ** 4).mean() - 3.0) < 0.1
    c = np.corrcoef(z.T)                       # the four elements of a block are uncorrelated
    assert np.abs(c - np.eye(4)).max() < 4 / math.sqrt(len(z))
    # tails: P(|z| > 3) = 2.6998e-3
    assert abs((np.abs(flat) > 3).mean() - 2.6998e-3) < 6e-4


# ---- GBM ---------------------------------------------------------------------------------------
def test_gbm_oracle_black_scholes(orc):
    """Config C1 exactly (100k x 252, seed 20251031): BS = 9.9251, bar |z| <= 2."""
    n = 100_000
    paths = orc.paths_gbm(20251031, 100.0, 0.04, 0.2, DT, 252, 0, n)
    m, se = orc.price_european(paths, 100.0, 0.04, 1.0, True)
    assert abs(m - 9.9251) <= 2.0 * se
    # shards reproduce the same ids
    part = orc.paths_gbm(20251031, 100.0, 0.04, 0.2, DT, 252, 1000, 10)
    assert np.array_equal(part, paths[:, 1000:1010])


def test_gbm_oracle_z_scores_over_many_seeds(orc):
    """Health of the draw stream: over 30 seeds the z-scores of the sample mean and variance of
    log S_T behave like N(0,1) (mean ~ 0, mean square ~ 1)."""
    zs, vs = [], []
    steps, n = 32, 10_000
    T = steps * DT
    for seed in range(500, 530):
        lz = np.log(orc.paths_gbm(seed, 100.0, 0.04, 0.2, DT, steps, 0, n)[-1] / 100.0)
        zs.append((lz.mean() - 0.02 * T) / (0.2 * math.sqrt(T / n)))
        vs.append((lz.var(ddof=1) / (0.04 * T) - 1.0) / math.sqrt(2.0 / n))
    for a in (np.array(zs), np.array(vs)):
        assert abs(a.mean()) < 0.6 and 0.5 < (a ** 2).mean() < 1.7, (a.mean(), (a ** 2).mean())


# ---- rBergomi: the spectral-pair synthesis has the reference's law ------------------------------
@pytest.mark.parametrize("steps,H,eta", [(252, 0.1, 1.9), (512, 0.1, 1.9), (7, 0.57, 0.03), (50, 0.3, 1.0), (1, 0.2, 0.5)])
def test_spectral_pairs_reproduce_reference_covariance(orc, steps, H, eta):
    """With Y_k = a_k (g_k + i h_k) and x_n = sum_k Y_k e^{2 pi i k n/M}:
    Cov(Re x_n, Re x_{n+d}) = Cov(Im x_n, Im x_{n+d}) = sum_k a_k^2 cos(2 pi k d/M) must equal the covariance of the
    reference's X at every lag, and Cov(Re x_n, Im x_{n+d}) = sum_k a_k^2 sin(2 pi k d/M)
    must vanish at every lag (the two paths of a pair are independent).  Zero-mean Gaussian vectors: equal
    covariance <=> equal law."""
    lam = orc.lam(steps, H)
    phi = orc.phi(lam)                       # pinned bit-exact to the compiled reference
    M = orc.next_pow2(steps)
    P = np.zeros(M)
    P[:min(steps, M)] = np.abs(phi[:min(steps, M)]) ** 2
    d = np.arange(M)
    k = np.arange(M)
    ang = 2 * np.pi * np.outer(d, k) / M
    cov_ref = (2 * H * eta ** 2 / M ** 2) * (P[None, :] * np.cos(ang)).sum(axis=1)
    amp, comp = orc.rbergomi_spectrum(H, eta, DT, steps)
    cov_ours = ((amp ** 2)[None, :] * np.cos(ang)).sum(axis=1)
    cross = ((amp ** 2)[None, :] * np.sin(ang)).sum(axis=1)
    scale = max(1e-300, abs(cov_ref[0]))
    assert np.allclose(cov_ours, cov_ref, rtol=1e-10, atol=1e-13 * scale)
    assert np.all(np.abs(cross) <= 1e-12 * scale)
    assert np.allclose(comp, -0.5 * eta ** 2 * (np.arange(steps) * DT) ** (2 * H), rtol=1e-14)
    if steps == 252:
        assert abs(cov_ref[0] - 0.12552) < 5e-5      # SURVEY-verified Var(X_n)
    if steps == 512:
        assert abs(cov_ref[0] - 0.25480) < 5e-5      # the M_phi=1024 / M_z=512 quirk


def test_spectral_sample_covariance_vs_reference_transform(orc):
    """Sampled X from the device algorithm AND X from the reference's own transform on Gaussian noise both
    reproduce the closed-form covariance (each within 4 standard errors of the estimator)."""
    steps, H, eta, n = 64, 0.1, 1.9, 6000
    _, X = orc.paths_rbergomi(3, 100.0, 0.04, 0.04, H, eta, -0.9, DT, steps, 0, n, want_X=True)
    phi = orc.phi(orc.lam(steps, H))
    rs = np.random.RandomState(1)
    Xr = np.array([orc.fractional_gaussian(phi, rs.standard_normal(steps) + 1j * rs.standard_normal(steps), H, eta)
                   for _ in range(n)])
    M = orc.next_pow2(steps)
    P = np.zeros(M)
    P[:steps] = np.abs(phi[:steps]) ** 2
    k = np.arange(M)
    cov = lambda d: (2 * H * eta ** 2 / M ** 2) * (P * np.cos(2 * np.pi * k * d / M)).sum()  # noqa: E731
    for lag in (0, 1, 5, 31):
        c0, cd = cov(0), cov(lag)
        se = math.sqrt((c0 * c0 + cd * cd) / n)
        for sample in (X, Xr):
            est = (sample[:, 10] * sample[:, (10 + lag) % steps]).mean()
            assert abs(est - cd) <= 4 * se, (lag, est, cd, se)
        # the two paths of a pair (Re / Im of one transform) are uncorrelated at every lag
        cross = (X[0::2, 10] * X[1::2, (10 + lag) % steps]).mean()
        assert abs(cross) <= 4 * c0 / math.sqrt(n / 2), (lag, cross)


def test_rbergomi_oracle_martingale(orc):
    steps, n = 64, 30_000
    paths = orc.paths_rbergomi(5, 100.0, 0.04, 0.04, 0.1, 1.9, -0.9, DT, steps, 0, n)
    ST = paths[-1]
    assert abs(ST.mean() - 100.0 * math.exp(0.04 * steps * DT)) <= 3 * ST.std() / math.sqrt(n)


# ---- reference-faithful generator vs the compiled reference ------------------------------------
@pytest.mark.skipif(not have_ref(), reason="compiled reference not present")
def test_mt_mode_statistics_match_compiled_reference(orc):
    """Class-level entry point (history-estimated parameters): 60 000 paths of the compiled reference (unseeded) against
    the same number from the restatement's "mt" mode -- mean of the log-return within 4 standard errors, variance
    ratio within 3 % (the standard error of a variance ratio of two Gaussian samples of this size is 0.8 %)."""
    ref = Reference()
    hist = synthetic_history(1001, seed=42)
    n, steps = 60_000, 40
    a = ref.generate_paths(hist, steps, n)           # unseeded std::random_device
    b = orc.generate_paths_mt_hist(hist, steps, n, 77)
    assert a.shape == b.shape == (n, steps + 1)
    assert (a[:, 0] == hist[-1]).all() and (b[:, 0] == hist[-1]).all()
    la, lb = np.log(a[:, -1] / a[:, 0]), np.log(b[:, -1] / b[:, 0])
    assert abs(la.mean() - lb.mean()) <= 4 * math.sqrt(la.var() / n + lb.var() / n)
    assert abs(la.var() / lb.var() - 1) < 0.03
    with pytest.raises(RuntimeError, match="Historical prices vector too small."):
        ref.generate_paths([100.0], 5, 5)


@pytest.mark.skipif(not have_ref(), reason="compiled reference not present")
def test_philox_mode_statistics_match_compiled_reference(orc):
    """Device algorithm (Philox mode) vs the compiled reference itself, class-level parameters."""
    ref = Reference()
    hist = synthetic_history(1001, seed=42)
    p = orc.estimate_params(hist)
    n, steps = 60_000, 40
    a = ref.generate_paths(hist, steps, n)
    b = orc.paths_rbergomi(123, p["S0"], 0.04, p["xi"], p["H"], p["eta"], p["rho"], DT, steps, 0, n).T
    la, lb = np.log(a[:, -1] / a[:, 0]), np.log(b[:, -1] / b[:, 0])
    assert abs(la.mean() - lb.mean()) <= 4 * math.sqrt(la.var() / n + lb.var() / n)
    assert abs(la.var() / lb.var() - 1) < 0.03


# ---- the rough regime (C4 / C5 parameters) against a sample drawn with the compiled reference ---
def _rough_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rough_regime_reference.json")))


def _z_scores(sums, sums2, n, fix):
    from oracle.binding import mean_and_se
    m, se = mean_and_se(sums, sums2, n)
    k = len(m)
    fm, fse = np.array(fix["mean"][:k]), np.array(fix["std_err"][:k])
    return (m - fm) / np.sqrt(se * se + fse * fse), m, se


@pytest.mark.parametrize("steps,n", [(252, 200_000), (512, 200_000)])
def test_rough_regime_oracle_modes_vs_compiled_reference_sample(orc, steps, n):
    """H = 0.1, eta = 1.9 (BASELINE.json configs C4 / C5): tests/golden/rough_regime_reference.json holds the statistics
    of 2e6 (252 steps) / 1e6 (512 steps) paths drawn through the compiled reference's own private members
    (oracle/gen_rough_fixture.py).  Both modes of the restatement -- "mt" (the reference's algorithm and RNG consumption,
    complex FFT per path) and "philox" (the device algorithm: one transform per PAIR of paths from symmetrised
    amplitudes) -- must reproduce, within 2 combined standard errors: E[S_T], the call and the put price, the realised
    variance and the clustering of squared returns at lags 1, 8 and 64.  The last four see the Volterra / forward-variance
    structure directly; a wrong spectrum, compensator or pairing shows there long before it moves a price."""
    from oracle.binding import STAT_NAMES, path_stats
    fx = _rough_fixture()
    P, fix = fx["params"], fx["samples"][str(steps)]
    chunk = 25_000
    for mode in ("mt", "philox"):
        s, s2 = np.zeros(7), np.zeros(7)
        for c in range(n // chunk):
            if mode == "mt":
                m = orc.generate_paths_mt(P["S0"], P["r"], P["xi"], P["H"], P["eta"], P["rho"], steps, chunk, 9000 + c).T
            else:
                m = orc.paths_rbergomi(4242, P["S0"], P["r"], P["xi"], P["H"], P["eta"], P["rho"], DT, steps, c * chunk, chunk)
            a, b, _ = path_stats(m, P["strike"])
            s += a
            s2 += b
        z, mean, se = _z_scores(s, s2, n, fix)
        assert (np.abs(z) <= 2.0).all(), (mode, steps, dict(zip(STAT_NAMES, np.round(z, 2))), mean, se)


# ---- LSM ---------------------------------------------------------------------------------------
def lsm_numpy(paths_pm, r, K, maturity, dt, is_call, poly):
    """Independent restatement of LSMPricer.cpp:19-102 with LAPACK gelsd (SVD, min-norm)."""
    pay = (lambda s: np.maximum(0.0, s - K)) if is_call else (lambda s: np.maximum(0.0, K - s))
    N, M = paths_pm.shape
    V = pay(paths_pm[:, M - 1])
    disc = math.exp(-r * dt)
    for j in range(M - 2, -1, -1):
        if j * dt > maturity:
            V = V * disc
            continue
        s = paths_pm[:, j]
        p = pay(s)
        itm = p > 1e-14
        Vn = np.zeros(N)
        if itm.any():
            A = np.vander(s[itm], poly + 1, increasing=True)
            c, *_ = np.linalg.lstsq(A, V[itm] * disc, rcond=min(A.shape) * np.finfo(float).eps)
            Vn[itm] = np.maximum(p[itm], A @ c)
        otm = p < 1e-14
        Vn[otm] = V[otm] * disc
        V = Vn
    return V.mean()


def _exact_eigen_rule_fit(S, b, p):
    """Eigen's bdcSvd().solve(b) on raw monomials of S (LSMPricer.cpp:61-76) in 60-digit arithmetic: minimum-norm least
    squares with singular values <= min(rows, cols) eps sigma_max dropped.  Returns the fitted values at the data."""
    import mpmath as mp
    mp.mp.dps = 60
    n = len(S)
    A = mp.matrix(n, p + 1)
    for i in range(n):
        for k in range(p + 1):
            A[i, k] = mp.mpf(float(S[i])) ** k
    U, sv, V = mp.svd_r(A, full_matrices=False, compute_uv=True)
    thr = min(n, p + 1) * mp.mpf(np.finfo(float).eps) * sv[0]
    coef = mp.matrix(p + 1, 1)
    for j in range(len(sv)):
        if sv[j] > thr:
            proj = sum(U[i, j] * mp.mpf(float(b[i])) for i in range(n)) / sv[j]
            for t in range(p + 1):
                coef[t] += proj * V[j, t]
    fit = A * coef
    return np.array([float(fit[i]) for i in range(n)])


def test_lsm_oracle_rank_rule_on_near_degenerate_dates_vs_exact_arithmetic(orc):
    """The oracle's least-squares solve (one-sided Jacobi SVD, Eigen's default threshold) on dates whose in-the-money
    prices nearly coincide, against the same rule evaluated in 60-digit arithmetic: two-date matrices whose single
    regression is the system under test (r = 0, terminal payoff K - S_1 chosen to be the right-hand side b)."""
    rs = np.random.RandomState(1)
    K = 100.0
    for base in (90.0, 99.0, 60.0):
        for n in (2, 3, 4, 5):
            for spread in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
                S = base * (1 + spread * rs.uniform(-1, 1, n))
                b = rs.uniform(45.0, 60.0, n)                      # above every immediate payoff: V_0 = fitted value
                fit = _exact_eigen_rule_fit(S, b, 2)
                _, v0 = orc.lsm_price(np.stack([S, K - b]), 0.0, K, 1.0, 1.0, False, 2, want_v0=True)
                want = np.maximum(K - S, fit)
                assert np.max(np.abs(v0 - want) / np.abs(want)) <= 2e-5, (base, n, spread, v0, want)


@pytest.mark.parametrize("is_call,poly", [(False, 2), (True, 2), (False, 3), (False, 0)])
def test_lsm_oracle_vs_lapack(orc, is_call, poly):
    n, steps, dt = 4000, 50, 0.02
    paths = orc.paths_gbm(9, 100.0, 0.04, 0.2, dt, steps, 0, n)      # step-major
    got = orc.lsm_price(paths, 0.04, 100.0, 1.0, dt, is_call, poly)
    want = lsm_numpy(paths.T, 0.04, 100.0, 1.0, dt, is_call, poly)
    assert abs(got - want) <= 1e-9 * abs(want), (got, want)
    # layout-independent
    assert orc.lsm_price(np.ascontiguousarray(paths.T), 0.04, 100.0, 1.0, dt, is_call, poly, step_major=False) == got


def test_lsm_oracle_edge_cases(orc):
    rs = np.random.RandomState(3)
    otm = 150.0 + rs.rand(50, 6)                                   # all OTM: pure discount (:89-94)
    v = orc.lsm_price(otm, 0.04, 100.0, 1.0, 0.2, False, 2, step_major=False)
    assert v == 0.0
    itm_end = np.full((10, 4), 150.0)
    itm_end[:, -1] = 90.0                                          # payoff 10 at the end only
    v = orc.lsm_price(itm_end, 0.04, 100.0, 1.0, 0.25, False, 2, step_major=False)
    assert abs(v - 10.0 * math.exp(-0.04 * 0.75)) < 1e-12
    # grid longer than maturity (:43-49): dates past maturity only discount
    mixed = 100.0 * np.exp(np.cumsum(0.1 * rs.standard_normal((300, 9)), axis=1))
    a = orc.lsm_price(mixed, 0.04, 100.0, 0.35, 0.1, False, 2, step_major=False)
    b = lsm_numpy(mixed, 0.04, 100.0, 0.35, 0.1, False, 2)
    assert abs(a - b) <= 1e-9 * abs(b)
    # N = 1 (rank-1 fit reproduces the single discounted value), and empty input throws
    single = np.array([[100.0, 90.0, 95.0, 85.0]])
    a = orc.lsm_price(single, 0.04, 100.0, 1.0, 0.25, False, 2, step_major=False)
    b = lsm_numpy(single, 0.04, 100.0, 1.0, 0.25, False, 2)
    assert abs(a - b) <= 1e-12 * abs(b)
    with pytest.raises(RuntimeError, match="LSM::PredictOptionPrice: Empty pricePaths."):
        orc.lsm_price(np.zeros((0, 0)), 0.04, 100.0, 1.0, 0.25, False, 2)


def test_lsm_american_put_above_european(orc):
    n, steps, dt = 30_000, 50, 0.02
    paths = orc.paths_gbm(11, 100.0, 0.04, 0.2, dt, steps, 0, n)
    am = orc.lsm_price(paths, 0.04, 100.0, 1.0, dt, False, 2)
    eu, se = orc.price_european(paths, 100.0, 0.04, 1.0, False)
    assert am > eu - 2 * se and am < eu + 1.5


# ---- MartingaleOptimization --------------------------------------------------------------------
def martingale_numpy(P, r, K, maturity, dt, is_call, poly, iters):
    """Independent restatement of MartingaleOptimizationPricer.cpp:21-189 with LAPACK gelsd."""
    pay = (lambda s: np.maximum(0.0, s - K)) if is_call else (lambda s: np.maximum(0.0, K - s))
    N, M = P.shape
    t = np.arange(M) * dt
    dates = int(np.argmax(t > maturity)) if (t > maturity).any() else M
    disc = np.exp(-r * np.minimum(t, maturity))
    d = pay(P) * disc[None, :]
    coef, offset = np.zeros(poly + 1), 0.0
    lo = up = 0.0
    for _ in range(iters):
        dd = d[:, :dates]
        best = np.maximum(dd.max(axis=1), 0.0) if dates else np.zeros(N)
        stop = np.where(best > 0, dd.argmax(axis=1), 0) if dates else np.zeros(N, int)
        lo = best.mean()
        Mv = np.polynomial.polynomial.polyval(P[:, :dates], coef) - offset
        up = np.maximum((dd - Mv).max(axis=1), 0.0).mean() if dates else 0.0
        other = (stop + M // 2) % M
        rows = np.arange(N)
        X = np.empty(2 * N)
        Y = np.empty(2 * N)
        X[0::2], Y[0::2] = P[rows, stop], 0.5 * d[rows, stop]
        X[1::2], Y[1::2] = P[rows, other], 0.2 * d[rows, other]
        if 2 * N >= poly + 1:
            A = np.vander(X, poly + 1, increasing=True)
            coef, *_ = np.linalg.lstsq(A, Y, rcond=min(A.shape) * np.finfo(float).eps)
            offset = np.polynomial.polynomial.polyval(P[:, 0], coef).mean()
    return 0.5 * (lo + up), lo, up


@pytest.mark.parametrize("is_call,poly,iters", [(False, 2, 5), (True, 2, 5), (False, 3, 2), (False, 2, 1), (False, 0, 3)])
def test_martingale_oracle_vs_lapack(orc, is_call, poly, iters):
    P = orc.paths_gbm(21, 100.0, 0.04, 0.2, DT, 40, 0, 3000).T.copy()     # [paths][cols]
    for maturity in (40 * DT, 25.5 * DT):                                  # second: grid longer than maturity
        got = orc.martingale_price(P, 0.04, 100.0, maturity, DT, is_call, poly, iters, step_major=False)
        want = martingale_numpy(P, 0.04, 100.0, maturity, DT, is_call, poly, iters)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12), (got, want)
    with pytest.raises(RuntimeError, match="MartingaleOptimization: Empty pricePaths."):
        orc.martingale_price(np.zeros((0, 0)), 0.04, 100.0, 1.0, DT, False, 2)
    with pytest.raises(RuntimeError, match="MartingaleOptimization: maxIterations must be positive."):
        orc.martingale_price(P, 0.04, 100.0, 1.0, DT, False, 2, 0, step_major=False)


# ---- BranchingProcesses ------------------------------------------------------------------------
def _branching_fixture():
    import os
    g = os.path.join(os.path.dirname(__file__), "golden")
    return np.load(os.path.join(g, "asymptotic.npz"))["paths"], np.load(os.path.join(g, "branching.npz"))


def test_branching_lower_bound_matches_compiled_reference(orc):
    """The deterministic half (BranchingProcessPricer.cpp:41-72) against values captured from the compiled
    reference (its OpenMP reduction fixes no summation order: 1e-13, not bit-exact)."""
    paths, d = _branching_fixture()
    ex_all = d["ex_all"]
    for is_call, maturity, K, n_ex, want in d["cases"]:
        ex = ex_all if int(n_ex) == len(ex_all) else ex_all[::5]
        for mode in ("mt", "philox"):
            _, lo, up = orc.branching_price(paths, 0.04, K, maturity, 1 / 252.0, bool(is_call), 10, ex, 7, mode=mode,
                                            step_major=False)
            assert abs(lo - want) <= 1e-13 * max(want, 1e-300), (mode, lo, want)
            assert up >= lo - 1e-12
    for bad, msg in [((np.zeros((0, 0)), 100.0, ex_all), "Empty pricePaths."), ((paths, 100.0, []), "No exercise times."),
                     ((paths, 0.0, ex_all), "Strike must be positive.")]:
        with pytest.raises(RuntimeError, match="BranchingProcesses: " + msg):
            orc.branching_price(bad[0], 0.04, bad[1], 1.0, 1 / 252.0, False, 10, bad[2], 1, step_major=False)


@pytest.mark.skipif(not have_ref(), reason="compiled reference not present")
def test_branching_upper_bound_statistics(orc):
    """Upper bound: unseeded resampling in the reference, so compare the mean over repeated calls --
    compiled reference vs oracle "mt" (same algorithm, explicit seed) vs oracle "philox" (device algorithm)."""
    ref = Reference()
    paths, d = _branching_fixture()
    ex = d["ex_all"]
    args = (0.04, 100.0, 40 / 252.0, 1 / 252.0, False, 10, ex)
    a = np.array([ref.branching_price(paths, *args)[2] for _ in range(24)])
    b = np.array([orc.branching_price(paths, *args, s, mode="mt", step_major=False)[2] for s in range(24)])
    c = np.array([orc.branching_price(paths, *args, s, mode="philox", step_major=False)[2] for s in range(24)])
    for x in (b, c):
        se = math.sqrt(a.var(ddof=1) / len(a) + x.var(ddof=1) / len(x))
        assert abs(a.mean() - x.mean()) <= 4 * se, (a.mean(), x.mean(), se)
    assert 0.5 < b.std(ddof=1) / c.std(ddof=1) < 2.0


def test_cpu_baseline_harness_prices_what_the_single_calls_price(orc):
    """bench.py's CPU baselines of the widened rows run the pricers over a resident sample in driver rows of 250 paths under
    omp dynamic (oracle/ref_harness.cpp: ref_pricer_chunks_omp, oracle/mcg_oracle.cpp: orc_pricer_chunks_omp).  What they
    time must be the pricers themselves: the sum of prices they return equals the sum over the same rows priced one call at
    a time (exactly for the deterministic pricers; BranchingProcesses resamples with an unseeded generator in the reference)."""
    from oracle.binding import Reference, have_ref
    S = np.ascontiguousarray(orc.paths_gbm(11, 100.0, 0.04, 0.2, 0.02, 50, 0, 1000).T)   # [1000][51]
    arg = (0.04, 100.0, 1.0, 0.02, False)
    th, sec, chk = orc.pricer_chunks_omp("lsm", S, 250, *arg, 2)
    one = sum(orc.lsm_price(S[k:k + 250], *arg, 2, step_major=False) for k in range(0, 1000, 250))
    assert th >= 1 and sec > 0 and abs(chk - one) <= 1e-12 * one
    th, sec, chk = orc.pricer_chunks_omp("martingale", S, 250, *arg, 2)
    one = sum(orc.martingale_price(S[k:k + 250], *arg, 2, 5, step_major=False)[0] for k in range(0, 1000, 250))
    assert abs(chk - one) <= 1e-12 * one
    if not have_ref():
        pytest.skip("compiled reference not built here")
    ref = Reference()
    th, sec, chk = ref.pricer_chunks_omp("asymptotic", S, 250, *arg, 0.2, 0.0)
    one = sum(ref.asymptotic_price(S[k:k + 250], *arg, 0.2, 0.0) for k in range(0, 1000, 250))
    assert abs(chk - one) <= 1e-12 * one
    th, sec, chk = ref.pricer_chunks_omp("branching", S, 250, *arg, 0.2, 0.0, 10)
    lo = sum(ref.branching_price(S[k:k + 250], *arg, 10, np.arange(50, dtype=np.int32))[1] for k in range(0, 1000, 250))
    assert chk > 0.5 * lo and np.isfinite(chk)       # midpoint of the deterministic lower bound and a resampled upper bound
    # whole driver rows: four sums, all positive, LSM and MartingaleOptimization through the restatement's entry points
    from oracle.binding import synthetic_history
    hist = synthetic_history(300, seed=3)
    th, sec, sums = ref.driver_rows_omp(hist, [20, 40, 11], [float(hist[-1])] * 3, [0, 1, 0], 250, 0.2, 0.08, orc)
    assert th >= 1 and sec > 0 and np.all(sums > 0) and np.all(np.isfinite(sums))
