"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Tolerances (all arithmetic is binary64):
  * path values, GPU vs oracle "philox" mode (same draws, libm vs device log/sincos/exp):
      GBM       rel 1e-11   (252 steps x a few ulp each)
      rBergomi  rel 1e-9    (plus a 256..512-term FMA chain per step)
  * prices vs closed forms / independent samples: |z| <= 2 MC standard errors (north-star bar)
  * LSM given identical paths (deterministic): rel 1e-8
  * layout conversions, sharding by path_begin: bit-exact
"""
import math

import numpy as np
import pytest

import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd import _native as N
from oracle.binding import Oracle, synthetic_history

pytestmark = pytest.mark.gpu

SEED = 20251031
RB_S0 = 100.0
DT = 1.0 / 252.0


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
    c = S0 * N(d1) - K * math.exp(-r * T) * N(d2)
    return c if call else c - S0 + K * math.exp(-r * T)


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


# ------------------------------------------------------------------------------------------------
# GBM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_paths,n_steps", [(1000, 252), (64, 1), (257, 7), (1, 50), (4096, 50)])
def test_gbm_paths_match_oracle(eng, orc, n_paths, n_steps):
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, n_steps, n_paths, path_begin=12345)
    got = P.to_host_step_major()
    want = orc.paths_gbm(SEED, 100.0, 0.04, 0.2, DT, n_steps, 12345, n_paths)
    assert got.shape == want.shape == (n_steps + 1, n_paths)
    assert (got[0] == 100.0).all()
    assert rel_err(got, want) < 1e-11
    P.free()


@pytest.mark.parametrize("n_paths,n_steps,sigma", [(1_049_700, 7, 0.2), (1_050_001, 5, 0.2), (1_049_700, 4, 1.5)])
def test_gbm_two_paths_per_lane_matches_oracle(eng, orc, n_paths, n_steps, sigma):
    """From 1M paths (256 CUs x 8 workgroups of 512) a lane carries two adjacent paths and stores 16 bytes per step.
    1 049 700 paths pad to 4101 x 256 columns: the upper two waves of the last 512-column workgroup lie beyond the row.
    Steps with a tail of 3 / 1 / 0 after the whole Philox blocks; sigma = 1.5 takes the general exponential.  Paths and
    the fused payoff sums against the oracle; odd path_begin (the pair is a pair of COLUMNS, not of ids)."""
    T = n_steps * DT
    P = eng.gbm(SEED, 100.0, 0.04, sigma, DT, n_steps, n_paths, path_begin=777, payoff=(101.0, True))
    got = P.to_host_step_major()
    want = orc.paths_gbm(SEED, 100.0, 0.04, sigma, DT, n_steps, 777, n_paths)
    assert got.shape == want.shape and rel_err(got, want) < 1e-11
    m, se = eng.price_european(P, 101.0, 0.04, T, True)
    om, ose = orc.price_european(want, 101.0, 0.04, T, True)
    assert abs(m - om) <= 1e-10 * om and abs(se - ose) <= 1e-8 * ose
    P.free()


def test_fused_payoff_sums_repeat_bit_for_bit():
    """The sums a generator leaves for price_european are a fixed-order reduction (per workgroup / per share of the
    persistent rBergomi kernel, then in chunks of 8192 partials, then one block): two contexts, and two runs on one,
    return the same bits -- whichever workgroup drew which share."""
    import montecarlooptionspricer_amd as mc
    res = []
    for rep in range(2):
        e = mc.PathEngine(0)
        for again in range(2):
            P = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 9, 3_000_001, payoff=(100.0, True))     # 5860 partials x 2 paths per lane
            g = e.price_european(P, 100.0, 0.04, 9 * DT, True)
            P.free()
            Q = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 3, 9_000_001, payoff=(100.0, False))    # 17579 partials: two chunk levels
            g2 = e.price_european(Q, 100.0, 0.04, 3 * DT, False)
            Q.free()
            R = e.rbergomi(SEED, RB_S0, 0.04, 0.04, 0.1, 1.9, -0.9, DT, 40, 600_001, payoff=(100.0, False))  # 4688 shares on 512 workgroups
            r = e.price_european(R, 100.0, 0.04, 40 * DT, False)
            R.free()
            res.append((g, g2, r))
        e.close()
    assert all(x == res[0] for x in res[1:]), res


@pytest.mark.parametrize("n_steps", [2, 3, 4, 5, 6, 9, 11])
def test_gbm_block_loop_tails_and_64_bit_path_ids(eng, orc, n_steps):
    """The generator runs per Philox block (4 steps) with a tail of 1-3 steps, and the Philox counter carries the full
    64-bit global path id: ids beyond 2^32 and a large 64-bit seed must match the oracle as well."""
    begin = (1 << 33) + 7
    seed = 0xFEDCBA9876543210
    P = eng.gbm(seed, 100.0, 0.04, 0.2, DT, n_steps, 130, path_begin=begin)
    want = orc.paths_gbm(seed, 100.0, 0.04, 0.2, DT, n_steps, begin, 130)
    assert rel_err(P.to_host_step_major(), want) < 1e-11
    P.free()
    Q = eng.rbergomi(seed, RB_S0, 0.04, 0.04, 0.1, 1.9, -0.9, DT, n_steps, 66, path_begin=begin + 1)
    want = orc.paths_rbergomi(seed, RB_S0, 0.04, 0.04, 0.1, 1.9, -0.9, DT, n_steps, begin + 1, 66)
    assert rel_err(Q.to_host_step_major(), want) < 1e-9
    Q.free()


def test_gbm_layouts_and_sharding_are_exact(eng):
    """to_host is the transpose of the stored matrix; a shard [b, b+n) reproduces the same ids."""
    full = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 33, 1000)
    sm = full.to_host_step_major()
    pm = full.to_host()
    assert pm.shape == (1000, 34) and np.array_equal(pm, sm.T)
    parts = []
    for b, n in [(0, 300), (300, 1), (301, 699)]:
        Q = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 33, n, path_begin=b)
        parts.append(Q.to_host_step_major())
        Q.free()
    assert np.array_equal(np.concatenate(parts, axis=1), sm)
    full.free()


def test_gbm_empty_and_bad_arguments(eng):
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 10, 0)
    assert P.n_paths == 0 and P.to_host().shape == (0, 11)
    with pytest.raises(mc.McgError, match="no paths"):
        eng.price_european(P, 100.0, 0.04, 1.0, True)
    P.free()
    with pytest.raises(mc.McgError, match="n_steps"):
        eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 0, 10)
    with pytest.raises(mc.McgError, match="n_paths"):
        eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 10, -1)
    with pytest.raises(mc.McgError, match="sigma"):
        eng.gbm(SEED, 100.0, 0.04, -0.2, DT, 10, 10)


def test_gbm_zero_vol_is_deterministic_growth(eng):
    P = eng.gbm(SEED, 100.0, 0.04, 0.0, DT, 252, 128)
    a = P.to_host()
    assert np.allclose(a[:, -1], 100.0 * math.exp(0.04), rtol=1e-13)
    assert (a == a[0]).all()
    P.free()


def test_european_call_c1_config_vs_black_scholes_and_oracle(eng, orc):
    """Config C1 (100k x 252, S0=K=100, r=0.04, sigma=0.2, T=1): BS = 9.9251."""
    n = 100_000
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, n, payoff=(100.0, True))
    mean, se = eng.price_european(P, 100.0, 0.04, 1.0, True)          # fused sums
    ref = bs_price(100.0, 100.0, 0.04, 0.2, 1.0)
    assert abs(ref - 9.9251) < 1e-4
    assert abs(mean - ref) <= 2.0 * se, (mean, se, ref)
    # same number from the stored matrix, from the put side, and from the oracle on the same paths
    mean2, se2 = eng.price_european(P, 100.0, 0.04, 1.0, False)       # different payoff -> re-reads last row
    host = P.to_host_step_major()
    om, ose = orc.price_european(host, 100.0, 0.04, 1.0, True)
    assert abs(om - mean) <= 1e-11 * abs(om) and abs(ose - se) <= 1e-9 * ose
    # The put on the same draws is the oracle's number too (no separate bar here: this seed's put happens to sit 2.6 of
    # its standard errors from Black-Scholes; calls AND puts are judged over 24 seeds in tests/test_gpu_statistics.py)
    om2, ose2 = orc.price_european(host, 100.0, 0.04, 1.0, False)
    assert abs(om2 - mean2) <= 1e-11 * abs(om2) and abs(ose2 - se2) <= 1e-9 * ose2
    Q = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, n)                    # unfused path, same draws
    mean3, se3 = eng.price_european(Q, 100.0, 0.04, 1.0, True)
    assert abs(mean3 - mean) <= 1e-12 * mean and abs(se3 - se) <= 1e-9 * se
    P.free()
    Q.free()


def test_gbm_martingale_property_large(eng):
    """Size-independent property at a larger size: E[S_T] = S0 e^{rT}."""
    n = 2_000_000
    P = eng.gbm(7, 100.0, 0.04, 0.2, DT, 252, n)
    # forward = e^{rT} * (call - put) + K  (put-call parity on the same paths)
    c, cse = eng.price_european(P, 100.0, 0.04, 1.0, True)
    p, pse = eng.price_european(P, 100.0, 0.04, 1.0, False)
    fwd = math.exp(0.04) * (c - p) + 100.0
    se = math.exp(0.04) * math.hypot(cse, pse)
    assert abs(fwd - 100.0 * math.exp(0.04)) <= 2.0 * se
    P.free()


# ------------------------------------------------------------------------------------------------
# rBergomi
# ------------------------------------------------------------------------------------------------
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)


@pytest.mark.parametrize("n_paths,n_steps", [(192, 64), (100, 252), (70, 7), (65, 1), (64, 512), (41, 17), (21, 33), (33, 100),
                                             (7, 1000), (5, 2048), (600, 16)])
def test_rbergomi_paths_match_oracle(eng, orc, n_paths, n_steps):
    P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, n_steps, n_paths,
                     path_begin=998)
    got = P.to_host_step_major()
    want = orc.paths_rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, n_steps, 998,
                              n_paths)
    assert rel_err(got, want) < 1e-9
    P.free()


@pytest.mark.parametrize("n_paths,n_steps", [(40_001, 150), (24_001, 300)])
def test_rbergomi_persistent_shares_match_oracle(eng, orc, n_paths, n_steps):
    """More shares than resident workgroups (each works through several, drawn from the ticket counter) AND an odd
    number of live tiles per share (150 of 256 / 300 of 512 steps: three of four), so that the last tile of one share
    and the first of the next use the same staging buffer; the last tile is also unfinished (S_T comes from a lane in the
    middle of the tile).  Paths and the fused payoff sums against the oracle."""
    T = n_steps * DT
    P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, n_steps, n_paths,
                     path_begin=2, payoff=(95.0, False))
    got = P.to_host_step_major()
    want = orc.paths_rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, n_steps, 2, n_paths)
    assert rel_err(got, want) < 1e-9
    m, se = eng.price_european(P, 95.0, RB["r"], T, False)     # from the sums the generator left
    om, ose = orc.price_european(want, 95.0, RB["r"], T, False)
    assert abs(m - om) <= 1e-9 * om and abs(se - ose) <= 1e-8 * ose
    P.free()


@pytest.mark.parametrize("n_paths,n_steps", [(6, 2500), (3, 4100)])
def test_rbergomi_beyond_2048_steps_matches_oracle(eng, orc, n_paths, n_steps):
    """The reference has no size limit (RoughVolatility.cpp:337-344): transforms longer than a wavefront holds (Mz = 4096,
    8192) take the direct-summation route -- same Philox draws, same matrix to rounding, and the payoff sums the fused
    entry point leaves agree with a pricing pass over the stored matrix."""
    P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, n_steps, n_paths, path_begin=10,
                     payoff=(90.0, True))
    got = P.to_host_step_major()
    want = orc.paths_rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, n_steps, 10, n_paths)
    assert np.isfinite(got).all() and rel_err(got, want) < 1e-9
    T = n_steps * DT
    m, se = eng.price_european(P, 90.0, RB["r"], T, True)
    om, ose = orc.price_european(want, 90.0, RB["r"], T, True)
    assert abs(m - om) <= 1e-9 * abs(om) + 1e-12 and abs(se - ose) <= 1e-8 * abs(ose) + 1e-12
    P.free()


def test_rbergomi_spectrum_matches_oracle(orc):
    from montecarlooptionspricer_amd.engine import rbergomi_spectrum
    for steps, H, eta in [(252, 0.1, 1.9), (512, 0.1, 1.9), (7, 0.57, 0.03), (1, 0.3, 1.0)]:
        a, c = rbergomi_spectrum(H, eta, DT, steps)
        ao, co = orc.rbergomi_spectrum(H, eta, DT, steps)
        assert np.allclose(a, ao, rtol=1e-12, atol=1e-300)
        assert np.allclose(c, co, rtol=1e-15, atol=0)


def test_rbergomi_eta_zero_is_gbm_with_sigma2_xi(eng):
    """constant variance v == xi turns the reference's stepping loop into GBM."""
    A = eng.rbergomi(SEED, 100.0, 0.04, 0.04, 0.3, 0.0, -0.5, DT, 50, 500)
    B = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 50, 500)
    assert rel_err(A.to_host_step_major(), B.to_host_step_major()) < 1e-12
    A.free()
    B.free()


def test_rbergomi_martingale_and_mixing_price(eng, orc):
    """Known answers of the reference's dynamics: E[S_T] = S0 e^{rT}, and the
    European price equals the oracle's reference-faithful ("mt") sample within 2 combined std-errs."""
    steps, T = 64, 64 * DT
    n = 400_000
    P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, n,
                     payoff=(100.0, True))
    c, cse = eng.price_european(P, 100.0, RB["r"], T, True)
    p, pse = eng.price_european(P, 100.0, RB["r"], T, False)
    fwd = math.exp(RB["r"] * T) * (c - p) + 100.0
    assert abs(fwd - 100.0 * math.exp(RB["r"] * T)) <= 2.0 * math.exp(RB["r"] * T) * math.hypot(cse, pse)
    ref_paths = orc.generate_paths_mt(RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], steps, 60_000, 11)
    rm, rse = orc.price_european(ref_paths, 100.0, RB["r"], T, True, step_major=False)
    z = abs(c - rm) / math.hypot(cse, rse)
    assert z <= 2.0, (c, cse, rm, rse, z)
    P.free()


# ------------------------------------------------------------------------------------------------
# LSM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("is_call,poly", [(False, 2), (False, 3), (True, 2), (False, 0), (False, 1)])
def test_lsm_matches_oracle_on_same_paths(eng, orc, is_call, poly):
    """C3-shaped (GBM, 50 exercise dates, dt = 0.02) at a size the oracle finishes in seconds."""
    n, steps, dt = 20_000, 50, 0.02
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, dt, steps, n)
    host = P.to_host_step_major()
    got, se = eng.price_lsm(P, 0.04, 100.0, 1.0, dt, is_call, poly)
    want, v0 = orc.lsm_price(host, 0.04, 100.0, 1.0, dt, is_call, poly, want_v0=True)
    assert abs(got - want) <= 1e-8 * abs(want), (got, want)
    assert abs(se - v0.std(ddof=1) / math.sqrt(n)) <= 1e-6 * se
    P.free()


@pytest.mark.parametrize("n,steps,poly,call", [(300_000, 20, 2, False), (1_200_000, 12, 3, False), (3_000_000, 6, 2, False),
                                               (70_000, 30, 4, False), (5_000, 9, 0, False), (2_400_001, 7, 2, True),
                                               (5_000_000, 5, 2, False), (8_300_003, 4, 1, False), (6_100_000, 4, 4, False),
                                               (524_288, 8, 2, False), (524_289, 8, 2, False), (800_001, 10, 3, True),
                                               (1_048_576, 6, 2, False), (1_048_577, 6, 2, False),
                                               (2_097_152, 5, 2, False), (2_097_153, 5, 2, False)])
def test_lsm_single_launch_sweep_equals_per_date_kernels(eng, n, steps, poly, call):
    """Single GPU runs the whole sweep as one launch (V in registers, workgroups exchanging moments and coefficients
    through sentinel slots); a context with a collective installed takes the per-date kernels.  Same arithmetic,
    different summation order of the moments: prices agree to rounding.  The sizes hit the register-resident variants
    (4, 8 and 16 paths per thread: 512 workgroups x 1024 / 2048 / 4096 paths, each boundary from both sides) and both
    LDS-ring variants (32 and 64 paths per thread, up to 8.37M paths; odd path counts leave the shard's last unit half
    empty and its last workgroups short)."""
    import montecarlooptionspricer_amd as mc
    dt = 1.0 / steps
    P = eng.gbm(SEED + 1, 100.0, 0.04, 0.25, dt, steps, n)
    got, se = eng.price_lsm(P, 0.04, 100.0, 1.0, dt, call, poly)
    assert eng.lsm_one_launch_enabled()
    other = mc.PathEngine(0)
    try:
        other.set_allreduce(lambda ptr, count, stream: None)  # world of one: the sum over ranks is the local value
        Q = other.gbm(SEED + 1, 100.0, 0.04, 0.25, dt, steps, n)
        want, se2 = other.price_lsm(Q, 0.04, 100.0, 1.0, dt, call, poly)
        Q.free()
    finally:
        other.close()
    P.free()
    assert abs(got - want) <= 1e-9 * abs(want), (got, want)
    assert abs(se - se2) <= 1e-9 * abs(se2)


def test_lsm_single_launch_sweep_from_concurrent_host_threads():
    """The reference's driver prices rows from an OpenMP region, one pricer object per thread: cooperative sweeps
    issued at the same time from several contexts must neither dead-lock nor disturb each other."""
    import threading
    import montecarlooptionspricer_amd as mc
    n, steps, dt = 60_000, 25, 0.04
    ref_eng = mc.PathEngine(0)
    P = ref_eng.gbm(SEED, 100.0, 0.04, 0.3, dt, steps, n)
    want = ref_eng.price_lsm(P, 0.04, 100.0, 1.0, dt, False, 2)
    P.free()
    ref_eng.close()
    out, errs = [None] * 4, []

    def work(i):
        try:
            e = mc.PathEngine(0)
            for _ in range(5):
                Q = e.gbm(SEED, 100.0, 0.04, 0.3, dt, steps, n)
                out[i] = e.price_lsm(Q, 0.04, 100.0, 1.0, dt, False, 2)
                Q.free()
            e.close()
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs, errs
    assert all(o == want for o in out), (out, want)


def test_lsm_one_launch_timeout_falls_back_to_per_date_kernels():
    """The hand-shake of the one-launch sweep gives up (forced: spin limit 0 through mcg_debug_lsm_hooks): the void
    result is discarded, the call answers from the per-date kernels, the ctx stays on them for the next eight prices and
    then tries the one-launch sweep again (or at once after mcg_lsm_one_launch_reset); nothing hangs."""
    base = mc.PathEngine(0)
    base.set_allreduce(lambda ptr, count, stream: None)     # a collective (identity) forces the per-date kernels
    P = base.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 20, 300_000)
    want = base.price_lsm(P, 0.04, 100.0, 0.4, 0.02, False, 2)
    P.free()
    base.close()
    e = mc.PathEngine(0)
    Q = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 20, 300_000)
    assert e.lsm_one_launch_enabled()
    one = e.price_lsm(Q, 0.04, 100.0, 0.4, 0.02, False, 2)   # healthy: the one-launch sweep answers
    assert e.lsm_one_launch_enabled() and one == pytest.approx(want, rel=1e-11)
    e.debug_lsm_hooks(spin_limit=0)
    got = e.price_lsm(Q, 0.04, 100.0, 0.4, 0.02, False, 2)
    assert not e.lsm_one_launch_enabled()
    assert got == want                                       # identical kernels, identical summation order
    e.debug_lsm_hooks()
    for _ in range(7):                                       # the next prices stay on the per-date kernels ...
        assert e.price_lsm(Q, 0.04, 100.0, 0.4, 0.02, False, 2) == want and not e.lsm_one_launch_enabled()
    assert e.price_lsm(Q, 0.04, 100.0, 0.4, 0.02, False, 2) == one and e.lsm_one_launch_enabled()   # ... then it is back
    e.debug_lsm_hooks(spin_limit=0)
    assert e.price_lsm(Q, 0.04, 100.0, 0.4, 0.02, False, 2) == want and not e.lsm_one_launch_enabled()
    e.debug_lsm_hooks()
    e.lsm_one_launch_reset()                                 # or at once, on request
    assert e.price_lsm(Q, 0.04, 100.0, 0.4, 0.02, False, 2) == one and e.lsm_one_launch_enabled()
    Q.free()
    e.close()


@pytest.mark.parametrize("poly", [0, 2])
def test_lsm_one_launch_coefficient_blocks_survive_late_readers(poly):
    """The reducing workgroup re-arms the previous date's coefficient block only once ALL moments of the next date are in
    (behind its barrier), never on the strength of the power sums alone, which workgroups send a round ahead: with every
    other workgroup arriving ~20 us late at its coefficient poll the sweep still completes in one launch, without a
    time-out, at the price of the undelayed run.  Orders 0 and 2 are the ones whose first polling wave owns power sums
    only."""
    e = mc.PathEngine(0)
    try:
        P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 1_000_000)
        e.timing_enable(True)
        want = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, poly)
        e.debug_lsm_hooks(poll_delay=5)
        e.timing_reset()
        got = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, poly)
        assert e.timing_get(N.K_LSM_SWEEP)[1] == 1 and e.lsm_one_launch_enabled()
        assert got == want
        P.free()
    finally:
        e.close()


@pytest.mark.parametrize("n", [300_000, 1_000_000, 3_000_000])
def test_lsm_one_launch_sweep_repeats_bit_for_bit(n):
    """The one-launch sweeps sum every moment in a fixed order (threads, waves, workgroups), whatever the timing of the
    hand-shakes (4, 8 paths per thread and the LDS-ring variant): the same paths give the same bits, run after run and
    context after context."""
    import montecarlooptionspricer_amd as mc
    res = []
    for rep in range(2):
        e = mc.PathEngine(0)
        P = e.gbm(SEED + 5, 100.0, 0.04, 0.3, 0.05, 20, n)
        for again in range(2):
            res.append(e.price_lsm(P, 0.04, 100.0, 1.0, 0.05, False, 2))
        assert e.lsm_one_launch_enabled()
        P.free()
        e.close()
    assert all(x == res[0] for x in res[1:]), res


def test_lsm_american_put_bounds(eng):
    """Sanity (not parity): American put >= European put (BS 6.0040 at these parameters)."""
    n, steps, dt = 200_000, 50, 0.02
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, dt, steps, n)
    am, se = eng.price_lsm(P, 0.04, 100.0, 1.0, dt, False, 2)
    eu = bs_price(100.0, 100.0, 0.04, 0.2, 1.0, call=False)
    assert eu - 3 * se < am < eu + 1.5
    P.free()


def test_lsm_edge_cases_match_oracle(eng, orc):
    rs = np.random.RandomState(3)
    # (a) every column OTM for a put -> pure discounting (LSMPricer.cpp:89-94)
    otm = 150.0 + rs.rand(50, 6)
    # (b) maturity shorter than the grid -> j*dt > maturity branch (:43-49)
    mixed = 100.0 * np.exp(np.cumsum(0.1 * rs.standard_normal((300, 9)), axis=1))
    mixed[:, 0] = 100.0
    # (c) a single ITM path per date (rank-1 regression), (d) a single path, (e) one column only
    one_itm = 120.0 + rs.rand(40, 5)
    one_itm[7, :] = 80.0
    single = np.array([[100.0, 90.0, 95.0, 85.0]])
    onecol = np.array([[90.0], [110.0], [100.0]])
    cases = [(otm, 1.0, 0.2, 2), (mixed, 0.35, 0.1, 2), (mixed, 10.0, 0.1, 3), (one_itm, 1.0, 0.25, 2),
             (single, 1.0, 0.25, 2), (onecol, 1.0, 0.25, 2)]
    for arr, maturity, dt, poly in cases:
        P = eng.from_host(arr)
        got, _ = eng.price_lsm(P, 0.04, 100.0, maturity, dt, False, poly)
        want = orc.lsm_price(arr, 0.04, 100.0, maturity, dt, False, poly, step_major=False)
        assert abs(got - want) <= 1e-8 * max(abs(want), 1e-12), (arr.shape, maturity, got, want)
        P.free()
    with pytest.raises(mc.McgError, match="poly_order"):
        P = eng.from_host(mixed)
        eng.price_lsm(P, 0.04, 100.0, 1.0, 0.1, False, 16)


def _near_degenerate_matrix(rs, n_total, n_itm, base, spread, K=100.0):
    """[n_total][3] price matrix (path-major): at the middle date exactly n_itm paths are in the money for a put, within
    a relative spread `spread` of `base`; everything else stays out of the money there.  Column 0 is out of the money
    too (no regression at j = 0), the last column gives every path its own terminal payoff."""
    m = np.empty((n_total, 3))
    m[:, 0] = 1.05 * K
    m[:, 1] = K * (1.02 + 0.2 * rs.rand(n_total))
    m[:, 2] = K * (0.7 + 0.5 * rs.rand(n_total))
    idx = rs.choice(n_total, n_itm, replace=False)
    m[idx, 1] = base * (1.0 + spread * rs.uniform(-1.0, 1.0, n_itm))
    return m


@pytest.mark.parametrize("n_total", [40, 600, 5000])
def test_lsm_near_degenerate_itm_sets_follow_the_reference_rank_rule(eng, orc, n_total):
    """The reference solves every date by Eigen's bdcSvd on raw monomials of S with its default rank threshold
    min(rows, cols) eps sigma_max (LSMPricer.cpp:76).  Dates whose in-the-money prices nearly coincide sit where that rule
    decides the fit: full rank down to a relative spread of about 1e-5 (the interpolating / least-squares quadratic),
    rank 2 below it.  The device has to follow it through every execution shape: one wavefront (<= 256 paths), one
    workgroup (<= 1024), the one-launch sweep, and -- with a collective installed -- the per-date kernels, which re-fit
    such a date inside the solve kernel.  Tolerance 1e-6 relative on the price: the reference-side solve itself is only
    defined to eps x cond(A) ~ 1e-16 x 1e10 there (oracle vs 60-digit arithmetic: tests/test_oracle_models.py)."""
    rs = np.random.RandomState(11)
    worst = 0.0
    for base in (90.0, 99.9, 60.0):
        for n_itm in (2, 3, 4, 5):
            for spread in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
                m = _near_degenerate_matrix(rs, n_total, n_itm, base, spread)
                P = eng.from_host(m)
                got, _ = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.5, False, 2)
                P.free()
                want = orc.lsm_price(m, 0.04, 100.0, 1.0, 0.5, False, 2, step_major=False)
                err = abs(got - want) / abs(want)
                worst = max(worst, err)
                assert err <= 1e-6, (n_total, base, n_itm, spread, got, want)
    # the same through the per-date kernels (a collective, here the identity, selects them): they re-fit such a date
    # from the one set of moments they carry (taken about the previous date's mean) -- same rule, same price
    other = mc.PathEngine(0)
    try:
        other.set_allreduce(lambda ptr, count, stream: None)
        for base, n_itm, spread in ((90.0, 3, 1e-4), (99.9, 2, 1e-6), (60.0, 5, 1e-3), (90.0, 4, 1e-7), (99.9, 3, 1e-5)):
            m = _near_degenerate_matrix(rs, n_total, n_itm, base, spread)
            Q = other.from_host(m)
            b = other.price_lsm(Q, 0.04, 100.0, 1.0, 0.5, False, 2)[0]
            Q.free()
            want = orc.lsm_price(m, 0.04, 100.0, 1.0, 0.5, False, 2, step_major=False)
            assert abs(b - want) <= 1e-6 * abs(want), ("per-date kernels", n_total, base, n_itm, spread, b, want)
    finally:
        other.close()


def test_lsm_near_degenerate_large_shards(eng, orc):
    """The same rank rule in the one-launch sweeps for big shards (16 and 32-64 paths per thread) and in the per-date
    kernels WITHOUT a collective (forced by switching the one-launch sweep off), on 2.2M paths."""
    rs = np.random.RandomState(12)
    n_total = 2_200_000
    for base, n_itm, spread in ((90.0, 3, 1e-4), (99.9, 5, 1e-6), (60.0, 4, 1e-3)):
        m = _near_degenerate_matrix(rs, n_total, n_itm, base, spread)
        want = orc.lsm_price(m, 0.04, 100.0, 1.0, 0.5, False, 2, step_major=False)
        P = eng.from_host(m)
        got = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.5, False, 2)[0]
        P.free()
        assert abs(got - want) <= 1e-6 * abs(want), ("one launch", base, n_itm, spread, got, want)
    big = _near_degenerate_matrix(rs, 5_000_000, 4, 90.0, 1e-5)          # 64 paths per thread
    want_big = orc.lsm_price(big, 0.04, 100.0, 1.0, 0.5, False, 2, step_major=False)
    P = eng.from_host(big)
    got = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.5, False, 2)[0]
    P.free()
    assert abs(got - want_big) <= 1e-6 * abs(want_big), ("one launch, 5M", got, want_big)
    e = mc.PathEngine(0)
    e.debug_lsm_hooks(spin_limit=0)                           # first call times out -> per-date kernels answer
    try:
        P = e.from_host(m)
        got = e.price_lsm(P, 0.04, 100.0, 1.0, 0.5, False, 2)[0]
        assert not e.lsm_one_launch_enabled()
        assert abs(got - want) <= 1e-6 * abs(want), ("per-date", got, want)
        P.free()
    finally:
        e.close()


@pytest.mark.parametrize("poly", [4, 5, 6, 8])
def test_lsm_high_orders_follow_the_reference_rank_rule(eng, orc, poly):
    """Raw monomials up to S^8 at S ~ 100 span sixteen orders of magnitude: Eigen's threshold truncates them on EVERY date
    (the fitted polynomial is not the degree-p least-squares fit any more but its projection on the leading singular
    directions).  The device recognises these dates (lsm_solve_nb's estimate) and reproduces the truncated solve."""
    n, steps, dt = 3000, 12, 1.0 / 12
    P = eng.gbm(SEED + 5, 100.0, 0.04, 0.3, dt, steps, n)
    host = P.to_host_step_major()
    got, _ = eng.price_lsm(P, 0.04, 100.0, 1.0, dt, False, poly)
    P.free()
    want = orc.lsm_price(host, 0.04, 100.0, 1.0, dt, False, poly)
    assert abs(got - want) <= 2e-6 * abs(want), (poly, got, want)


@pytest.mark.parametrize("poly", [9, 12, 15])
def test_lsm_orders_beyond_eight_are_served(eng, orc, poly):
    """LSM::PredictOptionPrice takes any polyOrder (LSMPricer.cpp:9-17).  Orders above 8 run one launch per exercise date
    (k_lsm_date) whatever the path count, every date re-fitted by the reference's rank rule -- here also through the class
    API (mcg_compat_lsm_price), which used to refuse them."""
    n, steps, dt = 3000, 12, 1.0 / 12
    P = eng.gbm(SEED + 5, 100.0, 0.04, 0.3, dt, steps, n)
    host = P.to_host_step_major()
    got, _ = eng.price_lsm(P, 0.04, 100.0, 1.0, dt, False, poly)
    P.free()
    want = orc.lsm_price(host, 0.04, 100.0, 1.0, dt, False, poly)
    assert abs(got - want) <= 5e-6 * abs(want), (poly, got, want)
    from montecarlooptionspricer_amd import compat
    small = np.ascontiguousarray(host[:, :300].T)
    got_c = compat.LSM().PredictOptionPrice(small.tolist(), 0.04, 100.0, 1.0, dt, False, poly)
    want_c = orc.lsm_price(small, 0.04, 100.0, 1.0, dt, False, poly, step_major=False)
    assert abs(got_c - want_c) <= 5e-6 * abs(want_c), (poly, got_c, want_c)


def test_from_host_roundtrip_is_exact(eng):
    rs = np.random.RandomState(0)
    a = rs.rand(777, 13) * 100
    P = eng.from_host(a)
    assert np.array_equal(P.to_host(), a)
    assert np.array_equal(P.to_host_step_major(), a.T)
    P.free()


# ------------------------------------------------------------------------------------------------
# the reference's class API (drop-in boundary)
# ------------------------------------------------------------------------------------------------
def test_class_api_generate_paths(eng, orc):
    hist = synthetic_history(1001, seed=42)
    mc.set_compat_seed(SEED)
    rv = mc.RoughVolatility()
    a = rv.GenerateStockPricePaths(hist, 30, 250)        # 250 paths/row is the reference's production size
    assert a.shape == (250, 31) and np.isfinite(a).all() and (a[:, 0] == hist[-1]).all()
    assert np.array_equal(a, rv.GenerateStockPricePaths(hist, 30, 250))  # fixed seed -> reproducible
    p = orc.estimate_params(hist)                         # host estimators == reference-pinned oracle
    P = eng.rbergomi(SEED, p["S0"], 0.04, p["xi"], p["H"], p["eta"], p["rho"], DT, 30, 250)
    assert np.array_equal(P.to_host(), a)
    P.free()
    mc.set_compat_seed(None)
    b = rv.GenerateStockPricePaths(hist, 30, 250)         # unseeded like the reference: differs run to run
    assert not np.array_equal(a, b)
    with pytest.raises(mc.McgError, match="Historical prices vector too small."):
        rv.GenerateStockPricePaths([100.0], 30, 250)
    assert rv.GenerateStockPricePaths(hist, 30, 0).shape == (0, 31)
    z = rv.GenerateStockPricePaths(hist, 0, 5)
    assert z.shape == (5, 1) and (z == hist[-1]).all()
    # two-point history: rho = NaN in the reference -> every step is NaN there too
    d = rv.GenerateStockPricePaths([100.0, 101.0], 4, 3)
    assert (d[:, 0] == 101.0).all() and np.isnan(d[:, 1:]).all()


def test_class_api_statistics_match_reference_faithful_sample(orc):
    """Statistical parity with the reference's own algorithm (mt mode): terminal mean and the mean
    of log-returns' variance agree within sampling error on 20k paths."""
    hist = synthetic_history(1001, seed=42)
    mc.set_compat_seed(99)
    a = mc.RoughVolatility().GenerateStockPricePaths(hist, 60, 20_000)
    mc.set_compat_seed(None)
    b = orc.generate_paths_mt_hist(hist, 60, 20_000, 5)
    for x in (a, b):
        assert np.isfinite(x).all()
    za = np.log(a[:, -1] / a[:, 0])
    zb = np.log(b[:, -1] / b[:, 0])
    se_m = math.sqrt(za.var() / len(za) + zb.var() / len(zb))
    assert abs(za.mean() - zb.mean()) <= 3.0 * se_m
    assert abs(za.var() / zb.var() - 1.0) <= 0.06


def test_class_api_lsm(eng, orc):
    hist = synthetic_history(300, seed=1)
    mc.set_compat_seed(5)
    paths = mc.RoughVolatility().GenerateStockPricePaths(hist, 40, 250)
    mc.set_compat_seed(None)
    K = float(hist[-1])
    got = mc.LSM().PredictOptionPrice(paths, 0.04, K, 40 / 365.0, DT, False, 2)   # PredictionGen.cpp:700-704,:790
    want = orc.lsm_price(paths, 0.04, K, 40 / 365.0, DT, False, 2, step_major=False)
    assert abs(got - want) <= 1e-8 * abs(want)
    with pytest.raises(mc.McgError, match="LSM::PredictOptionPrice: Empty pricePaths."):
        mc.LSM().PredictOptionPrice(np.zeros((0, 0)), 0.04, K, 1.0, DT, False, 2)


def test_timing_counters(eng):
    from montecarlooptionspricer_amd import _native as N
    eng.timing_enable(True)
    eng.timing_reset()
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, 100_000, payoff=(100.0, True))
    ms, n = eng.timing_get(N.K_GBM)
    assert n == 1 and ms > 0
    eng.timing_enable(False)
    P.free()


# ------------------------------------------------------------------------------------------------
# collectives (single GPU box: world_size 1, plus a callback that emulates a second identical rank)
# ------------------------------------------------------------------------------------------------
def test_allreduce_callback_plumbing():
    """A callback that doubles the buffer == two ranks holding identical shards: same mean, n doubles,
    std-err shrinks by sqrt(2); LSM moments doubled leave the fit (hence the price) unchanged."""
    import torch
    e = mc.PathEngine(0, stream=torch.cuda.current_stream().cuda_stream)
    calls = []

    def doubler(ptr, count, stream):
        from montecarlooptionspricer_amd.engine import _DevView
        t = torch.as_tensor(_DevView(ptr, count), device="cuda:0")
        t.mul_(2.0)
        calls.append(count)

    P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 30_000)
    m1, se1 = e.price_european(P, 100.0, 0.04, 1.0, False)
    l1, lse1 = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    e.set_allreduce(doubler)
    m2, se2 = e.price_european(P, 100.0, 0.04, 1.0, False)
    l2, lse2 = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    assert abs(m2 - m1) <= 1e-13 * m1 and abs(se2 * math.sqrt(2.0) - se1) <= 1e-4 * se1
    assert abs(l2 - l1) <= 1e-12 * l1 and abs(lse2 * math.sqrt(2.0) - lse1) <= 1e-4 * lse1
    # 3p+2 = 8 moments between two launches of the per-date kernel -- 51 columns, 51 launches, none spare -- and the
    # sweep's fault flag once
    assert calls.count(3) == 2 and calls.count(8) == 50 and calls.count(1) == 1
    e.set_allreduce(None)
    m3, _ = e.price_european(P, 100.0, 0.04, 1.0, False)
    assert m3 == m1
    P.free()
    e.close()


def test_torch_distributed_and_builtin_rccl_world_size_one():
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        e = mc.PathEngine(0, stream=torch.cuda.current_stream().cuda_stream)
        P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 20_000, payoff=(100.0, False))
        base = mc.PathEngine(0)
        Q = base.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 20_000)
        want_e = base.price_european(Q, 100.0, 0.04, 1.0, False)
        want_l = base.price_lsm(Q, 0.04, 100.0, 1.0, 0.02, False, 2)
        e.use_torch_distributed()
        # (the collective path runs the per-date kernels, the single-GPU base the one-launch sweep: same arithmetic,
        # the moments are summed in a different order)
        got_l = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
        assert got_l == pytest.approx(want_l, rel=1e-11)
        R = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 20_000, payoff=(100.0, False))
        assert e.price_european(R, 100.0, 0.04, 1.0, False) == want_e
        e.set_allreduce(None)
        e.init_rccl(0, 1, lambda uid: uid)                   # built-in RCCL communicator, 1 rank
        assert e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2) == got_l   # both collectives: identical kernels
        for x in (P, Q, R):
            x.free()
        e.close()
        base.close()
    finally:
        dist.destroy_process_group()


def test_gbm_large_steps_take_the_general_exp_path(eng, orc):
    """sigma*sqrt(dt) large enough that the host cannot bound |drift + vol z| by 0.125: the kernel
    variant with full range reduction runs, and must match the oracle just the same."""
    for sigma, dt, steps in [(0.8, 0.05, 40), (3.0, 0.25, 12), (0.2, 0.02, 50)]:
        P = eng.gbm(SEED, 100.0, 0.04, sigma, dt, steps, 777, path_begin=5)
        want = orc.paths_gbm(SEED, 100.0, 0.04, sigma, dt, steps, 5, 777)
        assert rel_err(P.to_host_step_major(), want) < 1e-11
        P.free()


# ------------------------------------------------------------------------------------------------
# AsymptoticAnalysis
# ------------------------------------------------------------------------------------------------
def test_asymptotic_matches_reference_goldens_and_oracle(eng, orc):
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "asymptotic.npz"))
    aa = mc.AsymptoticAnalysis()
    for which, is_call, maturity, dt, sigma, div, K, r, want in d["cases"]:
        m = d["paths_dirty"] if which else d["paths"]
        got = aa.PredictOptionPrice(m, r, K, maturity, dt, bool(is_call), sigma, div)   # class API, host paths
        assert abs(got - want) <= 1e-13 * max(abs(want), 1e-300), (got, want)
    # device-resident path, larger: 200k x 64 GBM
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 64, 200_000)
    host = P.to_host_step_major()
    for is_call in (False, True):
        got = eng.price_asymptotic(P, 0.04, 100.0, 64 * DT, DT, is_call, 0.2, 0.08)
        want = orc.asymptotic_price(host, 0.04, 100.0, 64 * DT, DT, is_call, 0.2, 0.08)
        assert abs(got - want) <= 1e-12 * want, (got, want)
    P.free()
    assert aa.PredictOptionPrice(np.zeros((0, 0)), 0.04, 100.0, 1.0, DT, False, 0.2, 0.0) == 0.0
    with pytest.raises(mc.McgError, match="AsymptoticAnalysis: Volatility must be positive."):
        aa.PredictOptionPrice(d["paths"], 0.04, 100.0, 1.0, DT, False, 0.0, 0.0)


# ------------------------------------------------------------------------------------------------
# MartingaleOptimization
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("is_call,poly,iters", [(False, 2, 5), (True, 2, 5), (False, 3, 2), (False, 2, 1), (False, 0, 4),
                                                (False, 4, 5), (False, 6, 3), (True, 8, 5), (False, 10, 5), (True, 15, 2)])
def test_martingale_matches_oracle(eng, orc, is_call, poly, iters):
    """Orders >= 4: Eigen's rank threshold truncates the raw monomials of the refit (MartingaleOptimizationPricer.cpp:166);
    the device re-fits those about the samples' mean and reproduces the truncated solve (lsm_solve_centered), tolerance
    as for LSM's high orders."""
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 40, 20_000)
    host = P.to_host_step_major()
    for maturity in (40 * DT, 25.5 * DT):
        got = eng.price_martingale(P, 0.04, 100.0, maturity, DT, is_call, poly, iters)
        want = orc.martingale_price(host, 0.04, 100.0, maturity, DT, is_call, poly, iters)
        assert np.allclose(got, want, rtol=1e-8 if poly < 4 else (2e-6 if poly <= 8 else 5e-6), atol=1e-12), (got, want)
    P.free()


def test_batch_rows_order_four_follow_the_rank_rule(eng, orc):
    """The batched rows at polyOrder 4 (the batch's maximum): LSM and MartingaleOptimization columns against the oracle."""
    rs = np.random.RandomState(9)
    rows = _driver_rows(8, rs)
    got = eng.batch_price_rows(rows, n_paths=250, r=0.04, dt=DT, num_branches=10, poly_order=4, max_iterations=5, seed=5)
    for i, d in enumerate(rows):
        P = eng.rbergomi(5, d["S0"], 0.04, d["xi"], d["H"], d["eta"], d["rho"], DT, d["n_steps"], 250, path_begin=i << 32)
        host = P.to_host_step_major()
        P.free()
        call = bool(d["is_call"])
        want = [orc.lsm_price(host, 0.04, d["strike"], d["maturity"], DT, call, 4),
                orc.martingale_price(host, 0.04, d["strike"], d["maturity"], DT, call, 4, 5)[0]]
        assert np.allclose(got[i, 2:], want, rtol=5e-6, atol=1e-9), (i, d, got[i], want)


def test_martingale_class_api_and_errors(orc):
    hist = synthetic_history(300, seed=1)
    mc.set_compat_seed(5)
    paths = mc.RoughVolatility().GenerateStockPricePaths(hist, 40, 250)      # production shape: 250 paths per row
    mc.set_compat_seed(None)
    K = float(hist[-1])
    mo = mc.MartingaleOptimization()
    got = mo.PredictOptionPrice(paths, 0.04, K, 40 / 365.0, DT, False, 2)   # PredictionGen.cpp:791 (default 5 iterations)
    want = orc.martingale_price(paths, 0.04, K, 40 / 365.0, DT, False, 2, 5, step_major=False)[0]
    assert abs(got - want) <= 1e-8 * abs(want)
    with pytest.raises(mc.McgError, match="MartingaleOptimization: Empty pricePaths."):
        mo.PredictOptionPrice(np.zeros((0, 0)), 0.04, K, 1.0, DT, False, 2)
    with pytest.raises(mc.McgError, match="MartingaleOptimization: maxIterations must be positive."):
        mo.PredictOptionPrice(paths, 0.04, K, 1.0, DT, False, 2, 0)


# ------------------------------------------------------------------------------------------------
# BranchingProcesses
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("is_call,branches,stride", [(False, 10, 1), (True, 10, 1), (False, 3, 1), (False, 7, 4), (False, 0, 1)])
def test_branching_matches_oracle_philox_mode(eng, orc, is_call, branches, stride):
    steps = 40
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, steps, 5000, path_begin=11)
    host = P.to_host_step_major()
    ex = np.arange(0, steps, stride, dtype=np.int32)          # the driver passes 0..steps-1
    for maturity in (steps * DT, 22.5 * DT):
        got = eng.price_branching(P, 0.04, 100.0, maturity, DT, is_call, branches, ex, seed=99)
        want = orc.branching_price(host, 0.04, 100.0, maturity, DT, is_call, branches, ex, 99, mode="philox", path_begin=11)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (got, want)
        assert got[1] <= got[2] + 1e-12 and abs(got[0] - 0.5 * (got[1] + got[2])) < 1e-13
    P.free()


def test_branching_trailing_exercise_index_beyond_the_matrix(eng, orc):
    """An exercise list whose tail lies behind the maturity may name columns that do not exist: the reference never
    touches them (its `t > maturity` break comes first, BranchingProcessPricer.cpp:57-59, :97-99) but still compares
    every date with exerciseTimes.back() (:104), so the last real column takes the continuation branch with an empty
    scan (:110).  Oracle "mt" (the reference's loops), oracle "philox" and the device agree on the lower bound; the
    device equals oracle "philox" on the upper bound, with no out-of-range gather."""
    steps = 12
    P = eng.gbm(SEED, 100.0, 0.04, 0.25, DT, steps, 3000)
    host = P.to_host_step_major()
    ex = np.array(list(range(steps + 1)) + [steps + 5, 10_000], dtype=np.int32)   # 0..12 exist; 17 and 10000 do not
    got = eng.price_branching(P, 0.04, 100.0, steps * DT, DT, False, 6, ex, seed=3)
    want = orc.branching_price(host, 0.04, 100.0, steps * DT, DT, False, 6, ex, 3, mode="philox")
    ref_like = orc.branching_price(host, 0.04, 100.0, steps * DT, DT, False, 6, ex, 3, mode="mt")
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (got, want)
    assert got[1] == pytest.approx(ref_like[1], rel=1e-13)
    with pytest.raises(mc.McgError, match="outside"):                              # inside the maturity it IS an error
        eng.price_branching(P, 0.04, 100.0, 1.0, DT, False, 6, np.array([0, 5, 40], dtype=np.int32), seed=3)
    P.free()


def test_handles_survive_their_ctx_and_narrow_matrices_upload(eng):
    """mcg_finalize before mcg_paths_free (an exception skipped P.free()): the handle is orphaned, a late free only
    deletes it.  And a two-column matrix with more than 65535*32 paths goes through the layout kernels in slabs."""
    e = mc.PathEngine(0)
    P = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 8, 1000)
    L, h = e._L, P._h
    e._live.discard(P)                 # bypass the Python-side bookkeeping: exercise the C ABI's own guard
    e.close()
    out = np.empty((1000, 9))
    import ctypes as C
    assert L.mcg_paths_to_host(h, out.ctypes.data_as(C.POINTER(C.c_double))) != 0
    assert b"outlived" in L.mcg_last_error()
    assert L.mcg_paths_free(h) == 0
    P._h = None
    n = 65535 * 32 + 777
    m = np.random.RandomState(2).rand(n, 2)
    Q = eng.from_host(m)
    assert np.array_equal(Q.to_host(), m) and np.array_equal(Q.to_host_step_major(), m.T)
    Q.free()


def test_branching_class_api_and_errors(orc):
    hist = synthetic_history(300, seed=1)
    mc.set_compat_seed(5)
    paths = mc.RoughVolatility().GenerateStockPricePaths(hist, 40, 250)
    K = float(hist[-1])
    ex = list(range(40))                                       # PredictionGen.cpp:780-783
    bp = mc.BranchingProcesses()
    got = bp.PredictOptionPrice(paths, 0.04, K, 40 / 365.0, DT, False, 10, ex)      # :789
    want = orc.branching_price(paths, 0.04, K, 40 / 365.0, DT, False, 10, ex, 5, mode="philox", step_major=False)[0]
    assert abs(got - want) <= 1e-12 * abs(want)
    mc.set_compat_seed(None)
    for bad, msg in [((np.zeros((0, 0)), K, ex), "Empty pricePaths."), ((paths, K, []), "No exercise times."),
                     ((paths, 0.0, ex), "Strike must be positive.")]:
        with pytest.raises(mc.McgError, match="BranchingProcesses: " + msg):
            bp.PredictOptionPrice(bad[0], 0.04, bad[1], 1.0, DT, False, 10, bad[2])


# ------------------------------------------------------------------------------------------------
# batched driver rows
# ------------------------------------------------------------------------------------------------
def _driver_rows(n, rs):
    rows = []
    for i in range(n):
        steps = int(rs.choice([5, 16, 21, 40, 63, 64, 100, 130]))
        S0 = float(rs.uniform(20, 400))
        rows.append(dict(S0=S0, xi=float(rs.uniform(0.01, 0.3)), H=float(rs.uniform(0.05, 0.6)),
                         eta=float(rs.uniform(0.0, 2.0)), rho=-0.3, strike=S0 * float(rs.uniform(0.9, 1.1)),
                         maturity=steps / 252.0 * float(rs.choice([1.0, 1.0, 0.69])), sigma=float(rs.uniform(0.1, 0.6)),
                         dividend=float(rs.uniform(0.0, 0.1)), n_steps=steps, is_call=int(rs.randint(0, 2))))
    return rows


def test_batch_rows_match_oracle_row_by_row(eng, orc):
    """Row i of mcg_batch_price_rows against the ORACLE: the row's 250 paths are the rBergomi paths (i << 32) + p of the
    same seed (generated by the single-contract entry point -- itself pinned to the oracle element-wise in
    test_rbergomi_paths_match_oracle -- and downloaded), and each of the driver's four columns
    (src/core/PredictionGen.cpp:788-791, :809-814) is recomputed on them by the CPU restatement: AsymptoticAnalysis
    (pinned bit-exact to the compiled reference), BranchingProcesses in philox mode (same resampling draws), LSM and
    MartingaleOptimization.  (kappa comes from a device DFT in the batch and from the host FFT in the single-contract
    generator: paths ~1e-13 apart, which the regressions amplify: rtol 1e-7.)"""
    rs = np.random.RandomState(4)
    rows = _driver_rows(24, rs)
    rows[3]["sigma"] = 0.0          # AsymptoticAnalysis would throw  -> the driver writes zeros for the row
    rows[7]["n_steps"] = 0          # "No time steps"                 -> zeros
    rows[11]["rho"] = float("nan")  # two-point history               -> NaN paths -> zeros
    got = eng.batch_price_rows(rows, n_paths=250, r=0.04, dt=DT, num_branches=10, poly_order=2, max_iterations=5, seed=77)
    assert got.shape == (24, 4)
    for i, d in enumerate(rows):
        if i in (3, 7, 11):
            assert (got[i] == 0.0).all()
            continue
        P = eng.rbergomi(77, d["S0"], 0.04, d["xi"], d["H"], d["eta"], d["rho"], DT, d["n_steps"], 250, path_begin=i << 32)
        host = P.to_host_step_major()
        call = bool(d["is_call"])
        ex = np.arange(d["n_steps"], dtype=np.int32)
        want = [orc.asymptotic_price(host, 0.04, d["strike"], d["maturity"], DT, call, d["sigma"], d["dividend"]),
                orc.branching_price(host, 0.04, d["strike"], d["maturity"], DT, call, 10, ex, 77, mode="philox",
                                    path_begin=i << 32)[0],
                orc.lsm_price(host, 0.04, d["strike"], d["maturity"], DT, call, 2),
                orc.martingale_price(host, 0.04, d["strike"], d["maturity"], DT, call, 2, 5)[0]]
        assert np.allclose(got[i], want, rtol=1e-7, atol=1e-9), (i, d, got[i], want)
        # and the engine's own single-contract entry points on the same device matrix agree with the batch kernels
        single = [eng.price_asymptotic(P, 0.04, d["strike"], d["maturity"], DT, call, d["sigma"], d["dividend"]),
                  eng.price_branching(P, 0.04, d["strike"], d["maturity"], DT, call, 10, ex, 77)[0],
                  eng.price_lsm(P, 0.04, d["strike"], d["maturity"], DT, call, 2)[0],
                  eng.price_martingale(P, 0.04, d["strike"], d["maturity"], DT, call, 2, 5)[0]]
        P.free()
        assert np.allclose(got[i], single, rtol=1e-7, atol=1e-9), (i, d, got[i], single)


def test_batch_rows_longer_than_the_row_kernels_are_priced_not_zeroed(eng):
    """A row of more than 1020 steps (PredictionGen.cpp:718 sets steps = floor(maturity * 252) with no cap) does not fit
    the row kernels' tables: it is priced through the single-contract entry points on the same Philox ids -- four real
    prices, equal to calling those entry points by hand -- while its neighbours still take the batch kernels."""
    rows = _driver_rows(3, np.random.RandomState(9))
    rows[1].update(n_steps=1100, maturity=1100 / 252.0, is_call=0)
    rows[1]["strike"] = rows[1]["S0"] * 1.05
    got = eng.batch_price_rows(rows, n_paths=250, r=0.04, dt=DT, num_branches=10, poly_order=2, max_iterations=5, seed=31)
    assert (got[1][1:] > 0.0).all() and np.isfinite(got).all()
    d = rows[1]
    P = eng.rbergomi(31, d["S0"], 0.04, d["xi"], d["H"], d["eta"], d["rho"], DT, d["n_steps"], 250, path_begin=1 << 32)
    ex = np.arange(d["n_steps"], dtype=np.int32)
    single = [eng.price_asymptotic(P, 0.04, d["strike"], d["maturity"], DT, False, d["sigma"], d["dividend"]),
              eng.price_branching(P, 0.04, d["strike"], d["maturity"], DT, False, 10, ex, 31)[0],
              eng.price_lsm(P, 0.04, d["strike"], d["maturity"], DT, False, 2)[0],
              eng.price_martingale(P, 0.04, d["strike"], d["maturity"], DT, False, 2, 5)[0]]
    P.free()
    assert np.array_equal(got[1], single), (got[1], single)
    alone = eng.batch_price_rows([rows[0], rows[2]], n_paths=250, r=0.04, dt=DT, num_branches=10, poly_order=2, max_iterations=5, seed=31)
    assert np.array_equal(got[0], alone[0])      # (row 2's Philox ids depend on its index: only row 0 is comparable)


def test_batch_rows_arguments(eng):
    rows = _driver_rows(2, np.random.RandomState(1))
    assert eng.batch_price_rows([], n_paths=250).shape == (0, 4)
    with pytest.raises(mc.McgError, match="n_paths"):
        eng.batch_price_rows(rows, n_paths=0)
    # beyond the row kernels' limits (256 paths per row, order 4) every row takes the single-contract entry points
    big = eng.batch_price_rows(rows, n_paths=300, poly_order=5, seed=5)
    for i, d in enumerate(rows):
        P = eng.rbergomi(5, d["S0"], 0.04, d["xi"], d["H"], d["eta"], d["rho"], DT, d["n_steps"], 300, path_begin=i << 32)
        ex = np.arange(d["n_steps"], dtype=np.int32)
        call = bool(d["is_call"])
        single = [eng.price_asymptotic(P, 0.04, d["strike"], d["maturity"], DT, call, d["sigma"], d["dividend"]),
                  eng.price_branching(P, 0.04, d["strike"], d["maturity"], DT, call, 10, ex, 5)[0],
                  eng.price_lsm(P, 0.04, d["strike"], d["maturity"], DT, call, 5)[0],
                  eng.price_martingale(P, 0.04, d["strike"], d["maturity"], DT, call, 5, 5)[0]]
        P.free()
        assert np.array_equal(big[i], single), (i, big[i], single)
    with pytest.raises(mc.McgError, match="maxIterations must be positive"):
        eng.batch_price_rows(rows, max_iterations=0)
    a = eng.batch_price_rows(rows, seed=5)
    assert np.array_equal(a, eng.batch_price_rows(rows, seed=5)) and not np.array_equal(a, eng.batch_price_rows(rows, seed=6))


# ------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes, through size-independent properties
# ------------------------------------------------------------------------------------------------
def test_full_size_c2_properties(eng):
    """C2 (10M x 252): Black-Scholes within 2 std-errs, forward = S0 e^{rT} by put-call parity on the same
    paths, and two 5M shards combine to exactly the sums of the 10M run (checksum of checksums)."""
    n = 10_000_000
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, n, payoff=(100.0, True))
    c, cse = eng.price_european(P, 100.0, 0.04, 1.0, True)
    p, pse = eng.price_european(P, 100.0, 0.04, 1.0, False)
    P.free()
    assert abs(c - bs_price(100.0, 100.0, 0.04, 0.2, 1.0)) <= 2.0 * cse
    fwd = math.exp(0.04) * (c - p) + 100.0
    assert abs(fwd - 100.0 * math.exp(0.04)) <= 2.0 * math.exp(0.04) * math.hypot(cse, pse)
    disc = math.exp(-0.04)
    parts = []
    for b in (0, n // 2):
        Q = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, n // 2, path_begin=b, payoff=(100.0, True))
        m, se = eng.price_european(Q, 100.0, 0.04, 1.0, True)
        Q.free()
        k = n // 2
        mean = m / disc
        parts.append((mean * k, (se / disc) ** 2 * k * (k - 1) + k * mean * mean, float(k)))
    from montecarlooptionspricer_amd.sharding import combine_sums, price_from_sums
    m2, se2 = price_from_sums(combine_sums(parts), disc)
    assert abs(m2 - c) <= 1e-10 * c and abs(se2 - cse) <= 1e-6 * cse


def test_full_size_c3_c4_properties(eng):
    """C3 (1M x 50 LSM): American put above the European put on the same paths, below the strike.
    C4 (4M x 512 rBergomi): martingale E[S_T] = S0 e^{rT} via put-call parity; finite price."""
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 1_000_000)
    am, ase = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    eu, ese = eng.price_european(P, 100.0, 0.04, 1.0, False)
    P.free()
    assert eu - 3 * ese < am < 100.0 and am > 6.0
    T = 512 * DT
    R = eng.rbergomi(SEED, 100.0, 0.04, RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 512, 4_000_000, payoff=(100.0, True))
    c, cse = eng.price_european(R, 100.0, 0.04, T, True)
    p, pse = eng.price_european(R, 100.0, 0.04, T, False)
    R.free()
    fwd = math.exp(0.04 * T) * (c - p) + 100.0
    assert abs(fwd - 100.0 * math.exp(0.04 * T)) <= 2.0 * math.exp(0.04 * T) * math.hypot(cse, pse)
    assert math.isfinite(c) and c > 0


@pytest.mark.parametrize("steps", [252, 512])
def test_rough_regime_prices_and_structure_vs_compiled_reference_sample(eng, steps):
    """C4 / C5 parameters (H = 0.1, eta = 1.9) against the committed sample of the COMPILED REFERENCE
    (tests/golden/rough_regime_reference.json: 2e6 / 1e6 paths through the reference's own private members,
    oracle/gen_rough_fixture.py).  North-star bar: |price - ref| <= 2 MC standard errors (combined), call and put, on
    4M device paths.  (The statistics that see the Volterra / forward-variance structure in the price matrix itself --
    E[S_T], realised variance, clustering of squared returns -- are judged over 16 seeds in tests/test_gpu_statistics.py.)"""
    import json
    import os
    from oracle.binding import STAT_NAMES
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rough_regime_reference.json")))
    p, fix = fx["params"], fx["samples"][str(steps)]
    fm, fse = np.array(fix["mean"]), np.array(fix["std_err"])
    K, n = p["strike"], 4_000_000
    for is_call, idx in ((True, 1), (False, 2)):
        P = eng.rbergomi(SEED, p["S0"], p["r"], p["xi"], p["H"], p["eta"], p["rho"], DT, steps, n, payoff=(K, is_call))
        m, se = eng.price_european(P, K, 0.0, 0.0, is_call)        # r = 0, T = 0: the undiscounted mean payoff
        P.free()
        z = (m - fm[idx]) / math.hypot(se, fse[idx])
        assert abs(z) <= 2.0, (steps, STAT_NAMES[idx], m, se, fm[idx], fse[idx], z)
