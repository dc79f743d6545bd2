"""Round 4 (run with -m gpu on an MI355X): the per-date LSM route's self-announcing exchange, the batched rows in chunks
under a memory budget with the driver's six columns, the measurement aids of the bench line, and EIGHT ranks of one
sharded job as eight threads of this process on GPU 0 (the pool's process guard allows six GPU processes: eight rank
PROCESSES on one card are not possible there; eight rank threads are -- and one host thread per GPU is a deployment the
reference's own OpenMP driver suggests)."""
import math

import numpy as np
import pytest

import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd import _native as N

pytestmark = pytest.mark.gpu

SEED, DT = 20251031, 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)


@pytest.fixture()
def eng():
    e = mc.PathEngine(0)
    yield e
    e.close()


# ------------------------------------------------------------------------------------------------
# k_lsm_date: a partial moment that has not arrived cannot be mistaken for one
# ------------------------------------------------------------------------------------------------
def _per_date(e):
    e.set_allreduce(lambda ptr, count, stream: None)   # world size 1: the identity; forces the per-date route


def test_per_date_route_queues_exactly_the_launches_it_needs():
    """M = steps + 1 launches for a sweep no date of which is re-fitted, one all-reduce BETWEEN two of them and none
    spent on a sweep that is over (r3: M + 4 + M/32 launches, each with its collective)."""
    e = mc.PathEngine(0)
    calls = []
    e.set_allreduce(lambda ptr, count, stream: calls.append(count))
    P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 300_000)
    mc.stats(reset=True)
    e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    s = mc.stats()
    assert s["lsm_per_date_sweeps"] == 1 and s["lsm_per_date_launches"] == 51 and s["lsm_per_date_refits"] == 0
    assert calls.count(8) == 50 and calls.count(1) == 1 and calls.count(3) == 1   # moments, the fault flag, the final sums
    # order 5: the reference's rank rule truncates the raw monomials on every date -> every date with a path in the money
    # takes two launches; still no launch beyond the sweep's end
    calls.clear()
    mc.stats(reset=True)
    e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 5)
    s = mc.stats()
    assert 51 < s["lsm_per_date_launches"] <= 2 * 51 and s["lsm_per_date_refits"] == s["lsm_per_date_launches"] - 51
    assert calls.count(17) == s["lsm_per_date_launches"] - 1
    P.free()
    e.close()


def test_per_date_route_never_prices_from_partials_that_did_not_arrive():
    """One workgroup withholds its partial moments at one exercise date (a store that never lands): the consumer finds
    the reserved NaN, waits its bounded wait, raises the sweep's fault flag -- mcg_price_lsm fails with MCG_ERR_HIP.
    Before round 4 the slot would have held the PREVIOUS date's partial and the price would have been silently wrong."""
    e = mc.PathEngine(0)
    P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 400_000)
    want = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)          # the one-launch sweep
    _per_date(e)
    clean = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    assert abs(clean[0] - want[0]) <= 1e-9 * want[0]
    mc.stats(reset=True)
    for date, wg in ((37, 5), (12, 0), (1, 300)):
        e.debug_lsm_date_fault(mode=1, date=date, workgroup=wg, spin_limit=2000)
        with pytest.raises(mc.McgError, match="partial moments of a workgroup did not arrive") as ei:
            e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
        assert ei.value.status == 3
    assert mc.stats()["lsm_per_date_faults"] == 3
    e.debug_lsm_date_fault(mode=0)
    again = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)          # the ctx is usable afterwards, same bits
    assert again == clean
    P.free()
    e.close()


def test_per_date_route_waits_for_partials_that_land_late():
    """The same workgroup sends its partials ~0.4 ms AFTER it has drawn its ticket (a store that lands late): whoever
    reduces its group finds the reserved NaN first and waits for the value -- the price is the usual one, bit for bit."""
    e = mc.PathEngine(0)
    P = e.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 64, 300_000)
    _per_date(e)
    clean = e.price_lsm(P, RB["r"], 100.0, 64 * DT, DT, False, 2)
    for date, wg in ((40, 3), (63, 15), (2, 16)):     # (15 / 16: the last member of a ticket group and the first of the next)
        e.debug_lsm_date_fault(mode=2, date=date, workgroup=wg, delay=100)
        assert e.price_lsm(P, RB["r"], 100.0, 64 * DT, DT, False, 2) == clean
    e.debug_lsm_date_fault(mode=0)
    P.free()
    e.close()


# ------------------------------------------------------------------------------------------------
# batched driver rows: chunks under a memory budget, six columns (missing #4 / weak #3)
# ------------------------------------------------------------------------------------------------
def _rows(n, rs, lo=5, hi=127):
    st = rs.randint(lo, hi, n)
    S0 = rs.uniform(20, 400, n)
    return [dict(S0=float(S0[i]), xi=float(rs.uniform(0.01, 0.3)), H=float(rs.uniform(0.3, 0.6)), eta=float(rs.uniform(0.01, 0.06)),
                 rho=-0.3, strike=float(S0[i] * rs.uniform(0.9, 1.1)), maturity=int(st[i]) / 252.0, sigma=float(rs.uniform(0.1, 0.6)),
                 dividend=0.08, n_steps=int(st[i]), is_call=int(rs.randint(0, 2))) for i in range(n)]


def test_batch_rows_prices_do_not_depend_on_the_chunking(eng):
    """The same 600 rows (short ones and a few of every longer LDS class) in one chunk and in many (budget 8 MB): every
    price identical to the last bit -- a row's Philox ids are (i << 32) + p whatever chunk it falls into."""
    rs = np.random.RandomState(21)
    rows = _rows(600, rs)
    for i, st in ((17, 200), (99, 300), (250, 520), (251, 700), (400, 1000)):
        rows[i].update(n_steps=st, maturity=st / 252.0)
    mc.stats(reset=True)
    one = eng.batch_price_rows(rows, seed=9)
    s1 = mc.stats(reset=True)
    eng.debug_batch_budget(8 << 20)
    many = eng.batch_price_rows(rows, seed=9)
    s2 = mc.stats()
    eng.debug_batch_budget(0)
    assert np.array_equal(one, many)
    assert np.isfinite(one).all() and (one[:, 1:] >= 0.0).all()
    assert s1["batch_rows"] == 600 and s2["batch_rows"] == 600 and s1["batch_chunks"] == 4     # one per LDS class present
    assert s2["batch_chunks"] > 12 and s2["batch_peak_workspace_bytes"] <= (8 << 20) + 256 * 1001 * 8 + (1 << 16)


def test_batch_rows_one_long_row_does_not_inflate_the_workspace(eng):
    """20 000 short rows and ONE 1 000-step row: until round 3 the matrix was padded to the longest row (41 GB); now every
    row owns a block of its own size."""
    rs = np.random.RandomState(5)
    rows = _rows(20_000, rs)
    rows[12_345].update(n_steps=1000, maturity=1000 / 252.0)
    arr = mc.make_rows(rows)
    mc.stats(reset=True)
    out = eng.batch_price_rows(arr, seed=3)
    s = mc.stats()
    assert s["batch_peak_workspace_bytes"] < 4 << 30, s
    need = sum(256 * (r["n_steps"] + 1) * 8 for r in rows)
    assert s["batch_peak_workspace_bytes"] < 1.1 * need
    assert np.isfinite(out).all() and (out[12_345] > 0.0).all()


def test_batch_rows_four_hundred_thousand_rows_complete(eng):
    """The reference's production use is hundreds of thousands of rows (PredictionGen.cpp:542-546)."""
    n = 400_000
    rs = np.random.RandomState(8)
    arr = (N.Row * n)()
    a = np.frombuffer(arr, dtype=np.dtype([(k, "<f8") for k in ("S0", "xi", "H", "eta", "rho", "strike", "maturity", "sigma", "dividend")]
                                           + [("n_steps", "<i4"), ("is_call", "<i4")]))
    st = rs.randint(5, 127, n)
    a["S0"] = rs.uniform(20, 400, n)
    a["xi"] = rs.uniform(0.01, 0.3, n)
    a["H"] = rs.uniform(0.3, 0.6, n)
    a["eta"] = rs.uniform(0.01, 0.06, n)
    a["rho"] = -0.3
    a["strike"] = a["S0"] * rs.uniform(0.9, 1.1, n)
    a["maturity"] = st / 252.0
    a["sigma"] = rs.uniform(0.1, 0.6, n)
    a["dividend"] = 0.08
    a["n_steps"] = st
    a["is_call"] = rs.randint(0, 2, n)
    eng.debug_batch_budget(6 << 30)
    mc.stats(reset=True)
    out = eng.batch_price_rows(arr, seed=4)
    s = mc.stats()
    eng.debug_batch_budget(0)
    assert s["batch_rows"] == n and s["batch_chunks"] >= 9 and s["batch_peak_workspace_bytes"] <= (6 << 30) + (1 << 20)
    assert np.isfinite(out).all() and (out[:, 1:] >= 0.0).all() and (out[:, 2] > 0.0).mean() > 0.9
    # the first 64 rows alone give the same prices (their Philox ids do not depend on the rest of the call)
    head = (N.Row * 64)(*arr[:64])
    assert np.array_equal(eng.batch_price_rows(head, seed=4), out[:64])


def test_batch_rows_six_columns_like_the_driver(eng):
    """mcg_row_build + mcg_batch_price_rows6 == what PredictionGen.cpp writes per row (:471-477, :809-816): four model
    prices, twenty_day_vol, twenty_day_momentum -- and six zeros for a row it skips or whose pricers throw."""
    from oracle.binding import synthetic_history
    hists = [synthetic_history(400, seed=s, s0=80.0 + 10 * s, sigma=0.15 + 0.05 * s) for s in range(6)]
    built = [mc.row_build(h, float(h[-1]), dte, dist, typ, 0.03)
             for h, dte, dist, typ in zip(hists, (30, 60, 90, 45, 1, 200), (0.02, -0.05, 0.0, 0.1, 0.0, 0.03), (1, 0, 1, 0, 1, 0))]
    built.append(mc.row_build(hists[0][:15], float(hists[0][14]), 40, 0.0, 1))    # < 21 prices: sigma = 0 -> AsymptoticAnalysis throws -> zeros
    rows = [b[0] for b in built]
    feats = np.array([b[1] for b in built])
    four = eng.batch_price_rows(rows, seed=12)
    six = eng.batch_price_rows(rows, seed=12, features=feats)
    assert six.shape == (7, 6)
    for i in (0, 1, 2, 3, 5):
        assert np.array_equal(six[i, :4], four[i]) and tuple(six[i, 4:]) == built[i][1] and (six[i, :4] >= 0).all() and six[i, 2] > 0
        assert rows[i]["sigma"] == built[i][1][0] > 0
    assert (six[4] == 0.0).all() and rows[4]["n_steps"] == 0          # dte = 1: no time step (:721-731)
    assert (six[6] == 0.0).all() and feats[6, 0] == 0.0                # the pricer block throws: ",0,0,0,0,0,0" (:792-805)


# ------------------------------------------------------------------------------------------------
# measurement aids of the bench line
# ------------------------------------------------------------------------------------------------
def test_write_ceiling_probe_and_in_kernel_clock(eng):
    gbs, ms = eng.probe_write_ceiling(4_000_000, 252, reps=3)
    assert 2500.0 < gbs < 8000.0 and abs(gbs - 8.0 * 253 * 4_000_000 / (ms * 1e-3) / 1e9) < 1e-6 * gbs
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, 4_000_000, payoff=(100.0, True))
    assert eng.generator_clock()["stamping_workgroups"] == 0      # a measurement aid, off unless armed
    P.free()
    eng.generator_clock_arm(True)
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, 4_000_000, payoff=(100.0, True))
    c = eng.generator_clock()
    P.free()
    assert c["stamping_workgroups"] >= 16 and 0.5 < c["GHz_min"] <= c["GHz_median"] <= c["GHz_max"] < 2.7, c
    tiny = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 16, 1000)     # too few workgroups to stamp: zeros, not stale values
    assert eng.generator_clock()["stamping_workgroups"] == 0
    tiny.free()
    eng.generator_clock_arm(False)


def test_bench_line_attributes_its_own_variance():
    """bench.py's roofline object carries what separates a slow board from a regression: the board's own write ceiling,
    measured in the same process right after the timed region, the fraction of it the generator reaches, and the shader
    clock stamped inside the timed kernel."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                       cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert 3000.0 < r["board_write_ceiling_GBs"] < 8000.0
    assert abs(r["frac_of_board_ceiling"] - r["achieved"] / r["board_write_ceiling_GBs"]) < 1e-12 and 0.5 < r["frac_of_board_ceiling"] < 1.05
    c = r["shader_clock_GHz"]
    assert c["stamping_workgroups"] >= 32 and 0.8 < c["GHz_min"] <= c["GHz_median"] <= c["GHz_max"] < 2.7


# ------------------------------------------------------------------------------------------------
# world size 8: eight rank threads of one process on GPU 0
# ------------------------------------------------------------------------------------------------
def _single_rank_reference(JOBS):
    e = mc.PathEngine(0)
    P = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, JOBS["euro_paths"], payoff=(100.0, True))
    euro = e.price_european(P, 100.0, 0.04, 1.0, True)
    P.free()
    P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, JOBS["lsm_steps"], JOBS["lsm_paths"])
    lsm = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    P.free()
    T = JOBS["rb_steps"] * DT
    P = e.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, JOBS["rb_steps"], JOBS["rb_paths"])
    rb = e.price_lsm(P, RB["r"], 100.0, T, DT, False, 2)
    rb_eu = e.price_european(P, 100.0, RB["r"], T, False)
    P.free()
    e.close()
    return dict(euro=euro, gbm_lsm=lsm, rb_lsm=rb, rb_euro_put=rb_eu)


@pytest.mark.parametrize("mode", ["shm", "ipc", "callback"])
def test_eight_rank_threads_equal_single_rank(tmp_path, mode):
    """BASELINE.json configs[4]'s world size, on the one GPU there is: eight ranks of one sharded job as eight threads of
    ONE child process (tests/thread_ranks_worker.py; eight rank processes would exceed the pool's limit of six GPU
    processes), each with its own ctx and -- shm / ipc -- its own stream and hardware queue: the eight one-launch sweeps
    are resident on the GPU together and exchange their per-date moments through an 8-row mailbox, in the host segment
    or in device memory with every rank pushing into all eight copies.  "callback": the per-date route with an all-reduce
    over the eight threads.  Every rank must hold the single-rank price; shards are unequal and, for rBergomi,
    even-aligned."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from thread_ranks_worker import JOBS
    world = 8
    want = _single_rank_reference(JOBS)
    out_file = str(tmp_path / "ranks.json")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(here, "thread_ranks_worker.py"), str(world), mode, out_file], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:]
    j = json.load(open(out_file))
    ranks, calls, s = j["ranks"], j["calls"], j["stats"]
    for r, out in enumerate(ranks):
        for key, tol in (("euro", 1e-12), ("rb_euro_put", 1e-12), ("gbm_lsm", 1e-9), ("rb_lsm", 1e-9)):
            assert abs(out[key][0] - want[key][0]) <= tol * abs(want[key][0]), (mode, r, key, out[key], want[key])
            assert abs(out[key][1] - want[key][1]) <= max(tol, 1e-9) * abs(want[key][1]), (mode, r, key)
        assert out["shard"][0] % 2 == 0
        if mode == "callback":
            assert out["comm"]["kind"] == "callback"
            # exactly one all-reduce between two launches of the per-date kernel, one of the fault flag, one of the sums
            assert out["gbm_lsm_sweep_launches"] == JOBS["lsm_steps"] + 1 + 1
            assert calls[r] == {"8": JOBS["lsm_steps"] + JOBS["rb_steps"], "1": 2, "3": 4}
        else:
            assert out["one_launch_enabled"] and out["gbm_lsm_sweep_launches"] == 1 and out["rb_lsm_sweep_launches"] == 1, p.stdout[-2000:]
            assert out["comm"]["n_ranks"] == world and out["comm"]["seen_ranks"] == world and out["comm"]["rank"] == r
            assert out["comm"]["kind"] == ("shm+peer-memory mailbox" if mode == "ipc" else "shm")
            assert out["peer_mailbox"] == (mode == "ipc")
    assert sum(out["shard"][1] for out in ranks) == JOBS["rb_paths"]
    assert all(out["euro"] == ranks[0]["euro"] and out["rb_lsm"] == ranks[0]["rb_lsm"] for out in ranks)   # the same bits on every rank
    if mode != "callback":
        assert s["lsm_one_launch_sweeps"] == 2 * world and s["lsm_one_launch_timeouts"] == 0 and s["shm_barrier_failures"] == 0
        assert s["peer_mailbox_enabled"] == (world if mode == "ipc" else 0)


# ------------------------------------------------------------------------------------------------
# BranchingProcesses: rows of F beyond one 2 MB slice
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_paths,steps,branches,ex_kind", [
    (300_001, 8, 10, "all"),        # 3 slices: one launch per exercise date (k_branch_date<3>)
    (524_289, 6, 7, "sparse"),      # 3 slices, two Philox blocks per path and date, a sparse exercise list
    (300_001, 6, 3, "shuffled"),    # a list that is not ascending: its LAST entry, not its largest, ends the branching
    (1_100_000, 5, 10, "all"),      # 5 slices: beyond k_branch_date's range -- round 5: k_branch_date_binned (indices sorted by slice per thread)
    (4_000_001, 4, 10, "all"),      # rows of 32 MB (the shape VERDICT r4 / r5 name, fewer dates) -- round 6: the XCD-affine route
    #                                 (k_branch_date_xcd: every XCD gathers from its eighth of the row, eight cells per path summed in order)
    (1_300_000, 4, 7, "sparse4"),   # binned with two Philox blocks per path and date, a sparse list
    (2_500_000, 3, 3, "all"),       # XCD-affine with one block (three of its four indices used)
    (2_000_001, 3, 7, "all"),       # XCD-affine with two blocks, a ragged last tile
    (300_001, 5, 13, "all"),        # more than twelve branches: indices are not kept, no slices
])
def test_branching_rows_beyond_one_slice_match_oracle(n_paths, steps, branches, ex_kind):
    """k_branch_date / k_branch_date_binned / k_branch_date_xcd / the sliced k_branch_bounds against the oracle in philox mode (the same
    resampling draws): the slices only change WHEN an index is gathered, and the order in which a path's branch values are
    summed."""
    from oracle.binding import Oracle
    orc = Oracle()
    e = mc.PathEngine(0)
    P = e.gbm(SEED, 100.0, 0.04, 0.3, DT, steps, n_paths, path_begin=6)
    host = P.to_host_step_major()
    ex = {"all": np.arange(steps), "sparse": np.array([0, 2, 3, 5]), "shuffled": np.array([4, 1, 5, 0, 2]), "sparse4": np.array([0, 1, 3])}[ex_kind].astype(np.int32)
    for is_call, maturity in ((False, steps * DT), (True, (steps - 1.5) * DT)):
        got = e.price_branching(P, 0.04, 100.0, maturity, DT, is_call, branches, ex, seed=41)
        want = orc.branching_price(host, 0.04, 100.0, maturity, DT, is_call, branches, ex, 41, mode="philox", path_begin=6)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (n_paths, is_call, got, want)
    P.free()
    e.close()


def test_batch_rows6_arguments(eng):
    rows = _rows(3, np.random.RandomState(2))
    four = eng.batch_price_rows(rows, seed=5)
    import ctypes as C
    arr = mc.make_rows(rows)
    out = np.full((3, 6), -1.0)
    dp = C.POINTER(C.c_double)
    # features2 = NULL: the four prices, zeros in the feature columns
    assert eng._L.mcg_batch_price_rows6(eng._ctx, arr, None, 3, 250, 0.04, DT, 10, 2, 5, 5, out.ctypes.data_as(dp)) == 0
    assert np.array_equal(out[:, :4], four) and (out[:, 4:] == 0.0).all()
    assert eng._L.mcg_batch_price_rows6(eng._ctx, arr, None, 0, 250, 0.04, DT, 10, 2, 5, 5, out.ctypes.data_as(dp)) == 0   # no rows: nothing to do
    for bad in ((eng._ctx, arr, None, 3, 0, 0.04, DT, 10, 2, 5, 5), (eng._ctx, arr, None, 3, 250, 0.04, DT, 10, 16, 5, 5),
                (eng._ctx, arr, None, 3, 250, 0.04, DT, 10, 2, 0, 5), (eng._ctx, arr, None, 3, 250, 0.04, 0.0, 10, 2, 5, 5),
                (eng._ctx, None, None, 3, 250, 0.04, DT, 10, 2, 5, 5), (None, arr, None, 3, 250, 0.04, DT, 10, 2, 5, 5)):
        assert eng._L.mcg_batch_price_rows6(*bad, out.ctypes.data_as(dp)) == 1, bad
    with pytest.raises(mc.McgError, match="features must be"):
        eng.batch_price_rows(rows, features=np.zeros((2, 2)))
