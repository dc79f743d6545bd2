"""Round 6 (run with -m gpu on an MI355X): the kernel instantiations every headline number comes from, compared ELEMENT-WISE
with the oracle AT THE SHAPES THEY ARE TIMED AT.

bench.py times k_gbm_paths<true,3,2> on 10M x 252 (two adjacent paths per lane, the small-exponent mode, fused payoff) and
queues its store-only-twin <false,3,2> as the ramp; C4 is k_rbergomi_fft<5,2,true> on 4M x 512, the C5 shard's generator
k_rbergomi_fft<4,2,false> on 8M x 252.  Which instantiation a launch takes is a function of (n_paths, n_steps, parameters,
payoff) alone (kernels_gbm.hip launch_gbm, kernels_rbergomi.hip launch_variant), so launching with bench.py's arguments IS
launching what it times.  Rounds 1-5 compared these variants with the oracle at 4-7 steps, and 252 steps only through the
one-path-per-lane variants at <= 4 096 paths; round 4's store-data hazard lived in exactly <.,3,2> and corrupted rows 0, 1, 5.

No 20 GB download: column slices of the device matrix are read through mcg_paths_info's pointer (a zero-copy torch view)
and every row of each slice is compared with oracle.paths_*(seed, ..., path_begin=slice_begin, n).
Reference: src/models/RoughVolatility.cpp:354-364 (stepping loop), :264-309 (fractionalGaussian, forwardVariance)."""
import math
import os

import numpy as np
import pytest

import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd import _native as N
from montecarlooptionspricer_amd.engine import _DevView
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu

SEED, DT = 20251031, 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)


@pytest.fixture()
def eng():
    e = mc.PathEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _device_matrix(P):
    """The matrix where it lies: a torch view [n_steps + 1][ld] over mcg_paths_info's device pointer (no copy)."""
    import torch
    t = torch.as_tensor(_DevView(P.device_ptr, (P.n_steps + 1) * P.ld), device=torch.device("cuda", 0))
    return t.view(P.n_steps + 1, P.ld)


def _columns(P, eng, begin, count):
    """All n_steps + 1 rows of columns [begin, begin + count) as a host array."""
    eng.synchronize()
    return _device_matrix(P)[:, begin:begin + count].cpu().numpy()


def _max_rel(got, want):
    return float(np.max(np.abs(got - want) / np.abs(want)))


def _payoff_moments(last_row, K, is_call):
    """(mean, standard error) of the undiscounted payoff over a host row, by exactly rounded sums (math.fsum)."""
    pay = np.maximum(last_row - K, 0.0) if is_call else np.maximum(K - last_row, 0.0)
    n = pay.size
    m = math.fsum(pay) / n
    var = max(0.0, (math.fsum(pay * pay) - n * m * m) / (n - 1))
    return m, math.sqrt(var / n)


def test_c2_timed_launch_slices_match_oracle(eng, orc):
    """bench.py's one_pass() and ramp_launch() at C2 (bench.py `one_pass`, `ramp_launch`: 10M x 252, payoff=(K, True) ->
    k_gbm_paths<true,3,2>; no payoff -> <false,3,2>).  Three column slices of each launch, all 253 rows: the first 1 024
    columns (two workgroups of 512), 1 024 straddling two workgroup boundaries in the middle of the grid, and the last 513
    real columns together with the 128 padded ones behind them (ld = 10 000 128: the tail workgroup stores
    unconditionally and holds the stream's next paths there) -- against the oracle's Philox restatement at 1e-11.
    The fused payoff sums of the timed launch against (i) a k_payoff_sums pass over the twin's stored last row and (ii) exactly
    rounded host sums of the downloaded last row."""
    n, steps, K, r, sigma, S0 = 10_000_000, 252, 100.0, 0.04, 0.2, 100.0
    eng.timing_enable(True)
    eng.timing_reset()
    P = eng.gbm(SEED, S0, r, sigma, DT, steps, n, payoff=(K, True))     # the timed launch
    Q = eng.gbm(SEED, S0, r, sigma, DT, steps, n)                       # its ramp twin
    eng.synchronize()
    assert eng.timing_get(N.K_GBM)[1] == 2
    eng.timing_enable(False)
    assert P.ld == 10_000_128 and Q.ld == P.ld
    mid = ((n // 512) // 2) * 512 - 500                                 # [mid, mid + 1024) crosses columns 512 k and 512 (k + 1)
    slices = [(0, 1024), (mid, 1024), (n - 513, 513 + (P.ld - n))]
    assert mid % 512 != 0 and (mid // 512) != ((mid + 1023) // 512) - 1
    for M, name in ((P, "<true,3,2>"), (Q, "<false,3,2>")):
        for begin, count in slices:
            got = _columns(M, eng, begin, count)
            want = orc.paths_gbm(SEED, S0, r, sigma, DT, steps, begin, count)
            assert got.shape == want.shape == (steps + 1, count)
            assert np.all(got[0] == S0), (name, begin)
            err = _max_rel(got, want)
            assert err < 1e-11, (name, begin, count, err)
    # the two launches wrote the same matrix, bit for bit (the payoff epilogue does not touch the stores)
    import torch
    assert torch.equal(_device_matrix(P)[:, :n], _device_matrix(Q)[:, :n])
    # fused sums of the timed launch | k_payoff_sums over the twin's stored row | exactly rounded host sums
    T = steps * DT
    fused = eng.price_european(P, K, r, T, True)
    eng.timing_enable(True)
    eng.timing_reset()
    stored = eng.price_european(Q, K, r, T, True)
    assert eng.timing_get(N.K_PAYOFF)[1] >= 1                            # a pass over the stored row, not cached sums
    eng.timing_enable(False)
    last = _device_matrix(Q)[steps, :n].cpu().numpy()
    m, se = _payoff_moments(last, K, True)
    disc = math.exp(-r * T)
    for name, (gm, gse) in (("fused", fused), ("k_payoff_sums", stored)):
        assert abs(gm - disc * m) <= 1e-12 * disc * m, (name, gm, disc * m)
        assert abs(gse - disc * se) <= 1e-9 * disc * se, (name, gse, disc * se)
    assert abs(fused[0] - 9.9251) < 3 * fused[1]                         # Black-Scholes, d1 = 0.3, d2 = 0.1
    P.free()
    Q.free()


def _rb_share_columns(M):
    """Columns per workgroup share of the FFT generator: rb_pairs_per_block(Mz) pairs (rbergomi_device.hpp)."""
    lanes_per_pair = M // 16
    return 2 * 4 * (64 // lanes_per_pair)


@pytest.mark.parametrize("n,steps,payoff", [(4_000_000, 512, (100.0, True)), (8_000_000, 252, None)],
                         ids=["c4_fft<5,2,true>_4Mx512", "c5gen_fft<4,2,false>_8Mx252"])
def test_rbergomi_timed_launch_slices_match_oracle(eng, orc, n, steps, payoff):
    """C4 (4M x 512, fused call payoff -> k_rbergomi_fft<5,2,true>; Mphi = 1024 / Mz = 512) and the C5 shard's generator
    (8M x 252 -> k_rbergomi_fft<4,2,false>) as bench.py launches them.  The kernel is persistent: 2 workgroups per CU take
    the first gridDim.x shares by blockIdx and every later one from a ticket counter.  256-column slices, all rows, at the
    first share, at the first TICKET-drawn share (share index gridDim.x), in the middle, and at the last share: 1e-9 against
    the oracle's restatement of the same Philox draws."""
    import torch
    Mz = 1 << (steps - 1).bit_length()
    per_share = _rb_share_columns(Mz)
    grid = 2 * torch.cuda.get_device_properties(0).multi_processor_count
    n_shares = (n // 2 + per_share // 2 - 1) // (per_share // 2)
    assert n_shares > 4 * grid
    P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, n, payoff=payoff)
    assert P.ld == n
    begins = [0, grid * per_share, (n_shares // 2) * per_share - 128, n - 256]
    for begin in begins:
        got = _columns(P, eng, begin, 256)
        want = orc.paths_rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, begin, 256)
        assert got.shape == want.shape == (steps + 1, 256)
        assert np.all(got[0] == RB["S0"])
        err = _max_rel(got, want)
        assert err < 1e-9, (begin, err)
    if payoff is not None:
        K, is_call = payoff
        T = steps * DT
        fused = eng.price_european(P, K, RB["r"], T, is_call)
        last = _device_matrix(P)[steps, :n].cpu().numpy()
        m, se = _payoff_moments(last, K, is_call)
        disc = math.exp(-RB["r"] * T)
        assert abs(fused[0] - disc * m) <= 1e-12 * disc * m, (fused, disc * m)
        assert abs(fused[1] - disc * se) <= 1e-9 * disc * se, (fused, disc * se)
    P.free()


_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.skipif(not os.path.exists(os.path.join(_GOLDEN, "lsm.npz")),
                    reason="tests/golden/lsm.npz absent: captured only where an Eigen3 exists (oracle/gen_golden.py --eigen); this image "
                           "has none -- the device's LSM is pinned to the restatement, not to Eigen")
def test_device_lsm_and_martingale_match_the_eigen_fixtures(eng):
    """The day the Eigen-backed fixtures exist, the DEVICE is held to them directly (LSMPricer.cpp:76,
    MartingaleOptimizationPricer.cpp:166), case by case at the tolerance stored with the case."""
    d = np.load(os.path.join(_GOLDEN, "lsm.npz"))
    for name in d["names"]:
        r, K, maturity, dt, is_call, poly, tol = d[f"{name}_args"]
        P = eng.from_host(d[f"{name}_paths"])
        got, _ = eng.price_lsm(P, r, K, maturity, dt, bool(is_call), int(poly))
        P.free()
        want = float(d[f"{name}_price"])
        assert abs(got - want) <= max(tol, 1e-8) * abs(want), (str(name), got, want)
    d = np.load(os.path.join(_GOLDEN, "martingale.npz"))
    for name in d["names"]:
        r, K, maturity, dt, is_call, poly, iters, tol = d[f"{name}_args"]
        P = eng.from_host(d[f"{name}_paths"])
        got = eng.price_martingale(P, r, K, maturity, dt, bool(is_call), int(poly), int(iters))[0]
        P.free()
        want = float(d[f"{name}_price"])
        assert abs(got - want) <= max(tol, 1e-8) * abs(want), (str(name), got, want)


# ---- the reference's driver unchanged: calls of many host threads answered together (csrc/coalesce.hpp) -----------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_unchanged_driver(n_rows, threads, coalesce, out_file, timeout=600, max_slots=None):
    import json
    import subprocess
    subprocess.run(["make", "build/unchanged_driver"], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_DYNAMIC="false")
    cmd = [os.path.join(ROOT, "build", "unchanged_driver"), str(n_rows), str(coalesce), out_file]
    if max_slots is not None:
        cmd += ["20251031", str(max_slots)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    rows = np.loadtxt(out_file)
    assert rows.shape == (n_rows, 6)
    return line, rows


def test_unchanged_driver_128_threads_equals_the_single_thread_run(tmp_path):
    """tests/cpp/unchanged_driver.cpp = the row loop of src/core/PredictionGen.cpp:542-570 / :736-737 / :788-791 against the
    reference's own headers: 128 OpenMP threads, each constructing the five classes per row and calling them in turn, 1 500 rows
    of 250 paths x 5..126 steps under mcg_compat_set_seed.  Calls of different threads are answered in shared launches
    (coalesced rounds); the prices of every row must equal -- to the last bit -- those of the same program run on ONE thread
    (rounds of one call), rows that throw in the reference's classes (a one-price history, sigma = 0) must come out as
    exceptions per row, not as a failed run, and nothing may dead-lock (the run is timed out)."""
    n = 1500
    one, a = _run_unchanged_driver(n, 1, 1, str(tmp_path / "one.txt"))
    many, b = _run_unchanged_driver(n, 128, 1, str(tmp_path / "many.txt"))
    assert many["threads"] == 128 and one["threads"] == 1
    assert np.array_equal(a, b), np.argwhere(a != b)[:5]
    threw = a[:, 1] == 1
    expect = np.array([(i % 211 == 17) or (i % 257 == 29) for i in range(n)])
    assert np.array_equal(threw, expect)
    assert np.all(a[threw, 2:] == 0.0) and np.all(a[~threw, 2:] >= 0.0) and np.all(np.isfinite(a))
    assert one["priced"] == many["priced"] == n - int(expect.sum())
    # ... and against the per-thread route (every call a launch and a synchronisation on the calling thread's own context:
    # the single-contract kernels instead of the row kernels): the same prices to 1e-9
    _legacy, c = _run_unchanged_driver(300, 8, 0, str(tmp_path / "legacy.txt"))
    assert np.array_equal(c[:, :2], a[:300, :2])
    assert np.allclose(c[:, 2:], a[:300, 2:], rtol=1e-9, atol=1e-12), np.max(np.abs(c[:, 2:] - a[:300, 2:]))
    # more calling threads than the arena has slots (512 by default; 6 here, mcg_debug_coalesce_slots, with 24 threads): the threads
    # that get none price on a context of their own -- no error, no dead-lock, the same prices
    crowd, d = _run_unchanged_driver(600, 24, 1, str(tmp_path / "crowd.txt"), max_slots=6)
    assert crowd["threads"] == 24 and crowd["own_context_calls"] > 0 and crowd["calls"] > 0
    assert np.array_equal(d[:, :2], a[:600, :2]) and np.allclose(d[:, 2:], a[:600, 2:], rtol=1e-9, atol=1e-12)


def test_coalesced_calls_of_python_threads_match_sequential_calls(orc):
    """Sixteen host threads (ctypes releases the GIL around every call) hammer the class API with DIFFERENT matrices and kinds
    at once -- generated paths, uploaded matrices the slot has never seen, all four pricers, orders 0..4, calls the row
    kernels do not serve (300 paths; order 7; a custom exercise-date list) in between -- and every answer must equal the one
    the same call gives alone.  mcg_stats must show rounds that answered several calls."""
    import threading
    from oracle.binding import synthetic_history
    mc.set_compat_seed(77)
    mc.stats(reset=True)
    hists = [synthetic_history(300 + 40 * k, seed=k + 1) for k in range(16)]

    def work(k, out):
        rv, lsm, aa, mo, bp = mc.RoughVolatility(), mc.LSM(), mc.AsymptoticAnalysis(), mc.MartingaleOptimization(), mc.BranchingProcesses()
        res = []
        for it in range(6):
            steps = 5 + 17 * ((k + it) % 7)
            n_paths = 300 if (k + it) % 11 == 0 else 250 - (k % 5)
            p = rv.GenerateStockPricePaths(hists[k], steps, n_paths)
            K = float(hists[k][-1]) * (1.0 + 0.02 * ((k % 5) - 2))
            T = steps / 252.0
            res.append(float(p.sum()))
            res.append(aa.PredictOptionPrice(p, 0.04, K, T, DT, k % 2 == 0, 0.2 + 0.01 * k, 0.08))
            res.append(lsm.PredictOptionPrice(p, 0.04, K, T, DT, k % 2 == 1, (k + it) % 5))
            res.append(mo.PredictOptionPrice(p, 0.04, K, T, DT, False, 2, 1 + (it % 3)))
            res.append(bp.PredictOptionPrice(p, 0.04, K, T, DT, False, 10, list(range(steps))))
            q = p[::-1].copy() * 1.01                        # a matrix the thread's slot has never seen
            res.append(lsm.PredictOptionPrice(q, 0.04, K, T, DT, False, 2))
            res.append(lsm.PredictOptionPrice(q, 0.04, K, T, DT, False, 7))                      # order 7: the thread's own context
            res.append(bp.PredictOptionPrice(q, 0.04, K, T, DT, False, 10, list(range(0, steps, 2))))   # custom dates: likewise
            res.append(aa.PredictOptionPrice(p, 0.04, K, T, DT, False, 0.3, 0.0))                # back to the first matrix
        out[k] = res

    alone = {}
    for k in range(16):
        work(k, alone)
    s0 = mc.stats(reset=True)
    # (one thread alone: rounds of one call -- or of two, when its own call of a kind meets that kind's prefetched request in the queue)
    assert s0["coalesced_calls"] > 0 and s0["coalesced_peak_calls_per_round"] <= 2 and s0["coalesced_fallbacks"] > 0
    # the first pricer call on a generated matrix queued the driver's other three ahead of time; those whose arguments the later
    # calls repeated (BranchingProcesses always; LSM at order 2; MartingaleOptimization at >= 2 iterations) were answered from that
    assert s0["coalesced_prefetched"] >= 3 * 16 and 16 <= s0["coalesced_prefetch_hits"] < s0["coalesced_prefetched"]
    together = {}
    threads = [threading.Thread(target=work, args=(k, together)) for k in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads)
    s1 = mc.stats()
    mc.set_compat_seed(None)
    for k in range(16):
        assert together[k] == alone[k], (k, [i for i, (x, y) in enumerate(zip(together[k], alone[k])) if x != y][:5])
    assert s1["coalesced_calls"] == s0["coalesced_calls"] and s1["coalesced_rounds"] < s1["coalesced_calls"]
    assert s1["coalesced_prefetch_hits"] == s0["coalesced_prefetch_hits"]
    assert s1["coalesced_peak_calls_per_round"] >= 2


def test_rank_thread_contexts_closed_one_after_another_do_not_stall_or_leak(tmp_path):
    """rank THREADS of one process borrow each other's peer-memory mailbox by pointer.  Closed together after a barrier
    every owner finds its borrowers gone; closed ONE AFTER ANOTHER by a single thread (engines closed in a loop, the garbage
    collector, atexit) every owner but the last still has borrowers -- round 5 waited 5 s per owner and then kept the megabyte.
    Now the mailbox is handed to the last borrower, which frees it: eight sequential closes take well under a second, the
    hand-overs are counted (mcg_stats.peer_mailbox_kept), and the job's prices are what they are with the ranks closed together."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out_file = str(tmp_path / "seq.json")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(here, "thread_ranks_worker.py"), "8", "ipc", out_file, "seqclose"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    j = json.load(open(out_file))
    assert all(r["peer_mailbox"] for r in j["ranks"]), [r["comm"] for r in j["ranks"]]
    assert len(j["close_seconds"]) == 8 and max(j["close_seconds"]) < 1.0, j["close_seconds"]
    assert j["stats"]["peer_mailbox_kept"] == 7, j["stats"]            # every owner but the last left its mailbox to a borrower
    assert all(r["rb_lsm"] == j["ranks"][0]["rb_lsm"] and r["euro"] == j["ranks"][0]["euro"] for r in j["ranks"])


def test_sharded_per_date_route_at_order_5_equals_the_single_rank_price():
    """the per-date LSM route sizes its batches by the share of re-fitted dates it has seen and may overshoot the end
    of the sweep; sharded, every launch but the first is preceded by one all-reduce of the moments.  Two rank threads (callback
    all-reduce over the threads), order 5 -- every in-the-money date re-fits --, 200 001 GBM paths x 40 dates: both ranks hold
    the single-context price (1e-9: the sums are formed in a different order) and the same bits as each other, and on each rank
    launches = moment all-reduces + 1 (mcg_stats / the callback's own count), overshoot launches included."""
    import threading
    import torch
    from montecarlooptionspricer_amd.sharding import shard_range
    n, steps, dt, order, world = 200_001, 40, 0.025, 5, 2
    nm = 3 * (order + 1) - 1
    with mc.PathEngine(0) as e:
        e.set_allreduce(lambda ptr, count, stream: None)          # an identity collective selects the per-date route
        P = e.gbm(SEED, 100.0, 0.04, 0.2, dt, steps, n)
        want = e.price_lsm(P, 0.04, 100.0, 1.0, dt, False, order)
        P.free()
    bar, lock, parts, res, errs = threading.Barrier(world), threading.Lock(), {}, [None] * world, []
    calls = [[] for _ in range(world)]

    def allreduce(rank):
        def fn(ptr, count, _stream):
            t = torch.as_tensor(_DevView(ptr, count), device="cuda:0")
            h = t.cpu().numpy().copy()
            with lock:
                parts[rank] = h
            bar.wait()
            tot = sum(parts[r] for r in range(world))
            bar.wait()
            t.copy_(torch.from_numpy(tot))
            calls[rank].append(count)
        return fn

    def work(rank):
        try:
            torch.cuda.set_device(0)
            e = mc.PathEngine(0, stream=torch.cuda.current_stream().cuda_stream)
            e.set_allreduce(allreduce(rank))
            b, c = shard_range(n, rank, world)
            P = e.gbm(SEED, 100.0, 0.04, 0.2, dt, steps, c, path_begin=b)
            e.timing_enable(True)
            e.timing_reset()
            res[rank] = (e.price_lsm(P, 0.04, 100.0, 1.0, dt, False, order), e.timing_get(N.K_LSM_SWEEP)[1])
            P.free()
            e.synchronize()
            bar.wait()
            e.close()
        except BaseException as ex:   # noqa: BLE001
            errs.append(ex)
            bar.abort()

    mc.stats(reset=True)
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    assert not any(t.is_alive() for t in th)
    (p0, launches0), (p1, launches1) = res
    assert p0 == p1
    assert abs(p0[0] - want[0]) <= 1e-9 * want[0] and abs(p0[1] - want[1]) <= 1e-7 * want[1], (p0, want)
    for rank, launches in ((0, launches0), (1, launches1)):
        moment_allreduces = calls[rank].count(nm)
        # timing_get counts the per-date launches + k_lsm_final; every per-date launch but the first follows one all-reduce
        assert launches - 1 == moment_allreduces + 1, (rank, launches, moment_allreduces, calls[rank][:8])
        assert moment_allreduces >= steps + 1          # one per date at least; the re-fits' second launches on top


@pytest.mark.parametrize("steps,n_paths", [(1, 1), (2, 3), (15, 2), (16, 255), (17, 256), (33, 7), (130, 250), (1020, 64)])
def test_coalesced_route_at_the_edges_of_its_shapes(eng, orc, steps, n_paths):
    """The class API through the combiner at the corners of what its row kernels serve (1 .. 256 paths, 1 .. 1020 steps; Mz < 32 takes
    the direct transform, 1020 steps the widest tables): the generated matrix equals mcg_paths_rbergomi's bit for bit, and the four
    pricers on it equal the oracle on the same numbers."""
    from oracle.binding import synthetic_history
    hist = synthetic_history(400, seed=11)
    mc.stats(reset=True)
    mc.set_compat_seed(SEED)
    a = mc.RoughVolatility().GenerateStockPricePaths(hist, steps, n_paths)
    p = orc.estimate_params(hist)
    P = eng.rbergomi(SEED, p["S0"], 0.04, p["xi"], p["H"], p["eta"], p["rho"], DT, steps, n_paths)
    assert a.shape == (n_paths, steps + 1) and np.array_equal(P.to_host(), a)
    P.free()
    K, T = float(hist[-1]) * 1.02, steps * DT
    got = [mc.AsymptoticAnalysis().PredictOptionPrice(a, 0.04, K, T, DT, False, 0.25, 0.08),
           mc.BranchingProcesses().PredictOptionPrice(a, 0.04, K, T, DT, False, 10, list(range(steps))),
           mc.LSM().PredictOptionPrice(a, 0.04, K, T, DT, False, 2),
           mc.MartingaleOptimization().PredictOptionPrice(a, 0.04, K, T, DT, False, 2)]
    mc.set_compat_seed(None)
    want = [orc.asymptotic_price(a, 0.04, K, T, DT, False, 0.25, 0.08, step_major=False),
            orc.branching_price(a, 0.04, K, T, DT, False, 10, np.arange(steps, dtype=np.int32), SEED, mode="philox", step_major=False)[0],
            orc.lsm_price(a, 0.04, K, T, DT, False, 2, step_major=False),
            orc.martingale_price(a, 0.04, K, T, DT, False, 2, 5, step_major=False)[0]]
    assert np.allclose(got, want, rtol=1e-7, atol=1e-12), (got, want)
    s = mc.stats()
    assert s["coalesced_fallbacks"] == 0 and s["coalesced_calls"] >= 5      # all five calls took the coalesced route
