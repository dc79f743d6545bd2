"""The rows bench.py reports BESIDE its headline (extra.configs at N = 1, and the two parity legs that need the engine):
C3 / C4 / the C5 shard, the three other pricers and the batched driver rows, the reference's driver unchanged
through the drop-in classes (round 6).  Nothing here is inside bench.py's timed region; nothing here touches oracle/ (the CPU
baselines are bench.py's own leg)."""
from __future__ import annotations

import json
import math
import os
import time

from tools.bench_common import DT, HBM_PEAK_GBS, N_SIMDS, RB, ROOT, SEED


def reference_parity(eng, mc, ref_price: dict, n_steps: int, seed: int) -> dict:
    """|price - ref| / MC-std-err against the compiled reference itself: the engine prices the same contract
    (rBergomi with the parameters the reference estimates from the same history, same step count, K = S0) and is
    set beside the mean payoff of the reference's own sample from the cpu_baseline leg."""
    p = mc.estimate_params(ref_price["history"])
    n, parts = 4_000_000, 4   # 16M paths of the seed's stream, four launches (the matrix of one is 8 GB)
    K = ref_price["strike"]
    means, ses = [], []
    for k in range(parts):
        P = eng.rbergomi(seed, p["S0"], 0.04, p["xi"], p["H"], p["eta"], p["rho"], 1.0 / 252.0, n_steps, n, path_begin=k * n,
                         payoff=(K, True))
        m, s = eng.price_european(P, K, 0.0, 0.0, True)  # r = 0: undiscounted mean payoff
        P.free()
        means.append(m)
        ses.append(s)
    price = sum(means) / parts
    se = math.sqrt(sum(x * x for x in ses)) / parts
    n = n * parts
    comb = math.hypot(se, ref_price["std_err"])
    return {"contract": f"rBergomi European call, K = S0 = {K:.4f}, {n_steps} steps, parameters estimated from the "
                        "1001-point synthetic history (xi=%.5f H=%.4f eta=%.4f)" % (p["xi"], p["H"], p["eta"]),
            "gpu_mean_payoff": price, "gpu_std_err": se, "gpu_paths": n,
            "reference_mean_payoff": ref_price["mean_payoff"], "reference_std_err": ref_price["std_err"],
            "reference_paths": ref_price["paths"],
            "abs_diff_over_combined_std_err": abs(price - ref_price["mean_payoff"]) / comb if comb > 0 else None}


def rough_regime_parity(eng) -> dict:
    """C4 / C5 parameters against the committed sample of the compiled reference (tests/golden/
    rough_regime_reference.json, oracle/gen_rough_fixture.py): undiscounted call and put at 252 and 512 steps."""
    path = os.path.join(ROOT, "tests", "golden", "rough_regime_reference.json")
    if not os.path.exists(path):
        return {}
    fx = json.load(open(path))
    p, out = fx["params"], {}
    for steps in ("252", "512"):
        fix = fx["samples"][steps]
        for is_call, idx, name in ((True, 1, "call"), (False, 2, "put")):
            P = eng.rbergomi(SEED, p["S0"], p["r"], p["xi"], p["H"], p["eta"], p["rho"], DT, int(steps), 4_000_000,
                             payoff=(p["strike"], is_call))
            m, se = eng.price_european(P, p["strike"], 0.0, 0.0, is_call)
            P.free()
            out[f"{name}_{steps}_steps"] = {"gpu": m, "gpu_std_err": se, "reference": fix["mean"][idx],
                                            "reference_std_err": fix["std_err"][idx], "reference_paths": fix["paths"],
                                            "abs_diff_over_combined_std_err":
                                                abs(m - fix["mean"][idx]) / math.hypot(se, fix["std_err"][idx])}
    return out


def valu_profile(name: str):
    """Committed PMC summary of a kernel (profiles/*_valu_counters.json): VALU instructions per launch at the profiled
    path count and the shader clock measured in the same passes.  None when no profile is committed."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith("_valu_counters.json") and name in f:
            best = os.path.join(pdir, f)  # sorted: the latest round wins
    if not best:
        return None
    try:
        j = json.load(open(best))
        return {"insts": j["counters_mean_per_launch"]["SQ_INSTS_VALU"], "clock_GHz": j["derived"]["shader_clock_GHz"],
                "paths": j.get("paths_per_launch"), "source": os.path.relpath(best, ROOT)}
    except Exception:
        return None


def extra_configs(eng, N, baselines=None) -> list:
    """C3, C4 and the C5 shard on this GPU, once each (one untimed pass, then 5 timed for the wall time -- the median -- and 3 more with
    per-kernel HIP events), after the headline loop: ms per pass, Mpaths/s, the dominant kernel's average launch time and what it achieves against the HBM roofline
    (SURVEY 8d algorithmic bytes) and against the VALU issue rate (instruction count from the committed PMC profile)."""
    reps = 3
    out = []

    def c3():
        P = eng.gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 1_000_000)
        r = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
        P.free()
        return r

    def c4():
        P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 512, 4_000_000, payoff=(100.0, True))
        r = eng.price_european(P, 100.0, RB["r"], 512 * DT, True)
        P.free()
        return r

    def c5():
        P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 252, 8_000_000)
        r = eng.price_lsm(P, RB["r"], 100.0, 1.0, DT, False, 2)
        P.free()
        return r

    # (name, pass, paths, time steps / exercise dates, {kernel: SURVEY 8(d) algorithmic bytes of all its launches in one pass})
    specs = [
        ("C3: American put, LSM order 2, GBM, 1M paths x 50 exercise dates", c3, 1_000_000, 50,
         {"gbm": 8.0 * 51 * 1_000_000, "lsm_sweep": 40.0 * 50 * 1_000_000}),
        ("C4: rBergomi European call (H=0.1), 4M paths x 512 steps", c4, 4_000_000, 512, {"rbergomi": 8.0 * 513 * 4_000_000}),
        ("C5 shard: rBergomi American put LSM order 2, 8M paths x 252 steps (1/8 of the 64M job)", c5, 8_000_000, 252,
         {"rbergomi": 8.0 * 253 * 8_000_000, "lsm_sweep": 40.0 * 252 * 8_000_000}),
    ]
    for name, fn, paths, steps, alg in specs:
        fn()
        eng.synchronize()
        # wall time WITHOUT the library's event timing (a HIP-event pair per launch costs ~9 us: 7 % of a C3 pass), then
        # the same passes again with it, for the per-kernel breakdown
        eng.timing_enable(False)
        walls = []
        for _ in range(2 * reps - 1):   # every pass ends in the price coming back: it can be timed by itself; the MEDIAN of five, so
            t0 = time.perf_counter()    # that one host hiccup (3 ms once, on a 0.4-ms pass) does not become the row's number
            res = fn()
            walls.append((time.perf_counter() - t0) * 1e3)
        eng.synchronize()
        ms = sorted(walls)[len(walls) // 2]
        eng.timing_enable(True)
        eng.timing_reset()
        for _ in range(reps):
            fn()
        eng.synchronize()
        kernels = {}
        for k, kname in N.KERNEL_NAMES.items():
            tot, cnt = eng.timing_get(k)
            if cnt:
                kernels[kname] = {"ms_per_pass": tot / reps, "launches_per_pass": cnt // reps}
        # Against the HBM roofline by the bytes each kernel MOVES.  Generators: SURVEY 8(d)'s 8 (steps + 1) B per path, all
        # written (counters: 1.00x, profiles/*_pmc_traffic.json).  LSM sweep: what the one-launch kernels stream by construction
        # -- every row once with the values in registers (8 B per path and date, k_lsm_coop, <= 2.09M paths = 512 workgroups x
        # 4096) or twice through the LDS ring (16 B, k_lsm_big: counters 32.47 GB against 32.26, profiles/r04_c5_pmc_traffic.json);
        # V never touches memory.  SURVEY 8(d)'s 40 B per path and date is the two-pass formulation's traffic, which these
        # kernels do not generate: it is kept for context only and no fraction is formed with it.
        for kname, b in alg.items():
            if kname in kernels:
                moved = b
                if kname == "lsm_sweep":
                    kernels[kname]["survey_two_pass_bytes_per_pass"] = b
                    moved = (8.0 if paths <= 2_097_152 else 16.0) * steps * paths
                kernels[kname]["bytes_moved_per_pass"] = moved
                kernels[kname]["hbm_frac"] = moved / (kernels[kname]["ms_per_pass"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        dom = max(alg, key=lambda k: kernels.get(k, {}).get("ms_per_pass", 0.0))
        row = {"config": name, "paths": paths, "ms_per_pass": ms, "Mpaths_per_s": paths / ms / 1e3,
               "price": res[0], "std_err": res[1], "kernels": kernels, "dominant_kernel": dom,
               "dominant_kernel_ms_per_pass": kernels[dom]["ms_per_pass"], "hbm_frac": kernels[dom].get("hbm_frac"),
               "bound": "valu-issue (fp64; generation_valu_issue_frac)" if dom == "rbergomi" else "hbm"}
        if dom == "lsm_sweep" and paths <= 2_097_152:
            # one launch, one grid-wide exchange of the regression moments per exercise date: at 1M paths a date's 8 MB stream in
            # ~1 us and the exchange costs several -- the sweep is bound by that latency, not by HBM
            row["bound"] = "latency (one grid-wide moment exchange per exercise date inside the launch)"
            row["us_per_exercise_date"] = kernels[dom]["ms_per_pass"] * 1e3 / steps
        if baselines and "lsm_sweep" in kernels and "lsm" in baselines:
            row["cpu_baseline"] = dict(baselines["lsm"], gpu_comparable="paths / kernels.lsm_sweep.ms_per_pass",
                                       gpu_value=paths / kernels["lsm_sweep"]["ms_per_pass"] / 1e3)
        if "rbergomi" in kernels:  # the generator is issue-bound: VALU instructions x 4 cycles against SIMD-cycles available
            vp = valu_profile("c4" if steps == 512 else "c5gen")
            if vp and vp.get("paths"):
                g = kernels["rbergomi"]["ms_per_pass"]
                insts = vp["insts"] * paths / vp["paths"]
                row["generation_valu_issue_frac"] = insts * 4.0 / (N_SIMDS * vp["clock_GHz"] * 1e9 * g * 1e-3)
                row["valu_source"] = (f"{vp['source']}: SQ_INSTS_VALU per launch scaled to {paths} paths x 4 cycles / "
                                      f"({N_SIMDS} SIMDs x {vp['clock_GHz']:.2f} GHz measured there x kernel time measured here)")
        out.append(row)
    return out


def widening_configs(eng, N, mc, baselines=None) -> list:
    """SURVEY 8(f) rows in this round's terms: the three other pricers of the reference's driver on the C3 matrix
    (GBM, 1M paths x 50 dates, device-resident) and the batched driver rows (20 000 option rows x 250 rBergomi paths, four
    prices each), once each after one untimed pass: device ms of the pricer's kernels (HIP events), the bytes its
    streams move by construction and the HBM fraction that makes."""
    import numpy as np
    out, reps = [], 3
    n, steps, dt = 1_000_000, 50, 0.02
    mat = 8.0 * (steps + 1) * n   # one read of the matrix
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, dt, steps, n)
    ex = list(range(steps))       # the driver passes 0..steps-1 (PredictionGen.cpp:780-783)
    specs = [
        ("asymptotic", "AsymptoticAnalysis::PredictOptionPrice (put, sigma 0.2, dividend 0) on the C3 matrix", N.K_ASYM,
         lambda: eng.price_asymptotic(P, 0.04, 100.0, 1.0, dt, False, 0.2, 0.0), mat,
         "one read of the matrix (k_asym_scan)"),
        ("martingale", "MartingaleOptimization::PredictOptionPrice (put, order 2, 5 iterations) on the C3 matrix", N.K_MARTINGALE,
         lambda: eng.price_martingale(P, 0.04, 100.0, 1.0, dt, False, 2, 5)[0], 2.0 * mat + 8.0 * n,
         "two reads of the matrix (primal + moments, dual) and one of row 0"),
        ("branching", "BranchingProcesses::PredictOptionPrice (put, 10 branches, 50 exercise dates) on the C3 matrix", N.K_BRANCHING,
         lambda: eng.price_branching(P, 0.04, 100.0, 1.0, dt, False, 10, ex, SEED)[0], 3.0 * mat + 8.0 * 10 * steps * n,
         "suffix maxima: read S, write F; bounds: read S + 10 random 8-byte gathers in F per path and date (rows of F are 8 MB: L2 / MALL hits, counted as moved)"),
    ]
    for key, name, kid, fn, moved, what in specs:
        fn()
        eng.synchronize()
        eng.timing_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            price = fn()
        eng.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ms, cnt = eng.timing_get(kid)
        out.append({"config": name, "paths": n, "ms_per_call": wall, "price": price,
                    "kernel_ms_per_call": ms / reps, "launches_per_call": cnt // reps, "bytes_moved_per_call": moved,
                    "bytes_moved": what, "hbm_frac": moved / (ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "Mpaths_per_s_of_device_time": n / (ms / reps) / 1e3})
        if baselines and key in baselines:
            out[-1]["cpu_baseline"] = dict(baselines[key], gpu_comparable="Mpaths_per_s_of_device_time")
    P.free()
    # BranchingProcesses on rows of F beyond one L2: 4M paths x 50 dates, rows of 32 MB.  Round 6: the XCD-affine route
    # (k_branch_date_xcd + k_branch_date_xcd_finish: every XCD gathers only from its eighth of the row); what bounds it is stated
    # with the row.
    n4 = 4_000_000
    P4 = eng.gbm(SEED, 100.0, 0.04, 0.2, dt, steps, n4)
    eng.price_branching(P4, 0.04, 100.0, 1.0, dt, False, 10, ex, SEED)
    eng.synchronize()
    eng.timing_reset()
    for _ in range(reps):
        price4 = eng.price_branching(P4, 0.04, 100.0, 1.0, dt, False, 10, ex, SEED)[0]
    eng.synchronize()
    ms4, cnt4 = eng.timing_get(N.K_BRANCHING)
    P4.free()
    moved4 = 3.0 * 8.0 * (steps + 1) * n4 + 8.0 * 10 * steps * n4 + (64.0 + 64.0 + 40.0) * steps * n4
    out.append({"config": "BranchingProcesses::PredictOptionPrice (put, 10 branches, 50 exercise dates) on a 4M x 50 GBM matrix (rows of F: 32 MB, 16 slices)",
                "paths": n4, "price": price4, "kernel_ms_per_call": ms4 / reps, "launches_per_call": cnt4 // reps,
                "bytes_moved_per_call": moved4, "bytes_moved": "as the C3-matrix row above (S read twice, F written, 10 gathers of 8 B per path and date) + per path and date 64 B of "
                               "per-XCD cells written and read and 40 B of bounds and prices in the finishing pass",
                "hbm_frac": moved4 / (ms4 / reps * 1e-3) / 1e9 / HBM_PEAK_GBS, "Mpaths_per_s_of_device_time": n4 / (ms4 / reps) / 1e3,
                "bound": "VALU issue of the gather launches (every tile's Philox draws are made by eight workgroups, one per XCD, each keeping "
                         "its XCD's eighth of the row: 92 % L2 hits, 0.83 VALU busy) + the finishing pass's 104 B per path and date "
                         "(counters: profiles/r06_branching_xcd_counters.json; round 5's binned kernel: 21.4 ms, 53 % hits)"})
    rs = np.random.RandomState(0)   # the row mix of tools/bench_rows.py: 5..126 steps, calls and puts around the money
    rows = []
    for _ in range(20_000):
        st = int(rs.randint(5, 127))
        S0 = float(rs.uniform(20, 400))
        rows.append(dict(S0=S0, xi=float(rs.uniform(0.01, 0.3)), H=float(rs.uniform(0.3, 0.6)), eta=float(rs.uniform(0.01, 0.06)),
                         rho=-0.3, strike=S0 * float(rs.uniform(0.9, 1.1)), maturity=st / 252.0, sigma=float(rs.uniform(0.1, 0.6)),
                         dividend=0.08, n_steps=st, is_call=int(rs.randint(0, 2))))
    arr = mc.make_rows(rows)   # the C array of mcg_row, built ONCE: what is timed below is the entry point, not its marshalling
    eng.batch_price_rows(arr, seed=1)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        pr = eng.batch_price_rows(arr, seed=1)
    wall = (time.perf_counter() - t0) / reps * 1e3
    ms, cnt = eng.timing_get(N.K_BATCH)
    cols = sum(r["n_steps"] + 1 for r in rows)
    # row blocks written once by the generator; read once each by AsymptoticAnalysis, BranchingProcesses (its suffix maxima stay in
    # registers / LDS since round 3) and LSM, twice by MartingaleOptimization (primal and dual scan)
    moved = 8.0 * 250 * cols * (1 + 5)
    out.append({"config": "mcg_batch_price_rows: 20 000 driver rows x 250 rBergomi paths (5-126 steps), four prices per row",
                "rows": len(rows), "ms_per_call": wall, "rows_per_s": len(rows) / wall * 1e3, "kernel_ms_per_call": ms / reps,
                "rows_per_s_of_device_time": len(rows) / (ms / reps) * 1e3, "launches_per_call": 6 * cnt // reps, "chunks_per_call": cnt // reps,
                "timed": "mcg_batch_price_rows on a prebuilt array of mcg_row (upload, kernels, download, scatter); device time = the chunks' kernel spans",
                "bytes_moved_per_call": moved,
                "bytes_moved": "row blocks written once by the generator, read once each by AsymptoticAnalysis, BranchingProcesses and LSM, twice by "
                               "MartingaleOptimization (primal and dual scan)",
                "hbm_frac": moved / (ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "latency- and issue-bound small-row work: the HBM fraction is reported, not the bound",
                "mean_prices": [float(x) for x in pr.mean(axis=0)]})
    if baselines and "driver_rows" in baselines:
        out[-1]["cpu_baseline"] = dict(baselines["driver_rows"], gpu_comparable="rows_per_s")
    return out


def unchanged_driver_row(baselines=None) -> dict:
    """The reference's driver UNCHANGED as a measured workload: tests/cpp/unchanged_driver.cpp is the row
    loop of src/core/PredictionGen.cpp:542-570 / :736-737 / :788-791 written against the reference's own headers (include/models/*.h)
    -- one option row per OpenMP thread, the five classes constructed per row, GenerateStockPricePaths then the four pricers, 250
    paths x 5..126 steps, exceptions caught per row -- compiled with g++ and linked against libmcgpu.so.  Run in child processes at
    omp_get_max_threads(), 128 and 16 threads with the class API's cross-thread coalescing on (the default: csrc/coalesce.hpp; once
    more at 128 threads without its prefetch of a row's other pricers), and at 128 and 16 threads with it off (every call a launch
    + a synchronisation on the calling thread's own context: the route of rounds 1-5).  rows_per_s = the best coalesced run; every run is listed with its CPU seconds (the GPU boxes of this
    pool give a job 16 CPUs' worth of time whatever its thread count: cpu_quota_cores)."""
    import subprocess
    row = {"config": "the reference driver's row loop UNCHANGED through the drop-in classes (tests/cpp/unchanged_driver.cpp = PredictionGen.cpp:542-570, "
                     ":736-737, :788-791): one row per OpenMP thread, five class-API calls per row, 250 rBergomi paths x 5-126 steps",
           "runs": []}
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        row["cpu_quota_cores"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:   # noqa: BLE001
        row["cpu_quota_cores"] = None
    exe = os.path.join(ROOT, "build", "unchanged_driver")
    try:
        src = os.path.join(ROOT, "tests", "cpp", "unchanged_driver.cpp")
        if not (os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src)):   # (__graft_entry__.build() makes it; g++ only)
            subprocess.run(["make", "build/unchanged_driver"], cwd=ROOT, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)

        def run(threads, coalesce, n_rows):
            # (a profiler wrapped around bench.py must not follow into these children: their 10^5 launches from 256 threads are not
            #  what the profile of the headline command is about)
            env = {k: v for k, v in os.environ.items() if not (k.startswith(("ROCP", "ROCPROF")) or k == "LD_PRELOAD" or k == "HSA_TOOLS_LIB")}
            env["OMP_DYNAMIC"] = "false"
            if threads:
                env["OMP_NUM_THREADS"] = str(threads)
            else:
                env.pop("OMP_NUM_THREADS", None)
            p = subprocess.run([exe, str(n_rows), str(coalesce)], capture_output=True, text=True, env=env, timeout=300)
            if p.returncode != 0:
                raise RuntimeError(f"unchanged_driver exited {p.returncode}: {p.stderr[-400:]}")
            j = json.loads(p.stdout.strip().splitlines()[-1])
            j["route"] = {1: "coalesced + the row's other pricers prefetched (default)", 2: "coalesced, no prefetch (mcg_compat_set_coalescing(2))",
                          0: "per-thread contexts (mcg_compat_set_coalescing(0): rounds 1-5)"}[coalesce]
            row["runs"].append(j)
            return j
        co = [run(t, 1, 16000) for t in (0, 128, 16)]
        run(128, 2, 16000)   # (listed under runs: what the prefetch is worth)
        own = [run(128, 0, 2000), run(16, 0, 4000)]
        best, old = max(co, key=lambda j: j["rows_per_s"]), max(own, key=lambda j: j["rows_per_s"])
        at128 = [j for j in co if j["threads"] == 128][0]
        row.update({"rows_per_s": best["rows_per_s"], "threads": best["threads"], "rows": best["rows"],
                    "cpu_microseconds_per_row": best["cpu_seconds"] / best["rows"] * 1e6,
                    "calls_per_round": best["calls"] / max(best["rounds"], 1),
                    "rows_per_s_at_128_threads": at128["rows_per_s"],
                    "per_thread_route_rows_per_s": old["rows_per_s"], "per_thread_route_threads": old["threads"],
                    "per_thread_route_rows_per_s_at_128_threads": own[0]["rows_per_s"],
                    "per_thread_route_cpu_microseconds_per_row": old["cpu_seconds"] / old["rows"] * 1e6,
                    "speedup_at_128_threads": at128["rows_per_s"] / own[0]["rows_per_s"],
                    "speedup_best_vs_best": best["rows_per_s"] / old["rows_per_s"],
                    "prefetch_hits_per_row": best.get("prefetch_hits", 0) / best["rows"],
                    "bound": "host round trips: a row waits for its paths and then for its slowest pricer (the first pricer call queues the other "
                             "three beside itself; a row's LSM sweep alone is ~0.36 ms of dependent dates at 126 steps), each answered by the next round "
                             "of its kind -- and, on this pool's boxes, the 16-CPU quota (cpu_seconds / seconds of the runs)",
                    "checksums_equal": len({round(j["checksum"] / j["rows"], 6) for j in co}) == 1})
    except Exception as e:   # noqa: BLE001 -- a reported row, never required for the headline
        row["error"] = f"{type(e).__name__}: {e}"
    if baselines and "driver_rows" in baselines:
        row["cpu_baseline"] = dict(baselines["driver_rows"], gpu_comparable="rows_per_s")
    return row
