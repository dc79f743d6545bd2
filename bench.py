#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s at 252 steps (BASELINE.json metric).

Default workload (BASELINE.json configs[1], "C2"): European call, GBM, 10M paths x 252 steps per GPU,
S0 = K = 100, r = 0.04, sigma = 0.2, dt = 1/252, Philox seed 20251031.  One "step" = one full pass
of the hot path: Philox normals -> GBM stepping -> the (253 x 10M) fp64 matrix written to HBM ->
per-path payoff -> wavefront-shuffle reduction -> (N>1: one all-reduce of 3 doubles) -> price.
Inputs are parameters only, so everything is resident when the timed region starts.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--config c5` runs BASELINE.json configs[4] instead: rBergomi (H = 0.1, eta = 1.9) American put, Longstaff-Schwartz
order 2, 252 steps, 8M paths per GPU (N = 8: the 64M-path job), one step = generate the shard's matrix + the backward
sweep with one all-reduce of the 3p+2 regression moments per exercise date.

Weak scaling: every rank owns `--paths` global path ids [rank*paths, (rank+1)*paths) of ONE Philox
stream; no data-path collective except the payoff / moment all-reduces.  The collective is the library's built-in RCCL
communicator (mcg_comm_init_rank: issued from C on the ctx's stream, nothing on the host waits per date); `--collective
torch` routes it through torch.distributed instead.  Prints ONE JSON line on rank 0 with the driver's contract fields
plus "roofline", "cpu_baseline" (N=1 only) and "extra.configs":
  * N = 1 (default run): C3, C4 and the C5 shard timed once each after the headline loop, the three other pricers of
    the driver on the C3 matrix and the batched driver rows, the cold first launch of C2;
  * N > 1: BASELINE.json configs[4] (C5: rBergomi American put LSM, 8M paths per GPU -- N = 8 is the 64M-path job)
    timed after the C2 loop without any exchange ("none": every rank prices its own shard, the baseline) and through each
    collective in turn -- the node mailbox in host memory ("shm"), the same in peer-mapped device memory ("ipc") and the
    built-in RCCL communicator ("rccl") -- with the slowest and the fastest rank's ms per pass, the collective that
    actually ran and the number of ranks its communicator has SEEN.  Where they run is `--c5-rows`: in a child process per
    rank (`child`: 2 x N GPU processes meanwhile) or in the rank processes themselves (`inline`: N), `auto` = child while
    2 x N fits under the pool's process guard as measured, else inline.  Either way the headline cannot be lost to them:
    rank 0 hands its line to a guardian process (LastWill, forked before anything touches the GPU) as soon as the
    headline is complete and again after every row; the guardian prints it when rank 0 ends -- however it ends.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_SIMDS = 1024         # 256 CUs x 4 SIMDs
SEED = 20251031
DT = 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)


def bs_call(S0, K, r, sigma, T):
    d1 = (math.log(S0 / K) + (r + 0.5 * sigma * sigma) * T) / (sigma * math.sqrt(T))
    d2 = d1 - sigma * math.sqrt(T)
    N = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))  # noqa: E731
    return S0 * N(d1) - K * math.exp(-r * T) * N(d2)


def cpu_baseline(n_steps: int, budget_s: float = 15.0) -> dict:
    """The reference generator on this host's cores (oracle/_ref = the compiled reference,
    "reference"; else our restatement in reference-faithful mode, "port"), parallelised the way the
    reference's driver does (omp parallel for schedule(dynamic) over independent generator calls,
    src/core/PredictionGen.cpp:542-546; 250 paths per call, :719).  Bounded sample."""
    import numpy as np  # noqa: F401

    from oracle.binding import Oracle, Reference, have_ref, synthetic_history
    hist = synthetic_history(1001, seed=42)
    chunk = 250
    strike = float(hist[-1])  # at the money: the reference takes S0 = last history price (:331)
    if have_ref():
        ref = Reference()
        kind = "reference"

        def run(n):
            th, o = ref.generate_paths_omp_payoff(hist, n_steps, n, chunk, strike, True)
            return th, o
    else:
        orc = Oracle()
        p = orc.estimate_params(hist)
        kind = "port"
        run = lambda n: orc.generate_paths_mt_omp(p["S0"], 0.04, p["xi"], p["H"], p["eta"], p["rho"], n_steps,  # noqa: E731
                                                  n, chunk, 1)
    t0 = time.perf_counter()
    cores, _ = run(8000)
    pilot = time.perf_counter() - t0
    rate = 8000 / max(pilot, 1e-6)
    n = int(min(max(rate * budget_s, 20_000), 5_000_000)) // chunk * chunk
    t0 = time.perf_counter()
    cores, sums = run(n)
    dt = time.perf_counter() - t0
    out = {"value": n / dt / 1e6, "unit": "Mpaths/s", "cores": int(cores), "kind": kind,
           "sample": f"{n} paths x {n_steps} steps via RoughVolatility::GenerateStockPricePaths "
                     f"(rBergomi, 1001-point synthetic history), omp dynamic, {chunk} paths/call, {dt:.1f} s"}
    if kind == "reference":
        # the same entry point on ONE thread, ~2 s: what a host core delivers when the calls do not compete (every call of the
        # reference constructs std::random_device + mt19937 three times per path and allocates per path: on many cores the
        # parallel rate above is far below cores x this)
        t0, done = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            ref.generate_paths(hist, n_steps, chunk)
            done += chunk
        out["single_thread_value"] = done / (time.perf_counter() - t0) / 1e6
        # the reference's own sample, priced: undiscounted mean call payoff +- std-err at K = S0
        m = sums[1] / sums[3]
        var = max(0.0, (sums[2] - sums[3] * m * m) / (sums[3] - 1))
        out["_ref_price"] = {"history": hist, "strike": strike, "mean_payoff": m, "std_err": math.sqrt(var / sums[3]),
                             "paths": int(sums[3])}
    return out


def _timed_chunks(fn, sample, budget_s: float, what: str, kind: str) -> dict:
    """`fn(sample[:n]) -> (threads, seconds, checksum)` prices a resident [n][m] sample in driver rows of 250 paths under
    omp dynamic (oracle/ref_harness.cpp: ref_pricer_chunks_omp, oracle/mcg_oracle.cpp: orc_pricer_chunks_omp).  A pilot
    sizes the sample to the budget; the sample is re-priced until ~the budget is spent."""
    n0 = min(len(sample), 4000)
    th, sec, _ = fn(sample[:n0])
    n = int(min(len(sample), max(n0, n0 / max(sec, 1e-6) * budget_s))) // 250 * 250
    done, spent = 0, 0.0
    while spent < 0.8 * budget_s:
        th, sec, _ = fn(sample[:n])
        done, spent = done + n, spent + sec
    return {"value": done / spent / 1e6, "unit": "Mpaths/s", "cores": int(th), "kind": kind,
            "sample": f"{what}: {done} paths ({n}-path sample of the row's matrix shape, 51 columns, priced {done // n}x) in driver rows of 250 "
                      f"paths, one PredictOptionPrice call per row under omp parallel for schedule(dynamic) (PredictionGen.cpp:542-546, :719), {spent:.1f} s"}


def cpu_baselines_widened(budget_s: float = 3.0) -> dict:
    """CPU baselines beside the rows of extra.configs that have none of their own (missing #3), each bounded to
    ~budget_s of wall time on all host cores: the compiled reference ("reference") for AsymptoticAnalysis and
    BranchingProcesses, this repo's restatement ("port": LSMPricer.cpp and MartingaleOptimizationPricer.cpp need Eigen, which
    the image lacks) for LSM and MartingaleOptimization, and whole driver rows (generation + four pricers; reference +
    port mixed, reported as "port").  Reported baselines, not targets."""
    import numpy as np

    from oracle.binding import Oracle, Reference, have_ref, synthetic_history
    orc = Oracle()
    ref = Reference() if have_ref() else None
    sample = np.ascontiguousarray(orc.paths_gbm(SEED, 100.0, 0.04, 0.2, 0.02, 50, 0, 60_000).T)   # [n][51]: the C3 matrix's law
    out = {}
    arg = (250, 0.04, 100.0, 1.0, 0.02, False)
    out["lsm"] = _timed_chunks(lambda m: orc.pricer_chunks_omp("lsm", m, *arg, 2), sample, budget_s,
                               "LSM::PredictOptionPrice (restated: oracle/mcg_oracle.cpp orc_lsm_price, order 2, put)", "port")
    out["martingale"] = _timed_chunks(lambda m: orc.pricer_chunks_omp("martingale", m, *arg, 2), sample, budget_s,
                                      "MartingaleOptimization::PredictOptionPrice (restated: orc_martingale_price, order 2, 5 iterations, put)", "port")
    if ref is not None:
        out["asymptotic"] = _timed_chunks(lambda m: ref.pricer_chunks_omp("asymptotic", m, *arg, 0.2, 0.0), sample, budget_s,
                                          "AsymptoticAnalysis::PredictOptionPrice (compiled reference, put, sigma 0.2)", "reference")
        out["branching"] = _timed_chunks(lambda m: ref.pricer_chunks_omp("branching", m, *arg, 0.2, 0.0, 10), sample, budget_s,
                                         "BranchingProcesses::PredictOptionPrice (compiled reference, put, 10 branches, 50 exercise dates)", "reference")
        hist = synthetic_history(1001, seed=42)
        rs = np.random.RandomState(0)

        def rows(n):
            st = rs.randint(5, 127, size=n)
            return ref.driver_rows_omp(hist, st, float(hist[-1]) * rs.uniform(0.9, 1.1, size=n), rs.randint(0, 2, size=n), 250, 0.2, 0.08, orc)
        th, sec, _ = rows(64)
        n = int(max(64, min(64 / max(sec, 1e-6) * budget_s * 1.5, 2_000_000)))
        th, sec, _ = rows(n)
        out["driver_rows"] = {"value": n / sec, "unit": "rows/s", "cores": int(th), "kind": "port",
                              "sample": f"{n} driver rows (RoughVolatility::GenerateStockPricePaths on a 1001-point synthetic history, 250 paths x 5-126 "
                                        "steps, then AsymptoticAnalysis and BranchingProcesses of the compiled reference and the restated LSM and "
                                        f"MartingaleOptimization), omp parallel for schedule(dynamic) over rows as PredictionGen.cpp:542-823, {sec:.1f} s"}
    return out


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
    # BranchingProcesses on rows of F beyond one L2: 4M paths x 50 dates, rows of 32 MB, the binned
    # per-date kernel (k_branch_date_binned).  What bounds it is stated with the row: every generation of resident paths
    # pulls the whole row through the L2 of every XCD.
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
    moved4 = 3.0 * 8.0 * (steps + 1) * n4 + 8.0 * 10 * steps * n4
    out.append({"config": "BranchingProcesses::PredictOptionPrice (put, 10 branches, 50 exercise dates) on a 4M x 50 GBM matrix (rows of F: 32 MB, 16 slices)",
                "paths": n4, "price": price4, "kernel_ms_per_call": ms4 / reps, "launches_per_call": cnt4 // reps,
                "bytes_moved_per_call": moved4, "bytes_moved": "as the C3-matrix row above: S read twice, F written, 10 gathers of 8 B per path and date",
                "hbm_frac": moved4 / (ms4 / reps * 1e-3) / 1e9 / HBM_PEAK_GBS, "Mpaths_per_s_of_device_time": n4 / (ms4 / reps) / 1e3,
                "bound": "L2 fill: the gathers of a generation of resident paths (590k-786k) touch every line of the 32 MB row in every XCD's L2 "
                         "(counters: profiles/r05_branching_binned_counters.json)"})
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


class Gpu:
    """What this script asks of torch.cuda (tests/bench_rehearsal.py has the CPU stand-in of --rehearsal)."""
    name = "cuda"

    @staticmethod
    def set_device(d):
        import torch
        torch.cuda.set_device(d)

    @staticmethod
    def synchronize():
        import torch
        torch.cuda.synchronize()

    @staticmethod
    def current_stream_handle():
        import torch
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def device_count():
        import torch
        return torch.cuda.device_count()


def agree(flags, dist, torch, dev) -> list:
    """Element-wise AND of `flags` over the ranks (one all-reduce).  Every rank enters it -- from its except branch too."""
    t = torch.tensor([1 if f else 0 for f in flags], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [int(x) == 1 for x in t.tolist()]


def everyone(ok: bool, dist, torch, dev) -> bool:
    """True iff `ok` on EVERY rank.  Every rank enters it -- from its except branch too -- so it doubles as the point where
    the ranks of a step that may fail locally meet again.  That is sound only where the step itself cannot leave a peer
    inside ANOTHER collective for good: set-up steps (nothing collective inside), and passes over the node mailbox (shm /
    ipc: the segment barrier times out, the peers raise and arrive here too).  A pass over the built-in RCCL communicator or
    over torch.distributed is not bounded like that -- see c5_sharded_rows.phase for what a failing rank does there."""
    return agree([ok], dist, torch, dev)[0]


class LastWill:
    """Rank 0 of an N > 1 run: a guardian process, forked before anything touches the GPU, that owns the job's ONE JSON
    line.  The rank sends it the line as soon as the headline is complete ("WILL", re-sent after every C5 row) and the
    finished line at the end ("FINAL"); when the pipe closes -- the rank returned, raised, was killed by the launcher after
    a peer died, or took a device fault in a C5 row -- the guardian prints FINAL, or else the last WILL with an "aborted"
    note.  Nothing after the headline can cost the line any more, whatever the C5 rows do."""

    def __init__(self, json_out):
        import signal
        r, w = os.pipe()
        self.pid = os.fork()
        if self.pid == 0:
            code = 0
            try:
                os.close(w)
                for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                    signal.signal(sg, signal.SIG_IGN)
                will = final = None
                with os.fdopen(r, "r") as f:
                    for ln in f:
                        if not ln.endswith("\n"):
                            break                      # (the rank died inside a write)
                        if ln.startswith("WILL "):
                            will = ln[5:]
                        elif ln.startswith("FINAL "):
                            final = ln[6:]
                line = final
                if line is None and will is not None:
                    j = json.loads(will)
                    j["aborted"] = ("rank 0 ended before the line was finished (a peer rank failed and the launcher ended the job, or "
                                    "a C5 row took the process down); the headline was complete, extra.configs holds the rows that were")
                    line = json.dumps(j)
                if line:
                    json_out.write(line.rstrip("\n") + "\n")
                    json_out.flush()
            except BaseException:   # noqa: BLE001
                code = 1
            finally:
                os._exit(code)
        os.close(r)
        self._w = os.fdopen(w, "w")
        self._out, self._lost = json_out, False

    def _send(self, tag: str, out: dict) -> None:
        try:
            self._w.write(tag + " " + json.dumps(out) + "\n")
            self._w.flush()
        except OSError:   # the guardian is gone (it should never be): this process prints the line itself at the end
            self._lost = True

    def update(self, out: dict):
        if not self._lost:
            self._send("WILL", out)

    def final(self, out: dict):
        if not self._lost:
            self._send("FINAL", out)
        try:
            self._w.close()
        except OSError:
            pass
        _, status = os.waitpid(self.pid, 0)
        if self._lost or status != 0:
            print(json.dumps(out), file=self._out, flush=True)


GPU_PROCESS_GUARD = 6   # what the pool's process guard allowed on the builder's one-GPU box; a node's is not stated


def c5_rows_mode(args, world: int) -> str:
    """child: every rank starts a child process for the C5 rows while it still holds the device (2 x world GPU processes; a
    fault in a row cannot touch the parent).  inline: the rows run in the rank processes, after the headline is safe with
    the guardian (world GPU processes).  auto: child where 2 x world fits under the process guard measured, else inline."""
    if args.c5_rows != "auto":
        return args.c5_rows
    return "child" if 2 * world <= GPU_PROCESS_GUARD else "inline"


def rccl_forms_in_time(make_scratch, rank: int, world: int, bcast, limit_s: float) -> bool:
    """Form the built-in RCCL communicator on a scratch context inside a time box; True iff it formed within limit_s.  Only
    ncclCommInitRank itself runs in the worker thread: the id's broadcast stays in the MAIN thread (torch's current device is
    thread-local -- a collective issued from a fresh thread would run on device 0 on every rank), which enters it ALWAYS,
    whatever the worker did (an empty id from rank 0 makes every rank's init raise together)."""
    import threading
    box = {}
    uid_ready, uid_back = threading.Event(), threading.Event()

    def bcast_in_main(uid):   # called by init_rccl inside the worker: park the id, wait for the main thread's broadcast
        box["uid_in"] = uid
        uid_ready.set()
        uid_back.wait()
        return box.get("uid_out")

    def work():
        try:
            e = make_scratch()
            box["scratch"] = e
            e.init_rccl(rank, world, bcast_in_main)
            box["formed"] = True
        except Exception as ex:   # noqa: BLE001
            box["err"] = ex
        finally:
            uid_ready.set()       # (a worker that failed before it had an id must not keep the main thread from the broadcast)
    t = threading.Thread(target=work, daemon=True)
    t.start()
    uid_ready.wait(limit_s)       # creating the id is local
    box["uid_out"] = bcast(box.get("uid_in", b"") if rank == 0 else None)
    uid_back.set()
    t.join(limit_s)
    if t.is_alive():
        print(f"bench: rank {rank}: the built-in RCCL communicator did not form within {limit_s:.0f} s; falling back to torch.distributed",
              file=sys.stderr, flush=True)
        return False
    if "err" in box or not box.get("formed"):
        print(f"bench: built-in RCCL communicator unavailable ({box.get('err')}); using torch.distributed", file=sys.stderr)
        if box.get("scratch") is not None:
            try:
                box["scratch"].close()
            except Exception:   # noqa: BLE001
                pass
        return False
    box["scratch"].close()
    return True


def install_collective(eng, mc, want: str, dist, torch, rank: int, world: int, dev="cuda", scratch=None) -> str:
    """Give `eng` the collective `want` ("ipc", "shm", "rccl", "torch") -- every rank ends up on the SAME one: a set-up
    that fails on any rank sends all of them one step down (ipc -> shm -> rccl -> torch).  Returns what is installed.
    No rank can be left alone in a collective: whatever a rank does before a broadcast cannot fail (the segment's name is
    a string; the RCCL id is created inside a try and an empty one is broadcast on failure, PathEngine.init_rccl), and
    every local step that can fail is followed by everyone()."""
    got = want
    if want in ("shm", "ipc"):
        box = [f"/mcg_bench_{os.getpid()}_{time.time_ns()}" if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ok = True
        try:
            eng.init_shm(box[0], rank, world)
        except mc.McgError as e:
            print(f"bench: shared-memory communicator unavailable ({e}); using RCCL", file=sys.stderr)
            ok = False
        if not everyone(ok, dist, torch, dev):
            eng.set_allreduce(None)
            got = f"rccl ({want} init failed" + ("" if not ok else " on a peer") + ")"
        elif want == "ipc":
            try:     # (collective over the segment: the ranks agree inside; an error poisons the segment for all of them)
                peer = eng.shm_peer_mailbox(True)
            except mc.McgError as e:
                print(f"bench: peer-memory mailbox failed ({e})", file=sys.stderr)
                peer, ok = False, False
            if not everyone(ok, dist, torch, dev):
                eng.set_allreduce(None)
                got = "rccl (ipc set-up failed" + ("" if not ok else " on a peer") + ")"
            elif not peer:
                got = "shm (peer-memory mailbox unavailable: export, open or in-kernel ping failed on some rank)"
    if got.startswith("rccl"):
        def bcast(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        ok = True
        try:                               # can EVERY rank load librccl?  (agreed before anybody enters ncclCommInitRank, where a
            eng.rccl_probe()               #  rank whose peer never arrives would wait)
        except mc.McgError as e:
            print(f"bench: librccl unavailable on rank {rank} ({e})", file=sys.stderr)
            ok = False
        if everyone(ok, dist, torch, dev):
            # ncclCommInitRank with more than one rank has never run in this repo's history (every GPU box had one GPU): a
            # communicator that does not FORM must cost the run its collective, not its line.  So it is formed once on a
            # scratch context inside a time box; only if every rank's formed in time does the real context get its own.  A
            # scratch context that is still inside ncclCommInitRank when the box closes is abandoned (daemon thread).
            ok = rccl_forms_in_time(scratch or (lambda: mc.PathEngine(eng.device)), rank, world, bcast,
                                    float(os.environ.get("MCG_BENCH_RCCL_INIT_LIMIT", "90")))
            if everyone(ok, dist, torch, dev):
                try:
                    eng.init_rccl(rank, world, bcast)
                except mc.McgError as e:       # communicator set-up failed on this node: use torch's, and say so
                    print(f"bench: built-in RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
                    ok = False
            else:
                ok = False
        else:
            ok = False
        if not everyone(ok, dist, torch, dev):               # all ranks take the same route
            got = "torch (built-in RCCL init failed" + ("" if not ok else " on a peer") + ")"
            eng.use_torch_distributed()
    elif got == "torch":
        eng.use_torch_distributed()
    return got


def c5_sharded_rows(args, mc, N, dist, torch, device, stream, rank, world, hw=Gpu, make_engine=None, on_row=None,
                    deadline=None) -> list:
    """BASELINE.json configs[4] on this run's N ranks, after the headline loop: rBergomi (H = 0.1, eta = 1.9) American put,
    LSM order 2, 252 steps, --c5-paths (8M) paths per GPU of ONE Philox stream, timed through each collective of
    --c5-collectives in turn on a fresh context.  One untimed pass, then 3 timed between barriers; per row: the slowest
    and the fastest rank's ms per pass, the collective that ran, what its communicator has seen (mcg_comm_info), the
    launches of the LSM sweep per pass (1 = the one-launch sweep exchanged inside the kernel) and the global price.
    A row is a sequence of local phases; after each the ranks meet in agree(): a rank that raised is there too, so the
    row is recorded as failed on ALL ranks at once -- WHERE the peers can get there: set-up phases and passes over the node
    mailbox (its barrier times out).  A rank that raises inside a pass over the built-in RCCL communicator or over
    torch.distributed leaves its peers inside an all-reduce that never completes (or, worse, would pair its own agreement
    all-reduce with their data all-reduce): there it ends the job instead -- exit code 17, the launcher (or the parents of
    the child job) take the peers down, the rows finished so far and the headline are already with rank 0's guardian.
    `deadline` (time.time() value): once any rank is past it the remaining rows are abandoned by all ranks together."""
    from montecarlooptionspricer_amd.sharding import shard_range
    rows, reps, steps = [], 3, 252
    total = args.c5_paths * world
    begin, count = shard_range(total, rank, world, align=2)
    dev = torch.device("cuda", device) if hw is Gpu else torch.device("cpu")
    make_engine = make_engine or (lambda: mc.PathEngine(device, stream=stream))
    out_of_time = False
    for want in [c for c in args.c5_collectives.split(",") if c]:
        if out_of_time:
            break
        e5, err, row = None, None, None
        if args.rehearsal:
            os.environ["MCG_REHEARSAL_ROW"] = want   # (read by the rehearsal's failure injection only)
        st = {"unbounded": False}

        def phase(fn):
            """Run a local step; every rank then learns whether it worked everywhere (and whether there is time left)."""
            nonlocal err, out_of_time
            ok = True
            if err is None:
                try:
                    fn()
                except Exception as ex:   # noqa: BLE001
                    err, ok = f"{type(ex).__name__}: {ex}", False
                    if st["unbounded"]:
                        print(f"bench: rank {rank} failed inside a pass over '{st.get('got')}' ({err}); its peers cannot leave that "
                              "collective, so this rank ends the job (exit code 17)", file=sys.stderr, flush=True)
                        os._exit(17)
            else:
                ok = False
            in_time = deadline is None or time.time() < deadline
            ok, in_time = agree([ok, in_time], dist, torch, dev)
            if not in_time:
                out_of_time = True
            return ok and in_time

        def setup():
            nonlocal e5
            e5 = make_engine()

        def one_pass():
            P = e5.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, count, path_begin=begin)
            r = e5.price_lsm(P, RB["r"], 100.0, steps * DT, DT, False, 2)
            P.free()
            return r

        def warm():
            one_pass()
            e5.synchronize()
            hw.synchronize()

        def timed():
            e5.timing_enable(True)
            e5.timing_reset()
            t0 = time.perf_counter()
            for _ in range(reps):
                st["price"], st["se"] = one_pass()
            e5.synchronize()
            hw.synchronize()
            st["mine"] = (time.perf_counter() - t0) / reps * 1e3

        try:
            good = phase(setup)
            if good:
                # (install_collective agrees among the ranks inside; an exception there is the same on every rank)
                st["got"] = "none (every rank prices its own shard alone: a local price, the baseline the routes below add their exchange to)" \
                    if want == "none" else install_collective(e5, mc, want, dist, torch, rank, world, dev, scratch=make_engine)
                st["info"] = e5.comm_info()
                st["unbounded"] = st["got"].startswith(("rccl", "torch"))
            good = good and phase(warm) and phase(timed)
            st["unbounded"] = False
            if good:
                t = torch.tensor([st["mine"], -st["mine"]], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms_max, ms_min = float(t[0].item()), -float(t[1].item())
                gen_ms, _ = e5.timing_get(N.K_RBERGOMI)
                sw_ms, sw_n = e5.timing_get(N.K_LSM_SWEEP)
                seen = torch.tensor([st["info"]["seen_ranks"]], device=dev)
                dist.all_reduce(seen, op=dist.ReduceOp.MIN)
                row = {
                    "config": f"C5: rBergomi American put LSM order 2, {args.c5_paths} paths x {steps} steps per GPU, {world} rank(s) "
                              f"= {total} paths of one Philox stream",
                    "collective_requested": want, "collective": st["got"],
                    "comm": dict(st["info"], seen_ranks_min_over_ranks=int(seen.item())),
                    "paths_per_gpu": args.c5_paths, "global_paths": total,
                    "ms_per_pass_slowest_rank": ms_max, "ms_per_pass_fastest_rank": ms_min,
                    "Mpaths_per_s": total / ms_max / 1e3, "price": st["price"], "std_err": st["se"],
                    "rank0_generator_ms_per_pass": gen_ms / reps, "rank0_lsm_sweep_ms_per_pass": sw_ms / reps,
                    "rank0_lsm_sweep_launches_per_pass": sw_n // reps,
                    "lsm_one_launch": e5.lsm_one_launch_enabled() and sw_n // reps <= 2, "rank0_stats": mc.stats()}
            elif out_of_time:
                row = {"config": "C5", "collective_requested": want,
                       "error": "the wall-clock budget of the C5 rows was used up: this row and the remaining ones were abandoned by all ranks together"}
            else:
                row = {"config": "C5", "collective_requested": want,
                       "error": err or "a peer rank failed in this row (its own stderr says why); all ranks abandoned it together"}
        except Exception as ex:   # (outside the phases: the collectives of this function itself)
            row = {"config": "C5", "collective_requested": want, "error": f"{type(ex).__name__}: {ex}"}
        finally:
            if e5 is not None:
                e5.close()
        rows.append(row)
        if on_row is not None:
            on_row(rows)
    return rows


def c5_rows_in_child_job(args, dist, torch, rank: int, world: int, budget_s: float):
    """Every rank of this job starts `bench.py --c5-child` as a child process (same RANK / LOCAL_RANK / WORLD_SIZE, a
    rendezvous port of its own) and the parents WATCH the children together: four times a second they exchange (over a gloo
    group of their own: no device work beside the children's timing) who is still running and who has failed -- a child
    that could not be started (spawn refused), one that exited non-zero, or the budget running out.  On the first failure
    every parent kills its child: no parent waits for a child whose peer is gone.  Rank 0's child prints each finished row
    as a line of its own, so the rows before a failure are kept.  Returns (on rank 0) the rows plus, after a failure, one
    row that says what went wrong."""
    import socket
    import subprocess
    import tempfile
    from datetime import timedelta
    mon = dist.new_group(backend="gloo", timeout=timedelta(seconds=120))
    box = [None]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            box[0] = sk.getsockname()[1]
    dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(box[0]),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in list(env):   # the child is not an elastic worker of the parent's agent
        if k.startswith("TORCHELASTIC_") or k in ("TORCH_NCCL_ASYNC_ERROR_HANDLING",):
            env.pop(k)
    cmd = [sys.executable, os.path.abspath(__file__), "--c5-child", "--gpus", str(world), "--backend", args.backend,
           "--c5-paths", str(args.c5_paths), "--c5-collectives", args.c5_collectives, "--c5-budget", str(budget_s)] + (["--rehearsal"] if args.rehearsal else [])
    fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
    proc, why = None, None
    try:
        if os.environ.get("MCG_BENCH_SPAWN_FAIL") in (str(rank), "all"):   # test hook: the pool refuses the process
            raise OSError(11, "Resource temporarily unavailable (injected)")
        proc = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, text=True)
    except OSError as e:
        why = f"rank {rank}: the child process could not be started ({e})"
        print("bench: " + why, file=sys.stderr, flush=True)
    t_end = time.time() + budget_s + 60.0   # (the child abandons its rows at budget_s by itself; this is for one that hangs)
    failed_any = False
    while True:
        rc = proc.poll() if proc is not None else 1
        failed = proc is None or (rc is not None and rc != 0) or time.time() > t_end
        t = torch.tensor([1 if failed else 0, 1 if (proc is not None and rc is None) else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=mon)
        if int(t[0]):
            failed_any = True
            if proc is not None and proc.poll() is None:
                proc.kill()
            break
        if not int(t[1]):
            break
        time.sleep(0.25)
    if proc is not None:
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pass
    rows = None
    if rank == 0:
        fo.seek(0)
        rows = [json.loads(ln[4:]) for ln in fo.read().splitlines() if ln.startswith("ROW ")]
        if failed_any:
            fe.seek(0)
            rc = proc.returncode if proc is not None else None
            rows.append({"config": "C5", "error": why or (f"child job failed (exit code {rc})" if rc not in (None, 0, -9) else
                                                          "child job ended by its parents: a peer rank's child failed, could not be started, or "
                                                          f"the job ran past {budget_s + 60:.0f} s (every rank's own stderr says which)"),
                         "stderr_tail": fe.read()[-1500:]})
    dist.barrier()
    return rows


def c5_rows_inline(args, mc, N, dist, torch, device, stream, rank, world, hw, make_engine, will, out, budget_s: float):
    """The C5 rows in the rank processes themselves (world GPU processes, not 2 x world).  The headline is with the guardian
    already; every finished row is sent after it.  A wall-clock budget: past it the ranks abandon the remaining rows
    together (checked in every agreement); a rank still inside a row a minute after that -- a collective that never
    returns -- ends the job (exit code 18), which the guardian's line survives."""
    import threading
    dog = threading.Timer(budget_s + 60.0, lambda: (print(f"bench: rank {rank}: the inline C5 rows hang past their budget; ending the job",
                                                          file=sys.stderr, flush=True), os._exit(18)))
    dog.daemon = True
    dog.start()

    def on_row(rows):
        if will is not None:
            out.setdefault("extra", {})["configs"] = list(rows)
            will.update(out)
    try:
        return c5_sharded_rows(args, mc, N, dist, torch, device, stream, rank, world, hw, make_engine, on_row=on_row,
                               deadline=time.time() + budget_s)
    finally:
        dog.cancel()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed passes of the hot path")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c2", "c5"], help="c2: GBM European (headline); c5: rBergomi LSM shard")
    ap.add_argument("--paths", type=int, default=0, help="paths per GPU (default: 10M for c2, 8M for c5)")
    ap.add_argument("--time-steps", type=int, default=252)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3/C4/C5-shard timings after the headline loop")
    ap.add_argument("--collective", default=os.environ.get("MCG_COLLECTIVE", "auto"), choices=["auto", "shm", "ipc", "rccl", "torch"],
                    help="auto: ipc for c5 (the LSM sweeps exchange inside the kernel through a mailbox in peer-mapped device memory; "
                         "falls back to shm = the same mailbox in host memory, then rccl, then torch), rccl for c2")
    ap.add_argument("--c5-paths", type=int, default=8_000_000, help="paths per GPU of the C5 rows under extra.configs at N > 1")
    ap.add_argument("--c5-collectives", default="none,shm,ipc,rccl",
                    help="collectives the C5 rows at N > 1 are timed through (none: every rank prices its own shard alone -- the "
                         "time the exchange adds nothing to)")
    ap.add_argument("--backend", default=os.environ.get("MCG_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets several ranks share one GPU (rehearsal only)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="control-flow rehearsal of an N > 1 run on CPU ranks (gloo; tests/bench_rehearsal.py stands in for the GPU "
                         "and the engine): every collective of this script runs, nothing is computed or timed, the line says so")
    ap.add_argument("--c5-rows", default=os.environ.get("MCG_BENCH_C5_ROWS", "auto"), choices=["auto", "child", "inline", "off"],
                    help="where the C5 rows of an N > 1 run are timed: child = a child process per rank (2 x N GPU processes while they "
                         "run; a fault there cannot touch the parent), inline = in the rank processes after the headline is safe with rank "
                         "0's guardian (N GPU processes), off = not at all; auto = child while 2 x N <= %d, else inline" % GPU_PROCESS_GUARD)
    ap.add_argument("--c5-budget", type=float, default=float(os.environ.get("MCG_BENCH_C5_BUDGET", "600")),
                    help="wall-clock seconds for all C5 rows of an N > 1 run; past it the ranks abandon the remaining rows together")
    ap.add_argument("--c5-child", action="store_true",
                    help="(internal) this process is one rank of the child job that times the C5 rows of an N > 1 run")
    args = ap.parse_args()
    if args.paths <= 0:
        args.paths = 10_000_000 if args.config == "c2" else 8_000_000

    # ONE JSON line on stdout, nothing else: libraries write banners there from C (RCCL prints its version block when a
    # communicator is created), so the process's stdout is pointed at stderr and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rank 0 of an N > 1 job hands its line to a guardian process, forked HERE: before torch is imported, before anything
    # has touched the GPU, while this process has one thread
    will = LastWill(json_out) if (world > 1 and rank == 0 and not args.c5_child) else None

    import torch

    import montecarlooptionspricer_amd as mc
    from montecarlooptionspricer_amd import _native as N
    from montecarlooptionspricer_amd.sharding import shard_range

    hw, make_engine = Gpu, None
    if args.rehearsal:
        if world < 2:
            raise SystemExit("--rehearsal rehearses the N > 1 control flow: start it with several ranks")
        from tests.bench_rehearsal import Cpu, RehearsalEngine
        hw, args.backend = Cpu, "gloo"
        make_engine = lambda: RehearsalEngine()   # noqa: E731
    dev_name = hw.name

    dist = None
    force_dist = os.environ.get("MCG_FORCE_DIST") == "1"   # rehearse the collective path with one rank
    device = local_rank % max(hw.device_count(), 1)
    hw.set_device(device)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        from datetime import timedelta
        limit = timedelta(seconds=300)   # a rank that dies must not leave the others waiting for half an hour
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device), timeout=limit)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=limit)

    if args.c5_child:   # one rank of the child job: the C5 rows, nothing else; rank 0 prints them as one JSON list
        if os.environ.get("MCG_BENCH_C5_CHILD_FAIL") == "1":   # test hook: the child job dies; the parent's line must survive it
            os._exit(3)
        stream = hw.current_stream_handle()

        def row_out(rows):   # each finished row at once: what is done stays done if a later row takes the job down
            if rank == 0:
                print("ROW " + json.dumps(rows[-1]), file=json_out, flush=True)
        c5_sharded_rows(args, mc, N, dist, torch, device, stream, rank, world, hw, make_engine, on_row=row_out,
                        deadline=time.time() + args.c5_budget)
        dist.barrier()
        dist.destroy_process_group()
        return

    S0, K, r, sigma, dt = 100.0, 100.0, 0.04, 0.2, DT
    n_steps, seed = args.time_steps, SEED
    T = n_steps * dt
    total_paths = args.paths * world
    begin, count = shard_range(total_paths, rank, world, align=2 if args.config == "c5" else 1)

    stream = hw.current_stream_handle() if dist is not None else None
    eng = make_engine() if make_engine else mc.PathEngine(device, stream=stream)
    collective = "none"
    if dist is not None:
        collective = args.collective
        if collective == "auto":   # c5: the in-kernel mailbox in peer memory (xGMI on a node) first; it falls back by itself
            collective = "ipc" if args.config == "c5" else "rccl"
        collective = install_collective(eng, mc, collective, dist, torch, rank, world, dev_name, scratch=make_engine)

    if args.config == "c2":
        k_main = N.K_GBM
        alg_bytes = 8.0 * (n_steps + 1) * count          # 8*(steps+1) B written per path

        def one_pass():
            P = eng.gbm(seed, S0, r, sigma, dt, n_steps, count, path_begin=begin, payoff=(K, True))
            price, se = eng.price_european(P, K, r, T, True)
            P.free()
            return price, se

        def ramp_launch():
            eng.gbm(seed, S0, r, sigma, dt, n_steps, count, path_begin=begin).free()
    else:
        k_main = N.K_RBERGOMI
        alg_bytes = 8.0 * (n_steps + 1) * count

        def one_pass():
            P = eng.rbergomi(seed, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], dt, n_steps, count, path_begin=begin)
            price, se = eng.price_lsm(P, RB["r"], K, T, dt, False, 2)
            P.free()
            return price, se

        def ramp_launch():
            eng.rbergomi(seed, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], dt, n_steps, count, path_begin=begin).free()

    RAMP_LAUNCHES = 12 if args.config == "c2" else 4

    def run_ramp():
        for _ in range(RAMP_LAUNCHES):
            ramp_launch()  # asynchronous, one reused buffer

    def fence(ramp=False):
        """barrier + synchronize on both sides of the timed region.
        ramp: an MI355X that has been idle needs ~12 launches of this kernel (50 ms) to settle at its clock under this
        load -- in a fresh process the per-launch time runs 5.7, 5.0, 4.7, 4.4, 4.3, 4.2, ... 4.02 ms (tools/ramp_exp.py) --
        and it drops out of that state again during the 0.2-2 ms the host spends in the barrier.  The throughput of
        interest is the steady one, so before the start barrier every rank queues RAMP_LAUNCHES untimed generator
        launches (asynchronously, one reused buffer): the device is at load while the host sits in the barrier and the
        queue has drained before the clock starts.  Same at every N; reported as config.untimed_ramp_launches (the same
        ramp also precedes the W warm-up steps)."""
        eng.synchronize()
        hw.synchronize()
        if ramp:
            run_ramp()
        if dist is not None:
            dist.barrier()
        eng.synchronize()
        hw.synchronize()

    # the very first launch of the measured kernel in this process, on a device that has been idle: reported, not timed
    eng.timing_enable(True)
    eng.timing_reset()
    ramp_launch()
    eng.synchronize()
    cold_ms = eng.timing_get(k_main)[0]
    eng.timing_enable(False)
    run_ramp()  # also ahead of the W warm-up steps: every launch of the measured kernel variant runs at the steady clock,
    #             so the rocprofv3 average over all of them agrees with the average over the K timed ones
    for _ in range(args.warmup):
        one_pass()
    fence(ramp=True)
    # HIP events around the kernels the roofline is computed from only (an event pair costs several microseconds on the
    # stream; the reductions behind the generator and the transfers are not bracketed inside the timed region)
    eng.timing_select([k_main] if args.config == "c2" else [k_main, N.K_LSM_SWEEP])
    eng.timing_enable(True)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        price, se = one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    eng.timing_select(None)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_name)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mode = c5_rows_mode(args, world) if (dist is not None and args.config == "c2" and not args.no_extra) else "off"

    def c5_rows_after(out):
        """BASELINE.json configs[4] at this N, through every collective in turn (all ranks take part; rank 0 reports) -- AFTER
        the headline dict is complete and with rank 0's guardian: these routes have not run on more than one GPU, and
        whatever goes wrong in them -- an exception, a time-out, a refused process, a device fault -- must not cost the line."""
        if mode == "off":
            return None
        if will is not None:
            will.update(out)
        eng.trim()
        if mode == "child":
            return c5_rows_in_child_job(args, dist, torch, rank, world, args.c5_budget)
        return c5_rows_inline(args, mc, N, dist, torch, device, stream, rank, world, hw, make_engine, will, out, args.c5_budget)

    if args.rehearsal:   # nothing was computed or timed: say what ran, and stop
        out = {"rehearsal": True, "metric": "Mpaths/sec at 252 steps", "value": None, "unit": "Mpaths/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "data": "none (control-flow rehearsal on CPU ranks: nothing computed, nothing timed)",
               "config": {"workload": "the N > 1 control flow of this script", "paths_per_gpu": args.paths, "global_paths": total_paths,
                          "collective": collective, "comm": eng.comm_info(), "c5_rows": mode},
               "ids_counted": price, "ids_summed": se}     # every rank's shard went through the collective once
        c5_rows = c5_rows_after(out)
        if rank == 0:
            out["extra"] = {"configs": c5_rows}
            will.final(out)
        eng.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    k_ms, k_n = eng.timing_get(k_main)
    sweep_ms, sweep_n = eng.timing_get(N.K_LSM_SWEEP)
    solve_ms, solve_n = eng.timing_get(N.K_LSM_SOLVE)
    # What lets a reader tell a slow board from a regression (boards of one pool differ by ~10 % on this power-limited
    # kernel): the shader clock the timed launches ran at, stamped inside the last one by ~60 workgroups, and -- right after
    # the timed region, device still at load -- what THIS board writes with the matrix's store pattern and no arithmetic.
    clock, ceiling = None, None
    if rank == 0:
        try:
            eng.timing_enable(False)
            if args.config == "c2":   # one ARMED launch right behind the timed ones (device at load): the timed launches carry no stamps
                eng.generator_clock_arm(True)
                ramp_launch()
                eng.synchronize()
                eng.generator_clock_arm(False)
                clock = eng.generator_clock()
            ceiling = eng.probe_write_ceiling(count, n_steps, reps=5)
        except Exception as e:   # noqa: BLE001 -- measurement aids only
            print(f"bench: board probe failed ({e})", file=sys.stderr)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_paths * args.steps / elapsed / 1e6
        k_avg_ms = k_ms / max(k_n, 1)
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.config == "c2" and os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("paths") == count and j.get("time_steps") == n_steps:
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/pmc_traffic.json (committed: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes "
                                      "of this command, tools/pmc_traffic.sh; not re-measured in this run)")
            except Exception:
                traffic = None
        if args.config == "c2":
            workload = (f"C2: European call, GBM, {args.paths // 1_000_000}M paths x {n_steps} steps per GPU, fp64 matrix written" if args.paths % 1_000_000 == 0
                        else f"C2: European call, GBM, {args.paths} paths x {n_steps} steps per GPU, fp64 matrix written")
            sharding = f"contiguous path ids over {world} rank(s); one 3-double all-reduce"
            ref = bs_call(S0, K, r, sigma, T)
            parity = {"price": price, "std_err": se, "black_scholes": ref,
                      "abs_err_over_std_err": abs(price - ref) / se if se > 0 else None}
            kernel_name = "k_gbm_paths"
        else:
            workload = ("C5: rBergomi (H=0.1, eta=1.9) American put, Longstaff-Schwartz order 2, 8M paths x 252 steps per GPU "
                        "(8 GPUs: the 64M-path job), fp64 matrix written then swept backwards")
            sharding = (f"contiguous even-aligned path ids over {world} rank(s); per exercise date 8 regression moments are "
                        "summed over the ranks (shm / ipc: inside the one-launch sweep through the node mailbox in host memory / in "
                        "peer-mapped device memory; rccl / torch: one all-reduce between two launches of the per-date kernel), then "
                        "3 doubles of final sums")
            parity = {"price": price, "std_err": se}
            kernel_name = "k_rbergomi_fft"
        out = {
            "metric": "Mpaths/sec at 252 steps", "value": value, "unit": "Mpaths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "paths_per_gpu": args.paths, "time_steps": n_steps, "global_paths": total_paths,
                       "sharding": sharding, "collective": collective,
                       "S0": S0, "K": K, "r": r, "sigma": sigma if args.config == "c2" else None, "seed": seed,
                       "untimed_ramp_launches": RAMP_LAUNCHES},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name,
                         "kernel_avg_ms": k_avg_ms, "launches": int(k_n),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "board_write_ceiling_GBs": ceiling[0] if ceiling else None,
                         "board_write_ceiling_ms_per_launch": ceiling[1] if ceiling else None,
                         "frac_of_board_ceiling": achieved / ceiling[0] if ceiling else None,
                         "board_write_ceiling_source": "mcg_probe_write_ceiling: 5 launches, right after the timed region, of a kernel that "
                                                       "stores the same matrix with the generator's store pattern and no arithmetic",
                         "shader_clock_GHz": clock,
                         "shader_clock_source": "s_memtime / s_memrealtime stamps of ~40 workgroups of one armed k_gbm_paths launch right behind the timed "
                                                "ones (mcg_generator_clock_arm / mcg_generator_clock)" if clock else None},
            "parity": parity,
        }
        if args.config == "c5":
            per_pass = max(args.steps, 1)
            one_launch = sweep_n // per_pass <= 2
            design = (16.0 if one_launch else 32.0) * n_steps * count
            out["roofline"]["lsm"] = {
                "sweep_ms_per_pass": sweep_ms / per_pass, "sweep_launches_per_pass": sweep_n // per_pass,
                "solve_ms_per_pass": solve_ms / per_pass, "solve_launches_per_pass": solve_n // per_pass,
                "survey_two_pass_bytes_per_pass": 40.0 * n_steps * count,   # SURVEY 8(d)'s formulation; context only, not what is moved
                "bytes_moved_per_pass": design,   # what this execution shape moves: 16 B one-launch (k_lsm_big), 32 B per-date kernels
                "hbm_frac": design / max(sweep_ms / per_pass * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                "shape": "one launch (V in registers; beyond 2.09M paths the matrix streams through an LDS-DMA ring)" if one_launch
                         else "per-date route: one kernel + one all-reduce of 8 moments per exercise date; sweep_ms is the span of the "
                              "queued sequence (launches, dispatch gaps, collectives)"}
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_c5_pmc_traffic.json")))   # the latest round's
            pmc5 = cand[-1] if cand else ""
            if pmc5 and count == 8_000_000 and n_steps == 252:
                try:
                    j5 = json.load(open(pmc5))
                    out["roofline"]["traffic"] = j5["generator"]["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = (f"profiles/{os.path.basename(pmc5)} (committed: rocprofv3 --pmc WRITE_SIZE / "
                                                         "FETCH_SIZE passes of this command, tools/profile_r0N.sh; not re-measured here)")
                    if one_launch:
                        out["roofline"]["lsm"]["traffic"] = j5["lsm_one_launch"]["hbm_bytes_per_launch"]
                    elif "lsm_per_date_launch" in j5:
                        out["roofline"]["lsm"]["traffic_per_launch"] = j5["lsm_per_date_launch"]["hbm_bytes_per_launch"]
                except Exception:
                    pass
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(n_steps)
                ref_price = cb.pop("_ref_price", None)
                out["cpu_baseline"] = cb
                if ref_price is not None:
                    out["parity"]["vs_reference"] = reference_parity(eng, mc, ref_price, n_steps, seed)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mpaths/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        if world == 1 and dist is None and args.config == "c2" and not args.no_extra:
            try:
                out["parity"]["rough_regime_vs_reference_sample"] = rough_regime_parity(eng)
                base = None
                if not args.no_cpu_baseline:
                    try:
                        base = cpu_baselines_widened()
                    except Exception as e:   # noqa: BLE001 -- reported baselines, never required
                        print(f"bench: CPU baselines of the widened rows failed ({e})", file=sys.stderr)
                out["extra"] = {"configs": extra_configs(eng, N, base) + widening_configs(eng, N, mc, base)}
                out["extra"]["c2_cold_first_launch_ms"] = cold_ms
            except Exception as e:
                out["extra"] = {"configs": [], "error": str(e)}
    c5_rows = None
    if mode != "off":
        if rank == 0:
            out["config"]["comm"] = eng.comm_info()
            out["config"]["c5_rows"] = mode
        c5_rows = c5_rows_after(out if rank == 0 else None)
    if rank == 0:
        if c5_rows is not None:
            out.setdefault("extra", {})["configs"] = c5_rows
        out["config"]["comm"] = eng.comm_info()
        if will is not None:
            will.final(out)
        else:
            print(json.dumps(out), file=json_out, flush=True)
    eng.timing_enable(False)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
