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

from tools.bench_common import DT, HBM_PEAK_GBS, RB, SEED  # noqa: E402
from tools.bench_extra import (extra_configs, reference_parity, rough_regime_parity, unchanged_driver_row,  # noqa: E402
                               widening_configs)
from tools.bench_multirank import (ABANDONED, GPU_PROCESS_GUARD, Gpu, LastWill, c5_rows_in_child_job, c5_rows_inline,  # noqa: E402
                                   c5_rows_mode, c5_sharded_rows, install_collective)


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


def leave(eng, dist, json_out) -> None:
    """The end of a rank.  Normally: close the context, meet the peers, leave the process group.  With a scratch context
    abandoned inside ncclCommInitRank (tools/bench_multirank.py: rccl_forms_in_time) a thread of this process still holds a
    GPU context and a half-formed communicator: closing contexts, destroying the process group or running library
    destructors at interpreter exit could block on it.  The line is out by now (rank 0: printed, or with the guardian, which
    has been waited for) -- so such a rank still meets its peers in the barrier (torch's own group is not affected) and then
    leaves through os._exit(0)."""
    abandoned = ABANDONED["rccl_init"]
    if not abandoned:
        eng.timing_enable(False)
        eng.close()
    if dist is not None:
        dist.barrier()
    if abandoned:
        json_out.flush()
        sys.stderr.flush()
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


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
            if ABANDONED["rccl_init"]:
                out["config"]["rccl_init_abandoned"] = True
            will.final(out)
        leave(eng, dist, json_out)
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
                eng.trim()   # (the child processes of the next row bring their own contexts)
                out["extra"]["configs"].append(unchanged_driver_row(base))
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
        if ABANDONED["rccl_init"]:
            out["config"]["rccl_init_abandoned"] = True   # a scratch context is still inside ncclCommInitRank on this rank
        if will is not None:
            will.final(out)
        else:
            print(json.dumps(out), file=json_out, flush=True)
    leave(eng, dist, json_out)


if __name__ == "__main__":
    main()
