#!/usr/bin/env python3
"""Per-config timings (C2..C5 shapes on ONE GPU) with per-kernel HIP-event breakdown.
Dev tool: the judged number comes from bench.py; this shows where each config spends its time."""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlooptionspricer_amd as mc  # noqa: E402
from montecarlooptionspricer_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="c2,c3,c4,c5")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--scale", type=float, default=1.0, help="scale path counts")
args = ap.parse_args()
DT = 1.0 / 252.0
RB = dict(xi=0.04, H=0.1, eta=1.9, rho=-0.9)
eng = mc.PathEngine(0)
eng.timing_enable(True)


def run(name, fn, n_paths, bytes_per_path):
    fn()
    eng.synchronize()
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        res = fn()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    ks = {N.KERNEL_NAMES[k]: eng.timing_get(k) for k in N.KERNEL_NAMES}
    ks = {k: (round(ms / args.reps, 3), n // args.reps) for k, (ms, n) in ks.items() if n}
    print(json.dumps({"config": name, "paths": n_paths, "wall_ms": round(dt * 1e3, 3),
                      "Mpaths_per_s": round(n_paths / dt / 1e6, 1), "result": res,
                      "kernels_ms_per_rep(launches)": ks,
                      "alg_GB": round(bytes_per_path * n_paths / 1e9, 2)}), flush=True)


for c in args.configs.split(","):
    if c == "c2":
        n = int(10_000_000 * args.scale)

        def f():
            P = eng.gbm(20251031, 100.0, 0.04, 0.2, DT, 252, n, payoff=(100.0, True))
            r = eng.price_european(P, 100.0, 0.04, 1.0, True)
            P.free()
            return r
        run("C2 GBM euro call 10Mx252", f, n, 8 * 253)
    elif c == "c3":
        n = int(1_000_000 * args.scale)

        def f():
            P = eng.gbm(20251031, 100.0, 0.04, 0.2, 0.02, 50, n)
            r = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
            P.free()
            return r
        run("C3 GBM american put LSM 1Mx50", f, n, 8 * 51 + 32 * 50)
    elif c == "c4":
        n = int(4_000_000 * args.scale)

        def f():
            P = eng.rbergomi(20251031, 100.0, 0.04, RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 512, n,
                             payoff=(100.0, True))
            r = eng.price_european(P, 100.0, 0.04, 512 * DT, True)
            P.free()
            return r
        run("C4 rBergomi euro call 4Mx512", f, n, 8 * 513)
    elif c == "c5":
        n = int(8_000_000 * args.scale)

        def f():
            P = eng.rbergomi(20251031, 100.0, 0.04, RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 252, n)
            r = eng.price_lsm(P, 0.04, 100.0, 1.0, DT, False, 2)
            P.free()
            return r
        run("C5 shard: rBergomi american put LSM 8Mx252", f, n, 8 * 253 + 32 * 252)
    elif c == "c5pd":   # the same through the per-date route (an identity collective selects it)
        n = int(8_000_000 * args.scale)
        eng.set_allreduce(lambda ptr, count, stream: None)

        def f():
            P = eng.rbergomi(20251031, 100.0, 0.04, RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, 252, n)
            r = eng.price_lsm(P, 0.04, 100.0, 1.0, DT, False, 2)
            P.free()
            return r
        run("C5pd shard, per-date route: rBergomi american put LSM 8Mx252", f, n, 8 * 253 + 32 * 252)
        eng.set_allreduce(None)
eng.close()
