#!/usr/bin/env python3
"""Rows per second of the batched driver entry point (mcg_batch_price_rows): the reference driver's
per-row work (250 rBergomi paths + AsymptoticAnalysis + BranchingProcesses(10) + LSM(2) + Martingale(2))."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlooptionspricer_amd as mc  # noqa: E402
from montecarlooptionspricer_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=20000)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
rs = np.random.RandomState(0)
rows = []
for i in range(args.rows):
    steps = int(rs.randint(5, 127))                 # dte 7..183 days -> floor(dte/365*252) steps
    S0 = float(rs.uniform(20, 400))
    rows.append(dict(S0=S0, xi=float(rs.uniform(0.01, 0.3)), H=float(rs.uniform(0.3, 0.6)), eta=float(rs.uniform(0.01, 0.06)),
                     rho=-0.3, strike=S0 * float(rs.uniform(0.9, 1.1)), maturity=steps / 252.0 * 252.0 / 365.0 * 365.0 / 252.0,
                     sigma=float(rs.uniform(0.1, 0.6)), dividend=0.08, n_steps=steps, is_call=int(rs.randint(0, 2))))
eng = mc.PathEngine(0)
arr = mc.make_rows(rows) if hasattr(mc, "make_rows") else rows   # the C array, built once (an older build through MCG_LIB: the dicts)
eng.batch_price_rows(arr, seed=1)
eng.timing_enable(True)
eng.timing_reset()
t0 = time.perf_counter()
for _ in range(args.reps):
    out = eng.batch_price_rows(arr, seed=1)
dt = (time.perf_counter() - t0) / args.reps
ms, n = eng.timing_get(N.K_BATCH)
print(f"{args.rows} rows: wall {dt*1e3:.1f} ms/call = {args.rows/dt:.0f} rows/s  (device span {ms/max(n,1):.1f} ms; "
      f"mean prices {out.mean(axis=0)})")
