"""Fixture matrices for the Eigen-backed pin of LSM and MartingaleOptimization.  TEST INFRASTRUCTURE ONLY.

`python oracle/gen_golden.py --eigen` runs every case below through the reference's LSMPricer.cpp /
MartingaleOptimizationPricer.cpp compiled in place against an Eigen3 (oracle/_ref/libmcref_eigen.so) and stores inputs and
outputs in tests/golden/{lsm,martingale}.npz.  The cases are the ones the oracle's own LSM tests already lean on
(tests/test_oracle_models.py): spread GBM matrices at the orders the driver and the device use (0, 2, 3, 5), an all-OTM
column (LSMPricer.cpp:89-94 alone), a grid longer than the maturity (:43-49), a date with a single in-the-money path (rank-1
system), S0 in the money at j = 0 (all rows identical: the rank rule of bdcSvd decides the fit), N = 1, near-coincident
in-the-money prices, and the driver's own row shape (250 paths, order 2, PredictionGen.cpp:719, :790).
Inputs are generated from numpy's RandomState only -- nothing here needs the oracle or the product."""
from __future__ import annotations

import numpy as np


def _gbm(rs, n, steps, dt, S0=100.0, r=0.04, sigma=0.2):
    z = rs.standard_normal((n, steps))
    inc = (r - 0.5 * sigma * sigma) * dt + sigma * np.sqrt(dt) * z
    P = np.empty((n, steps + 1))
    P[:, 0] = S0
    P[:, 1:] = S0 * np.exp(np.cumsum(inc, axis=1))
    return P


def lsm_cases():
    """name -> dict(paths [n][m], r, K, maturity, dt, is_call, poly, tol): tol = the relative tolerance the restatement (and
    the device) will be held to against Eigen on that case."""
    rs = np.random.RandomState(20251031)
    c = {}
    g = _gbm(rs, 2000, 50, 0.02)
    for tag, is_call, poly, tol in (("gbm_put_o2", 0, 2, 1e-9), ("gbm_call_o2", 1, 2, 1e-9), ("gbm_put_o3", 0, 3, 1e-8),
                                    ("gbm_put_o0", 0, 0, 1e-12), ("gbm_put_o5", 0, 5, 5e-6)):
        c[tag] = dict(paths=g, r=0.04, K=100.0, maturity=1.0, dt=0.02, is_call=is_call, poly=poly, tol=tol)
    row = _gbm(rs, 250, 60, 1 / 252.0, S0=166.5, sigma=0.21)                      # the driver's row shape
    c["driver_row_put"] = dict(paths=row, r=0.04, K=166.5, maturity=60 / 252.0, dt=1 / 252.0, is_call=0, poly=2, tol=1e-8)
    c["driver_row_call"] = dict(paths=row, r=0.04, K=170.0, maturity=60 / 252.0, dt=1 / 252.0, is_call=1, poly=2, tol=1e-8)
    otm = _gbm(rs, 400, 12, 0.05)
    otm[:, 6] = 150.0 + rs.rand(400)                                               # one column entirely out of the money
    c["all_otm_column"] = dict(paths=otm, r=0.04, K=100.0, maturity=0.6, dt=0.05, is_call=0, poly=2, tol=1e-9)
    mixed = 100.0 * np.exp(np.cumsum(0.1 * rs.standard_normal((300, 9)), axis=1))
    c["grid_past_maturity"] = dict(paths=mixed, r=0.04, K=100.0, maturity=0.35, dt=0.1, is_call=0, poly=2, tol=1e-9)
    one = np.full((40, 5), 130.0) + rs.rand(40, 5)
    one[:, -1] = 80.0 + 10.0 * rs.rand(40)
    one[7, 2] = 91.0                                                               # the only in-the-money path of date 2
    c["single_itm_path"] = dict(paths=one, r=0.04, K=100.0, maturity=1.0, dt=0.25, is_call=0, poly=2, tol=1e-6)
    itm0 = _gbm(rs, 500, 20, 0.05, S0=90.0)
    c["s0_in_the_money"] = dict(paths=itm0, r=0.04, K=100.0, maturity=1.0, dt=0.05, is_call=0, poly=2, tol=1e-6)
    c["n_equals_1"] = dict(paths=np.array([[100.0, 90.0, 95.0, 85.0]]), r=0.04, K=100.0, maturity=1.0, dt=0.25, is_call=0, poly=2,
                           tol=1e-6)
    k = 0
    for base in (90.0, 99.0, 60.0):
        for n, spread in ((3, 1e-3), (4, 1e-5), (5, 1e-7)):
            S = base * (1 + spread * rs.uniform(-1, 1, n))
            b = rs.uniform(45.0, 60.0, n)
            c[f"near_degenerate_{k}"] = dict(paths=np.stack([S, 100.0 - b], axis=1), r=0.0, K=100.0, maturity=1.0, dt=1.0, is_call=0,
                                             poly=2, tol=2e-5)
            k += 1
    return c


def martingale_cases():
    """name -> dict(paths, r, K, maturity, dt, is_call, poly, iters, tol)."""
    rs = np.random.RandomState(20251032)
    c = {}
    g = _gbm(rs, 2000, 50, 0.02)
    for tag, is_call, poly, iters, tol in (("gbm_put_o2", 0, 2, 5, 1e-8), ("gbm_call_o2", 1, 2, 5, 1e-8), ("gbm_put_o3_i3", 0, 3, 3, 1e-7),
                                           ("gbm_put_o5", 0, 5, 5, 5e-6), ("gbm_put_o2_i1", 0, 2, 1, 1e-8)):
        c[tag] = dict(paths=g, r=0.04, K=100.0, maturity=1.0, dt=0.02, is_call=is_call, poly=poly, iters=iters, tol=tol)
    row = _gbm(rs, 250, 60, 1 / 252.0, S0=166.5, sigma=0.21)
    c["driver_row_put"] = dict(paths=row, r=0.04, K=166.5, maturity=60 / 252.0, dt=1 / 252.0, is_call=0, poly=2, iters=5, tol=1e-7)
    c["driver_row_call"] = dict(paths=row, r=0.04, K=170.0, maturity=60 / 252.0, dt=1 / 252.0, is_call=1, poly=2, iters=5, tol=1e-7)
    mixed = 100.0 * np.exp(np.cumsum(0.1 * rs.standard_normal((300, 9)), axis=1))
    c["grid_past_maturity"] = dict(paths=mixed, r=0.04, K=100.0, maturity=0.35, dt=0.1, is_call=0, poly=2, iters=5, tol=1e-7)
    return c
