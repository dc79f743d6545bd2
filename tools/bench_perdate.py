#!/usr/bin/env python3
"""Per-date LSM route (world size 1, identity collective) on the C5 shard at orders 2 and 5: ms per sweep, launches and
read-back rounds (the route's batches at orders >= 4).  tools/ab_libs.sh-style: MCG_LIB selects the library."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlooptionspricer_amd as mc  # noqa: E402
from montecarlooptionspricer_amd import _native as N  # noqa: E402

SEED, DT = 20251031, 1.0 / 252.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
e = mc.PathEngine(0)
calls = []
e.set_allreduce(lambda ptr, count, stream: calls.append(count))
P = e.rbergomi(SEED, 100.0, 0.04, 0.04, 0.1, 1.9, -0.9, DT, 252, n)
for order in (2, 5):
    e.price_lsm(P, 0.04, 100.0, 1.0, DT, False, order)
    calls.clear()
    mc.stats(reset=True)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        price, _ = e.price_lsm(P, 0.04, 100.0, 1.0, DT, False, order)
    e.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    s = mc.stats()
    print(f"order {order}: {ms:.3f} ms per sweep, {s['lsm_per_date_launches'] // 3} launches, {calls.count(1) // 3} read-back rounds, price {price:.6f}")
P.free()
e.close()
