#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/rough_regime_reference.json: statistics of a sample drawn with the
COMPILED REFERENCE (oracle/_ref/libmcref.so) at BASELINE.json's rough-volatility parameters (C4: 512 steps,
C5: 252 steps; xi = 0.04, H = 0.1, eta = 1.9, rho = -0.9, S0 = K = 100, r = 0.04).

The reference's public entry point cannot reach this regime (it estimates eta = 2 stdev(returns) from a price
history); oracle/ref_harness.cpp: ref_explicit_stats_omp chains the reference's own private members per path.  The
reference seeds from std::random_device, so every run of this script gives a different (equally valid) sample; the
committed file is one of them and is only ever compared statistically (|diff| <= 2 combined standard errors).

    python oracle/gen_rough_fixture.py [paths_252] [paths_512] [--add]

--add draws that many MORE paths and merges them into the committed file (the file keeps the per-statistic sums and sums
of squares, so samples combine exactly); round 3 did so, 2e6 + 8e6 paths at 252 steps and 1e6 + 8e6 at 512.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.binding import STAT_NAMES, Reference, mean_and_se  # noqa: E402

PARAMS = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9, strike=100.0)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    add = "--add" in sys.argv[1:]
    n252 = int(args[0]) if len(args) > 0 else 2_000_000
    n512 = int(args[1]) if len(args) > 1 else 1_000_000
    ref = Reference()
    path = os.path.join(ROOT, "tests", "golden", "rough_regime_reference.json")
    old = json.load(open(path)) if add else None
    out = {"generator": "oracle/gen_rough_fixture.py via oracle/_ref/libmcref.so (ref_explicit_stats_omp)",
           "params": PARAMS, "stat_names": list(STAT_NAMES), "samples": {}}
    for steps, n in ((252, n252), (512, n512)):
        t0 = time.time()
        th, s, s2 = ref.explicit_stats(PARAMS["S0"], PARAMS["r"], PARAMS["xi"], PARAMS["H"], PARAMS["eta"], PARAMS["rho"],
                                       steps, n, PARAMS["strike"])
        if old is not None:
            assert old["params"] == PARAMS and old["stat_names"] == list(STAT_NAMES)
            prev = old["samples"][str(steps)]
            s, s2, n = s + np.array(prev["sums"]), s2 + np.array(prev["sums_sq"]), n + prev["paths"]
        m, se = mean_and_se(s, s2, n)
        out["samples"][str(steps)] = {"paths": n, "sums": s.tolist(), "sums_sq": s2.tolist(), "mean": m.tolist(),
                                      "std_err": se.tolist()}
        print(f"steps {steps}: {n} paths, {th} threads, {time.time() - t0:.1f} s")
        for name, a, b in zip(STAT_NAMES, m, se):
            print(f"   {name:15s} {a:.8g} +- {b:.3g}")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
