#!/usr/bin/env python3
"""Diagnosis of the round-4 incident: WHERE do the paths of the GBM generator built with its clock
stamps in scalar registers (libmcgpu_sgprstamps.so) differ from the product's?  Each build runs in a process of its own
(MCG_LIB) and dumps its matrix; the parent compares element by element: which rows (steps), which columns (workgroups:
stamping ones or all), by how much."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import montecarlooptionspricer_amd as mc
e = mc.PathEngine(0)
n, steps, sigma = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
P = e.gbm(20251031, 100.0, 0.04, sigma, 1.0 / 252.0, steps, n, path_begin=777, payoff=(101.0, True))
np.save(sys.argv[4], P.to_host_step_major())
print("clock", e.generator_clock())
''' % ROOT
for n, steps, sigma in ((1_049_700, 7, 0.2), (1_049_700, 4, 1.5), (1_049_700, 8, 0.2)):
    out = {}
    for tag, lib in (("product", None), ("sgprstamps", os.path.join(ROOT, "montecarlooptionspricer_amd", "lib", "libmcgpu_sgprstamps.so"))):
        env = dict(os.environ)
        if lib:
            env["MCG_LIB"] = lib
        f = f"/tmp/diag_{tag}.npy"
        r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(steps), str(sigma), f], env=env, capture_output=True, text=True)
        print(tag, r.stdout.strip(), r.stderr.strip()[-300:])
        out[tag] = np.load(f)
    a, b = out["product"], out["sgprstamps"]
    rel = np.abs(a - b) / np.abs(a)
    print(f"n={n} steps={steps} sigma={sigma}: max rel {rel.max():.3e}")
    for j in range(steps + 1):
        bad = np.nonzero(rel[j] > 1e-13)[0]
        wg = np.unique(bad // 512)
        print(f"  step {j}: {len(bad)} columns differ (max {rel[j].max():.2e}) in {len(wg)} workgroups of {(n + 511) // 512}; first wgs {wg[:8]}, lanes(within wg) sample {np.unique(bad % 512)[:10]}")
