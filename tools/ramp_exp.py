"""Kernel duration of consecutive headline launches in a fresh process (clock / power ramp of the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd import _native as N
eng = mc.PathEngine(0)
eng.timing_enable(True)
out = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    eng.timing_reset()
    P = eng.gbm(1, 100.0, 0.04, 0.2, 1 / 252, 252, 10_000_000, payoff=(100.0, True))
    eng.price_european(P, 100.0, 0.04, 1.0, True)
    P.free()
    ms, n = eng.timing_get(N.K_GBM)
    out.append(ms / n)
print(" ".join("%.2f" % x for x in out))
