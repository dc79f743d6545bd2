#!/usr/bin/env python3
"""Device time of BranchingProcesses::PredictOptionPrice (mcg_price_branching: suffix maxima + bounds kernels) on GBM
matrices of a few shapes; dev tool (the judged rows are bench.py's extra.configs)."""
import sys,time
sys.path.insert(0,'.')
import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd import _native as N
e=mc.PathEngine(0); e.timing_enable(True)
import os
shapes=[tuple(int(x) for x in s.split("x")) for s in os.environ.get("BRANCH_SHAPES","1000000x50,4000000x50,250000x252").split(",")]
for n,steps in shapes:
    P=e.gbm(20251031,100.0,0.04,0.2,1.0/steps,steps,n)
    ex=list(range(steps))
    e.price_branching(P,0.04,100.0,1.0,1.0/steps,False,10,ex,7)
    e.timing_reset()
    r=[e.price_branching(P,0.04,100.0,1.0,1.0/steps,False,10,ex,7) for _ in range(3)]
    ms,c=e.timing_get(N.K_BRANCHING)
    print(n,steps,'branching kernels ms per call',round(ms/3,3),'launches',c//3,r[0])
    P.free()
