import sys,time
sys.path.insert(0,'.')
import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd import _native as N
e=mc.PathEngine(0); e.timing_enable(True)
DT=1/252
for _ in range(14): e.gbm(20251031,100.0,0.04,0.2,DT,252,10_000_000).free()
e.synchronize()
for rep in range(3):
    for n in (9_961_472, 10_000_000, 10_485_760, 9_437_184, 9_700_000):
        e.timing_reset()
        for _ in range(6):
            P=e.gbm(20251031,100.0,0.04,0.2,DT,252,n,payoff=(100.0,True)); P.free()
        e.synchronize()
        ms,c=e.timing_get(N.K_GBM)
        print(n, 'gens', round(((n+511)//512)/1024,3), 'ms', round(ms/c,4), 'ns per path', round(ms/c*1e6/n,4))
