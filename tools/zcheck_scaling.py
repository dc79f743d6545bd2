import sys, math; sys.path.insert(0,'.')
import montecarlooptionspricer_amd as mc
from montecarlooptionspricer_amd.sharding import combine_sums, price_from_sums
eng=mc.PathEngine(0)
import ctypes as C
DT=1/252; n=10_000_000
def bs(S0,K,r,s,T):
    d1=(math.log(S0/K)+(r+.5*s*s)*T)/(s*math.sqrt(T)); d2=d1-s*math.sqrt(T); N=lambda x:.5*math.erfc(-x/math.sqrt(2))
    return S0*N(d1)-K*math.exp(-r*T)*N(d2)
ref=bs(100,100,.04,.2,1.0)
parts=[]
for g in range(8):
    P=eng.gbm(20251031,100.,.04,.2,DT,252,n,path_begin=g*n,payoff=(100.,True))
    m,se=eng.price_european(P,100.,.04,1.0,True)
    # recover sums
    disc=math.exp(-.04); mean=m/disc
    var=(se/disc)**2*n
    s=mean*n; s2=var*(n-1)+n*mean*mean
    parts.append((s,s2,float(n))); P.free()
    if g+1 in (1,2,4,8):
        mm,ss=price_from_sums(combine_sums(parts),disc)
        print(g+1,'GPUs-equivalent: price %.5f se %.5f z %.2f'%(mm,ss,(mm-ref)/ss))
