#!/bin/bash
# A/B on ONE box: alternate tools/bench_configs.py between montecarlooptionspricer_amd/lib/libmcgpu_base.so and the in-tree build.
#   tools/ab_configs.sh c4,c5 [rounds]
CFG=${1:-c4,c5}; N=${2:-3}
for i in $(seq $N); do
  for which in base new; do
    if [ $which = base ]; then export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_base.so; else unset MCG_LIB; fi
    python tools/bench_configs.py --configs $CFG --reps 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); k=j['kernels_ms_per_rep(launches)']
    print('$which', j['config'][:10], 'wall %.3f' % j['wall_ms'], ' '.join('%s %.3f' % (a,b[0]) for a,b in k.items()), 'price %.6f' % j['result'][0])"
  done
done
