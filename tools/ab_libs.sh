#!/bin/bash
# A/B/C... on ONE box: alternate tools/bench_configs.py between montecarlooptionspricer_amd/lib/libmcgpu_<name>.so builds
# ("new" = the in-tree libmcgpu.so).   tools/ab_libs.sh c4,c5 3 base new w3
CFG=${1:-c4,c5}; N=${2:-3}; shift 2
for i in $(seq $N); do
  for which in "$@"; do
    if [ $which = new ]; then unset MCG_LIB; else export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_$which.so; fi
    python tools/bench_configs.py --configs $CFG --reps 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); k=j['kernels_ms_per_rep(launches)']
    print('$which', j['config'][:10], 'wall %.3f' % j['wall_ms'], ' '.join('%s %.3f' % (a,b[0]) for a,b in k.items()), 'price %.6f' % j['result'][0])"
  done
done
