#!/bin/bash
# A/B on ONE box: alternate the headline bench between the in-tree build and montecarlooptionspricer_amd/lib/libmcgpu_base.so
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_base.so; else unset MCG_LIB; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$which  Mpaths/s %.1f  kernel %.3f ms  frac %.3f' % (j['value'], r['kernel_avg_ms'], r['frac']))"
  done
done
