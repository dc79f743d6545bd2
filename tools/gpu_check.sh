#!/bin/bash
# quick GPU loop: math + parity tests, then the headline bench without the CPU baseline leg
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout -k 10 300 python bench.py --steps ${1:-5} --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('Mpaths/s %.1f  kernel %.3f ms  %.0f GB/s  frac %.3f  z %.2f' % (j['value'], r['kernel_avg_ms'], r['achieved'], r['frac'], j['parity']['abs_err_over_std_err']))"
