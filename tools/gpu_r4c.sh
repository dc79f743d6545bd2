#!/bin/bash
# GPU box: which build of the GBM generator passes the two-paths-per-lane parity test (clock stamps held in SGPRs / none /
# held in LDS), then the suite on the in-tree build.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r4c}
mkdir -p gpurun_out
for v in sgprstamps nostamps new; do
  if [ $v = new ]; then unset MCG_LIB; else export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_$v.so; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "two_paths_per_lane or gbm_paths_match or gbm_layouts" > gpurun_out/${T}_gbm_$v.log 2>&1; echo "$v rc=$? $(tail -1 gpurun_out/${T}_gbm_$v.log)"
done
unset MCG_LIB
timeout -k 10 300 tools/ab_libs.sh c2 3 new nostamps 2>&1 | tee gpurun_out/${T}_ab_stamps.log
