#!/bin/bash
# GPU box: parity of the BranchingProcesses slice routes and of the batched rows with the pricers on side streams; A/B of
# the batched rows (pricers one after the other / side by side) on one board.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r4g}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -k "branching or batch" --durations=8 > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -16 gpurun_out/${T}_pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  for v in serial new; do
    if [ $v = new ]; then unset MCG_LIB; else export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_$v.so; fi
    echo "== $v $(timeout -k 10 200 python tools/bench_rows.py 2>/dev/null | tail -1)"
  done
done 2>&1 | tee gpurun_out/${T}_ab_rows.log
unset MCG_LIB
