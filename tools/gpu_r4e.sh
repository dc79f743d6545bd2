#!/bin/bash
# GPU box: the sliced BranchingProcesses gathers -- parity tests, A/B against the previous build on one board, counters.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r4e}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "branching or batch_rows" --durations=8 > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/${T}_pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1; do
  for v in new; do
    if [ $v = new ]; then unset MCG_LIB; else export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_$v.so; fi
    echo "== $v"; timeout -k 10 200 python tools/bench_branching.py 2>/dev/null
  done
done 2>&1 | tee gpurun_out/${T}_ab_branching.log
unset MCG_LIB
for c in TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $PWD/gpurun_out/${T}_pmc_branch_$c -- python3 tools/bench_branching.py > gpurun_out/${T}_pmc_branch_$c.log 2>&1 || echo "pass $c failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/${T}_branch_stats -- python3 tools/bench_branching.py > gpurun_out/${T}_branch_stats.log 2>&1 || echo "stats pass failed"
echo done
