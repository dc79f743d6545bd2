#!/bin/bash
# GPU box, round 4, first call: the GPU suite with per-test durations, the GBM generator's LDS-table variants A/B/C on one
# board (34 KiB of tables leave 4 workgroups per CU), and the cache counters of k_branch_bounds.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r4a}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/${T}_pytest.log
tail -25 gpurun_out/${T}_pytest.log
[ $rc -eq 0 ] || exit $rc
echo "== GBM table variants (new = 34 KiB, gt1 = 32 KiB, gt2 = 24 KiB)"
timeout -k 10 600 tools/ab_libs.sh c2 4 new gt1 gt2 2>&1 | tee gpurun_out/${T}_ab_gbm_tables.log
echo "== counters of the branching kernels"
rocprofv3 -L > gpurun_out/${T}_counters_avail.txt 2>&1 || true
for c in TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum FETCH_SIZE TCC_REQ_sum; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $PWD/gpurun_out/${T}_pmc_branch_$c -- python3 tools/bench_branching.py > gpurun_out/${T}_pmc_branch_$c.log 2>&1 || echo "pass $c failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/${T}_branch_stats -- python3 tools/bench_branching.py > gpurun_out/${T}_branch_stats.log 2>&1 || echo "stats pass failed"
echo done
