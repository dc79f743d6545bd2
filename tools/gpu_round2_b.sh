#!/bin/bash
# GPU box: the LSM one-launch variants (equality with the per-date kernels), the C5 shard test, C5 timing.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q -k "single_launch or one_launch or c5_shard or lsm" > gpurun_out/r2b_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r2b_pytest.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_configs.py --configs c3,c5 --reps 3 > gpurun_out/r2b_configs.log 2>&1; echo "configs rc=$?"; cat gpurun_out/r2b_configs.log | tail -5
