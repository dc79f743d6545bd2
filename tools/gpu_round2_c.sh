#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/r2c_pytest.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_configs.py --configs c3,c5 --reps 3 > gpurun_out/r2c_configs.log 2>&1; echo "configs rc=$?"; tail -3 gpurun_out/r2c_configs.log
[ $rc -eq 0 ] && timeout -k 10 200 python tools/bench_rows.py > gpurun_out/r2c_rows.log 2>&1; tail -3 gpurun_out/r2c_rows.log
