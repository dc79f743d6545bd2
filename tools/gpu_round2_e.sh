#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k "two_rank" > gpurun_out/r2e_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -40 gpurun_out/r2e_pytest.log
