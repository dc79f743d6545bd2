#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "martingale or batch" > gpurun_out/r2g_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/r2g_pytest.log
