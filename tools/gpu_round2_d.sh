#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rbergomi or rough or class_api or batch" > gpurun_out/r2d_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r2d_pytest.log
timeout -k 10 300 bash tools/ab_configs.sh c4,c5 3 > gpurun_out/r2d_ab.log 2>&1; cat gpurun_out/r2d_ab.log
