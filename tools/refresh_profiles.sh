#!/bin/bash
# Run on the GPU box (gpurun): re-collect the rocprofv3 kernel statistics that profiles/ holds.
# Output goes to gpurun_out/prof_*; copy the *_kernel_stats.csv files into profiles/ afterwards.
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_c2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c345 -- python3 tools/bench_configs.py --configs c3,c4,c5 --reps 2 > gpurun_out/prof_c345.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_rows -- python3 tools/bench_rows.py > gpurun_out/prof_rows.log 2>&1 &&
python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err &&
python3 tools/bench_configs.py --configs c1,c3,c4,c5 --reps 3 > gpurun_out/configs.log 2>&1 &&
python3 tools/bench_rows.py > gpurun_out/rows.log 2>&1
echo "rc=$?"
find gpurun_out -name "*kernel_stats.csv" | head
