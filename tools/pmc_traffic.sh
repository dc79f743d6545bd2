#!/bin/bash
# Run on the GPU box: HBM traffic of the headline kernel, WRITE_SIZE and FETCH_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md, HBM section).  tools/pmc_traffic_summary.py turns the CSVs into profiles/pmc_traffic.json.
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_traffic_w -- $CMD > gpurun_out/pmc_traffic_w.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_traffic_r -- $CMD > gpurun_out/pmc_traffic_r.log 2>&1
echo "rc=$?"
