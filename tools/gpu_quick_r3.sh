#!/bin/bash
# quick GPU check: the GPU tests, then C5 at world size 1 through the collectives named in $2 (default: rccl)
set -o pipefail
export TMPDIR=/tmp
T=${1:-r3q}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -8 gpurun_out/${T}_pytest.log
for c in ${2:-rccl}; do
  timeout -k 10 300 env MCG_FORCE_DIST=1 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective $c > gpurun_out/${T}_c5_${c}1.json 2> gpurun_out/${T}_c5_${c}1.err; echo "c5 $c rc=$?"
  python - <<PY
import json
j=json.load(open("gpurun_out/${T}_c5_${c}1.json"))
print("$c", round(j["value"],1), "Mpaths/s", round(j["ms_per_step"],3), "ms", j["config"].get("collective"), {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"].get("lsm",{}).items() if k in ("sweep_ms_per_pass","sweep_launches_per_pass","solve_ms_per_pass")}, "gen ms", round(j["roofline"]["kernel_avg_ms"],3), j["parity"])
PY
done
