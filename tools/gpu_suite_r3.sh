#!/bin/bash
# GPU box (gpurun), round 3: the GPU tests, the default bench line, the C5 lines at world size 1 through every collective
# (RCCL = the per-date route: one kernel + one all-reduce per date; shm / ipc = the one-launch sweep with the mailbox).
# Output in gpurun_out/r3a_*.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r3a}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/${T}_pytest.log
tail -15 gpurun_out/${T}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
for c in rccl shm ipc; do
  timeout -k 10 300 env MCG_FORCE_DIST=1 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective $c > gpurun_out/${T}_c5_${c}1.json 2> gpurun_out/${T}_c5_${c}1.err; echo "c5 $c rc=$?"
done
timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5_n1.json 2> gpurun_out/${T}_c5_n1.err; echo "c5 n1 rc=$?"
python - <<PY
import json
for f in ("bench","c5_rccl1","c5_shm1","c5_ipc1","c5_n1"):
    try:
        j=json.load(open(f"gpurun_out/${T}_%s.json"%f))
        print(f, round(j["value"],1), "Mpaths/s", round(j["ms_per_step"],3), "ms", j["config"].get("collective"), j["config"].get("comm"), {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"].get("lsm",{}).items() if k in ("sweep_ms_per_pass","sweep_launches_per_pass","solve_ms_per_pass")}, "frac", round(j["roofline"]["frac"],3))
    except Exception as e:
        print(f, "failed", e)
PY
