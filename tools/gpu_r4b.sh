#!/bin/bash
# GPU box, round 4: the round's new tests first (fail fast), then the whole GPU suite with durations and the library's
# event counters, then the default bench line and the C5 lines at world size 1 through every collective.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r4b}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q --durations=10 > gpurun_out/${T}_new.log 2>&1; rc=$?; echo "new tests rc=$rc"
tail -40 gpurun_out/${T}_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/${T}_pytest.log
tail -30 gpurun_out/${T}_pytest.log
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
        r=j["roofline"]
        print(f, round(j["value"],1), "Mpaths/s", round(j["ms_per_step"],3), "ms", j["config"].get("collective"), {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.get("lsm",{}).items() if k in ("sweep_ms_per_pass","sweep_launches_per_pass","solve_ms_per_pass")}, "frac", round(r["frac"],3), "ceiling", r.get("board_write_ceiling_GBs"), "frac_of_ceiling", r.get("frac_of_board_ceiling"), "clock", r.get("shader_clock_GHz"))
    except Exception as e:
        print(f, "failed", e)
j=json.load(open("gpurun_out/${T}_bench.json"))
for row in j.get("extra",{}).get("configs",[]):
    print(row["config"][:60], {k:(round(v,3) if isinstance(v,float) else v) for k,v in row.items() if k in ("ms_per_pass","ms_per_call","kernel_ms_per_call","rows_per_s","rows_per_s_of_device_time","hbm_frac","chunks_per_call")})
PY
