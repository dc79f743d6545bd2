#!/bin/bash
# GPU box: the multi-rank tests, then C5 at world size 1 through every collective and without one (ms per pass, sweep ms)
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "multirank" 2>&1 | tail -3
show='import sys,json; j=json.loads(sys.stdin.read()); print(j["config"]["collective"], round(j["ms_per_step"],3), "ms per pass; sweep", round(j["roofline"]["lsm"]["sweep_ms_per_pass"],3), "generator", round(j["roofline"]["kernel_avg_ms"],3))'
for c in ipc shm rccl; do
  MCG_FORCE_DIST=1 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective $c 2>/dev/null | python -c "$show"
done
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$show"
