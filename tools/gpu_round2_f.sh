#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for coll in shm torch; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --config c5 --paths 1000000 --steps 5 --warmup 2 --backend gloo --collective $coll > gpurun_out/r2f_c5_$coll.json 2> gpurun_out/r2f_c5_$coll.err; echo "c5 $coll rc=$?"
done
python - <<'P'
import json
for c in ("shm","torch"):
    try:
        j=json.loads(open(f"gpurun_out/r2f_c5_{c}.json").read().strip().split("\n")[-1])
        print(c, "ms/step", round(j["ms_per_step"],3), "collective", j["config"]["collective"], "price", j["parity"]["price"], "lsm", {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"]["lsm"].items() if k in ("sweep_ms_per_pass","sweep_launches_per_pass","solve_ms_per_pass","shape")})
    except Exception as e:
        print(c, "ERR", e); print(open(f"gpurun_out/r2f_c5_{c}.err").read()[-1500:])
P
timeout -k 10 200 env MCG_FORCE_DIST=1 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective shm > gpurun_out/r2f_c5_shm1.json 2> gpurun_out/r2f_c5_shm1.err; echo "c5 shm world1 rc=$?"; python -c "
import json; j=json.loads(open('gpurun_out/r2f_c5_shm1.json').read().strip().split('\n')[-1]); print('world1 shm ms', j['ms_per_step'], j['config']['collective'], j['roofline']['lsm']['sweep_launches_per_pass'], j['parity']['price'])"
