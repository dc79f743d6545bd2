#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --paths 2000000 --steps 5 --warmup 2 --backend gloo > gpurun_out/r2i_c2_auto.json 2> gpurun_out/r2i_c2_auto.err; echo "c2 auto rc=$?"
tail -c 400 gpurun_out/r2i_c2_auto.json; echo; grep -i "bench:" gpurun_out/r2i_c2_auto.err | head -5
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29619 bench.py --gpus 2 --paths 2000000 --steps 5 --warmup 2 --backend gloo --collective shm > gpurun_out/r2i_c2_shm.json 2> gpurun_out/r2i_c2_shm.err; echo "c2 shm rc=$?"
python3 -c "
import json
for f in ('r2i_c2_auto','r2i_c2_shm'):
    j=json.loads(open('gpurun_out/'+f+'.json').read().strip().split('\n')[-1]); print(f, j['value'], j['ms_per_step'], j['config']['collective'], j['parity']['price'], j['parity']['abs_err_over_std_err'])"
