#!/bin/bash
# GPU box (gpurun): the GPU tests, the default bench line, the C5 lines (no collective, RCCL and shared memory at world size 1)
# and a two-rank rehearsal of C5 on the one GPU.  Output in gpurun_out/r2a_*.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a_pytest.log
tail -5 gpurun_out/r2a_pytest.log
timeout -k 10 400 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; echo "bench rc=$?"
timeout -k 10 300 env MCG_FORCE_DIST=1 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective rccl > gpurun_out/r2a_c5_rccl1.json 2> gpurun_out/r2a_c5_rccl1.err; echo "c5 rccl rc=$?"
timeout -k 10 300 env MCG_FORCE_DIST=1 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective shm > gpurun_out/r2a_c5_shm1.json 2> gpurun_out/r2a_c5_shm1.err; echo "c5 shm rc=$?"
timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a_c5_n1.json 2> gpurun_out/r2a_c5_n1.err; echo "c5 n1 rc=$?"
# (two ranks on ONE GPU: both persistent grids must be resident together -- 16 paths per thread keeps each at 244 workgroups)
export MCG_LSM_COOP_MIN_PPT=16
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --config c5 --paths 1000000 --steps 3 --warmup 1 --backend gloo --collective shm > gpurun_out/r2a_c5_gloo2.json 2> gpurun_out/r2a_c5_gloo2.err; echo "c5 gloo2 rc=$?"
tail -c 600 gpurun_out/r2a_c5_gloo2.err
