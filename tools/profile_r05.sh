#!/bin/bash
# Run on the GPU box (gpurun): what profiles/r05_* holds beyond the limiter passes (tools/gpu_task.sh limiter).  Output goes to
# gpurun_out/p5_*; tools/profile_r05_summary.py condenses it into profiles/r05_*.  Kernel statistics and PMC counters are
# collected in separate rocprofv3 runs (--kernel-trace --stats only / --pmc only), as the pool requires; the program after `--`
# is python3 itself (environment variables are exported in this shell, never through `env`).  No TA_* counters (they hang).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
BENCH="python3 bench.py --steps 10 --warmup 3"
C2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
C5="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline"
C5D="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective rccl"
C5I="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective ipc"
C5S="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective shm"
st() { timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1; rc=$?; [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$1 killed at its limit"; exit $rc; }; return $rc; }
pm() { d=$1; shift; c=$1; shift; timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; rc=$?; [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$d killed at its limit"; exit $rc; }; return $rc; }
VA="SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
VB="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS"
st p5_stats_bench "$BENCH" && echo "stats bench done" &&
st p5_stats_c5 "$C5" &&
pm p5_pmc_c2_w "WRITE_SIZE" $C2 && pm p5_pmc_c2_r "FETCH_SIZE" $C2 && pm p5_pmc_c2_va "$VA" $C2 && echo "c2 pmc done" &&
pm p5_pmc_c5_w "WRITE_SIZE" $C5 && pm p5_pmc_c5_r "FETCH_SIZE" $C5 &&
pm p5_pmc_c5_va "$VA" $C5 && pm p5_pmc_c5_vb "$VB" $C5 && echo "c5 pmc done" &&
pm p5_pmc_c4_va "$VA" python3 tools/bench_configs.py --configs c4 --reps 2 &&
pm p5_pmc_c4_vb "$VB" python3 tools/bench_configs.py --configs c4 --reps 2 && echo "c4 pmc done"
echo "profile rc=$?"
# the bench lines themselves (C2 default, C5 alone and through every collective at world size 1)
timeout -k 10 600 $BENCH > $O/p5_bench_n1.json 2> $O/p5_bench_n1.err; echo "bench rc=$?"
timeout -k 10 300 $C5 > $O/p5_bench_c5_n1.json 2>> $O/p5_bench_n1.err
export MCG_FORCE_DIST=1
timeout -k 10 300 $C5D > $O/p5_bench_c5_rccl1.json 2>> $O/p5_bench_n1.err
timeout -k 10 300 $C5I > $O/p5_bench_c5_ipc1.json 2>> $O/p5_bench_n1.err
timeout -k 10 300 $C5S > $O/p5_bench_c5_shm1.json 2>> $O/p5_bench_n1.err
st p5_stats_c5_rccl "$C5D"
unset MCG_FORCE_DIST
python3 tools/profile_r05_summary.py > $O/p5_summary.log 2>&1; echo "summary rc=$?"; tail -30 $O/p5_summary.log
