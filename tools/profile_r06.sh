#!/bin/bash
# Run on the GPU box (gpurun): what profiles/r06_* holds beyond the rBergomi limiter session (tools/gpu_task.sh rblimiter).
# Output goes to gpurun_out/p6_*; tools/profile_r06_summary.py condenses it into profiles/r06_*.  Kernel statistics and PMC counters
# are collected in separate rocprofv3 runs (--kernel-trace --stats only / --pmc only), as the pool requires; the program after `--`
# is python3 itself.  No TA_* counters (they hang on this pool).   tools/profile_r06.sh [part]   part = a | b | all
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
part=${1:-all}
BENCH="python3 bench.py --steps 10 --warmup 3"
C2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
C5="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline"
st() { timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1; rc=$?; [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$1 killed at its limit"; exit $rc; }; return $rc; }
pm() { d=$1; shift; c=$1; shift; timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; rc=$?; [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$d killed at its limit"; exit $rc; }; return $rc; }
VA="SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
VB="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS"
if [ $part = a ] || [ $part = all ]; then
  # the headline kernel's books on THIS board: limiter passes (k_gbm_paths and the store-only probe in the same passes), traffic
  tools/gpu_task.sh limiter p6 || exit $?
  pm p6_pmc_c2_w "WRITE_SIZE" $C2 && pm p6_pmc_c2_r "FETCH_SIZE" $C2 && pm p6_pmc_c2_va "$VA" $C2 && echo "c2 pmc done"
fi
if [ $part = b ] || [ $part = all ]; then
  st p6_stats_bench "$BENCH" && echo "stats bench done" &&
  st p6_stats_c5 "$C5" &&
  pm p6_pmc_c5_w "WRITE_SIZE" $C5 && pm p6_pmc_c5_r "FETCH_SIZE" $C5 &&
  pm p6_pmc_c5_va "$VA" $C5 && pm p6_pmc_c5_vb "$VB" $C5 && echo "c5 pmc done" &&
  pm p6_pmc_c4_va "$VA" python3 tools/bench_configs.py --configs c4 --reps 2 &&
  pm p6_pmc_c4_vb "$VB" python3 tools/bench_configs.py --configs c4 --reps 2 && echo "c4 pmc done"
  echo "profile rc=$?"
  timeout -k 10 600 $BENCH > $O/p6_bench_n1.json 2> $O/p6_bench_n1.err; echo "bench rc=$?"
  timeout -k 10 300 $C5 > $O/p6_bench_c5_n1.json 2>> $O/p6_bench_n1.err
fi
python3 tools/profile_r06_summary.py > $O/p6_summary.log 2>&1; echo "summary rc=$?"; tail -30 $O/p6_summary.log
