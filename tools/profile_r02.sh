#!/bin/bash
# Run on the GPU box (gpurun): everything profiles/ holds for round 2.  Output goes to gpurun_out/p2_*;
# tools/profile_r02_summary.py condenses it into profiles/r02_*.  Kernel statistics and PMC counters are collected
# in separate rocprofv3 runs (--kernel-trace --stats only / --pmc only), as the pool requires.
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out
mkdir -p $O
C2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
C5="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline"
CFG="python3 tools/bench_configs.py --configs c3,c4 --reps 3"
st() { rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1; }
pm() { d=$1; shift; c=$1; shift; rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
st p2_stats_c2 "$C2" && st p2_stats_c5 "$C5" && st p2_stats_c34 "$CFG" &&
pm p2_pmc_c2_w "WRITE_SIZE" $C2 && pm p2_pmc_c2_r "FETCH_SIZE" $C2 &&
pm p2_pmc_c5_w "WRITE_SIZE" $C5 && pm p2_pmc_c5_r "FETCH_SIZE" $C5 &&
pm p2_pmc_c5_va "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" $C5 &&
pm p2_pmc_c5_vb "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS" $C5 &&
pm p2_pmc_c4_va "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" python3 tools/bench_configs.py --configs c4 --reps 2 &&
pm p2_pmc_c4_vb "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS" python3 tools/bench_configs.py --configs c4 --reps 2
echo "profile rc=$?"
python3 tools/profile_r02_summary.py > $O/p2_summary.log 2>&1; echo "summary rc=$?"; tail -40 $O/p2_summary.log
