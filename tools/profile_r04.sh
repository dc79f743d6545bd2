#!/bin/bash
# Run on the GPU box (gpurun): everything profiles/ holds for round 4.  Output goes to gpurun_out/p4_*;
# tools/profile_r04_summary.py condenses it into profiles/r04_*.  Kernel statistics and PMC counters are collected
# in separate rocprofv3 runs (--kernel-trace --stats only / --pmc only), as the pool requires; the program after `--` is
# python3 itself (environment variables are exported in this shell, never through `env`).
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out
mkdir -p $O
BENCH="python3 bench.py --steps 10 --warmup 3"                                   # the driver's default line: C2 + C3/C4/C5 + the 8(f) rows
C2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
C5="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline"         # one-launch sweep, no collective
C5D="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective rccl"   # (MCG_FORCE_DIST=1) per-date route
C5I="python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --collective ipc"    # (MCG_FORCE_DIST=1) peer-memory mailbox
st() { rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- $2 > $O/$1.log 2>&1; }
pm() { d=$1; shift; c=$1; shift; rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
VA="SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
VB="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS"
st p4_stats_bench "$BENCH" && echo "stats bench done" &&
st p4_stats_c5 "$C5" &&
pm p4_pmc_c2_w "WRITE_SIZE" $C2 && pm p4_pmc_c2_r "FETCH_SIZE" $C2 && pm p4_pmc_c2_va "$VA" $C2 && echo "c2 pmc done" &&
pm p4_pmc_c5_w "WRITE_SIZE" $C5 && pm p4_pmc_c5_r "FETCH_SIZE" $C5 &&
pm p4_pmc_c5_va "$VA" $C5 && pm p4_pmc_c5_vb "$VB" $C5 && echo "c5 pmc done" &&
pm p4_pmc_c4_va "$VA" python3 tools/bench_configs.py --configs c4 --reps 2 &&
pm p4_pmc_c4_vb "$VB" python3 tools/bench_configs.py --configs c4 --reps 2 && echo "c4 pmc done"
rc=$?
export MCG_FORCE_DIST=1
[ $rc -eq 0 ] && st p4_stats_c5_rccl "$C5D" && st p4_stats_c5_ipc "$C5I" &&
pm p4_pmc_c5d_w "WRITE_SIZE" $C5D && pm p4_pmc_c5d_r "FETCH_SIZE" $C5D && echo "c5 per-date done"
echo "profile rc=$?"
unset MCG_FORCE_DIST
for c in TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum; do pm p4_pmc_branch_$c $c python3 tools/bench_branching.py; done
st p4_branch_stats "python3 tools/bench_branching.py"
python3 tools/profile_r04_summary.py > $O/p4_summary.log 2>&1; echo "summary rc=$?"; tail -40 $O/p4_summary.log
