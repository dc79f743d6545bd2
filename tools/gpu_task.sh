#!/bin/bash
# The GPU-box sessions of this repo, one script (it replaces the one-off tools/gpu_r4?.sh of round 4):
#     gpurun --timeout N -- 'tools/gpu_task.sh <task> [tag] [args...]'
# Every task writes under gpurun_out/<tag>_*; steps are joined so that a step killed at its time limit ends the session.
#   suite      the whole GPU test suite (one process), slowest tests listed
#   tests K    pytest -m gpu -k "K"
#   bench      the driver's default line (bench.py, N = 1) -> <tag>_bench.json
#   limiter    PMC passes that attribute the headline kernel's non-issue cycles (tools/pmc_passes.py, C2 command).  No TA_*
#              counters: a pass with TA_TA_BUSY_sum / TA_*_STALLED_BY_TC_CYCLES_sum never returned on this pool (r5a: killed
#              by the pass time-out after 240 s; the SQ passes before it take 2 s each)
#   rblimiter  the same for the rBergomi generator (tools/bench_configs.py c5,c4); also lists the device's counters
#   stats      rocprofv3 --kernel-trace --stats of the default bench -> <tag>_stats/
#   ab A B [configs] [reps]   tools/ab_libs.sh between two built libraries
#   tool / toolstats   a dev tool under tools/ (bench_branching.py, bench_rows.py, ...) plainly / under rocprofv3 --stats
#   pmc        counter passes (SQ / LDS / TCC / TCP groups) of named kernels under a dev tool
#   perdate    tools/bench_perdate.py, product against the study build with MCG_LSM_DATE_ADAPTIVE=0
#   two-rank   bench.py with two ranks sharing the card (gloo), inline / child / off C5 rows
set -o pipefail
export TMPDIR=/tmp
task=${1:?task}; T=${2:-r5}; shift; shift || true
O=gpurun_out
mkdir -p $O
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its limit (rc=$rc): stopping"; exit $rc; fi; return $rc; }
C2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
case $task in
suite)
  timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > $O/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -14 $O/${T}_pytest.log; exit $rc ;;
tests)
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$1" > $O/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -14 $O/${T}_pytest.log; exit $rc ;;
bench)
  timeout -k 10 900 python3 bench.py "$@" > $O/${T}_bench.json 2> $O/${T}_bench.err; rc=$?; echo "bench rc=$rc"; tail -3 $O/${T}_bench.err; head -c 1500 $O/${T}_bench.json; exit $rc ;;
limiter)
  timeout -k 10 1100 python3 tools/pmc_passes.py --tag ${T}_c2lim --kernels "k_gbm_paths<true,k_probe_write" \
    --group "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
    --group "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE" \
    --group "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
    --group "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE" \
    --group "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
    --group "TCP_PENDING_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE" \
    --group "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
    --group "TCC_BUSY_sum TCC_CYCLE_sum TCC_WRITE_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
    --group "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE" \
    --group "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum GRBM_GUI_ACTIVE" \
    --group "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_STREAMING_REQ_sum GRBM_GUI_ACTIVE" \
    -- $C2 2>&1 | tee $O/${T}_c2lim.log; guard ;;
rblimiter) # round 6: the same attribution for the rBergomi generator (C5 shard's <4,2,false> at 8M x 252, C4's <5,2,true> at 4M x 512)
  rocprofv3 -L > $O/${T}_counters_avail.txt 2>&1
  timeout -k 10 1100 python3 tools/pmc_passes.py --tag ${T}_rblim --kernels "k_rbergomi_fft<4,k_rbergomi_fft<5" \
    --group "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
    --group "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE" \
    --group "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
    --group "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SENDMSG GRBM_GUI_ACTIVE" \
    --group "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MISC GRBM_GUI_ACTIVE" \
    --group "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
    --group "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 GRBM_GUI_ACTIVE" \
    --group "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
    -- python3 tools/bench_configs.py --configs c5,c4 --reps 4 2>&1 | tee $O/${T}_rblim.log; guard ;;
stats)
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 bench.py "$@" > $O/${T}_stats.log 2>&1; rc=$?; echo "stats rc=$rc"; tail -2 $O/${T}_stats.log; exit $rc ;;
ab)
  timeout -k 10 900 tools/ab_libs.sh ${3:-c2} ${4:-4} $1 $2 2>&1 | tee $O/${T}_ab.log; guard ;;
two-rank)
  for mode in inline child off; do
    timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --paths 2000000 --c5-paths 1000000 --c5-rows $mode "$@" \
      > $O/${T}_two_rank_$mode.json 2> $O/${T}_two_rank_$mode.err; rc=$?; echo "two-rank $mode rc=$rc"; guard || exit $rc
    head -c 600 $O/${T}_two_rank_$mode.json; echo
  done ;;
tool)      # a dev tool under tools/ plainly, its output kept: gpu_task.sh tool <tag> bench_branching.py [args]
  timeout -k 10 600 python3 tools/"$@" 2>&1 | tee $O/${T}_tool.log; guard ;;
toolstats) # ... and under rocprofv3 --kernel-trace --stats (the per-kernel table is printed; mind that it averages over ALL calls of
           #     a kernel -- bench_rows.py's first call prices 64 rows: read full-call times from the kernel trace)
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_toolstats -- python3 tools/"$@" > $O/${T}_toolstats.log 2>&1; guard || exit $?
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/${T}_toolstats/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s calls %6s  avg %10.1f us  total %9.3f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  ;;
pmc)       # counters of a gather / latency-bound kernel: gpu_task.sh pmc <tag> <kernel substrings> <tool.py> [args]
  K=$1; shift
  timeout -k 10 900 python3 tools/pmc_passes.py --tag ${T}_pmc --kernels "$K" \
    --group "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
    --group "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
    --group "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE" \
    --group "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
    --group "TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
    -- python3 tools/"$@" 2>&1 | tee $O/${T}_pmc.log; guard ;;
perdate)   # the per-date route's batches at order 5, adaptive (product) against one launch per remaining date (study build)
  timeout -k 10 300 python3 tools/bench_perdate.py 2>&1 | tee $O/${T}_perdate.log; guard || exit $?
  export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_study.so MCG_LSM_DATE_ADAPTIVE=0
  timeout -k 10 300 python3 tools/bench_perdate.py 2>&1 | sed 's/^/one-per-date batches: /' | tee -a $O/${T}_perdate.log; guard ;;
*) echo "unknown task $task"; exit 2 ;;
esac
