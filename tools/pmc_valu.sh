#!/bin/bash
# Run on the GPU box: VALU-side PMC counters of the headline kernel (separate passes, --pmc only).
# Copies of the per-dispatch CSVs go to gpurun_out/pmc_valu_*; tools/pmc_valu_summary.py condenses them.
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_valu_a -- $CMD > gpurun_out/pmc_valu_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $ROOT/gpurun_out/pmc_valu_b -- $CMD > gpurun_out/pmc_valu_b.log 2>&1 &&
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU2 --output-format csv -d $ROOT/gpurun_out/pmc_valu_c -- $CMD > gpurun_out/pmc_valu_c.log 2>&1
echo "rc=$?"
find gpurun_out -name "*counter_collection.csv" | head
