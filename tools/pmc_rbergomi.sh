#!/bin/bash
# Run on the GPU box: VALU-side PMC counters of the rBergomi generator on C4 (separate --pmc passes).
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
CMD="python3 tools/bench_configs.py --configs c4 --reps 2"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_rb_a -- $CMD > gpurun_out/pmc_rb_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS --output-format csv -d $ROOT/gpurun_out/pmc_rb_b -- $CMD > gpurun_out/pmc_rb_b.log 2>&1
echo "rc=$?"
