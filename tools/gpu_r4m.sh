#!/bin/bash
# per-kernel times of the batched rows, two builds
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in serial new; do
  if [ $v = new ]; then unset MCG_LIB; else export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/r4m_rows_$v -- python3 tools/bench_rows.py --reps 5 > gpurun_out/r4m_rows_$v.log 2>&1
  echo "== $v"; grep -h "k_batch" gpurun_out/r4m_rows_$v/*/*kernel_stats.csv | cut -d, -f1,2,4,6,7 | cut -c1-120
done
