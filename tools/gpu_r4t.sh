#!/bin/bash
# how much does k_batch_paths live off its occupancy?  (2 workgroups per CU as built; extra dynamic LDS leaves 1)
export TMPDIR=/tmp
export MCG_LIB=$PWD/montecarlooptionspricer_amd/lib/libmcgpu_study.so
for kb in 0 40; do
  export MCG_BATCH_PATHS_EXTRA_LDS_KB=$kb
  rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/r4t_$kb -- python3 tools/bench_rows.py --reps 3 > gpurun_out/r4t_$kb.log 2>&1
  echo "extra LDS $kb KB: $(grep -h 'k_batch_paths' gpurun_out/r4t_$kb/*/*kernel_stats.csv | cut -d, -f3,6,7)"
done
