#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/p2_rows -- python3 tools/bench_rows.py > gpurun_out/p2_rows.log 2>&1; echo rc=$?
f=$(ls -t gpurun_out/p2_rows/*/*kernel_stats.csv | head -1); python3 - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "avg_us %10.1f" % (float(r["AverageNs"])/1e3), r["Percentage"])
P
