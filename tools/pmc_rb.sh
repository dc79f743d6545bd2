#!/bin/bash
# PMC passes over the C5-shard generator (dev tool): tools/pmc_rb.sh <outdir-prefix>
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-pmc_rb}
CMD="python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d ${O}_a -- $CMD > ${O}_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d ${O}_b -- $CMD > ${O}_b.log 2>&1
echo rc=$?
python3 - <<P
import csv,glob,collections
for tag in "ab":
    for f in glob.glob("${O}_%s/**/*counter_collection.csv" % tag, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "rbergomi" in k or "lsm_big" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,d in acc.items():
            print(k)
            for c,v in d.items(): print("   %-24s %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
P
