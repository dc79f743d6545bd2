#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $PWD/gpurun_out/r4s_rows_va -- python3 tools/bench_rows.py --reps 2 > gpurun_out/r4s_rows_va.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r4s_rows_va/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_batch" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 1000000: continue
    k=r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    m={a:sum(b)/len(b) for a,b in c.items()}
    cyc=m["GRBM_GUI_ACTIVE"]/8
    print(k, "cycles", round(cyc/1e6,2),"M  valu_insts/wave", round(m["SQ_INSTS_VALU"]/m["SQ_WAVES"],1), "valu busy", round(m["SQ_ACTIVE_INST_VALU"]*4/1024/cyc,3), "lds/wave", round(m["SQ_INSTS_LDS"]/m["SQ_WAVES"],1), "waves", int(m["SQ_WAVES"]))
PY
