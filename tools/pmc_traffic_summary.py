#!/usr/bin/env python3
"""profiles/pmc_traffic.json (read by bench.py for roofline.traffic) and profiles/r01_c2_pmc_counters.csv from the two
passes of tools/pmc_traffic.sh.  Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB;
FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 and is doubled."""
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows, vals = [], {"WRITE_SIZE": [], "FETCH_SIZE": []}
for tag in "wr":
    f = max(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_traffic_{tag}/*/*counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if "k_gbm_paths" in r["Kernel_Name"]:
            rows.append(r)
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name",
        "Counter_Value", "Start_Timestamp", "End_Timestamp"]
with open(os.path.join(ROOT, "profiles/r01_c2_pmc_counters.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=keep)
    w.writeheader()
    for r in rows:
        w.writerow({k: r[k] for k in keep})
wr = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024.0
rd = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024.0 * 2.0
out = {"paths": 10_000_000, "time_steps": 252, "hbm_bytes_per_launch": wr + rd, "write_bytes": wr, "fetch_bytes_corrected_x2": rd,
       "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/pmc_traffic.sh), profiles/r01_c2_pmc_counters.csv; "
                 "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section",
       "kernel": rows[0]["Kernel_Name"], "algorithmic_bytes": 8 * 253 * 10_000_000}
json.dump(out, open(os.path.join(ROOT, "profiles/pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
