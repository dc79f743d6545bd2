#!/usr/bin/env python3
"""profiles/r04_branching_counters_before.json from the separate `rocprofv3 --pmc` passes of tools/bench_branching.py
(tools/gpu_r4a.sh): per shape, the cache counters of k_branch_bounds per launch, the request traffic they imply and what
that is of the guide's random-row rates (MI355X_MICROARCH.md, 'Indexed rows')."""
import csv, glob, json, os, sys
from collections import defaultdict

T = sys.argv[1] if len(sys.argv) > 1 else "r4a"
# tools/bench_branching.py: grid threads of a launch (one thread per 4 paths, at most 2048 workgroups) -> (paths of the
# matrix, dates, paths a launch covers): the 4M-path matrix takes two launches of 2 097 152 paths each
shapes = {250_112: (1_000_000, 50, 1_000_000), 524_288: (4_000_000, 50, 2_097_152), 62_720: (250_000, 252, 250_000)}
out = {"command": "rocprofv3 --pmc <one counter per pass> -- python3 tools/bench_branching.py", "kernel": "k_branch_bounds",
       "note": "4 launches per shape and pass (1 untimed + 3 timed); values are means per launch", "shapes": {}}
acc = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(f"gpurun_out/{T}_pmc_branch_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "k_branch_bounds" not in row["Kernel_Name"]:
                continue
            acc[int(row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(f"gpurun_out/{T}_branch_stats/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "k_branch_bounds" in row["Kernel_Name"]:
            dur[int(row["Grid_Size_X"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
for grid, cs in sorted(acc.items()):
    paths, dates, covered = shapes.get(grid, (None, None, None))
    c = {k: sum(v) / len(v) for k, v in cs.items()}
    ms = sorted(dur.get(grid, [0.0]))
    ms = ms[len(ms) // 2]
    row = {"grid_threads": grid, "paths": paths, "dates": dates, "kernel_ms_median": ms, "counters_mean_per_launch": c}
    if paths:
        gathers = 10.0 * min(covered, paths / max(1, round(paths / covered))) * (dates - 1)   # per LAUNCH: branches x its paths x dates with a later column
        row["gathers"] = gathers
        row["F_row_MB"] = paths * 8 / 1e6
        if ms > 0:
            row["G_gathers_per_s"] = gathers / (ms * 1e-3) / 1e9
        hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            row["L2_hit_rate"] = hit / (hit + miss)
            row["L2_lookups_per_gather"] = (hit + miss) / gathers
        if c.get("TCC_EA0_RDREQ_sum") and ms > 0:
            row["beyond_L2_requests_per_gather"] = c["TCC_EA0_RDREQ_sum"] / gathers
            row["beyond_L2_TBps_at_64B_per_request"] = c["TCC_EA0_RDREQ_sum"] * 64 / (ms * 1e-3) / 1e12
        if c.get("TCP_TCC_READ_REQ_sum") and ms > 0:
            row["L1_to_L2_read_requests_per_gather"] = c["TCP_TCC_READ_REQ_sum"] / gathers
            row["L2_request_TBps_at_64B_per_request"] = c["TCP_TCC_READ_REQ_sum"] * 64 / (ms * 1e-3) / 1e12
    out["shapes"][f"{paths}x{dates}" if paths else str(grid)] = row
json.dump(out, open("profiles/r04_branching_counters_before.json", "w"), indent=1)
print(json.dumps(out, indent=1))
