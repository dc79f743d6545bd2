#!/usr/bin/env python3
"""Condense the per-dispatch CSVs of tools/pmc_valu.sh (gpurun_out/pmc_valu_{a,b,c}) into
profiles/r01_c2_valu_counters.json: means per launch of the headline kernel and the derived VALU-busy figures."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, dur, regs = {}, [], None
for tag in "abc":
    f = max(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_valu_{tag}/*/*counter_collection.csv")), key=os.path.getmtime)
    acc, seen = collections.defaultdict(list), set()
    for r in csv.DictReader(open(f)):
        if "k_gbm_paths" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        regs = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"])}
        name = r["Kernel_Name"]
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
wave_steps = 10_000_000 * 252 / 64
d = sum(dur) / len(dur)
cyc = out["GRBM_GUI_ACTIVE"] / 8
fp64 = sum(out[k] for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
summary = {
    "command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (three separate passes, tools/pmc_valu.sh)",
    "kernel": name + ", 10M paths x 252 steps per launch", "registers": regs,
    "counters_mean_per_launch": out, "kernel_ms_in_profiled_runs": d,
    "derived": {
        "shader_clock_GHz": cyc / (d * 1e-3) / 1e9,
        "valu_instructions_per_wave_step": out["SQ_INSTS_VALU"] / wave_steps,
        "fp64_instructions_per_wave_step": fp64 / wave_steps,
        "valu_busy_fraction": out["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
        "cycles_per_valu_instruction": out["SQ_ACTIVE_INST_VALU"] * 4 / out["SQ_INSTS_VALU"],
        "note": "SQ_ACTIVE_INST_* count quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)"}}
json.dump(summary, open(os.path.join(ROOT, "profiles/r01_c2_valu_counters.json"), "w"), indent=1)
print(json.dumps(summary["derived"], indent=1), regs, "durations", [round(x, 3) for x in dur])
