#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_r03.sh (gpurun_out/p3_*) into profiles/r03_*:
  r03_{bench,c5,c5_rccl,c5_ipc}_kernel_stats.csv   the --kernel-trace --stats summaries, verbatim (bench = the driver's
                                     default command: C2, then C3 / C4 / C5 shard and the SURVEY 8(f) rows)
  pmc_traffic.json / r03_c2_pmc_traffic.json   HBM bytes per launch of k_gbm_paths (bench.py's roofline.traffic)
  r03_c5_pmc_traffic.json            HBM bytes per launch of the C5 kernels (generator, one-launch LSM sweep, per-date kernel)
  r03_c5gen_valu_counters.json, r03_c4_valu_counters.json   VALU-side counters of the rBergomi generator
Units and corrections as MI355X_MICROARCH.md prescribes: WRITE_SIZE / FETCH_SIZE are in KiB, FETCH_SIZE reports half
of the bytes of wide coalesced reads on gfx950 and is doubled; SQ_ACTIVE_INST_* count quad-cycles summed over the 1024
SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    f = glob.glob(os.path.join(G, pattern))
    return max(f, key=os.path.getmtime) if f else None


def counters(tag, kernel_substr):
    """mean counter value per launch and mean duration [ms] of kernels whose name contains kernel_substr"""
    f = newest(f"{tag}/*/*counter_collection.csv")
    acc, dur, seen, name, regs = collections.defaultdict(list), [], set(), None, None
    if not f:
        return {}, None, None, None
    for r in csv.DictReader(open(f)):
        if kernel_substr not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        name = r["Kernel_Name"]
        regs = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"])}
    return {k: sum(v) / len(v) for k, v in acc.items()}, (sum(dur) / len(dur) if dur else None), name, regs


for tag, out in (("p3_stats_bench", "r03_bench_kernel_stats.csv"), ("p3_stats_c5", "r03_c5_kernel_stats.csv"),
                 ("p3_stats_c5_rccl", "r03_c5_rccl_kernel_stats.csv"), ("p3_stats_c5_ipc", "r03_c5_ipc_kernel_stats.csv")):
    f = newest(f"{tag}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(P, out))
        print("copied", out)


def traffic(wtag, rtag, kernel, alg_bytes, extra):
    w, dw, name, _ = counters(wtag, kernel)
    r, dr, _, _ = counters(rtag, kernel)
    if "WRITE_SIZE" not in w or "FETCH_SIZE" not in r:
        return None
    wr, rd = w["WRITE_SIZE"] * 1024.0, r["FETCH_SIZE"] * 1024.0 * 2.0
    d = dict(extra)
    d.update({"kernel": name, "hbm_bytes_per_launch": wr + rd, "write_bytes": wr, "fetch_bytes_corrected_x2": rd,
              "algorithmic_bytes": alg_bytes, "traffic_over_algorithmic": (wr + rd) / alg_bytes,
              "kernel_ms_in_profiled_runs": [dw, dr],
              "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/profile_r03.sh: {wtag}, {rtag}); "
                        "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"})
    return d


t = traffic("p3_pmc_c2_w", "p3_pmc_c2_r", "k_gbm_paths", 8 * 253 * 10_000_000, {"paths": 10_000_000, "time_steps": 252})
if t:
    json.dump(t, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join(P, "r03_c2_pmc_traffic.json"), "w"), indent=1)
    print("C2 traffic", t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
c5 = {}
for key, kernel, alg in (("generator", "k_rbergomi_fft", 8 * 253 * 8_000_000), ("lsm_one_launch", "k_lsm_big", 16 * 252 * 8_000_000)):
    t = traffic("p3_pmc_c5_w", "p3_pmc_c5_r", kernel, alg, {"paths": 8_000_000, "time_steps": 252})
    if t:
        c5[key] = t
        print("C5", key, t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
t = traffic("p3_pmc_c5d_w", "p3_pmc_c5d_r", "k_lsm_date", 32 * 8_000_000, {"paths": 8_000_000, "time_steps": 252})
if t:
    t["note_per_date"] = ("k_lsm_date: mean over ALL its launches of a pass, 252 working ones (32 B per path: S_j, S_{j-1}, V read, V "
                          "written) and 13 that return at once (terminal payoff launch: 24 B; spare launches: nothing)")
    c5["lsm_per_date_launch"] = t
    print("C5 per-date", t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
if c5:
    c5["note"] = ("lsm_one_launch: algorithmic_bytes is what the kernel's design reads, 16 B per path and date (each row "
                  "twice, V in registers); SURVEY 8(d)'s two-pass figure for the same sweep is 40 B per path and date")
    json.dump(c5, open(os.path.join(P, "r03_c5_pmc_traffic.json"), "w"), indent=1)


def valu(atag, btag, kernel, paths, steps, out, cmd):
    a, da, name, regs = counters(atag, kernel)
    b, db, _, _ = counters(btag, kernel)
    if "SQ_INSTS_VALU" not in a:
        return
    c = dict(a)
    c.update(b)
    units = paths * steps / 64.0
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    fp64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
    s = {"command": cmd, "kernel": f"{name}, {paths} paths x {steps} steps per launch", "paths_per_launch": paths,
         "registers": regs, "counters_mean_per_launch": c, "kernel_ms_in_profiled_runs": da,
         "derived": {"shader_clock_GHz": cyc / (da * 1e-3) / 1e9,
                     "valu_instructions_per_64_path_steps": c["SQ_INSTS_VALU"] / units,
                     "fp64_instructions_per_64_path_steps": fp64 / units,
                     "lds_instructions_per_64_path_steps": c.get("SQ_INSTS_LDS", 0.0) / units,
                     "valu_busy_fraction": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                     "cycles_per_valu_instruction": c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"]}}
    json.dump(s, open(os.path.join(P, out), "w"), indent=1)
    print(out, json.dumps(s["derived"]))


valu("p3_pmc_c5_va", "p3_pmc_c5_vb", "k_rbergomi_fft", 8_000_000, 252, "r03_c5gen_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline (two passes, tools/profile_r03.sh)")
valu("p3_pmc_c4_va", "p3_pmc_c4_vb", "k_rbergomi_fft", 4_000_000, 512, "r03_c4_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 tools/bench_configs.py --configs c4 --reps 2 (two passes, tools/profile_r03.sh)")
