#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_r04.sh (gpurun_out/p4_*) into profiles/r04_*:
  r04_{bench,c5,c5_rccl,c5_ipc}_kernel_stats.csv   the --kernel-trace --stats summaries, verbatim (bench = the driver's
                                     default command: C2, then C3 / C4 / C5 shard and the SURVEY 8(f) rows)
  pmc_traffic.json / r04_c2_pmc_traffic.json   HBM bytes per launch of k_gbm_paths (bench.py's roofline.traffic)
  r04_c5_pmc_traffic.json            HBM bytes per launch of the C5 kernels (generator, one-launch LSM sweep, per-date kernel)
  r04_c5gen_valu_counters.json, r04_c4_valu_counters.json   VALU-side counters of the rBergomi generator
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


for tag, out in (("p4_stats_bench", "r04_bench_kernel_stats.csv"), ("p4_stats_c5", "r04_c5_kernel_stats.csv"),
                 ("p4_stats_c5_rccl", "r04_c5_rccl_kernel_stats.csv"), ("p4_stats_c5_ipc", "r04_c5_ipc_kernel_stats.csv")):
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
              "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/profile_r04.sh: {wtag}, {rtag}); "
                        "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"})
    return d


t = traffic("p4_pmc_c2_w", "p4_pmc_c2_r", "k_gbm_paths", 8 * 253 * 10_000_000, {"paths": 10_000_000, "time_steps": 252})
if t:
    json.dump(t, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join(P, "r04_c2_pmc_traffic.json"), "w"), indent=1)
    print("C2 traffic", t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
c5 = {}
for key, kernel, alg in (("generator", "k_rbergomi_fft", 8 * 253 * 8_000_000), ("lsm_one_launch", "k_lsm_big", 16 * 252 * 8_000_000)):
    t = traffic("p4_pmc_c5_w", "p4_pmc_c5_r", kernel, alg, {"paths": 8_000_000, "time_steps": 252})
    if t:
        c5[key] = t
        print("C5", key, t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
t = traffic("p4_pmc_c5d_w", "p4_pmc_c5d_r", "k_lsm_date", 32 * 8_000_000, {"paths": 8_000_000, "time_steps": 252})
if t:
    t["note_per_date"] = ("k_lsm_date: mean over ALL its launches of a pass: the terminal-payoff launch (24 B per path) and 252 working ones "
                          "(32 B per path: S_j, S_{j-1}, V read, V written); round 4 queues no spare launch")
    c5["lsm_per_date_launch"] = t
    print("C5 per-date", t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
if c5:
    c5["note"] = ("lsm_one_launch: algorithmic_bytes is what the kernel's design reads, 16 B per path and date (each row "
                  "twice, V in registers); SURVEY 8(d)'s two-pass figure for the same sweep is 40 B per path and date")
    json.dump(c5, open(os.path.join(P, "r04_c5_pmc_traffic.json"), "w"), indent=1)


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


valu("p4_pmc_c5_va", "p4_pmc_c5_vb", "k_rbergomi_fft", 8_000_000, 252, "r04_c5gen_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline (two passes, tools/profile_r04.sh)")
valu("p4_pmc_c4_va", "p4_pmc_c4_vb", "k_rbergomi_fft", 4_000_000, 512, "r04_c4_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 tools/bench_configs.py --configs c4 --reps 2 (two passes, tools/profile_r04.sh)")


# C2: the shader clock by the counters (GRBM_GUI_ACTIVE summed over the 8 XCDs / kernel time) beside the in-kernel stamps of
# the bench line (roofline.shader_clock_GHz, mcg_generator_clock)
a, da, name, regs = counters("p4_pmc_c2_va", "k_gbm_paths")
if a and da:
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    out = {"kernel": name, "registers": regs, "counters_mean_per_launch": a, "kernel_ms_in_profiled_runs": da,
           "derived": {"shader_clock_GHz_by_GRBM_GUI_ACTIVE": cyc / (da * 1e-3) / 1e9,
                       "valu_busy_fraction": a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                       "valu_instructions_per_wavefront_step": a["SQ_INSTS_VALU"] / (10_000_000 * 252 / 64.0)}}
    try:
        j = json.load(open(os.path.join(P, "r04_bench_n1.json")))
        out["bench_line_in_kernel_stamps_GHz"] = j["roofline"].get("shader_clock_GHz")
    except Exception:
        pass
    json.dump(out, open(os.path.join(P, "r04_c2_valu_counters.json"), "w"), indent=1)
    print("r04_c2_valu_counters.json", json.dumps(out["derived"]))

# BranchingProcesses: cache counters of the bounds kernels (tools/bench_branching.py: 1M x 50, 4M x 50, 250k x 252)
br = {"command": "rocprofv3 --pmc <one counter per pass> -- python3 tools/bench_branching.py (tools/profile_r04.sh)", "kernels": {}}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCP_TCC_READ_REQ_sum"):
    f = newest(f"p4_pmc_branch_{c}/*/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if "k_branch_bounds" in r["Kernel_Name"] or "k_branch_date" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
            acc[key][c].append(float(r["Counter_Value"]))
            acc[key]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for (kname, grid), cs in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    row = {"launches_seen_per_pass": len(cs.get("TCC_HIT_sum", [])), "mean_ms_per_launch": m.pop("_ms"), "counters_mean_per_launch": m}
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        row["L2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if m.get("TCC_EA0_RDREQ_sum"):
        row["beyond_L2_G_requests_per_s"] = m["TCC_EA0_RDREQ_sum"] / (row["mean_ms_per_launch"] * 1e-3) / 1e9
    br["kernels"][f"{kname} grid_threads={grid}"] = row
if br["kernels"]:
    json.dump(br, open(os.path.join(P, "r04_branching_counters_final.json"), "w"), indent=1)
    for k, v in br["kernels"].items():
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "counters_mean_per_launch"})
f = newest("p4_branch_stats/*/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(P, "r04_branching_kernel_stats.csv"))
