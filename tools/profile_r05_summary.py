#!/usr/bin/env python3
"""Condense round 5's GPU-box output (gpurun_out/r5*) into profiles/r05_*:
  r05_c2_limiter.json   what limits k_gbm_paths<true,3,2>: the counter passes of tools/gpu_task.sh limiter
                        (tools/pmc_passes.py: one rocprofv3 --pmc run per counter group) on TWO boards, with the same passes'
                        counters for k_probe_write -- the store-only kernel with the generator's store pattern -- beside them.
Units: SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over the 1024 SIMDs; SQ_BUSY_CU_CYCLES is summed
over the 256 CUs; GRBM_GUI_ACTIVE, TCC_* and TCP_*_sum are summed over the 8 XCDs (16 TCC channels each)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def board(tag):
    f = os.path.join(G, f"{tag}_c2lim_summary.json")
    if not os.path.exists(f):
        return None
    j = json.load(open(f))
    out = {"command": j["command"], "passes": len(j["passes"]), "dropped_counters": j["dropped_counters"], "kernels": {}}
    for k, v in j["kernels"].items():
        c = v["counters_mean_per_launch"]
        ms = sum(v["ms_per_pass"]) / len(v["ms_per_pass"])
        cyc = sum(v["GRBM_GUI_ACTIVE_per_pass"]) / len(v["GRBM_GUI_ACTIVE_per_pass"]) / 8.0
        d = {"kernel_cycles": cyc, "kernel_ms_mean_over_passes": ms, "shader_clock_GHz": cyc / (ms * 1e-3) / 1e9,
             "valu_busy_fraction": c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / 1024 / cyc,
             "valu_busy_cycles": c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / 1024,
             "lds_busy_fraction": c.get("SQ_ACTIVE_INST_LDS", 0.0) * 4 / 1024 / cyc,
             "scalar_busy_fraction": c.get("SQ_ACTIVE_INST_SCA", 0.0) * 4 / 1024 / cyc,
             "waves_waiting_on_any_instruction_fraction": c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
             "vmem_issue_fifo_full_events": sum(c.get(n, 0.0) for n in ("SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL"))}
        if "TCC_CYCLE_sum" in c:
            d.update({"tcc_write_request_stall_fraction": c["TCC_EA0_WRREQ_STALL_sum"] / c["TCC_CYCLE_sum"],
                      "tcc_dram_credit_stall_fraction": c["TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"] / c["TCC_CYCLE_sum"],
                      "tcc_busy_fraction": c["TCC_BUSY_sum"] / c["TCC_CYCLE_sum"],
                      "tcc_write_requests_in_flight_mean": c["TCC_EA0_WRREQ_LEVEL_sum"] / c["TCC_CYCLE_sum"],
                      "store_latency_cycles_tcp_to_tcc": c["TCP_TCC_WRITE_REQ_LATENCY_sum"] / c["TCP_TCC_WRITE_REQ_sum"],
                      "bytes_written_by_64B_requests": c["TCC_EA0_WRREQ_64B_sum"] * 64.0})
        out["kernels"][k] = {"kernel": v["kernel"], "launch_shape": v["launch_shape"], "launches_per_pass": v["launches_per_pass"],
                             "ms_per_pass": v["ms_per_pass"], "derived": d, "counters_mean_per_launch": c}
    return out


res = {"what": "limiter of the headline kernel, bench.py default workload C2 (10M paths x 252 steps, 20.24 GB written per launch)",
       "boards": {}}
for tag in sys.argv[1:] or ["r5a", "r5b"]:
    b = board(tag)
    if b:
        res["boards"][tag] = b
g = [b["kernels"]["k_gbm_paths"]["derived"] for b in res["boards"].values() if "k_gbm_paths" in b["kernels"]]
if g:
    cy = sorted(x["kernel_cycles"] / 1e6 for x in g)
    busy = sorted(x["valu_busy_fraction"] for x in g)
    clk = sorted(x["shader_clock_GHz"] for x in g)
    res["reading"] = (
        "k_gbm_paths takes %.2f-%.2fM shader cycles per launch, %.2fM of them with the VALU issuing (busy %.2f-%.2f): the kernel is "
        "VALU-issue-bound, and what a board makes of it is its clock under this load (%.2f-%.2f GHz: the power cap -- the store-only "
        "probe beside it runs at 2.39).  The store path is not the limiter: no issue-side FIFO-full event, write-request stalls in the "
        "L2 channels 2-4 %% of cycles where the probe -- which does saturate it -- shows 9-10 %%, store latency 187 cycles against the "
        "probe's 430.  Round 4's board ran the same VALU work at 1.83 GHz in 7.55M cycles (busy 0.80): at a higher clock the same stores "
        "take more cycles and the two limits meet.  time = max(6.04M cycles / clock, bytes / board write rate) + what does not overlap."
        % (cy[0], cy[-1], g[0]["valu_busy_cycles"] / 1e6, busy[0], busy[-1], clk[0], clk[-1]))
json.dump(res, open(os.path.join(P, "r05_c2_limiter.json"), "w"), indent=1)
print(res.get("reading"))
for t, b in res["boards"].items():
    for k, v in b["kernels"].items():
        print(t, k, json.dumps({a: round(x, 4) for a, x in v["derived"].items()}))


# ---- the rest of profiles/r05_*: what tools/profile_r05.sh collected under gpurun_out/p5_* (same condensation as round 4's
# tools/profile_r04_summary.py: WRITE_SIZE / FETCH_SIZE in KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md, SQ_ACTIVE_INST_* in
# quad-cycles over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs)
import collections
import csv
import glob
import shutil


def newest(pattern):
    f = glob.glob(os.path.join(G, pattern))
    return max(f, key=os.path.getmtime) if f else None


def counters(tag, kernel_substr):
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


for tag, out in (("p5_stats_bench", "r05_bench_kernel_stats.csv"), ("p5_stats_c5", "r05_c5_kernel_stats.csv"),
                 ("p5_stats_c5_rccl", "r05_c5_rccl_kernel_stats.csv")):
    f = newest(f"{tag}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(P, out))
        print("copied", out)
for src, out in (("p5_bench_n1.json", "r05_bench_n1.json"), ("p5_bench_c5_n1.json", "r05_bench_c5_n1.json"),
                 ("p5_bench_c5_rccl1.json", "r05_bench_c5_rccl1.json"), ("p5_bench_c5_ipc1.json", "r05_bench_c5_ipc1.json"),
                 ("p5_bench_c5_shm1.json", "r05_bench_c5_shm1.json")):
    f = os.path.join(G, src)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        try:
            json.dump(json.loads(open(f).read().strip().splitlines()[-1]), open(os.path.join(P, out), "w"), indent=1)
            print("copied", out)
        except Exception as e:   # noqa: BLE001
            print("skipped", src, e)


def traffic(wtag, rtag, kernel, alg_bytes, extra):
    w, dw, name, _ = counters(wtag, kernel)
    r, dr, _, _ = counters(rtag, kernel)
    if "WRITE_SIZE" not in w or "FETCH_SIZE" not in r:
        return None
    wr, rd = w["WRITE_SIZE"] * 1024.0, r["FETCH_SIZE"] * 1024.0 * 2.0
    d = dict(extra)
    d.update({"kernel": name, "hbm_bytes_per_launch": wr + rd, "write_bytes": wr, "fetch_bytes_corrected_x2": rd,
              "algorithmic_bytes": alg_bytes, "traffic_over_algorithmic": (wr + rd) / alg_bytes, "kernel_ms_in_profiled_runs": [dw, dr],
              "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/profile_r05.sh: {wtag}, {rtag}); "
                        "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"})
    return d


t = traffic("p5_pmc_c2_w", "p5_pmc_c2_r", "k_gbm_paths", 8 * 253 * 10_000_000, {"paths": 10_000_000, "time_steps": 252})
if t:
    json.dump(t, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join(P, "r05_c2_pmc_traffic.json"), "w"), indent=1)
    print("C2 traffic", t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
c5 = {}
for key, kernel, alg in (("generator", "k_rbergomi_fft", 8 * 253 * 8_000_000), ("lsm_one_launch", "k_lsm_big", 16 * 252 * 8_000_000)):
    t = traffic("p5_pmc_c5_w", "p5_pmc_c5_r", kernel, alg, {"paths": 8_000_000, "time_steps": 252})
    if t:
        c5[key] = t
        print("C5", key, t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
if c5:
    c5["note"] = ("lsm_one_launch: algorithmic_bytes is what the kernel's design reads, 16 B per path and date (each row twice, V in "
                  "registers); SURVEY 8(d)'s two-pass figure for the same sweep is 40 B per path and date")
    json.dump(c5, open(os.path.join(P, "r05_c5_pmc_traffic.json"), "w"), indent=1)


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
    sm = {"command": cmd, "kernel": f"{name}, {paths} paths x {steps} steps per launch", "paths_per_launch": paths,
          "registers": regs, "counters_mean_per_launch": c, "kernel_ms_in_profiled_runs": da,
          "derived": {"shader_clock_GHz": cyc / (da * 1e-3) / 1e9, "valu_instructions_per_64_path_steps": c["SQ_INSTS_VALU"] / units,
                      "fp64_instructions_per_64_path_steps": fp64 / units, "lds_instructions_per_64_path_steps": c.get("SQ_INSTS_LDS", 0.0) / units,
                      "valu_busy_fraction": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                      "cycles_per_valu_instruction": c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"]}}
    json.dump(sm, open(os.path.join(P, out), "w"), indent=1)
    print(out, json.dumps(sm["derived"]))


valu("p5_pmc_c5_va", "p5_pmc_c5_vb", "k_rbergomi_fft", 8_000_000, 252, "r05_c5gen_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline (two passes, tools/profile_r05.sh)")
valu("p5_pmc_c4_va", "p5_pmc_c4_vb", "k_rbergomi_fft", 4_000_000, 512, "r05_c4_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 tools/bench_configs.py --configs c4 --reps 2 (two passes, tools/profile_r05.sh)")
a, da, name, regs = counters("p5_pmc_c2_va", "k_gbm_paths")
if a and da:
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    json.dump({"kernel": name, "registers": regs, "counters_mean_per_launch": a, "kernel_ms_in_profiled_runs": da,
               "derived": {"shader_clock_GHz_by_GRBM_GUI_ACTIVE": cyc / (da * 1e-3) / 1e9, "valu_busy_fraction": a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                           "valu_instructions_per_64_path_steps": a["SQ_INSTS_VALU"] / (10_000_000 * 252 / 64.0)}},
              open(os.path.join(P, "r05_c2_valu_counters.json"), "w"), indent=1)

# the binned BranchingProcesses kernel (tools/gpu_task.sh pmc <tag> k_branch_date_binned,k_branch_suffix bench_branching.py, BRANCH_SHAPES=4000000x50)
for tag in ("r5h",):
    f = os.path.join(G, f"{tag}_pmc_summary.json")
    if os.path.exists(f):
        j = json.load(open(f))
        k = j["kernels"].get("k_branch_date_binned")
        if k:
            c = k["counters_mean_per_launch"]
            ms = sum(k["ms_per_pass"]) / len(k["ms_per_pass"])
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            k["derived"] = {"L2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
                            "bytes_fetched_beyond_L2_per_launch": c["TCC_EA0_RDREQ_sum"] * 64.0,
                            "row_bytes_times_8_XCDs": 32e6 * 8, "beyond_L2_G_requests_per_s": c["TCC_EA0_RDREQ_sum"] / (ms * 1e-3) / 1e9,
                            "valu_busy_fraction": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                            "mean_L2_read_latency_cycles": c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"],
                            "paths_per_launch": 786432, "gathers_per_launch": 7864320,
                            "reading": "a launch = one generation of resident paths (786 432 at four per thread) at one exercise date of a 4M-path matrix: "
                                       "its 7.9M gathers touch almost every line of the 32 MB row in the L2 of each of the 8 XCDs -- ~200 MB fetched "
                                       "beyond L2 per launch, 47 G requests/s, VALU 22 % busy.  Six such launches per date."}
        json.dump(j, open(os.path.join(P, "r05_branching_binned_counters.json"), "w"), indent=1)
        print("r05_branching_binned_counters.json", k.get("derived") if k else None)
