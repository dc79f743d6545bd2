#!/usr/bin/env python3
"""Condense round 6's GPU-box output (gpurun_out/p6_*, tools/profile_r06.sh) into profiles/r06_*:
  r06_c2_limiter.json   the headline kernel's books on the board the session landed on: VALU time, the
                        store-only probe's time in the same passes, the kernel's time, and the residual that is neither -- with
                        the counters that name it.  Appended per board (tag): run the session on several boards and each adds its row.
  pmc_traffic.json, r06_c2_pmc_traffic.json   HBM bytes of the TIMED variant k_gbm_paths<true,3,2> (round 5 tallied the ramp twin)
  r06_bench_kernel_stats.csv, r06_c5_kernel_stats.csv, r06_bench_n1.json, r06_bench_c5_n1.json, r06_c5_pmc_traffic.json,
  r06_c5gen_valu_counters.json, r06_c4_valu_counters.json, r06_c2_valu_counters.json
Units: SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE, TCC_* and
TCP_*_sum are summed over the 8 XCDs; WRITE_SIZE / FETCH_SIZE are KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


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


# ---- the limiter's books, per board ---------------------------------------------------------------------------------------
def books(tag):
    f = os.path.join(G, f"{tag}_c2lim_summary.json")
    if not os.path.exists(f):
        return None
    j = json.load(open(f))
    k = j["kernels"].get("k_gbm_paths<true") or j["kernels"].get("k_gbm_paths")   # the TIMED variant (the ramp twin <false,3,2> runs in the same command)
    pr = j["kernels"].get("k_probe_write")
    if not k or not pr:
        return None
    c = k["counters_mean_per_launch"]
    ms = sum(k["ms_per_pass"]) / len(k["ms_per_pass"])
    cyc = sum(k["GRBM_GUI_ACTIVE_per_pass"]) / len(k["GRBM_GUI_ACTIVE_per_pass"]) / 8.0
    clock = cyc / (ms * 1e-3) / 1e9
    valu_cycles = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024
    valu_ms = valu_cycles / (clock * 1e9) * 1e3
    pms = sum(pr["ms_per_pass"]) / len(pr["ms_per_pass"])
    pc = pr["counters_mean_per_launch"]
    pcyc = sum(pr["GRBM_GUI_ACTIVE_per_pass"]) / len(pr["GRBM_GUI_ACTIVE_per_pass"]) / 8.0
    residual = ms - max(valu_ms, pms)
    waves = c.get("SQ_WAVES", 0.0) or 1.0
    d = {"kernel": k["kernel"], "launch_shape": k["launch_shape"], "kernel_ms": ms, "kernel_cycles": cyc, "shader_clock_GHz": clock,
         "valu_busy_cycles": valu_cycles, "valu_busy_fraction": valu_cycles / cyc, "valu_time_ms": valu_ms,
         "store_probe_ms": pms, "store_probe_clock_GHz": pcyc / (pms * 1e-3) / 1e9, "store_probe_GBs": 20.24e9 / (pms * 1e-3) / 1e9,
         "residual_ms": residual, "residual_fraction_of_kernel": residual / ms,
         "time_model": "kernel_ms = max(valu_time_ms, store_probe_ms) + residual_ms",
         # what the residual is made of: cycles in which NO wave of a SIMD issues a VALU instruction
         "per_wave_fraction_waiting_SQ_WAIT_INST_ANY": c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
         "per_wave_fraction_SQ_WAIT_ANY": c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
         "vmem_instructions_in_flight_per_wave_SQ_INST_LEVEL_VMEM": c.get("SQ_INST_LEVEL_VMEM", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
         "store_issue_cycles_fraction_SQ_ACTIVE_INST_VMEM_plus_FLAT": (c.get("SQ_ACTIVE_INST_VMEM", 0.0) + c.get("SQ_ACTIVE_INST_FLAT", 0.0)) * 4 / 1024 / cyc,
         "scalar_busy_fraction": c.get("SQ_ACTIVE_INST_SCA", 0.0) * 4 / 1024 / cyc,
         "vmem_issue_fifo_full_events": sum(c.get(n, 0.0) for n in ("SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL")),
         "waves_per_launch": waves}
    if "TCC_CYCLE_sum" in c:
        d.update({"tcc_write_request_stall_fraction": c["TCC_EA0_WRREQ_STALL_sum"] / c["TCC_CYCLE_sum"],
                  "tcc_dram_credit_stall_fraction": c.get("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 0.0) / c["TCC_CYCLE_sum"],
                  "store_latency_cycles_tcp_to_tcc": c.get("TCP_TCC_WRITE_REQ_LATENCY_sum", 0.0) / max(c.get("TCP_TCC_WRITE_REQ_sum", 1.0), 1.0),
                  "probe_tcc_write_request_stall_fraction": pc.get("TCC_EA0_WRREQ_STALL_sum", 0.0) / max(pc.get("TCC_CYCLE_sum", 1.0), 1.0),
                  "probe_store_latency_cycles_tcp_to_tcc": pc.get("TCP_TCC_WRITE_REQ_LATENCY_sum", 0.0) / max(pc.get("TCP_TCC_WRITE_REQ_sum", 1.0), 1.0)})
    return {"command": j["command"], "passes": len(j["passes"]), "dropped_counters": j["dropped_counters"], "books": d,
            "ms_per_pass": k["ms_per_pass"], "probe_ms_per_pass": pr["ms_per_pass"], "counters_mean_per_launch": c,
            "probe_counters_mean_per_launch": pc}


out_file = os.path.join(P, "r06_c2_limiter.json")
res = json.load(open(out_file)) if os.path.exists(out_file) else {
    "what": "the headline kernel's books per board (VERDICT r5, next #5): k_gbm_paths<true,3,2>, bench.py default workload C2 (10M paths x 252 "
            "steps, 20.24 GB written per launch); eleven rocprofv3 --pmc passes (tools/gpu_task.sh limiter), the store-only k_probe_write in "
            "the same passes.  kernel_ms = max(VALU time, store-probe time) + residual.",
    "boards": {}}
for tag in sys.argv[1:] or ["p6"]:
    b = books(tag)
    if b:
        res["boards"][tag] = b
# round 5's three boards in the same terms (profiles/r05_c2_limiter.json: two boards; r05_c2_valu_counters.json: the third)
try:
    r5 = json.load(open(os.path.join(P, "r05_c2_limiter.json")))
    for t, b in r5["boards"].items():
        d, pd = b["kernels"]["k_gbm_paths"]["derived"], b["kernels"]["k_probe_write"]["derived"]
        vms = d["valu_busy_cycles"] / (d["shader_clock_GHz"] * 1e9) * 1e3
        res.setdefault("round5_boards", {})[t] = {"kernel_ms": d["kernel_ms_mean_over_passes"], "shader_clock_GHz": d["shader_clock_GHz"],
                                                  "valu_busy_fraction": d["valu_busy_fraction"], "valu_time_ms": vms,
                                                  "store_probe_ms": pd["kernel_ms_mean_over_passes"],
                                                  "residual_ms": d["kernel_ms_mean_over_passes"] - max(vms, pd["kernel_ms_mean_over_passes"])}
    v = json.load(open(os.path.join(P, "r05_c2_valu_counters.json")))
    clk = v["derived"]["shader_clock_GHz_by_GRBM_GUI_ACTIVE"]
    vms = v["counters_mean_per_launch"]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (clk * 1e9) * 1e3
    res["round5_boards"]["p5 (r05_c2_valu_counters.json; store probe of that session from r05_bench_kernel_stats.csv: 3.39 ms)"] = {
        "kernel_ms": v["kernel_ms_in_profiled_runs"], "shader_clock_GHz": clk, "valu_busy_fraction": v["derived"]["valu_busy_fraction"],
        "valu_time_ms": vms, "store_probe_ms": 3.39, "residual_ms": v["kernel_ms_in_profiled_runs"] - max(vms, 3.39)}
except Exception as e:   # noqa: BLE001
    print("round-5 boards not folded in:", e)
json.dump(res, open(out_file, "w"), indent=1)
for t, b in res["boards"].items():
    print(t, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in b["books"].items() if k not in ("launch_shape", "kernel", "time_model")}))
print("round 5:", json.dumps(res.get("round5_boards"), indent=1))

# ---- kernel statistics and bench lines -------------------------------------------------------------------------------------
for tag, out in (("p6_stats_bench", "r06_bench_kernel_stats.csv"), ("p6_stats_c5", "r06_c5_kernel_stats.csv")):
    f = newest(f"{tag}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(P, out))
        print("copied", out)
for src, out in (("p6_bench_n1.json", "r06_bench_n1.json"), ("p6_bench_c5_n1.json", "r06_bench_c5_n1.json")):
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
              "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/profile_r06.sh: {wtag}, {rtag}); "
                        "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"})
    return d


t = traffic("p6_pmc_c2_w", "p6_pmc_c2_r", "k_gbm_paths<true", 8 * 253 * 10_000_000, {"paths": 10_000_000, "time_steps": 252})   # the TIMED variant
if t:
    json.dump(t, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join(P, "r06_c2_pmc_traffic.json"), "w"), indent=1)
    print("C2 traffic", t["kernel"], t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
c5 = {}
for key, kernel, alg in (("generator", "k_rbergomi_fft", 8 * 253 * 8_000_000), ("lsm_one_launch", "k_lsm_big", 16 * 252 * 8_000_000)):
    t = traffic("p6_pmc_c5_w", "p6_pmc_c5_r", kernel, alg, {"paths": 8_000_000, "time_steps": 252})
    if t:
        c5[key] = t
        print("C5", key, t["hbm_bytes_per_launch"], t["traffic_over_algorithmic"])
if c5:
    c5["note"] = ("lsm_one_launch: algorithmic_bytes is what the kernel's design reads, 16 B per path and date (each row twice, V in "
                  "registers); SURVEY 8(d)'s two-pass figure for the same sweep is 40 B per path and date")
    json.dump(c5, open(os.path.join(P, "r06_c5_pmc_traffic.json"), "w"), indent=1)


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


valu("p6_pmc_c5_va", "p6_pmc_c5_vb", "k_rbergomi_fft", 8_000_000, 252, "r06_c5gen_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline (two passes, tools/profile_r06.sh)")
valu("p6_pmc_c4_va", "p6_pmc_c4_vb", "k_rbergomi_fft", 4_000_000, 512, "r06_c4_valu_counters.json",
     "rocprofv3 --pmc <counters> -- python3 tools/bench_configs.py --configs c4 --reps 2 (two passes, tools/profile_r06.sh)")
a, da, name, regs = counters("p6_pmc_c2_va", "k_gbm_paths<true")
if a and da:
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    json.dump({"kernel": name, "registers": regs, "counters_mean_per_launch": a, "kernel_ms_in_profiled_runs": da,
               "derived": {"shader_clock_GHz_by_GRBM_GUI_ACTIVE": cyc / (da * 1e-3) / 1e9, "valu_busy_fraction": a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                           "valu_instructions_per_64_path_steps": a["SQ_INSTS_VALU"] / (10_000_000 * 252 / 64.0)}},
              open(os.path.join(P, "r06_c2_valu_counters.json"), "w"), indent=1)
