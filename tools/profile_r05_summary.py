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
