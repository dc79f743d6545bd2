#!/usr/bin/env python3
"""Run one command under `rocprofv3 --pmc` once per counter group (one group per pass: the pool wants counters collected on
their own, and a pass holds only as many counters as one hardware block has slots) and condense the passes into ONE json:
per kernel (substring match) the mean of every counter per launch and the mean launch duration of each pass.

    python3 tools/pmc_passes.py --tag r5_c2lim --kernels k_gbm_paths,k_probe_write \
        --group "SQ_WAVES SQ_BUSY_CYCLES" --group "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
        -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra

Counters the device does not list (`rocprofv3 -L`) are dropped from their group and named in the summary.  The program
after `--` must be the interpreter / binary itself (no env, sh -c, ... in between: the profiler's preloaded library has the
GPU initialised before the program starts).  A pass that ends in a time-out stops the whole session."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def available():
    p = subprocess.run(["rocprofv3", "-L"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    names = set()
    for ln in p.stdout.splitlines():
        if ln.startswith("Counter_Name"):
            names.add(ln.split(":", 1)[1].strip())
    return names


def read_pass(d, kernels):
    out = {}
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv"))
    for k in kernels:
        acc, dur, seen, name, regs = collections.defaultdict(list), [], set(), None, None
        for f in files:
            for r in csv.DictReader(open(f)):
                if k not in r["Kernel_Name"]:
                    continue
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (f, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                name = r["Kernel_Name"]
                regs = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                        "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
        if name:
            out[k] = {"kernel": name, "launches": len(dur), "ms": sum(dur) / len(dur), "launch_shape": regs,
                      "counters": {c: sum(v) / len(v) for c, v in acc.items()}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kernels", required=True, help="comma-separated substrings of kernel names")
    ap.add_argument("--group", action="append", default=[], help="counters of one pass, space-separated")
    ap.add_argument("--pass-timeout", type=int, default=240)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    kernels = [k for k in a.kernels.split(",") if k]
    os.makedirs(G, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    have = available()
    summary = {"command": "rocprofv3 --pmc <group> -- " + " ".join(cmd), "passes": [], "dropped_counters": [], "kernels": {}}
    for i, g in enumerate(a.group):
        want = g.split()
        use = [c for c in want if c in have]
        summary["dropped_counters"] += [c for c in want if c not in have]
        if not use:
            continue
        d = os.path.join(G, f"{a.tag}_p{i}")
        t0 = time.time()
        try:
            p = subprocess.run(["rocprofv3", "--pmc", *use, "--output-format", "csv", "-d", d, "--", *cmd], cwd=ROOT,
                               stdout=open(d + ".log", "w"), stderr=subprocess.STDOUT, timeout=a.pass_timeout)
            rc = p.returncode
        except subprocess.TimeoutExpired:
            print(f"pass {i} timed out: stopping", flush=True)
            summary["passes"].append({"group": use, "rc": "timeout"})
            break
        got = read_pass(d, kernels) if rc == 0 else {}
        summary["passes"].append({"group": use, "rc": rc, "seconds": round(time.time() - t0, 1)})
        print(f"pass {i}: rc={rc} {time.time() - t0:.0f}s {' '.join(use)}", flush=True)
        for k, v in got.items():
            e = summary["kernels"].setdefault(k, {"kernel": v["kernel"], "launch_shape": v["launch_shape"], "launches_per_pass": v["launches"],
                                                  "ms_per_pass": [], "counters_mean_per_launch": {}})
            e["ms_per_pass"].append(round(v["ms"], 4))
            for c, x in v["counters"].items():
                e["counters_mean_per_launch"].setdefault(c, x)
                if c == "GRBM_GUI_ACTIVE":   # collected in several passes: keep them all (the clock of each pass)
                    e.setdefault("GRBM_GUI_ACTIVE_per_pass", []).append(x)
    json.dump(summary, open(os.path.join(G, f"{a.tag}_summary.json"), "w"), indent=1)
    print("summary:", os.path.join("gpurun_out", f"{a.tag}_summary.json"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
