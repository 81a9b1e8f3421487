#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (merged back under gpurun_out/) into the committed summaries:
profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats table) and profiles/<tag>_summary.json (per-kernel average
duration, HBM traffic per launch from FETCH_SIZE / WRITE_SIZE with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming
reads, so the read side is doubled; both counters are in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # hybrid steps in the traced run; 0: the number of k_readout launches (one per step)


def short(name):
    for key in ("k_readout", "k_update", "k_gridtend_physics", "k_physics", "k_gridtend", "k_spectral", "k_grid", "k_spec", "k_uvvds", "k_gather", "k_scatter", "k_gemm_nt_big", "k_gemm_big_reduce", "k_gemm_nt_dma", "k_gemm_acc"):
        if key in name:
            return key
    return None


out = {"tag": tag, "kernels": {}}
stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "*", "*kernel_stats.csv"))
stats = sorted(stats, key=os.path.getmtime, reverse=True)
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    if not steps:
        steps = max([int(row["Calls"]) for row in csv.DictReader(open(stats[0])) if "k_readout" in row["Name"]] or [12])
    for row in csv.DictReader(open(stats[0])):
        k = short(row["Name"])
        if k and k not in out["kernels"]:      # (the first, i.e. the largest, row of a name: k_spec before k_spectral's short() collisions do not arise, k_gather2 / k_gather do)
            out["kernels"][k] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                 "ms_per_step": float(row["TotalDurationNs"]) / steps / 1e6}
for ctr, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{ctr}_{tag}", "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)[:1]:
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k and row["Counter_Name"] == key:
                agg[k].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            kib = sum(v) / len(v)
            d = out["kernels"].setdefault(k, {})
            d[f"{key}_KiB_per_launch"] = kib
            d[f"{key}_bytes_per_launch" + ("_x2_corrected" if ctr == "fetch" else "")] = kib * 1024.0 * (2.0 if ctr == "fetch" else 1.0)
for k, d in out["kernels"].items():
    if "FETCH_SIZE_bytes_per_launch_x2_corrected" in d:
        d["hbm_traffic_bytes_per_launch"] = d["FETCH_SIZE_bytes_per_launch_x2_corrected"] + d.get("WRITE_SIZE_bytes_per_launch", 0.0)
bench = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_bench.json")
if os.path.exists(bench):
    try:
        out["bench_line_under_rocprof"] = json.loads(open(bench).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
