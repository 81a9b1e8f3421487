#!/usr/bin/env python3
"""profiles/collect_speedy_pmc.sh's rocprofv3 --pmc passes (merged back under gpurun_out/) -> profiles/<tag>_speedy_pmc.json:
per kernel of the SPEEDY window, the mean of every counter over the kernel's dispatches, the dispatch's wall time from the
counter pass's own timestamps (profiled passes run slower than unprofiled ones: they are evidence for WHERE the time goes, not
for how long a launch takes) and a few derived figures:
   clock_ghz          GRBM_GUI_ACTIVE / 8 XCDs / wall time (reads high on short dispatches, MI355X_MICROARCH.md "DVFS give-back")
   wave_cycles_per_wave, busy fraction = SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (share of the launch in which ANY wave was resident,
   summed over the shader engines the counter is summed over), the split of SQ_WAVE_CYCLES into WAIT_ANY / WAIT_INST_ANY /
   ACTIVE_INST_ANY (quad-cycles, disjoint), LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, L2 hit rate.
FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request) and both sizes are KiB."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
KEYS = ("k_gridtend_physics", "k_gridtend", "k_spectral", "k_grid", "k_spec", "k_readout", "k_update", "k_gather", "k_scatter", "k_handoff")


def short(name):
    for key in KEYS:
        if key in name:
            return key
    return None


out = {"tag": tag, "passes": {}, "kernels": {}}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for p in ("sq1", "sq2", "tcc", "fetch", "write"):
    files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"spmc_{p}_{tag}", "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not files:
        continue
    out["passes"][p] = os.path.relpath(files[0], ROOT)
    seen = set()
    for row in csv.DictReader(open(files[0])):
        k = short(row["Kernel_Name"])
        if not k:
            continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        key = (p, row["Dispatch_Id"])
        if p == "sq1" and key not in seen:
            seen.add(key)
            agg[k]["_wall_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
            meta[k] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]), "lds_bytes": int(row["LDS_Block_Size"]),
                       "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                       "scratch": int(row["Scratch_Size"])}
for k, ctrs in agg.items():
    d = dict(meta.get(k, {}))
    d["dispatches_in_pass"] = len(ctrs.get("_wall_ns", []))
    m = {c: sum(v) / len(v) for c, v in ctrs.items()}
    d["wall_us_under_pmc"] = m.pop("_wall_ns", 0.0) / 1e3
    d["counters_mean_per_dispatch"] = {c: round(v, 1) for c, v in sorted(m.items())}
    der = {}
    g = m.get("GRBM_GUI_ACTIVE")
    if g and d["wall_us_under_pmc"]:
        der["clock_ghz_from_grbm"] = g / 8.0 / (d["wall_us_under_pmc"] * 1e3)
    if m.get("SQ_WAVES"):
        der["workgroups"] = d.get("grid", 0) / max(d.get("workgroup", 1), 1)
        der["waves"] = m["SQ_WAVES"]
        if m.get("SQ_WAVE_CYCLES"):
            der["wave_lifetime_cycles"] = 4.0 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]        # quad-cycles -> cycles
            if g:
                der["wave_lifetime_share_of_launch"] = der["wave_lifetime_cycles"] / (g / 8.0)
    if m.get("SQ_WAVE_CYCLES"):
        wc = m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                der[c.lower() + "_share_of_wave_cycles"] = m[c] / wc
    if g and m.get("SQ_BUSY_CYCLES"):
        der["sq_busy_over_grbm_active"] = m["SQ_BUSY_CYCLES"] / g
    if m.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
        der["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "FETCH_SIZE" in m:
        der["hbm_read_bytes_x2_corrected"] = m["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in m:
        der["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024.0
    d["derived"] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in der.items()}
    out["kernels"][k] = d
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_speedy_pmc.json"), "w"), indent=1)
for k in ("k_grid", "k_spec", "k_gridtend_physics", "k_spectral"):
    if k in out["kernels"]:
        print(k, json.dumps(out["kernels"][k]["derived"]), out["kernels"][k]["wall_us_under_pmc"], out["kernels"][k].get("grid"), out["kernels"][k].get("workgroup"))
