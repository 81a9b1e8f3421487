#!/usr/bin/env python3
"""Summarises the counter run of profiles/gram_only.py into profiles/<tag>_gram_pmc.json:
    GRAM_REPS=5 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY \
        SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gram_pmc -- python3 profiles/gram_only.py
    python3 profiles/gram_pmc.py gpurun_out/gram_pmc r2 "<timing line of the same script without counters>" """
import collections, csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
timing = sys.argv[3] if len(sys.argv) > 3 else ""
f = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    name = "k_gemm_nt_big" if "gemm_nt_big" in k else "k_gemm_big_reduce" if "big_reduce" in k else None
    if name:
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in sorted(d.items())} for k, d in agg.items()}
b = out["k_gemm_nt_big"]
n, busy, gui = (b[c]["mean_per_launch"] for c in ("SQ_INSTS_VALU_MFMA_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
out["derived"] = {
    "workload": "one Gram update: n=5760, 132 model rows, 136 targets, m=2920 (624 tiles of 256x128, all products in one launch)",
    "mfma_f64_instructions_per_launch": n,
    "expected_624_tiles_x_365_ktiles_x_256_mfma_x_4_waves": 624 * 365 * 256 * 4,
    "mfma_busy_cycles_per_instruction": busy / n,
    "kernel_cycles_per_xcd (GRBM_GUI_ACTIVE / 8 XCDs)": gui / 8,
    "mfma_pipe_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)": busy / (gui / 8 * 1024),
    "wave_parked_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": b["SQ_WAIT_ANY"]["mean_per_launch"] / b["SQ_WAVE_CYCLES"]["mean_per_launch"],
    "issue_stall_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: waiting for the matrix pipe)": b["SQ_WAIT_INST_ANY"]["mean_per_launch"] / b["SQ_WAVE_CYCLES"]["mean_per_launch"],
    "steady_state_timing_same_script_without_counters": timing,
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_gram_pmc.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
