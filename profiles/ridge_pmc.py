#!/usr/bin/env python3
"""Summarises a counter run of the ridge solve into profiles/<tag>_ridge_pmc.json:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE \
        --output-format csv -d gpurun_out/ridge_pmc_<tag> -- python3 profiles/micro/fit_batch16.py
    python3 profiles/ridge_pmc.py gpurun_out/ridge_pmc_<tag> <tag>
Per kernel of the solver: launches, mean counters per launch, and the share of the launch's SIMD-cycles in which the matrix pipe was busy
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)); over all of them: the same share of the summed kernel time.
Counter passes serialise the two streams of the solver, so the sums describe the kernels, not the overlapped wall time."""
import collections, csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True))[-1]
KEYS = ("k_chol_panel", "k_lu_trsm_mfma", "k_gemm_nt_dma", "k_lu_backsub_step", "k_lu_backsub_near", "k_build_system", "k_extract_wout", "k_symmetrize", "k_chol_diag_to_w")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = next((k for k in KEYS if k in r["Kernel_Name"]), None)
    if name == "k_gemm_nt_dma":
        name += "<16>" if "<16>" in r["Kernel_Name"] else "<8>"
    if name:
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"kernels": {}}
tot_busy = tot_cyc = tot_mfma = 0.0
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    e = {"launches": n, "mean_per_launch": {c: round(v, 1) for c, v in sorted(m.items())}}
    if m.get("GRBM_GUI_ACTIVE"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_pipe_busy_fraction"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4)
        if m.get("SQ_WAVE_CYCLES"):
            e["wave_parked_fraction"] = round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
        tot_busy += m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * n
        tot_cyc += cyc * n
        tot_mfma += m.get("SQ_INSTS_VALU_MFMA_F64", 0.0) * n
    out["kernels"][k] = e
out["derived"] = {
    "workload": "16 ridge systems of 5892 x 5892 + 136 right-hand sides in lockstep, blocked Cholesky (profiles/micro/fit_batch16.py), all launches of the run",
    "mfma_f64_instructions_total": tot_mfma,
    "flops_from_mfma_instructions (4x4x4_4b: 512 flop each)": tot_mfma * 512,
    "mfma_pipe_busy_fraction_of_summed_kernel_cycles": round(tot_busy / (tot_cyc * 1024), 4) if tot_cyc else None,
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_ridge_pmc.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["launches"]):
    print(k, e["launches"], e.get("mfma_pipe_busy_fraction"), e.get("wave_parked_fraction"))
