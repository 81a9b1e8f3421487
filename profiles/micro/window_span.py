"""Where a SPEEDY window's time goes, seen from its wavefronts (VERDICT r3 "next round" 2c).

Needs the diagnostic library (make -C speedy-ml_amd/csrc span -> libspeedyml_hip_span.so: every wavefront of k_grid /
k_gridtend_physics / k_spec / k_spectral records its first and last instruction on the 100 MHz constant clock, csrc/span.h).  One
6-hour window (stepone + 24 leapfrog steps = 104 dependent launches) is run on a realistic state; per launch the records give

    ramp   first wave start -> last wave start          (the dispatcher placing the launch's workgroups)
    body   mean wave lifetime
    drain  first wave end   -> last wave end
    span   first wave start -> last wave end            (what the stream sees as the kernel, minus the boundary)
    gap    last wave end    -> first wave start of the NEXT launch   (the kernel boundary as dependent work experiences it)

and sum(span) + sum(gap) is the window.  Usage:  python profiles/micro/window_span.py [out.json [raw_records.npz]]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SML_LIB_PATH", os.path.join(ROOT, "speedy-ml_amd", "csrc", "libspeedyml_hip_span.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
from speedy_ml_amd import _lib, hybrid, synth  # noqa: E402

NAMES = {1: "k_grid", 2: "k_gridtend_physics", 3: "k_spec", 4: "k_spectral"}
WAVES_CAP, LAUNCHES_CAP, HEAD = 4096, 64, 64
REC = np.dtype([("start", "<u8"), ("end", "<u8"), ("hw", "<u4"), ("xcc", "<u4"), ("pad", "<u8")])
WAVES = {1: int(os.environ.get("SML_SPAN_GRID_WAVES", 231 * 16)), 2: int(os.environ.get("SML_SPAN_PHYS_WAVES", 72 * 3)), 3: int(os.environ.get("SML_SPAN_SPEC_WAVES", 219 * 11)), 4: 248}        # waves per launch of the shipped geometries (checked against the records)


def main():
    L = _lib.lib()
    assert hasattr(L, "sml_span_attach_dyn"), "not the span build: make -C speedy-ml_amd/csrc span"
    sea = synth.land_mask()
    m = hybrid.HybridRank(list(range(hybrid.NREG)), hybrid.region_classes(sea), sea_mask=sea, mode="hybrid", n_override=1)
    stream = torch.cuda.current_stream()
    if os.environ.get("SML_SPAN_NO_DIAG"):
        m.dyn.physics_diag(False)                # (experiment: the physics without its 23 diagnostic fields)
    for _ in range(3):
        m.step(stream)
    nrec = 4 * LAUNCHES_CAP * WAVES_CAP
    buf = torch.zeros(HEAD + nrec * REC.itemsize, dtype=torch.uint8, device="cuda")
    head = np.zeros(HEAD // 4, dtype=np.uint32)
    head[8], head[9] = WAVES_CAP, LAUNCHES_CAP
    for fn in ("sml_span_attach_dyn", "sml_span_attach_spectral"):
        _lib.check(getattr(L, fn)(C.c_void_p(buf.data_ptr())))
    out = {}
    variants = [("window alone", False), ("window right behind the full-size readout's stand-in (a 7.5 GB read)", True)]
    if os.environ.get("SML_SPAN_MORE"):      # what is it about the read?  the same window behind a second window (the tables just used), behind 1 ms of matrix-core work
        variants += [("window right behind another window", "window"), ("window right behind 1 ms of fp64 matrix products (no memory stream)", "mfma"),
                     ("window behind the 7.5 GB read and then another window", "read+window")]
    for label, pre in variants:
        buf.zero_()
        buf[:HEAD] = torch.from_numpy(head.view(np.uint8)).cuda()
        junk = torch.empty(int(7.5e9 // 8), dtype=torch.float64, device="cuda") if pre in (True, "read+window") else None
        mm = torch.randn((2048, 2048), dtype=torch.float64, device="cuda") if pre == "mfma" else None
        if pre in ("window", "read+window"):
            saved = m.state.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if pre in (True, "read+window"):
            junk.sum()
        if pre == "mfma":
            for _ in range(3):
                mm @ mm
        if pre in ("window", "read+window"):
            m.dyn.window(m.state, 24, start=True, stream=stream)
            m.state.copy_(saved)                               # (device copy in stream order: the second window follows at once; its records start at launch 26)
        e0.record(stream)
        m.dyn.window(m.state, 24, start=True, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        del junk, mm
        raw = buf.cpu().numpy()
        rec = raw[HEAD:].view(REC).reshape(4, LAUNCHES_CAP, WAVES_CAP)
        launches = []
        base = 26 if pre in ("window", "read+window") else 0
        for i in range(base, base + 26):
            for kid in (1, 2, 3, 4):
                r = rec[kid - 1, i + (1 if kid == 4 else 0)]      # (the window's first k_grid has already bumped k_spectral's counter)
                r = r[r["end"] != 0]
                assert len(r) == WAVES[kid], (kid, i, len(r))
                launches.append((kid, r))
        if len(sys.argv) > 2 and pre is False:          # raw records of the window (for offline study of which wavefronts are the slow ones)
            np.savez_compressed(sys.argv[2], rec=rec[:, :27])
        assert base or (not rec[:3, 26:]["end"].any() and not rec[3, 27:]["end"].any() and not rec[3, 0]["end"].any())
        rows = []
        for kid, r in launches:
            s, e = r["start"].astype(np.float64) / 100.0, r["end"].astype(np.float64) / 100.0      # microseconds
            ss = np.sort(s)
            rows.append(dict(kernel=kid, waves=len(r), s0=s.min(), s1=s.max(), e0=e.min(), e1=e.max(), body=float(np.mean(e - s)),
                             body_max=float(np.max(e - s)), s50=float(ss[len(ss) // 2] - ss[0]), s90=float(ss[int(len(ss) * 0.9)] - ss[0])))
        for a, b in zip(rows, rows[1:]):
            a["gap"] = b["s0"] - a["e1"]
        total = rows[-1]["e1"] - rows[0]["s0"]
        table = {}
        for kid, name in NAMES.items():
            rs = [r for r in rows if r["kernel"] == kid]
            f = lambda key: float(np.mean([r[key] for r in rs if key in r]))
            table[name] = dict(launches=len(rs), waves=int(rs[0]["waves"]), ramp_us=float(np.mean([r["s1"] - r["s0"] for r in rs])),
                               half_started_us=f("s50"), ninety_pct_started_us=f("s90"), body_mean_us=f("body"), body_max_us=f("body_max"),
                               drain_us=float(np.mean([r["e1"] - r["e0"] for r in rs])), span_us=float(np.mean([r["e1"] - r["s0"] for r in rs])),
                               gap_after_us=f("gap"))
        spans = sum(r["e1"] - r["s0"] for r in rows)
        gaps = sum(r.get("gap", 0.0) for r in rows)
        out[label] = dict(launches=len(rows), window_first_wave_to_last_wave_us=total, sum_of_spans_us=spans, sum_of_gaps_us=gaps,
                          window_by_hip_events_us=e0.elapsed_time(e1) * 1e3, per_kernel=table)
        print(f"== {label}: {len(rows)} launches, {total:.1f} us first wave -> last wave ({e0.elapsed_time(e1) * 1e3:.1f} us by HIP events); "
              f"spans {spans:.1f} + gaps {gaps:.1f}")
        print(f"{'kernel':22s} {'n':>3s} {'waves':>6s} {'ramp':>6s} {'50%up':>6s} {'90%up':>6s} {'body':>6s} {'bodymx':>6s} {'drain':>6s} {'span':>6s} {'gap>':>6s}")
        for name, t in table.items():
            print(f"{name:22s} {t['launches']:3d} {t['waves']:6d} {t['ramp_us']:6.2f} {t['half_started_us']:6.2f} {t['ninety_pct_started_us']:6.2f} "
                  f"{t['body_mean_us']:6.2f} {t['body_max_us']:6.2f} {t['drain_us']:6.2f} {t['span_us']:6.2f} {t['gap_after_us']:6.2f}")
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
