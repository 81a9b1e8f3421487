"""When does each wavefront of k_gridtend_physics3 reach the workgroup barrier?  From the raw records of profiles/micro/window_span.py
(diagnostic build: SML_SPAN_MARK stores the wave's clock at its arrival, no wait added): per launch class (short-wave step or not), mean
over workgroups of arrival and end, microseconds after the launch's first wave started.
    python profiles/micro/physics_barrier_arrivals.py gpurun_out/window_raw.npz [waves_per_workgroup]"""
import sys
import numpy as np
d = np.load(sys.argv[1])["rec"]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names = ["dynamics", "moist (+ finish)", "radiation", "vertical diffusion"][:W]
for label, launches in (("short-wave steps", [0, 1, 2, 5, 8, 11]), ("other steps", [3, 4, 6, 7, 9, 10, 12, 13])):
    arr, end = [], []
    for li in launches:
        r = d[1, li, :72 * W]
        s0 = r["start"].min() / 100.0
        arr.append(r["pad"].astype(np.float64).reshape(72, W) / 100.0 - s0)
        end.append(r["end"].astype(np.float64).reshape(72, W) / 100.0 - s0)
    arr, end = np.concatenate(arr), np.concatenate(end)
    print(f"{label}: launch span {end.max(1).mean():.2f} us (mean over workgroups of the last wave's end)")
    for w in range(W):
        print(f"   wave {w} {names[w]:20s} reaches the barrier at {arr[:, w].mean():5.2f} (max {arr[:, w].max():5.2f}), ends at {end[:, w].mean():5.2f}")
