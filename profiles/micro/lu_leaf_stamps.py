"""Reads the cycle stamps the LU leaf kernel writes under SML_LU_STAMP=1 (thread 0: start, after the loads, after each of the 8
pivots, after the stores were issued, end) and prints per-phase times for a few leaf heights.  Run: SML_LU_STAMP=1 SML_LU_STAMP_FILE=...
python profiles/micro/fit_only.py 1 ; python profiles/micro/lu_leaf_stamps.py <file> <n_aug>"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 32)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5892
for leaf in (0, 4, 32, 256, 640, 1024, 1408, 1792, 2176, 2560, 2800, 2920):
    if leaf >= len(a) or a[leaf, 0] == 0: continue
    cyc, rt = a[leaf, :12], a[leaf, 16:28]
    d = np.diff(cyc)
    tot_ns = (rt[11] - rt[0]) * 10.0
    ghz = (cyc[11] - cyc[0]) / max(tot_ns, 1)
    us = d / ghz / 1e3
    print(f"leaf {leaf:4d} rows {n - 2 * leaf:5d} total {tot_ns/1e3:6.2f} us clk {ghz:.2f} GHz | load {us[0]:5.2f} | pivots " +
          " ".join(f"{x:4.2f}" for x in us[1:9]) + f" | store-issue {us[9]:5.2f} | tail {us[10]:5.2f}")
