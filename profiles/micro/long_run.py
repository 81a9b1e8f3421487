"""BASELINE config 3 length: 120 closed-loop hybrid steps (30 days) with the full SPEEDY window; prints the range of the forecast fields"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package; load_package()
from speedy_ml_amd import domain, hybrid, synth
sea = synth.land_mask(); classes = hybrid.region_classes(sea)
m = hybrid.HybridRank(list(range(1152)), classes, sea_mask=sea, mode="hybrid", n_override=1)
st = torch.cuda.current_stream()
t0 = time.time()
for k in range(120):
    m.step(st)
    if (k + 1) % 20 == 0:
        torch.cuda.synchronize()
        F = m.F[:domain.G2_OFF].reshape(8, 48, 96, 4)
        print(k + 1, "safe", int(m.safe.item()), "T %.1f..%.1f" % (float(F[..., 0].min()), float(F[..., 0].max())),
              "u %.1f..%.1f" % (float(F[..., 1].min()), float(F[..., 1].max())), "q %.2f..%.2f" % (float(F[..., 3].min()), float(F[..., 3].max())),
              "finite", bool(torch.isfinite(m.F[:domain.GP_OFF]).all()), "precip(cnv) max %.2f" % float(m.phys.diag("precnv").max()), flush=True)
print("elapsed", time.time() - t0)
