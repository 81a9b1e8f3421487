"""Phase time stamps of one k_spectral workgroup (100) inside a window's leapfrog step.  Needs the diagnostic .so built with
-DSML_KSPEC_STAMPS:  rm span_obj/dynamics.o; make -C speedy-ml_amd/csrc span EXTRA=-DSML_KSPEC_STAMPS"""
import ctypes as C
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

sea = synth.land_mask()
m = hybrid.HybridRank(list(range(hybrid.NREG)), hybrid.region_classes(sea), sea_mask=sea, mode="hybrid", n_override=1)
stream = torch.cuda.current_stream()
for _ in range(2):
    m.step(stream)
names = ("loads issued + landed", "barrier", "sptend + geop (LDS sums)", "implic (3 barriers, two 8x8 mat-vecs)", "hordif + timint + stores issued", "geopotential of the new level + its store")
for nsteps in (2, 3, 4):
    m.dyn.window(m.state, nsteps, start=False, stream=stream)
    out = (C.c_ulonglong * 16)()
    _lib.check(_lib.lib().sml_dyn_kspec_stamps(out))
    v = np.array(list(out)[:7], dtype=np.float64) / 100.0
    print(f"total {v[6] - v[0]:.2f} us: " + " | ".join(f"{n} {d:.2f}" for n, d in zip(names, np.diff(v))))
