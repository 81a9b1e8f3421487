"""Phase time stamps of the two wavefronts of k_gridtend_physics (workgroup 36, a tropical row) inside a window's time steps.  Needs a
library built with -DSML_PHYS_STAMPS into the diagnostic .so:  rm span_obj/dynamics.o; make -C speedy-ml_amd/csrc span EXTRA=-DSML_PHYS_STAMPS"""
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
L = _lib.lib()
W0 = ("loads + grid-point dynamics", "barrier")
W1 = ("loads + column_thermo", "convmf + lscond", "vdifsc", "barrier (wait for the others)", "sums + stores")
W2 = ("loads + column_thermo", "convmf + lscond again (SW steps)", "cloud + radsw", "radlw down", "suflux", "radlw up", "park, barrier, diagnostics")
for nsteps, label in ((1, "short-wave step"), (2, "step without short-wave radiation"), (3, "step without short-wave radiation")):
    m.dyn.window(m.state, nsteps, start=False, stream=stream)
    out = (C.c_ulonglong * 48)()
    _lib.check(L.sml_dyn_phys_stamps(out))
    v = np.array(list(out), dtype=np.float64) / 100.0
    w0, w1, w2 = v[0:3], v[16:22], v[32:40]
    t0 = min(w0[0], w1[0], w2[0])
    print(f"== {label} (last of {nsteps}): wave 0 (dynamics) ends at {w0[2] - t0:.2f} us, wave 1 (moist + finish) at {w1[5] - t0:.2f}, wave 2 (radiation) at {w2[7] - t0:.2f}")
    print("  wave 0: " + " | ".join(f"{n} {d:.2f}" for n, d in zip(W0, np.diff(w0))))
    print("  wave 1: " + " | ".join(f"{n} {d:.2f}" for n, d in zip(W1, np.diff(w1))))
    print("  wave 2: " + " | ".join(f"{n} {d:.2f}" for n, d in zip(W2, np.diff(w2))))
