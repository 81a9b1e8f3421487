"""Phase time stamps of one k_spec workgroup inside the 73-field forward launch of a time step.  Needs a library built with
-DSML_GRID_STAMPS=<workgroup index> (the same switch as grid_phase_stamps.py; slots 8..13 belong to k_spec)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_package; load_package()
import test_physics_gpu as T
from make_physics_golden import coupled_inputs
from speedy_ml_amd import _lib
_, st, phis, surf = coupled_inputs(seed=2)
for rep in range(2):
    got, dyn, ph = T.device_window(st, phis, surf, 3)
    out = (C.c_ulonglong * 16)()
    _lib.lib().sml_spectral_debug_stamps(out)
    v = np.array(list(out)[8:14], dtype=np.float64) / 100.0
    d = np.diff(v)
    print("total %.2f us: stage (loads + LDS) %.2f | barrier %.2f | DFT %.2f | barrier + weights %.2f | Legendre + stores %.2f" % ((v[5] - v[0],) + tuple(d)))
