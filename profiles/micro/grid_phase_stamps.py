"""Phase time stamps of one k_grid workgroup inside the 77-field launch of a time step with physics.  Needs a library built with
-DSML_GRID_STAMPS=<workgroup index> (make -C speedy-ml_amd/csrc EXTRA=-DSML_GRID_STAMPS=100); thread 0 of that workgroup stamps."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_package; load_package()
import test_physics_gpu as T
from make_physics_golden import coupled_inputs
from speedy_ml_amd import _lib
_, st, phis, surf = coupled_inputs(seed=2)
for rep in range(3):
    got, dyn, ph = T.device_window(st, phis, surf, 3)
    out = (C.c_ulonglong * 16)()
    _lib.lib().sml_spectral_debug_stamps(out)
    v = np.array(list(out)[:6], dtype=np.float64) / 100.0
    d = np.diff(v)
    print("total %.2f us: issue loads %.2f | wait+barrier %.2f | Legendre %.2f | barrier %.2f | Fourier+stores %.2f" % ((v[5] - v[0],) + tuple(d)))
