"""Phase time stamps of the two-kernel SPEEDY window (k_mspace / k_latspace, speedy-ml_amd/csrc/dynamics.hip).  Needs a library
built with -DSML_DYN_STAMPS (make -C speedy-ml_amd/csrc CXXFLAGS+=-DSML_DYN_STAMPS); otherwise the stamps stay zero."""
import sys, ctypes as C, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package; load_package()
from speedy_ml_amd import _lib
from speedy_ml_amd.spectral import Spectral
from speedy_ml_amd.dynamics import Dynamics
sp = Spectral(); dyn = Dynamics(sp)
_lib.check(_lib.lib().sml_dyn_select_window_form(1))
z = torch.zeros((32, 62), dtype=torch.float64, device="cuda")
dyn.set_boundary(z, z, z)
s = np.zeros((2, 33, 32, 62)); s[:, 16:24, 0, 0] = 300.0
rng = np.random.default_rng(0); s[:, :, :10, :20] += rng.standard_normal((2, 33, 10, 20)) * 1e-6
state = torch.from_numpy(s).cuda()
dyn.window(state, 1, start=True)   # stamps keep the FIRST value: A0 is skipped by do_step-only slots
out = (C.c_ulonglong * 64)()
L = _lib.lib()
L.sml_dyn_debug_stamps(out)
v = np.array(list(out), dtype=np.float64)
# wall_clock64 ticks at 100 MHz on gfx9 (10 ns)
m = v[0:8]; l = v[10:16]
print("raw", (m - m[0]) / 100.0)
print("A0 (synthesis-only launch): load %.1f | first stepping launch: symasym..analysis %.1f step %.1f derived %.1f synth-loop %.1f stores %.1f" % ((m[1]-m[0])/100, (m[3]-m[2])/100, (m[4]-m[3])/100, (m[5]-m[4])/100, (m[7]-m[5])/100, (m[6]-m[7])/100))
print("k_latspace phases (us): load %.1f synth %.1f gridtend %.1f fold %.1f fwd %.1f total %.1f" % tuple(list(np.diff(l) / 100.0) + [(l[5] - l[0]) / 100.0]))
