"""fused k_gridtend_physics: repeatability and short-wave persistence on identical inputs"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_package; load_package()
import test_physics_gpu as T
from make_physics_golden import coupled_inputs
_, st, phis, surf = coupled_inputs(seed=2)
got, dyn, ph = T.device_window(st, phis, surf, 0)
def one_step():
    state = np.zeros((2, 33, 32, 62))
    for j in range(2):
        for off, k in ((0, "vor"), (8, "div"), (16, "t"), (24, "tr")):
            state[j, off:off + 8] = st[k][..., j].transpose(2, 1, 0)
        state[j, 32] = st["ps"][..., j].T
    d = torch.from_numpy(state).cuda()
    dyn.impint(1800.0)
    dyn.step(d, 2, 2, 1800.0)
    return d
from speedy_ml_amd import _lib
if len(sys.argv) > 1:
    _lib.check(_lib.lib().sml_dyn_select_physics_form(int(sys.argv[1])))
dyn.set_lradsw(True); a = one_step(); a2 = one_step()
dyn.set_lradsw(False); b = one_step(); b2 = one_step()
print("SW repeat equal", torch.equal(a, a2), "noSW repeat equal", torch.equal(b, b2), "SW vs noSW equal", torch.equal(a, b))
d = (a - b).abs()
print("max diff", float(d.max()), "rel", float(d.max() / a.abs().max()))
idx = torch.nonzero(d.reshape(2, 33, -1).max(dim=2).values)
print(idx[:20].tolist())
