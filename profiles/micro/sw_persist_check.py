"""k_physics: a short-wave step followed by a non-short-wave step on the same inputs must give the same bits (everything the
short-wave scheme produces is kept in the handle)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_package; load_package()
from make_physics_golden import TYEAR, gaussian_latitudes, physics_inputs
from speedy_ml_amd.physics import Physics
inp = physics_inputs()
ph = Physics(gaussian_latitudes())
g = lambda a: np.asarray(a).reshape(48, 96)
ph.set_surface(*[g(inp[k]) for k in ("fmask", "phis0", "tland", "tsea", "swav", "alb_l", "alb_s", "albsfc", "snowc")])
ph.sol_oz(TYEAR)
grids = np.zeros((41, 4608))
for i, k in enumerate(("ug", "vg", "tg", "qg", "phig")):
    grids[8 * i:8 * i + 8] = inp[k].T
grids[40] = inp["pslg"]
dg = torch.from_numpy(grids.reshape(41, 48, 96)).cuda()
a = torch.zeros((32, 48, 96), dtype=torch.float64, device="cuda"); b = a.clone()
ph.tendencies(dg, True, a)
da = {k: ph.diag(k).copy() for k in ("ssrd", "slrd", "olr", "ts", "shf")}
ph.tendencies(dg, False, b)
print("tend equal:", torch.equal(a, b), "max diff", float((a - b).abs().max()))
for k in da:
    print(k, np.abs(ph.diag(k) - da[k]).max())
d = (a - b).abs().reshape(32, -1).max(dim=1).values.cpu().numpy()
print(np.nonzero(d)[0], d[np.nonzero(d)[0]][:10])
