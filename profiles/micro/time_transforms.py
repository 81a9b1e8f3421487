import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd.spectral import Spectral
sp = Spectral()
s = torch.randn((91, 32, 62), dtype=torch.float64, device="cuda"); g = torch.zeros((91, 48, 96), dtype=torch.float64, device="cuda")
kc = torch.tensor([1] * 57 + [2] * 34, dtype=torch.int32, device="cuda")
sc = torch.tensor([1] * 48 + [0] * 25, dtype=torch.int32, device="cuda")
o = torch.zeros((73, 32, 62), dtype=torch.float64, device="cuda")
for name, fn in (("grid_mixed(91)", lambda: sp.grid_mixed(s, kc, out=g)), ("spec_mixed(73)", lambda: sp.spec_mixed(g[:73], sc, out=o))):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) / 200 * 1e3:.2f} us per launch")
