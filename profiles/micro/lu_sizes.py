"""Ridge solves at many sizes against numpy (debug aid for the LU: which n_aug fail, and how badly)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
from speedy_ml_amd._lib import SmlError
sizes = [int(x) for x in sys.argv[1:]] or [97, 300, 508, 512, 516, 520, 572, 580, 640, 708, 1000, 1030, 1100, 1536, 1540, 2050]
for n in sizes:
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n)); a = a + a.T
    b = rng.standard_normal((3, n))
    try:
        w = train.fit_chunk_hybrid(torch.from_numpy(a).cuda(), torch.from_numpy(np.ascontiguousarray(b.T)).cuda(), n, 0, 3, 0.0, 0.0, 0.0, False)
        wg = w.cpu().numpy().T
        want = np.linalg.solve(a.T, b.T).T
        print(n, "err", float(np.max(np.abs(wg - want)) / np.max(np.abs(want))))
    except SmlError as e:
        print(n, "FAIL", str(e)[-90:])
