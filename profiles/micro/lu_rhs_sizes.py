import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
for n, no in ((3968, 8), (3968, 3), (3840, 8), (4096, 8), (2048, 8), (1000, 8), (3968, 136)):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n)); a = a @ a.T / n + np.eye(n)
    b = rng.standard_normal((no, n))
    w = train.fit_chunk_hybrid(torch.from_numpy(a).cuda(), torch.from_numpy(np.ascontiguousarray(b.T)).cuda(), n, 0, no, 0.0, 0.0, 0.0, False)
    wg = w.cpu().numpy().T
    want = np.linalg.solve(a.T, b.T).T
    print(n, no, "err", float(np.max(np.abs(wg - want)) / np.max(np.abs(want))), flush=True)
