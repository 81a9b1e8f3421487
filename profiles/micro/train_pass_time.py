"""Wall time of one reservoir_layer_chunking_hybrid pass on the device (sml_bank_train_pass) for a few full-size reservoirs at the
shipped batch size (m = 98) -- the recurrence plus the grouped Gram updates."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package; load_package()
from speedy_ml_amd import train
from speedy_ml_amd.reservoir import ReservoirBank
from speedy_ml_amd.synth import make_reservoir
nres, batch, nb = int(os.environ.get("NRES", "8")), 98, 20
r = make_reservoir(seed=1, dense_win=False)
bank = ReservoirBank(nres)
rows = np.arange(1, r.n + 1, dtype=np.int32); cols = (np.arange(r.n, dtype=np.int32) // r.win_q + 1).astype(np.int32)
for s in range(nres):
    bank.load_sparse_win(s, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, rows, cols, r.win_vals, r.wout, r.mean, r.std, None)
T = 6 + batch * nb
rng = np.random.default_rng(0)
noisy = torch.from_numpy(rng.standard_normal((T, nres, 576))).cuda()
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
models = [to_dev(rng.standard_normal((r.n_model, T))) for _ in range(nres)]
targets = [to_dev(rng.standard_normal((r.n_out, T))) for _ in range(nres)]
cs = [train.fortran_zeros(r.n_aug, r.n_aug) for _ in range(nres)]
bs = [train.fortran_zeros(r.n_out, r.n_aug) for _ in range(nres)]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nbf = bank.train_pass(noisy, 6, batch, models, targets, cs, bs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pass: %d reservoirs, %d columns, %d batches of %d: %.1f ms = %.2f ms per reservoir and batch" % (nres, T, nbf, batch, dt * 1e3, dt * 1e3 / nres / nbf))
