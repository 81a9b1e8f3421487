"""Sixteen 5892-row ridge systems in lockstep (the training queue's production mode), twice, for rocprofv3 --kernel-trace --stats."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
n, n_model, n_out, m = 5760, 132, 136, 2920
n_aug = n + n_model
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(1)
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
c = train.fortran_zeros(n_aug, n_aug); b = train.fortran_zeros(n_out, n_aug)
for _ in range(3): train.chunking_matmul(states, model, y, c, b)
cs = [c.clone() for _ in range(nb)]
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train.fit_chunk_hybrid_batched(cs, [b] * nb, n, n_model, n_out)
    torch.cuda.synchronize()
    print(f"{nb} systems: {1e3*(time.perf_counter()-t0)/nb:.3f} ms per system")
