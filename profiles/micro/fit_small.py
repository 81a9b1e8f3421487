"""A small ridge solve (n_aug = 384: three 128-row blocks) repeated, for rocprofv3 --kernel-trace: the chain kernels of the Cholesky
(k_chol_potrf, k_lu_trsm_mfma<2>, the block-row GEMM, the back-substitution steps) on an otherwise idle chip."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
n, n_model, n_out, m = int(sys.argv[1]) if len(sys.argv) > 1 else 376, 8, 136, 1024
n_aug = n + n_model
torch.manual_seed(1)
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
c = train.fortran_zeros(n_aug, n_aug); b = train.fortran_zeros(n_out, n_aug)
train.chunking_matmul(states, model, y, c, b)
torch.cuda.synchronize()
for _ in range(5):
    t0 = time.perf_counter()
    w = train.fit_chunk_hybrid(c, b, n, n_model, n_out)
    torch.cuda.synchronize()
    print(f"fit {1e3*(time.perf_counter()-t0):.3f} ms")
