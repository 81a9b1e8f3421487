"""The 5892 x 5892 (+136 right-hand sides) ridge solve of config 4 with each solver (sml_train_select_solver: 0 = Cholesky with LU
fallback, 1 = pivoted LU), single and 16 in lockstep; prints wall times, the algorithmic rate (SURVEY 8d's dgesv count, 146 GFLOP) and
the backward error.  `python fit_solvers.py chol 3` runs only the Cholesky single solve 3 times (for rocprofv3 --kernel-trace)."""
import ctypes as C
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import _lib, train
n, n_model, n_out, m = 5760, 132, 136, 2920
n_aug = n + n_model
torch.manual_seed(1)
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
c = train.fortran_zeros(n_aug, n_aug); b = train.fortran_zeros(n_out, n_aug)
if os.environ.get("FIT_PRE98"):          # the bench's order: the m = 98 product (k_gemm_nt_dma + side streams) first
    s98, m98, y98 = states[:98].contiguous(), model[:98].contiguous(), y[:98].contiguous()
    for _ in range(23): train.chunking_matmul(s98, m98, y98, c, b)
for _ in range(3): train.chunking_matmul(states, model, y, c, b)
torch.cuda.synchronize()
flops = (2.0 / 3.0) * n_aug ** 3 + 2.0 * n_aug ** 2 * n_out
reg = torch.diag(torch.cat([torch.full((n_model,), 1.0), torch.full((n,), 1e-6)])).to("cuda", torch.float64)
L = _lib.lib()
only = sys.argv[1] if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def timed(fn, k):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for name, sel in (("chol", 0), ("lu", 1)):
    if only and only != name:
        continue
    L.sml_train_select_solver(sel)
    train.release_workspace()
    dt = timed(lambda: train.fit_chunk_hybrid(c, b, n, n_model, n_out), reps)
    w = train.fit_chunk_hybrid(c, b, n, n_model, n_out)
    resid = (c + reg) @ w - b
    berr = float(resid.norm() / (torch.linalg.matrix_norm(c + reg) * w.norm() + b.norm()))
    print(f"{name}: single {dt*1e3:.2f} ms = {flops/dt/1e12:.1f} TF/s (dgesv count) = {flops/dt/1e12/78.6:.3f} of 78.6; backward error {berr:.2e}", flush=True)
    if only:
        continue
    for nb in (8, 16):
        cs = [c.clone() for _ in range(nb)]
        dtb = timed(lambda: train.fit_chunk_hybrid_batched(cs, [b] * nb, n, n_model, n_out), 2)
        print(f"{name}: {nb} in lockstep {dtb*1e3/nb:.2f} ms per system = {nb*flops/dtb/1e12:.1f} TF/s = {nb*flops/dtb/1e12/78.6:.3f}", flush=True)
        del cs
L.sml_train_select_solver(0)
