import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
n, n_model, n_out, m = 5760, 132, 136, 2920
n_aug = n + n_model
torch.manual_seed(1)
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
c = train.fortran_zeros(n_aug, n_aug); b = train.fortran_zeros(n_out, n_aug)
for _ in range(3): train.chunking_matmul(states, model, y, c, b)
cnt = int(sys.argv[1])
cs = [c.clone() for _ in range(cnt)]
train.fit_chunk_hybrid_batched(cs, [b] * cnt, n, n_model, n_out)
torch.cuda.synchronize(); t0 = time.perf_counter()
train.fit_chunk_hybrid_batched(cs, [b] * cnt, n, n_model, n_out)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
flops = (2.0 / 3.0) * n_aug ** 3 + 2.0 * n_aug ** 2 * n_out
print(f"batch env {os.environ.get('SML_FIT_BATCH')} count {cnt}: {dt*1e3/cnt:.2f} ms per system, {cnt*flops/dt/1e12:.1f} TF/s")
