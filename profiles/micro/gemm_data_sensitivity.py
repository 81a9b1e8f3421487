import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
n, n_model, n_out, m = 5760, 132, 136, 2920
n_aug = n + n_model
for name, gen in (("zeros", lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")),
                  ("ones", lambda *s: torch.ones(*s, dtype=torch.float64, device="cuda")),
                  ("randn", lambda *s: torch.randn(*s, dtype=torch.float64, device="cuda"))):
    states, model, y = gen(m, n), gen(m, n_model), gen(m, n_out)
    c = train.fortran_zeros(n_aug, n_aug); b = train.fortran_zeros(n_out, n_aug)
    for _ in range(2): train.chunking_matmul(states, model, y, c, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): train.chunking_matmul(states, model, y, c, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(name, f"{dt*1e3:.3f} ms")
