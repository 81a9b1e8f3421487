#!/usr/bin/env python3
"""The Gram update of one full-size reservoir (n = 5760, 132 model rows, 136 targets, m = 2920 time columns), repeated: the
workload behind the k_gemm_nt_big rows of profiles/r2_*.  Prints the average wall time per update (HIP events on the launch stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
n, n_model, n_out, m = 5760, 132, 136, int(os.environ.get("GRAM_M", "2920"))
reps = int(os.environ.get("GRAM_REPS", "20"))
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
c = train.fortran_zeros(n + n_model, n + n_model); b = train.fortran_zeros(n_out, n + n_model)
for _ in range(3): train.chunking_matmul(states, model, y, c, b)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps): train.chunking_matmul(states, model, y, c, b)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / reps * 1e-3
nt = (n + 127) // 128
executed = 2.0 * 128 * 128 * m * (nt * (nt + 1) // 2) + 2.0 * m * (n + n_model) * (n_model + n_out)
print(f"{dt*1e3:.3f} ms per update, {executed/dt/1e12:.1f} TF/s executed (lower-triangle 128-blocks + skinny blocks), {executed/dt/1e12/78.6*100:.1f} % of 78.6")
