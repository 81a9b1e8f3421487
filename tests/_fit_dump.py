"""Helper of tests/test_train_gpu.py: one ridge solve of a seeded system in a fresh process (the LU's stream layout is chosen once
per process from the environment), W_out written to the .npy file named on the command line."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train

n, n_model, n_out, m = int(sys.argv[2]), 132, 136, 900
torch.manual_seed(11)
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
c = train.fortran_zeros(n + n_model, n + n_model)
b = train.fortran_zeros(n_out, n + n_model)
train.chunking_matmul(states, model, y, c, b)
w = train.fit_chunk_hybrid(c, b, n, n_model, n_out)
torch.cuda.synchronize()
np.save(sys.argv[1], w.cpu().numpy())
