"""SYRK-only timing of the Gram accumulation (n = 5760, m = 2920): the script behind the ablation table in the header of
k_gemm_nt_dma (speedy-ml_amd/csrc/train.hip).  The ablated variants (no DMA, no barrier, operands from registers, other ring
shapes, XCD-aware tile map, split-K tail) were compile-time switches of that kernel and are not kept in the tree; this
script times whatever kernel the library currently holds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from speedy_ml_amd import train
n, m = 5760, 2920
states = torch.randn((m, n), dtype=torch.float64, device="cuda")
y = torch.randn((m, 8), dtype=torch.float64, device="cuda")          # a token right-hand side: 45 extra skinny tiles
c = train.fortran_zeros(n, n); b = train.fortran_zeros(8, n)
for _ in range(2): train.chunking_matmul(states, None, y, c, b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): train.chunking_matmul(states, None, y, c, b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
tiles = 45 * 46 // 2
print(f"{dt*1e3:.3f} ms  {2*128*128*m*tiles/dt/1e12:.1f} TF/s (lower-triangle tiles only)")
