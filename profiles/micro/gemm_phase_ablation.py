"""SYRK-only timing of the Gram accumulation (n = 5760, m = 2920, 8 target rows): the script behind the ablation numbers in the
headers of k_gemm_nt_dma and k_gemm_nt_big (speedy-ml_amd/csrc/train.hip).
  SML_GEMM_BIG=0            the 128 x 128 kernel + side streams (its own ablated variants were compile-time switches that are
                            not kept in the tree)
  SML_GEMM_ABL=1|2|4|7      k_gemm_nt_big without DMA | without LDS reads | without barrier | without all three (results are wrong,
                            the timing is the point)
  SML_GEMM_STAMPS=1         shader cycles per full-K tile, clock, cycles per MFMA, cycles at the scalar point (stderr)
  SML_GEMM_XCD=0 / SML_GEMM_SPLIT=0   list not dealt per XCD / no K-split tail"""
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
