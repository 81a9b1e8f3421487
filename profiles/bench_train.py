#!/usr/bin/env python3
"""Auxiliary measurement for BASELINE config 4 (ridge-regression training kernels K8/K9) on one MI355X.
Prints one JSON line: achieved fp64 TFLOP/s of the Gram accumulation (executed-flop convention: lower-triangle tiles)
and the wall time of the 5892x5892 LU ridge solve.  Not the driver's bench (that is /bench.py)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from speedy_ml_amd import train  # noqa: E402


def main():
    n, n_model, n_out = 5760, 132, 136
    n_aug = n + n_model
    out = {}
    for m in (98, 2920):
        states = torch.randn((m, n), dtype=torch.float64, device="cuda")
        model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
        y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
        c = train.fortran_zeros(n_aug, n_aug)
        b = train.fortran_zeros(n_out, n_aug)
        for _ in range(3):
            train.chunking_matmul(states, model, y, c, b)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            train.chunking_matmul(states, model, y, c, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nt = (n + 127) // 128
        tiles = nt * (nt + 1) // 2
        executed = 2.0 * 128 * 128 * m * tiles + 2.0 * m * n_aug * (n_model + n_out)   # lower-triangle tiles + small blocks
        full = 2.0 * n_aug * n_aug * m + 2.0 * n_out * n_aug * m
        out[f"accumulate_m{m}"] = {"ms": dt * 1e3, "tflops_executed": executed / dt / 1e12,
                                   "tflops_full_gemm_convention": full / dt / 1e12,
                                   "c_traffic_GBps_lower": (n_aug * n_aug * 8.0) / dt / 1e9}
    # ridge solve at full size
    m = 2920
    states = torch.randn((m, n), dtype=torch.float64, device="cuda")
    model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
    y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
    c = train.fortran_zeros(n_aug, n_aug)
    b = train.fortran_zeros(n_out, n_aug)
    for _ in range(3):
        train.chunking_matmul(states, model, y, c, b)
    train.fit_chunk_hybrid(c, b, n, n_model, n_out)              # (allocates the workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        w = train.fit_chunk_hybrid(c, b, n, n_model, n_out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    flops = (2.0 / 3.0) * n_aug ** 3 + 2.0 * n_aug ** 2 * n_out
    out["fit_5892"] = {"ms": dt * 1e3, "tflops": flops / dt / 1e12}
    reg = torch.diag(torch.cat([torch.full((n_model,), 1.0), torch.full((n,), 1e-6)])).to("cuda", torch.float64)
    resid = (c + reg) @ w - b          # column-major buffers: torch [n_aug, n_out] = Z ; C symmetric
    out["fit_5892"]["backward_error"] = float(resid.norm() / b.norm())
    # eight ridge solves in flight (one size class): amortised time per reservoir
    cs = [c.clone() for _ in range(8)]
    train.fit_chunk_hybrid_batched(cs, [b] * 8, n, n_model, n_out)          # (grows the workspace to 8 systems)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        train.fit_chunk_hybrid_batched(cs, [b] * 8, n, n_model, n_out)
    torch.cuda.synchronize()
    dt8 = (time.perf_counter() - t0) / 2
    out["fit_5892_batched8"] = {"ms_total": dt8 * 1e3, "ms_per_system": dt8 * 1e3 / 8, "tflops": 8 * flops / dt8 / 1e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
