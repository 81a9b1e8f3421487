#!/usr/bin/env python3
"""Benchmark of the SPEEDY-ML hybrid-step hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over the 1152 local reservoirs of the T30L8 hybrid model (BASELINE.json
config 3), driven by the native engine -- one sml_hybrid_step call per step, the path the Fortran drop-in ships (--host python: the same
kernels issued call by call from Python): batched predict of every resident reservoir, the region exchange (pack / all-gather over RCCL when
N>1 / scatter + clamps), the SPEEDY hand-off and the 6-hour SPEEDY window on the device (26 time steps: spectral
transforms, grid-point tendencies with the column physics, semi-implicit spectral step), and the gather + standardisation
of the next inputs.  Regions are sharded over ranks exactly as processor_decomposition does
(src/res_domain.f90:31-62); the only data-path collective is the all-gather of the outvec slab.  Inputs are
synthetic (seeded, ERA5-shaped) and resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line with the contract's keys plus "roofline" and "cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed hybrid steps (120 = the 30 days of BASELINE config 3)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="hybrid", choices=["hybrid", "sweep", "ml_only"],
                    help="sweep = reservoir predict sweep only; ml_only = the reference's ML-only forecast loop (predict_ml + exchange, no "
                         "SPEEDY); development aids, the driver uses the default")
    ap.add_argument("--host", default="native", choices=["native", "python"],
                    help="who drives a hybrid step: native = ONE sml_hybrid_step call per step (csrc/hybrid.hip, the engine the Fortran drop-in's "
                         "sendrecievegrid drives; default), python = HybridRank.step issuing the same kernels call by call (development aid; "
                         "the only host of --mode sweep / ml_only and of SML_PIPELINE=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training", action="store_true", help="skip the BASELINE config 4 training-kernel block (N = 1 only) of the JSON line")
    ap.add_argument("--slab", action="store_true", help="BASELINE config 5: add the 1152-region slab-ocean reservoirs and their coupling")
    ap.add_argument("--float32-weights", action="store_true",
                    help="synthetic weights rounded to float32, as a reservoir read from the reference's NetCDF weight files holds them: the "
                         "banks then read their compact copies (DESIGN 4.13); the default line also carries this case as 'float32_weight_files'")
    ap.add_argument("--no-float32-block", action="store_true", help="skip the second model of the 'float32_weight_files' block (profiling runs)")
    ap.add_argument("--no-physics", action="store_true", help="adiabatic SPEEDY window (development aid: isolates the cost of the column physics)")
    ap.add_argument("--regions", type=int, default=1152, help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)     # launch + rendezvous only (CPU test of the N > 1 launch)
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the ranks ourselves, as a CHILD
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` (one rank per GPU over RCCL), relay its output and
    exit code.  Nothing here touches the GPU, and the launcher is a child process, never an exec of this one."""
    import subprocess
    if os.environ.get("SML_BENCH_CHILD") == "1":
        raise SystemExit("bench.py: a child of a self-launched run came up without WORLD_SIZE: refusing to launch again")
    # --standalone: torchrun picks AND HOLDS a free port for its own c10d rendezvous (no bind-close-reuse race on a port of ours)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, SML_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def cpu_baseline(model, budget_s=12.0):
    """Reference-faithful CPU path (the oracle, 1 core) on a bounded sample of the same workload.  The only place outside
    tests/ and smoke() that touches oracle/: it is the reported baseline, never part of the measured or shipped path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle
    from speedy_ml_amd.dynamics import DELT
    from speedy_ml_amd.hybrid import NREG
    from speedy_ml_amd.spectral import IL, IX, MX2, NX
    from speedy_ml_amd.synth import make_reservoir
    o = Oracle()
    r = make_reservoir(seed=20240954)          # interior + SST class, dense W_in as the reference stores it
    x = np.zeros(r.n)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s or n < 3:
        x, out = o.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, 1.0,
                               r.feedback, r.local_model, x)
        n += 1
    per_predict = (time.perf_counter() - t0) / n
    sample = (f"{n} reference-faithful predict calls (COO SpMV, dense 26.5 MB W_in matmul, W_out GEMV) of one interior "
              f"reservoir: {per_predict * 1e3:.3f} ms each, x1152 per step")
    total = per_predict * NREG
    # the same predict on every host core at once (one reservoir copy per thread, as every MPI rank of the reference owns its
    # matrices; ctypes releases the GIL): the aggregate rate prices the all-cores variant of the baseline below
    import threading
    ncores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))
    counts = [0] * ncores

    def worker(i):
        rr = make_reservoir(seed=20240954 + i)
        xx = np.zeros(rr.n)
        t_end = time.perf_counter() + 4.0
        while time.perf_counter() < t_end:
            xx, _ = o.predict_raw(rr.n, rr.d, rr.n_model, rr.n_out, rr.rows, rr.cols, rr.vals, rr.win, rr.wout, 1.0, rr.feedback,
                                  rr.local_model, xx)
            counts[i] += 1

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(ncores)]
    t1 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    par_rate = sum(counts) / (time.perf_counter() - t1)          # predicts per second, all cores
    total_par = NREG / par_rate
    if model.mode == "hybrid":
        from _oracle import DynOracle, RefSpectral
        rng = np.random.default_rng(1)
        v = rng.standard_normal((MX2, NX))
        gfield = rng.standard_normal((IX, IL))
        # transforms: the COMPILED REFERENCE (oracle/_ref: the reference's own FFTPACK + Legendre code) when its .so
        # travelled with the snapshot, else the oracle's direct-DFT restatement (5x slower: flattering)
        eng = RefSpectral() if RefSpectral.available() else o
        which = "compiled reference spe_spectral.f90/FFTPACK (oracle/_ref)" if RefSpectral.available() else "oracle direct-DFT restatement"

        def pair_time(e, budget):
            t1 = time.perf_counter()
            m = 0
            while time.perf_counter() - t1 < budget:
                e.grid(v, 1)
                e.spec(gfield)
                m += 1
            return (time.perf_counter() - t1) / m, m
        per_pair, m = pair_time(eng, 2.0)
        per_pair_oracle, _ = pair_time(o, 1.0)
        # one adiabatic time step with the oracle (dynamics_oracle.c); its own direct-DFT transforms are swapped for the
        # reference's FFTPACK ones in the estimate: 50 inverse + 73 forward per step
        dyn = DynOracle(o)
        dyn.impint(2 * DELT, 0.5)
        mask = np.repeat(o.table(11), 2, axis=0)
        small = lambda shape, sc: rng.standard_normal(shape) * sc * mask.reshape((MX2, NX) + (1,) * (len(shape) - 2))
        st = {"vor": small((MX2, NX, 8, 2), 1e-7), "div": small((MX2, NX, 8, 2), 1e-8), "t": small((MX2, NX, 8, 2), 0.1),
              "tr": small((MX2, NX, 8, 2), 1e-4), "ps": small((MX2, NX, 2), 1e-4)}
        st["t"][0, 0] += dyn.table(17)[:, None] * np.sqrt(2.0)
        zero = np.zeros((MX2, NX))
        t1 = time.perf_counter()
        ns = 0
        while time.perf_counter() - t1 < 3.0 or ns < 2:
            st = dyn.step_dry(2, 2, 2 * DELT, 0.5, 0.05, 0.53, st, zero, zero, zero)
            ns += 1
        per_step_oracle = (time.perf_counter() - t1) / ns
        per_tr, per_tr_oracle = per_pair / 2.0, per_pair_oracle / 2.0
        per_step = max(per_step_oracle - 123 * per_tr_oracle, 0.0) + 123 * per_tr
        nst = 0 if model.leapfrog_steps is None else model.leapfrog_steps + 2
        phys_note = ""
        if getattr(model, "phys", None) is not None:
            # the column physics: the reference's own phy_*.f90 (oracle/_ref/libref_phy.so) through the phypar call sequence, a
            # short-wave step and a step without; + phypar's 41 inverse transforms per step (the reference does all of them)
            from _oracle import RefPhys
            per_step += 41 * per_tr
            if RefPhys.available():
                sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
                from make_physics_golden import HSG, TYEAR, gaussian_latitudes, physics_inputs
                inp = physics_inputs()
                ref = RefPhys(HSG, gaussian_latitudes())
                ref.set_surface(inp["phis0"], inp["alb_l"], inp["alb_s"], inp["albsfc"], inp["snowc"])
                ref.sol_oz(TYEAR)
                a = [inp[k] for k in ("ug", "vg", "tg", "qg", "phig", "pslg", "fmask", "phis0", "tland", "tsea", "swav")]
                tt = {}
                for sw in (True, False):
                    t1 = time.perf_counter()
                    for _ in range(3):
                        ref.phypar(*a, sw, inp["utend"], inp["vtend"], inp["ttend"], inp["qtend"])
                    tt[sw] = (time.perf_counter() - t1) / 3
                nsw = 2 + len([i for i in range(nst - 2) if (i + 1) % 3 == 1])
                total += tt[True] * nsw + tt[False] * (nst - nsw)
                phys_note = (f"; column physics with the compiled reference parametrisations (oracle/_ref, called through the phypar "
                             f"sequence): {tt[True] * 1e3:.1f} ms per short-wave step (x{nsw}), {tt[False] * 1e3:.1f} ms otherwise (x{nst - nsw})")
            else:
                phys_note = "; column physics NOT priced (oracle/_ref/libref_phy.so absent): the baseline is flattered"
        total += per_tr * 99 + per_step * nst
        total_par += total - per_predict * NREG                   # the SPEEDY leg is serial in the reference (root rank only)
        sample += (f"; SPEEDY leg: {m} grid+spec pairs with the {which}: {per_pair * 1e6:.0f} us per pair (incl. ctypes overhead); "
                   f"{ns} adiabatic time steps with the oracle: {per_step_oracle * 1e3:.2f} ms each, of which its 123 direct-DFT "
                   f"transforms are re-priced at the reference's transform cost -> {per_step * 1e3:.2f} ms per step, x{nst} steps "
                   f"+ 99 hand-off transforms per hybrid step; exchange tilers not timed (small)" + phys_note)
    sample += (f"; all-cores variant: {ncores} threads predicting concurrently reach {par_rate:.0f} predicts/s in aggregate "
               f"(W_in is streamed from DRAM), SPEEDY leg serial as in the reference")
    return {"value": 1.0 / total, "unit": "steps/s", "cores": 1, "kind": "port", "sample": sample,
            "all_cores": {"value": 1.0 / total_par, "unit": "steps/s", "cores": ncores}}


def training_block(with_cpu):
    """BASELINE config 4 on this GPU, reported beside the hybrid step (rank 0, N = 1): the fp64-MFMA Gram accumulation of chunking_matmul
    (src/mod_reservoir.f90:1645-1701) at the shipped batch size m = 98 and at the 40-year size m = 2920, the ridge solve of
    fit_chunk_hybrid / dgesv (:1235-1334, src/mod_linalg.f90:109-151) for the 5892 x 5892 system with 136 right-hand sides, one at a
    time and eight in one batch, and the device training pass (recurrence + Gram flushes) of eight resident full-size reservoirs.
    Rooflines are EXECUTED flops (only tiles on / below the diagonal of C are computed) over the 78.6 TFLOP/s fp64 MFMA spec."""
    import torch
    from speedy_ml_amd import train
    from speedy_ml_amd.reservoir import ReservoirBank
    from speedy_ml_amd.synth import make_reservoir
    PEAK = 78.6
    n, d, n_model, n_out = 5760, 576, 132, 136
    n_aug = n + n_model
    dev = "cuda"

    def timed(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    out = {"peak_tflops_fp64_mfma_spec": PEAK, "flop_convention": "executed (lower-triangle tiles of C + the skinny blocks)"}
    nt = (n + 127) // 128
    tiles = nt * (nt + 1) // 2
    c = b = None
    for m in (98, 2920):
        states = torch.randn((m, n), dtype=torch.float64, device=dev)
        model = torch.randn((m, n_model), dtype=torch.float64, device=dev)
        y = torch.randn((m, n_out), dtype=torch.float64, device=dev)
        c, b = train.fortran_zeros(n_aug, n_aug), train.fortran_zeros(n_out, n_aug)
        for _ in range(3):
            train.chunking_matmul(states, model, y, c, b)
        dt = timed(lambda: train.chunking_matmul(states, model, y, c, b), 20)
        executed = 2.0 * 128 * 128 * m * tiles + 2.0 * m * n_aug * (n_model + n_out)
        out[f"gram_m{m}"] = {"ms": dt * 1e3, "tflops": executed / dt / 1e12, "frac": executed / dt / 1e12 / PEAK,
                             "tflops_full_dgemm_convention": (2.0 * n_aug * n_aug * m + 2.0 * n_out * n_aug * m) / dt / 1e12}
    # the ridge solve on the Gram matrix just accumulated (6 batches of 2920 columns) + the reference's regularisation.
    # Algorithmic flops: SURVEY 8d's K9 count, what the reference's dgesv does: (2/3) n_aug^3 + 2 n_aug^2 n_out = 146 GFLOP.  The
    # default solver is a blocked Cholesky of the symmetric positive definite system (pivoted LU where a pivot is not positive):
    # it EXECUTES (1/3) n_aug^3 + 2 n_aug^2 n_out = 77.6 GFLOP.  `frac` is executed flops / time / peak (MFMA utilisation); the dgesv-equivalent
    # rate is reported as tflops_dgesv_equiv.
    from speedy_ml_amd import _lib
    L = _lib.lib()
    flops_lu = (2.0 / 3.0) * n_aug ** 3 + 2.0 * n_aug ** 2 * n_out
    flops_chol = (1.0 / 3.0) * n_aug ** 3 + 2.0 * n_aug ** 2 * n_out
    reg = torch.diag(torch.cat([torch.full((n_model,), 1.0), torch.full((n,), 1e-6)])).to(dev, torch.float64)

    def solve_block(executed):
        train.fit_chunk_hybrid(c, b, n, n_model, n_out)                                   # allocates the workspace
        each = sorted(timed(lambda: train.fit_chunk_hybrid(c, b, n, n_model, n_out), 1) for _ in range(9))
        dt1 = each[len(each) // 2]                                                       # median of 9 single solves (each synchronised)
        w = train.fit_chunk_hybrid(c, b, n, n_model, n_out)
        resid = (c + reg) @ w - b                                                      # column-major buffers: torch [n_aug, n_out] = Z; C symmetric
        berr = float(resid.norm() / (torch.linalg.matrix_norm(c + reg) * w.norm() + b.norm()))
        # frac = EXECUTED flops / time / peak: the matrix cores' utilisation.  The reference's dgesv would execute flops_lu on the same
        # system; the rate at which this solver gets through that count is reported beside it as tflops_dgesv_equiv (not a roofline)
        out = {"ms": dt1 * 1e3, "tflops": executed / dt1 / 1e12, "frac": executed / dt1 / 1e12 / PEAK,
               "tflops_dgesv_equiv": flops_lu / dt1 / 1e12, "normwise_backward_error": berr,
               "ms_min_max_of_9": [each[0] * 1e3, each[-1] * 1e3]}
        for nsys in (8, 16):
            cs = [c.clone() for _ in range(nsys)]
            train.fit_chunk_hybrid_batched(cs, [b] * nsys, n, n_model, n_out)           # grows the workspace
            dtn = timed(lambda: train.fit_chunk_hybrid_batched(cs, [b] * nsys, n, n_model, n_out), 2)
            out[f"batched{nsys}"] = {"ms_per_system": dtn * 1e3 / nsys, "tflops": nsys * executed / dtn / 1e12, "frac": nsys * executed / dtn / 1e12 / PEAK,
                                     "tflops_dgesv_equiv": nsys * flops_lu / dtn / 1e12}
            del cs
        return out, out["batched16"]["ms_per_system"] * 16e-3

    prev = L.sml_train_select_solver(0)
    out["ridge_solve_5892"], dt16 = solve_block(flops_chol)
    out["ridge_solve_5892"]["solver"] = "blocked Cholesky (k_chol_panel + lower-trapezoid MFMA updates); pivoted LU on a non-positive pivot"
    out["ridge_solve_5892"]["algorithmic_gflop"] = flops_lu / 1e9
    out["ridge_solve_5892"]["executed_gflop"] = flops_chol / 1e9
    L.sml_train_select_solver(1)                                                       # dgesv's algorithm alone, for comparison
    train.release_workspace()
    out["ridge_solve_5892_pivoted_lu"], _ = solve_block(flops_lu)
    L.sml_train_select_solver(prev)
    train.release_workspace()
    del reg
    # the training pass (reservoir_layer_chunking_hybrid, :1067-1175) of resident full-size reservoirs: one pass of the shipped
    # configuration = 20 batches of 98 columns (traininglength 12000 h / 6 passes / timestep 6)
    # 32 resident reservoirs: the recurrence's one launch per time column is a fixed cost shared by the residents (per reservoir and
    # batch 0.18 / 0.13 / 0.10 / 0.09 ms at 8 / 16 / 32 / 64, profiles/micro/train_pass_time.py; training.py trains in groups of 64)
    nres, batch, discard = 32, 98, 40
    nbatch = 20
    T = discard + nbatch * batch
    bank = ReservoirBank(nres)
    rs = [make_reservoir(n=n, d=d, n_model=n_model, n_out=n_out, seed=20240954 + i) for i in range(4)] * (nres // 4)   # (values do not matter for the timing)
    for i, r in enumerate(rs):
        bank.load(i, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)
    noisy = torch.randn((T, nres, 576), dtype=torch.float64, device=dev) * 0.5
    models = [torch.randn((T, n_model), dtype=torch.float64, device=dev) for _ in range(nres)]
    targs = [torch.randn((T, n_out), dtype=torch.float64, device=dev) for _ in range(nres)]
    cs8 = [train.fortran_zeros(n_aug, n_aug) for _ in range(nres)]
    bs8 = [train.fortran_zeros(n_out, n_aug) for _ in range(nres)]
    bank.train_pass(noisy, discard, batch, models, targs, cs8, bs8)
    dtp = timed(lambda: bank.train_pass(noisy, discard, batch, models, targs, cs8, bs8), 2)
    per_batch = dtp / (nbatch * nres)
    shipped_batches = 120                                                              # traininglength 12000 h, timestep 6, 20 batches per pass
    per_res = shipped_batches * per_batch + dt16 / 16
    out["train_pass"] = {"ms_per_reservoir_batch": per_batch * 1e3, "resident_reservoirs": nres, "batch_size": batch,
                         "reservoirs_per_s_shipped_config": 1.0 / per_res,
                         "note": "shipped config: 120 batches of m = 98 per reservoir (12000 h, 6 interleaved passes) + one ridge solve (16 systems in lockstep, Cholesky)"}
    bank.close()
    del noisy, models, targs, cs8, bs8
    torch.cuda.empty_cache()
    # BASELINE config 4 END TO END at the size it is quoted on (40 years = 350 640 hourly columns, six passes of 20 batches of m = 2920
    # per reservoir + the ridge solves): 16 resident full-size reservoirs through training.train_reservoirs_device, inputs generated on
    # the device (profiles/train_40yr.py; 64 residents reach 0.28 s per reservoir, profiles/r4_train_40yr_64_residents.json)
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from train_40yr import run as train_40yr
    out["train_pass_40yr"] = train_40yr(16, verbose=False, compare_m98=False)
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _oracle import Oracle
        o = Oracle()
        rng = np.random.default_rng(4)
        mm = 4
        st, md, yy = rng.standard_normal((n, mm)), rng.standard_normal((n_model, mm)), rng.standard_normal((n_out, mm))
        co, bo = np.zeros((n_aug, n_aug), order="F"), np.zeros((n_out, n_aug), order="F")
        t0 = time.perf_counter()
        o.chunking_matmul(st, md, yy, co, bo)
        t_gram = (time.perf_counter() - t0) * 98.0 / mm
        ns = 1068
        na = ns + n_model
        a = rng.standard_normal((na, 2 * na))
        cc = np.asfortranarray(a @ a.T)
        bb = np.asfortranarray(rng.standard_normal((n_out, na)))
        t0 = time.perf_counter()
        o.fit_chunk_hybrid(ns, n_model, n_out, 1e-3, 1.0, 0.0, True, cc, bb)
        t_fit = (time.perf_counter() - t0) * (n_aug / na) ** 3
        # ... and the library the reference itself calls for the solve: LAPACK dgesv (numpy.linalg.solve -> the BLAS numpy ships with),
        # full size, all the threads that BLAS takes: the oracle's unblocked C loop above is several times slower than any LAPACK
        t_lapack, lapack_threads = None, None
        try:
            aa = rng.standard_normal((n_aug, 2048))
            cc_full = aa @ aa.T + np.diag(np.r_[np.full(n_model, 1.0), np.full(n, 1e-6)])
            bb_full = rng.standard_normal((n_aug, n_out))
            t0 = time.perf_counter()
            np.linalg.solve(cc_full, bb_full)
            t_lapack = time.perf_counter() - t0
            try:
                from threadpoolctl import threadpool_info
                lapack_threads = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
            except Exception:
                lapack_threads = None
            del aa, cc_full, bb_full
        except Exception:
            pass
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "gram_m98_s": t_gram, "ridge_solve_5892_s": t_fit,
                               "ridge_solve_5892_lapack_dgesv_s": t_lapack, "lapack_threads": lapack_threads,
                               "reservoirs_per_s_shipped_config": 1.0 / (shipped_batches * t_gram + t_fit),
                               "reservoirs_per_s_shipped_config_with_lapack_solve": (1.0 / (shipped_batches * t_gram + t_lapack)) if t_lapack else None,
                               "sample": f"oracle ro_chunking_matmul at full n with {mm} columns scaled to 98 (linear in m); oracle "
                                         f"ro_fit_chunk_hybrid (unblocked LU) at n_aug = {na} scaled by (5892/{na})^3; recurrence not priced"}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import _lib, domain, hybrid, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        # launch + rendezvous only, no GPU: every rank joins a gloo group and rank 0 prints who arrived
        if world > 1:
            dist.init_process_group("gloo")
            seen = [None] * world
            dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank})
            dist.barrier()
            dist.destroy_process_group()
        else:
            seen = [{"rank": rank, "local_rank": local_rank}]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": seen}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU over RCCL ("nccl" IS RCCL on ROCm).  SML_DIST_BACKEND=gloo lets several ranks share one GPU so that
    # the N>1 code path can be rehearsed on a 1-GPU box (the slab is then staged through the host); never used for numbers.
    backend = os.environ.get("SML_DIST_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    _lib.check(_lib.lib().sml_set_device(device_index))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = [int(r) for r in domain.processor_decomposition_manual(rank, world, hybrid.NREG)]
    if args.regions < len(regions):
        regions = regions[:args.regions]
    t0 = time.time()
    # SML_PIPELINE=1 selects the software-pipelined schedule (readout state block under the SPEEDY window, DESIGN.md
    # "Software pipeline"): measured +6 % only, because the latency-bound SPEEDY kernels slow down 2x next to an
    # HBM-saturating stream; the default is the reference's sequential order.
    model = hybrid.HybridRank(regions, classes, world=world, rank=rank, sea_mask=sea, mode=args.mode,
                              pipeline=os.environ.get("SML_PIPELINE", "0") == "1", slab=args.slab, physics=not args.no_physics,
                              speedy_cus=int(os.environ.get("SML_SPEEDY_CUS", "0")),
                              persistent_readout=os.environ.get("SML_PERSISTENT_READOUT", "1") == "1", float32_weights=args.float32_weights)
    model.stop_on_unsafe = not (world == 1 and args.regions != 1152)       # (--regions emulates one rank's load: its grid is not physical)
    if rank == 0:
        print(f"[bench] rank0 loaded {len(regions)} reservoirs in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    # The timed host: the native engine (what a Fortran host ships with) unless --host python or a schedule only the Python host has
    python_only = args.mode != "hybrid" or model.pipeline
    if args.host == "native" and python_only and rank == 0:
        print("[bench] --mode sweep / ml_only and SML_PIPELINE=1 exist in the Python host only: timing HybridRank.step", file=sys.stderr, flush=True)
    comm = None
    host = model
    if args.host == "native" and not python_only:
        # the rank exchange: torch.distributed's all-gather between the engine's two half-steps by default; SML_ENGINE_COMM=engine (or
        # the slab coupling, whose second all-gather lives in the engine) gives the engine its own RCCL communicator, as under a Fortran host
        if world > 1 and (os.environ.get("SML_ENGINE_COMM", "torch") == "engine" or args.slab):
            comm = hybrid.make_comm(world, rank)
        host = hybrid.NativeEngine(model, comm=comm)

    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        host.step(stream)
    barrier()
    # Inside the timed region only k_readout carries HIP events (the roofline is its measured duration); k_update's pair and the
    # per-phase split of the step need eight more event records per step on the step's stream, each a barrier packet in a chain of
    # dependent 5-10 us launches, so they are measured on a pass of its own right after the timed one (SML_BENCH_PHASES_INLINE=1: inside).
    phases_inline = os.environ.get("SML_BENCH_PHASES_INLINE", "0") == "1"
    host.timing(True, phases=phases_inline)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        host.step(stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    kern = host.timing_collect()
    if not phases_inline:
        host.timing(True, phases=True)
        for _ in range(min(args.steps, 20)):
            host.step(stream)
        barrier()
        after = host.timing_collect()
        kern["phases_ms_per_step"] = after.get("phases_ms_per_step", {})
        kern["update_ms"], kern["update_launches"] = after["update_ms"], after["update_launches"]      # (k_update carries no events inside the timed region)
    host.timing(False)
    emulated_rank = world == 1 and args.regions != 1152          # (--regions: one rank's load without its peers' outvecs -- the grid is not physical)
    if host.aborted(wait=True) and not emulated_rank:
        raise SystemExit("bench.py: the range guard of iogrid(30) tripped -- the forecast loop stopped (src/mpires.f90:744); no number")
    # every rank's view of the step: kernel and phase times (an N-GPU line is read through these: the SPEEDY leg is replicated)
    per_rank = {"rank": rank, "regions": len(regions),
                "update_ms": kern["update_ms"] / max(kern["update_launches"], 1), "readout_ms": kern["readout_ms"] / max(kern["readout_launches"], 1)}
    for k, v in kern.get("phases_ms_per_step", {}).items():
        per_rank[{"speedy": "speedy_ms", "allgather": "allgather_ms", "predict": "predict_ms", "scatter": "scatter_ms", "gather": "gather_ms"}[k]] = v
    ranks = [per_rank]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, per_rank)

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        # HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md): collected by
        # profiles/collect.sh with the same bank at N=1, so it is only quoted when this run is N=1 with all 1152 regions.
        traffic, traffic_from = None, None
        try:
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[-1]
            traffic = json.load(open(latest))["kernels"]["k_readout"].get("hbm_traffic_bytes_per_launch")
            traffic_from = os.path.relpath(latest, ROOT)
            if world != 1 or len(regions) != hybrid.NREG:
                traffic = traffic_from = None
        except Exception:
            traffic = traffic_from = None
        ms = elapsed / args.steps * 1e3
        upd_b, ro_b = model.bank.algorithmic_bytes()
        ro_ms = kern["readout_ms"] / max(kern["readout_launches"], 1)
        upd_ms = kern["update_ms"] / max(kern["update_launches"], 1)
        achieved = ro_b / (ro_ms * 1e-3) / 1e9 if ro_ms > 0 else 0.0
        line = {
            "metric": "hybrid forecast steps/sec at T30L8, 1152 N_res=6000 reservoirs",
            "value": args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": host.describe(),
            "per_rank": ranks,
            "per_rank_note": "readout_ms: HIP events inside the timed region; update_ms and the phase split (predict / allgather / scatter / speedy / gather) "
                             + ("inside it as well" if phases_inline else f"from a pass of {min(args.steps, 20)} steps right after it (eight more event records per step)"),
            "roofline": {"bound": "hbm", "kernel": "k_readout<4,512> (W_out [local_model;x~] GEMV, all resident reservoirs)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic,
                         "traffic_source": (f"{traffic_from}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/collect.sh), "
                                            "a committed measurement, not taken in this run") if traffic_from else None,
                         "algorithmic_bytes_per_launch": ro_b, "avg_launch_ms": ro_ms,
                         "secondary": {"kernel": "k_update (SELL-64 [A|Win]x + tanh)", "achieved":
                                       (upd_b / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0), "unit": "GB/s",
                                       "algorithmic_bytes_per_launch": upd_b, "avg_launch_ms": upd_ms}},
        }
        compact = model.bank.compact()
        if compact:
            line["roofline"]["kernel"] = "k_readout32<4,512> (the same GEMV from the 4-byte copy of W_out: every weight is exactly a float)"
            line["roofline"]["traffic"] = line["roofline"]["traffic_source"] = None
            line["config"]["weights"] = "rounded to float32 (as read from the reference's NetCDF weight files): compact copies in HBM, fp64 arithmetic"
        elif world == 1 and args.mode == "hybrid" and host is not model and len(regions) == hybrid.NREG and not args.no_float32_block:
            # The same step with weights as a reservoir read from the reference's weight files holds them (NF90_REAL: exactly floats).
            # A second model; reported beside the headline, which stays on arbitrary doubles.
            m32 = hybrid.HybridRank(regions, classes, world=1, rank=0, sea_mask=sea, mode="hybrid", slab=args.slab, physics=not args.no_physics,
                                    float32_weights=True)
            h32 = hybrid.NativeEngine(m32)
            n32 = min(args.steps, 60)
            for _ in range(args.warmup):
                h32.step(stream)
            barrier()
            h32.timing(True, phases=False)
            t32 = time.perf_counter()
            for _ in range(n32):
                h32.step(stream)
            barrier()
            e32 = time.perf_counter() - t32
            k32 = h32.timing_collect()
            h32.timing(True, phases=True)
            for _ in range(min(n32, 20)):
                h32.step(stream)
            barrier()
            a32 = h32.timing_collect()
            h32.timing(False)
            u32b, r32b = m32.bank.algorithmic_bytes()
            r32ms = k32["readout_ms"] / max(k32["readout_launches"], 1)
            u32ms = a32["update_ms"] / max(a32["update_launches"], 1)
            line["float32_weight_files"] = {
                "what": "the same hybrid step with every weight rounded to float32 -- what read_trained_res leaves after the reference's NetCDF "
                        "weight files (NF90_REAL, src/mod_io.f90 via src/mod_reservoir.f90:1727-1736); the banks detect it and read 4-byte copies, "
                        "arithmetic stays fp64 on the same numbers (tests/test_reservoir_gpu.py::test_compact_storage_of_float_weights)",
                "compact": bool(m32.bank.compact()), "value": n32 / e32, "unit": "steps/s", "ms_per_step": e32 / n32 * 1e3, "steps": n32,
                "readout_ms": r32ms, "update_ms": u32ms, "phases_ms_per_step": a32.get("phases_ms_per_step", {}),
                "roofline": {"bound": "hbm", "kernel": "k_readout32<4,512>", "algorithmic_bytes_per_launch": r32b,
                             "achieved": r32b / (r32ms * 1e-3) / 1e9 if r32ms > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                             "frac": (r32b / (r32ms * 1e-3) / 1e9 / 8000.0) if r32ms > 0 else 0.0,
                             "secondary": {"kernel": "k_update<512, float values>", "algorithmic_bytes_per_launch": u32b,
                                           "achieved": u32b / (u32ms * 1e-3) / 1e9 if u32ms > 0 else 0.0, "unit": "GB/s"}}}
            h32.close()
            del h32, m32
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:        # the CPU baseline is measured at N = 1 only (the other ranks would idle)
            line["cpu_baseline"] = cpu_baseline(model)
        if not args.no_training and world == 1 and args.mode == "hybrid":
            if host is not model:
                host.close()
            del host, model
            torch.cuda.empty_cache()
            line["training"] = training_block(not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if comm is not None:
        _lib.check(_lib.lib().sml_comm_destroy(comm))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
