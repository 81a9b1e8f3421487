"""BASELINE config 4 at the size it is quoted on, end to end on one MI355X: ridge-regression training of resident full-size
reservoirs (n = 5760, d = 576, 132 model + 136 target rows) on 40 years of hourly ERA5-shaped synthetic data -- traininglength
350 640 h, timestep 6 -> six interleaved passes of 58 440 columns, 40 discarded, 20 batches of m = 2920 each
(src/mod_reservoir.f90:289-301, 1067-1175, 1561-1592, 1645-1701), then fit_chunk_hybrid (:1235-1334).

Everything runs through speedy_ml_amd.training.train_reservoirs_device: the AR(1) inputs (phi = 0.98 / h) are generated
on the device, each pass's noise and imperfect model too; the recurrence is one k_update launch per time column shared by the
residents, every batch one fp64-MFMA Gram update per reservoir, the ridge systems are solved in lockstep.

    python profiles/train_40yr.py [residents] [out.json]

Reports seconds per reservoir, the executed TFLOP/s over the whole pass (recurrence included), the projected wall time for all 1152
reservoirs on 1 and 8 GPUs (training has no collective: 144 reservoirs per GPU), the ridge systems' backward error, and the one-step
prediction error of the trained readout on a stretch of the series against (i) the imperfect model it was given and (ii) a readout
trained through the shipped m = 98 path on the first 12 000 hours of the same series."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

TRAINLEN, DISCARD, TIMESTEP = 350640, 240, 6
N, D, N_MODEL, N_OUT = 5760, 576, 132, 136
PEAK = 78.6


def one_step_error(bank, slots, hourly, models_at, rows_t, t0, steps, timestep):
    """teacher-forced one-step predictions on columns t0, t0 + timestep, ...: mean squared error of outvec against the next clean
    target, and of the imperfect model against it.  bank's W_out as installed by the training."""
    import torch
    from speedy_ml_amd._lib import device_view
    cap = bank.capacity
    fb = device_view(bank.feedback_ptr, (cap, bank.max_d))
    lm = device_view(bank.local_model_ptr, (cap, bank.max_n_model))
    ov = device_view(bank.outvec_ptr, (cap, bank.max_n_out))
    rows = torch.as_tensor(np.asarray(rows_t), dtype=torch.long, device="cuda")
    for s in slots:
        bank.set_state(s, np.zeros(N))
    err_h = err_m = 0.0
    cnt = 0
    for k in range(steps):
        t = t0 + k * timestep
        fb.copy_(hourly[t])
        truth_next = hourly[t + timestep][:, rows]                                   # [cap, n_out]
        model_next = models_at(t + timestep, truth_next)
        lm.copy_(model_next)
        bank.predict(raw=True)
        if k >= 60:                                                                  # (spun up)
            err_h += float(((ov[slots] - truth_next[slots]) ** 2).mean())
            err_m += float(((model_next[slots] - truth_next[slots][:, :N_MODEL]) ** 2).mean())
            cnt += 1
    return err_h / cnt, err_m / cnt


def run(residents=16, verbose=True, compare_m98=True):
    import torch
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import domain, synth, train, training
    from speedy_ml_amd.reservoir import ReservoirBank
    rows_t = domain.target_map(1152, 954)
    say = (lambda *a: print(*a, file=sys.stderr, flush=True)) if verbose else (lambda *a: None)
    base = [synth.make_reservoir(n=N, d=D, n_model=N_MODEL, n_out=N_OUT, seed=20240954 + i) for i in range(min(4, residents))]

    def make_bank():
        bank = ReservoirBank(residents, max_d=D, max_n_model=N_MODEL, max_n_out=N_OUT)
        for s in range(residents):
            r = base[s % len(base)]
            bank.load(s, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, np.zeros((N_OUT, N + N_MODEL)), r.mean, r.std, None)
        return bank
    specs = [dict(n=N, n_model=N_MODEL, n_out=N_OUT, target_rows=rows_t) for _ in range(residents)]
    t0 = time.perf_counter()
    hourly = synth.ar1_series_device(TRAINLEN, residents * D, phi=0.98, seed=7).reshape(TRAINLEN, residents, D)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    say(f"[train_40yr] {TRAINLEN} hourly columns x {residents} reservoirs x {D} inputs generated on the device in {t_gen:.2f} s "
        f"({hourly.numel() * 8 / 1e9:.1f} GB)")
    batch = training.chunk_batch_size(TRAINLEN, DISCARD, TIMESTEP)
    assert batch == 2920
    bank = make_bank()
    marks = [time.perf_counter()]

    def progress(i, nb):
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
        say(f"[train_40yr] pass {i + 1}/6: {nb} batches of m = {batch} in {marks[-1] - marks[-2]:.2f} s")
    torch.cuda.synchronize()
    marks[0] = time.perf_counter()
    res = training.train_reservoirs_device(bank, specs, hourly, TRAINLEN, DISCARD, TIMESTEP, noisemag=0.2, model_sigma=0.3, seed=11, keep_gram=True,
                                           progress=progress)
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    t_pass = marks[-1] - marks[0]
    t_fit = t_end - marks[-1]
    total = t_end - marks[0]
    nt = (N + 127) // 128
    n_aug = N + N_MODEL
    executed = 2.0 * 128 * 128 * batch * (nt * (nt + 1) // 2) + 2.0 * batch * n_aug * (N_MODEL + N_OUT)        # per reservoir-batch, as bench.py's gram block
    nbatches = res[0]["batches"] * TIMESTEP
    assert nbatches == 120 and res[0]["batch_size"] == batch
    # ridge systems: normwise backward error on the accumulated Gram matrices
    berr = []
    reg = torch.cat([torch.full((N_MODEL,), 1.0), torch.full((N,), 1e-6)]).to("cuda", torch.float64)
    for s in range(min(residents, 4)):
        c, b, w = res[s]["c"], res[s]["b"], res[s]["wout_dev"]
        a = c + torch.diag(reg)
        berr.append(float(((a @ w - b).norm() / (torch.linalg.matrix_norm(a) * w.norm() + b.norm()))))
        del a
    for s in range(residents):
        res[s].pop("c"); res[s].pop("b"); res[s].pop("wout_dev")
    torch.cuda.empty_cache()
    out = {"residents": residents, "traininglength_h": TRAINLEN, "timestep": TIMESTEP, "passes": TIMESTEP, "batch_size": batch, "batches_per_reservoir": nbatches,
           "data_generation_s": t_gen, "passes_s": t_pass, "ridge_solves_s": t_fit, "total_s": total, "s_per_reservoir": total / residents,
           "ms_per_reservoir_batch": t_pass / (nbatches * residents) * 1e3,
           "tflops_executed_whole_pass": residents * nbatches * executed / t_pass / 1e12,
           "frac_of_fp64_mfma_peak_whole_pass": residents * nbatches * executed / t_pass / 1e12 / PEAK,
           "projected_1152_reservoirs_s": {"1_gpu": 1152 / residents * total, "8_gpus": 144 / residents * total},
           "ridge_normwise_backward_error_max": max(berr),
           "note": "recurrence (one k_update launch per time column, shared by the residents) + one Gram update per reservoir and batch + "
                   "ridge solves in lockstep; data generation not included in the projection (the reference reads ERA5 from disk)"}
    assert max(berr) < 1e-15, berr
    # one-step prediction on a stretch of the series (teacher-forced), against the imperfect model and the m = 98 readout
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)

    def models_at(t, truth_next):
        return (truth_next[:, :N_MODEL] + 0.3 * torch.randn((residents, N_MODEL), dtype=torch.float64, device="cuda", generator=gen)).contiguous()
    slots = list(range(min(residents, 4)))
    e40, em = one_step_error(bank, slots, hourly, models_at, rows_t, 20000, 260, TIMESTEP)
    out["one_step_mse"] = {"readout_40yr": e40, "imperfect_model": em}
    assert e40 < em, (e40, em)
    if compare_m98:
        bank98 = make_bank()
        short = hourly[:12000].contiguous()
        training.train_reservoirs_device(bank98, [specs[s] if s in slots else None for s in range(residents)], short, 12000, DISCARD, TIMESTEP, noisemag=0.2,
                                         model_sigma=0.3, seed=11)
        gen.manual_seed(99)
        e98, _ = one_step_error(bank98, slots, hourly, models_at, rows_t, 20000, 260, TIMESTEP)
        out["one_step_mse"]["readout_12000h_m98"] = e98
        # more data, same model class: the 40-year readout predicts the held-out stretch at least as well as the 12 000-hour one
        assert e40 <= 1.02 * e98, (e40, e98)
        bank98.close()
    bank.close()
    del hourly
    torch.cuda.empty_cache()
    train.release_workspace()
    return out


if __name__ == "__main__":
    residents = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    r = run(residents)
    print(json.dumps(r, indent=1))
    if len(sys.argv) > 2:
        json.dump(r, open(sys.argv[2], "w"), indent=1)
