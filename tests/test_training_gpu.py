"""GPU: train_reservoir end to end (src/mod_reservoir.f90:214-320) for a region-954 reservoir through the device path --
initialize_chunk_training's batch size, the `timestep` interleaved passes of reservoir_layer_chunking_hybrid, chunking_matmul's
targets (tile_full_input_to_target_data), fit_chunk_hybrid, write_trained_res / read back -- against the same sequence issued through
the CPU oracle with identical noise realisations.

Tolerances: accumulated C, B 1e-12 of their max (GEMM summation order); W_out through its ridge system's residual 1e-9 and directly
1e-6 of max|W_out| (the Gram matrix of a driven reservoir is ill-conditioned; beta_res^2 = 1e-6 regularises it)."""
import numpy as np
import pytest
import torch

from speedy_ml_amd import domain, training, weights
from speedy_ml_amd.reservoir import ReservoirBank
from speedy_ml_amd.synth import make_reservoir

pytestmark = pytest.mark.gpu
NREG, REGION = 1152, 954
TRAINLEN, DISCARD, TIMESTEP = 1236, 36, 6


def synthetic_series(d, n_model, rows_target, rng):
    """standardised AR(1) inputs with a slow oscillation; the imperfect model forecasts the targets with a bias and noise"""
    L = TRAINLEN
    t = np.arange(L)
    base = 0.6 * np.sin(2 * np.pi * t[None, :] / 96.0 + rng.uniform(0, 6.28, (d, 1)))
    ar = np.zeros((d, L))
    e = rng.standard_normal((d, L)) * 0.25
    for k in range(1, L):
        ar[:, k] = 0.95 * ar[:, k - 1] + e[:, k]
    truth = base + ar
    model = truth[rows_target[:n_model], :] * 0.9 + 0.1 + 0.2 * rng.standard_normal((n_model, L))
    return truth, model


def test_train_reservoir_end_to_end(oracle, tmp_path):
    g = oracle.initializedomain(NREG, REGION)
    s = oracle.allocate_sizes(g)
    d, n_out, n_model = s.reservoir_numinputs, s.chunk_size_prediction, s.chunk_size_speedy
    assert (d, n_out, n_model) == (576, 136, 132)
    n = d                                                    # one node per input keeps the oracle's dense products cheap
    r = make_reservoir(n=n, d=d, n_model=n_model, n_out=n_out, seed=31)
    rng = np.random.default_rng(31)
    r.win = training.make_win(n, d, 0.5, rng.random((d, n // d)))
    rows_t = domain.target_map(NREG, REGION)
    truth, model = synthetic_series(d, n_model, rows_t, rng)
    noisemag = 0.2
    gauss = rng.standard_normal(truth.shape)
    mean35, std35 = 0.1, 1.3
    noisy = training.add_input_noise(truth, gauss, noisemag, precip_slice=(s.precip_start - 1, s.precip_end), precip_mean=mean35,
                                     precip_std=std35, precip_epsilon=0.001)
    # the noise follows the reference: relative to the value, precipitation perturbed in physical space
    assert np.allclose(noisy[:s.precip_start - 1], truth[:s.precip_start - 1] * (1 + noisemag * gauss[:s.precip_start - 1]))
    p = slice(s.precip_start - 1, s.precip_end)
    phys = 0.001 * (np.exp(truth[p] * std35 + mean35) - 1)
    assert np.allclose(noisy[p], (np.log(1 + np.abs(phys + gauss[p] * noisemag * phys) / 0.001) - mean35) / std35)

    batch = training.chunk_batch_size(TRAINLEN, DISCARD, TIMESTEP)
    assert batch == oracle.find_closest_divisor((TRAINLEN - DISCARD) // (20 * TIMESTEP), (TRAINLEN - DISCARD) // TIMESTEP) == 10

    bank = ReservoirBank(2, max_d=d, max_n_model=n_model, max_n_out=n_out)
    stat = np.full(n_out, -1, dtype=np.int32)
    bank.load(1, n, d, n_model, n_out, r.rows, r.cols, r.vals, r.win, np.zeros((n_out, n + n_model)), r.mean, r.std, stat)
    spec = dict(n=n, n_model=n_model, n_out=n_out, trainingdata=noisy, clean=truth, imperfect_model=model, target_rows=rows_t)
    res = training.train_reservoirs(bank, [None, spec], TRAINLEN, DISCARD, TIMESTEP, beta_res=1e-3, beta_model=1.0, prior_val=0.0)[1]
    assert res["batch_size"] == 10 and res["batches"] == 20

    # ---- the same through the oracle: six passes of reservoir_layer_chunking_hybrid, chunking_matmul with the oracle's tiler ----
    n_aug = n + n_model
    co, bo = np.zeros((n_aug, n_aug), order="F"), np.zeros((n_out, n_aug), order="F")
    for i in range(TIMESTEP):
        td, cl, md = noisy[:, i::TIMESTEP], truth[:, i::TIMESTEP], model[:, i::TIMESTEP]
        targ = oracle.tile_target(g, s, cl, n_out)
        nb = oracle.train_states(n, d, r.rows, r.cols, r.vals, r.win, 1.0, np.asfortranarray(td), DISCARD // TIMESTEP, batch,
                                 np.asfortranarray(md), targ, co, bo)
        assert nb == 20
    info, wo = oracle.fit_chunk_hybrid(n, n_model, n_out, 1e-3, 1.0, 0.0, True, co.copy(order="F"), bo.copy(order="F"))
    assert info == 0
    wg = res["wout"]
    assert wg.shape == (n_out, n_aug)
    scale = np.max(np.abs(wo))
    assert np.max(np.abs(wg - wo)) <= 1e-6 * scale, np.max(np.abs(wg - wo)) / scale
    # the fit solves (C + reg) W^T = (B + prior)^T: residual of the device W_out in the oracle's system
    low = np.tril(co)
    cs = low + np.tril(co, -1).T
    reg = np.concatenate([np.full(n_model, 1.0 ** 2), np.full(n, 1e-3 ** 2)])
    lhs = wg @ (cs + np.diag(reg))
    assert np.max(np.abs(lhs - bo)) <= 1e-9 * np.max(np.abs(bo))

    # ---- the trained slot itself predicts with the W_out the fit installed (src/mod_reservoir.f90:1312-1330: predict uses the
    # fit's wout): sml_bank_set_wout must lay it out in the device's state order ----
    xs = rng.standard_normal(n) * 0.2
    us, ms = np.ascontiguousarray(truth[:, 11]), np.ascontiguousarray(model[:, 12])
    bank.set_state(1, xs)
    bank.set_feedback(1, us)
    bank.set_local_model(1, ms)
    bank.predict()
    torch.cuda.synchronize()
    xw, ow = oracle.predict_raw(n, d, n_model, n_out, r.rows, r.cols, r.vals, r.win, wg, 1.0, us, ms, xs)
    assert np.max(np.abs(bank.get_state(1) - xw)) <= 1e-13
    assert np.max(np.abs(bank.get_outvec(1) - ow)) <= 1e-11 * np.max(np.abs(ow)), np.max(np.abs(bank.get_outvec(1) - ow)) / np.max(np.abs(ow))

    # ---- what was learnt: on the training columns the hybrid readout beats the imperfect model it was given ----
    cl = truth[:, 0::TIMESTEP]
    md = model[:, 0::TIMESTEP]
    x = np.zeros(n)
    err_h, err_m, cnt = 0.0, 0.0, 0
    for k in range(cl.shape[1] - 1):
        x, out = oracle.predict_raw(n, d, n_model, n_out, r.rows, r.cols, r.vals, r.win, wg, 1.0, np.ascontiguousarray(cl[:, k]),
                                    np.ascontiguousarray(md[:, k + 1]), x)
        if k >= DISCARD // TIMESTEP:
            tgt = cl[rows_t, k + 1]
            err_h += np.mean((out - tgt) ** 2)
            err_m += np.mean((md[:, k + 1] - tgt[:n_model]) ** 2)
            cnt += 1
    assert err_h < 0.8 * err_m, (err_h / cnt, err_m / cnt)

    # ---- write_trained_res -> read_trained_res -> bank: the device predicts with the float32-rounded trained weights ----
    path = str(tmp_path / weights.trained_res_filename(REGION, "e2e"))
    weights.write_trained_res(path, r.win, wg, r.rows, r.cols, r.vals, r.mean, r.std)
    w = weights.load_trained_res(bank, 0, path, n_model, stat)
    x0 = rng.standard_normal(n) * 0.2
    bank.set_state(0, x0)
    u7, m8 = np.ascontiguousarray(truth[:, 7]), np.ascontiguousarray(model[:, 8])
    bank.set_feedback(0, u7)
    bank.set_local_model(0, m8)
    bank.predict()
    torch.cuda.synchronize()
    x1, out = oracle.predict_raw(n, d, n_model, n_out, w["rows"], w["cols"], w["vals"], w["win"], w["wout"], 1.0, u7, m8, x0)
    assert np.max(np.abs(bank.get_state(0) - x1)) <= 1e-13
    assert np.max(np.abs(bank.get_outvec(0) - out)) <= 1e-11 * np.max(np.abs(out))


def test_device_resident_training_equals_the_host_fed_path():
    """train_reservoirs_device (the data resident on the device: the form BASELINE config 4 needs at 350 640 hourly columns) against
    train_reservoirs fed from host arrays with the SAME noise and imperfect-model numbers: same kernels, same order -> W_out bit for bit.
    Two slots of different size classes, one empty slot between them; the series comes from the device AR(1) generator."""
    from speedy_ml_amd import synth
    L, disc, ts = 1236, 36, 6
    d, n_model, n_out = 576, 132, 136
    rows_t = domain.target_map(NREG, REGION)
    sizes = {0: 576, 2: 1152}
    bank_d = ReservoirBank(3, max_d=d, max_n_model=n_model, max_n_out=n_out)
    bank_h = ReservoirBank(3, max_d=d, max_n_model=n_model, max_n_out=n_out)
    stat = np.full(n_out, -1, dtype=np.int32)
    specs = [None, None, None]
    for slot, n in sizes.items():
        r = make_reservoir(n=n, d=d, n_model=n_model, n_out=n_out, seed=40 + slot)
        for b in (bank_d, bank_h):
            b.load(slot, n, d, n_model, n_out, r.rows, r.cols, r.vals, r.win, np.zeros((n_out, n + n_model)), r.mean, r.std, stat)
        specs[slot] = dict(n=n, n_model=n_model, n_out=n_out, target_rows=rows_t)
    hourly = synth.ar1_series_device(L, 3 * d, phi=0.95, seed=3, chunk=256).reshape(L, 3, d).contiguous()
    # stationary, unit variance, the right memory
    h = hourly.reshape(L, -1)
    assert abs(float(h.var()) - 1.0) < 0.1 and abs(float((h[1:] * h[:-1]).mean() / h.var()) - 0.95) < 0.02
    drawn = {}

    def noisy_of_pass(i, clean):
        g = torch.Generator(device="cuda")
        g.manual_seed(100 + i)
        drawn[("noisy", i)] = clean + torch.randn(clean.shape, dtype=torch.float64, device="cuda", generator=g) * 0.2 * clean
        return drawn[("noisy", i)]

    def model_of_pass(i, slot, truth):
        g = torch.Generator(device="cuda")
        g.manual_seed(1000 + 10 * i + slot)
        drawn[("model", i, slot)] = (truth[:, :n_model] + 0.3 * torch.randn((truth.shape[0], n_model), dtype=torch.float64, device="cuda", generator=g)).contiguous()
        return drawn[("model", i, slot)]

    res_d = training.train_reservoirs_device(bank_d, specs, hourly, L, disc, ts, noisy_of_pass=noisy_of_pass, model_of_pass=model_of_pass, keep_gram=True)
    host = hourly.cpu().numpy()
    hspecs = [None, None, None]
    for slot in sizes:
        noisy = np.zeros((d, L))
        model = np.zeros((n_model, L))
        for i in range(ts):
            noisy[:, i::ts] = drawn[("noisy", i)][:, slot, :].cpu().numpy().T
            model[:, i::ts] = drawn[("model", i, slot)].cpu().numpy().T
        hspecs[slot] = dict(specs[slot], trainingdata=noisy, clean=np.ascontiguousarray(host[:, slot, :].T), imperfect_model=model)
    res_h = training.train_reservoirs(bank_h, hspecs, L, disc, ts)
    for slot in sizes:
        assert res_d[slot]["batch_size"] == res_h[slot]["batch_size"] == 10 and res_d[slot]["batches"] == 20
        assert np.array_equal(res_d[slot]["wout"], res_h[slot]["wout"]), slot
        # the ridge system it solves: normwise backward error of the device W_out on the device's own Gram matrix
        c, b, w = res_d[slot]["c"], res_d[slot]["b"], res_d[slot]["wout_dev"]
        n = sizes[slot]
        reg = torch.diag(torch.cat([torch.full((n_model,), 1.0), torch.full((n,), 1e-6)])).to("cuda", torch.float64)
        resid = (c + reg) @ w - b
        berr = float(resid.norm() / (torch.linalg.matrix_norm(c + reg) * w.norm() + b.norm()))
        assert berr < 1e-15, berr
