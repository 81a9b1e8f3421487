"""train_reservoir (src/mod_reservoir.f90:214-320) for the reservoirs resident in a bank: the orchestration around the device kernels.

    gen_res / W_in                         -> done by the caller (speedy_ml_amd.reservoir.gen_res, make_win) and loaded into the bank
    initialize_chunk_training (:1561-1592) -> chunk_batch_size(): 20 batches, find_closest_divisor
    do i = 1, timestep                     -> `timestep` interleaved passes over the columns i, i+timestep, ... of the hourly data
      reservoir_layer_chunking_hybrid (:1067-1175)  -> ReservoirBank.train_pass (recurrence + MFMA Gram updates on the device)
        gaussian_noise_1d_function[_precip] (src/mod_utilities.f90:1387-1462) -> add_input_noise() (the normal deviates are an input:
                                              the reference's RANDOM_NUMBER stream cannot be reproduced)
        chunking_matmul (:1645-1701): targets = tile_full_input_to_target_data(trainingdata) -> domain.target_map rows
    fit_chunk_hybrid (:1235-1334)          -> train.fit_chunk_hybrid_batched (LU on the device, size classes solved together)
    write_trained_res (:1703-1737)         -> weights.write_trained_res

All reservoirs of the bank advance together (one launch per time column); passes and batches follow the reference's index
arithmetic, so the accumulated matrices equal the reference's up to the order of the floating-point sums inside the GEMM."""
import numpy as np

from . import domain, train

NUM_OF_BATCHES = 20          # src/mod_reservoir.f90:1570


def chunk_batch_size(traininglength, discardlength, timestep):
    """initialize_chunk_training (src/mod_reservoir.f90:1570-1576)"""
    approx = (traininglength - discardlength) // (NUM_OF_BATCHES * timestep)
    return domain.find_closest_divisor(approx, (traininglength - discardlength) // timestep)


def make_win(n, numinputs, sigma, uniform):
    """W_in as train_reservoir builds it (src/mod_reservoir.f90:262-283): q = n / numinputs nodes per input, column i holds
    sigma * (-1 + 2 * rand) in rows (i-1) q + 1 .. i q.  uniform: (numinputs, q) U(0,1) numbers (row i = the i-th random_number call)."""
    q = n // numinputs
    uniform = np.asarray(uniform, dtype=np.float64)
    assert uniform.shape == (numinputs, q)
    win = np.zeros((n, numinputs), order="F")
    for i in range(numinputs):
        win[i * q:(i + 1) * q, i] = sigma * (-1.0 + 2.0 * uniform[i])
    return win


def add_input_noise(inputdata, gauss, noisemag, precip_slice=None, precip_mean=0.0, precip_std=1.0, precip_epsilon=0.001):
    """gaussian_noise_1d_function (src/mod_utilities.f90:1387-1409): x + g * noisemag * x, column by column; with precip_slice =
    (precip_start - 1, precip_end) the precipitation entries get their noise in physical space as gaussian_noise_1d_function_precip
    does (:1411-1462): un-standardise, invert log(1 + p / eps), perturb, abs, transform back.  inputdata, gauss: (d, T)."""
    x, g = np.asarray(inputdata, dtype=np.float64), np.asarray(gauss, dtype=np.float64)
    out = x + g * noisemag * x
    if precip_slice is not None:
        a, b = precip_slice
        t = x[a:b] * precip_std + precip_mean
        t = precip_epsilon * (np.e ** t - 1.0)
        t = t + g[a:b] * noisemag * t
        t = np.abs(t)
        t = np.log(1.0 + t / precip_epsilon)
        t = t - precip_mean
        out[a:b] = t / precip_std
    return out


def train_reservoirs(bank, specs, traininglength, discardlength, timestep, beta_res=0.001, beta_model=1.0, prior_val=0.0,
                     using_prior=True, ml_only=False, stream=None):
    """specs: one dict per slot to train (others None): n, n_model, n_out, trainingdata (d, traininglength) already noisy where the
    reference adds noise (see add_input_noise; the clean copy provides the targets), clean (d, traininglength) or None (= same),
    imperfect_model (n_model, traininglength) or None when ml_only, target_rows (0-based rows of the input vector: domain.target_map).
    Runs the `timestep` interleaved passes, accumulates C and B on the device, solves for W_out, stores it in the bank and returns
    {slot: dict(wout=(n_out, n_aug) Fortran-ordered host array, batch_size, batches)}."""
    import torch
    assert traininglength % timestep == 0 and discardlength % timestep == 0
    batch = chunk_batch_size(traininglength, discardlength, timestep)
    discard = discardlength // timestep
    cols = traininglength // timestep
    cap, maxd = bank.capacity, bank.max_d
    cs, bs = [None] * cap, [None] * cap
    for slot, sp in enumerate(specs):
        if sp is None:
            continue
        n_aug = sp["n"] + (0 if ml_only else sp["n_model"])
        cs[slot] = train.fortran_zeros(n_aug, n_aug)
        bs[slot] = train.fortran_zeros(sp["n_out"], n_aug)
    nb_total = 0
    for i in range(timestep):                                    # trainingdata(:, i:traininglength:timestep)
        noisy = np.zeros((cols, cap, maxd))
        models, targets = [None] * cap, [None] * cap
        for slot, sp in enumerate(specs):
            if sp is None:
                continue
            td = np.asarray(sp["trainingdata"])[:, i::timestep]
            clean = td if sp.get("clean") is None else np.asarray(sp["clean"])[:, i::timestep]
            noisy[:, slot, :td.shape[0]] = td.T
            # chunking_matmul: the targets of batch b are the clean columns discard + (b-1) m + 1 .. discard + b m, i.e. column for
            # column the input that the state was driven TOWARDS; train_pass indexes them from the start of the pass
            targets[slot] = torch.from_numpy(np.ascontiguousarray(clean[sp["target_rows"], :].T)).cuda()      # column-major (n_out, T)
            if not ml_only:
                models[slot] = torch.from_numpy(np.ascontiguousarray(np.asarray(sp["imperfect_model"])[:, i::timestep].T)).cuda()
        nb = bank.train_pass(torch.from_numpy(noisy).cuda(), discard, batch, models, targets, cs, bs, stream=stream, ml_variant=ml_only)
        nb_total += nb
    # fit_chunk_hybrid for all trained slots: reservoirs of one size class are solved together (up to 8 LU factorisations in
    # flight: a single factorisation is latency-bound on its panel kernel)
    out, classes = {}, {}
    for slot, sp in enumerate(specs):
        if sp is not None:
            classes.setdefault((sp["n"], 0 if ml_only else sp["n_model"], sp["n_out"]), []).append(slot)
    for (n, n_model, n_out), slots in classes.items():
        wouts = train.fit_chunk_hybrid_batched([cs[s] for s in slots], [bs[s] for s in slots], n, n_model, n_out, beta_res, beta_model,
                                               prior_val, using_prior, stream=stream)
        torch.cuda.synchronize()
        for slot, wout in zip(slots, wouts):
            host = np.asfortranarray(wout.cpu().numpy().T)      # device buffer is column-major (n_out, n_aug) = torch [n_aug, n_out]
            bank.set_wout(slot, host)
            out[slot] = dict(wout=host, batch_size=batch, batches=nb_total // timestep)
    return out


def train_reservoirs_device(bank, specs, hourly, traininglength, discardlength, timestep, noisemag=0.2, model_sigma=0.3, seed=0,
                            beta_res=0.001, beta_model=1.0, prior_val=0.0, using_prior=True, noisy_of_pass=None, model_of_pass=None,
                            keep_gram=False, stream=None, progress=None):
    """train_reservoirs with the training data RESIDENT on the device -- the form for BASELINE config 4 at its quoted size (40 years =
    350 640 hourly columns: 1.6 GB per reservoir, which a host loop of numpy slices cannot feed).  hourly: float64 CUDA tensor
    [traininglength, capacity, max_d], the clean standardised inputs of every slot (slot s uses its first d columns).  Per pass i of the
    `timestep` interleaved passes (trainingdata(:, i:traininglength:timestep), src/mod_reservoir.f90:289-301) the strided slice is made
    contiguous, the input noise is applied (gaussian_noise_1d_function: x + g noisemag x, src/mod_utilities.f90:1387-1409; deviates from
    a seeded device generator), the imperfect model is truth + N(0, model_sigma^2) on the slot's target rows, and
    ReservoirBank.train_pass runs the recurrence + Gram updates; then one ridge solve per slot, size classes in lockstep.
    specs: per slot dict(n, n_model, n_out, target_rows) or None.  noisy_of_pass(i, clean_pass) / model_of_pass(i, slot, truth_pass)
    override the two random draws (tests feed the host path the same numbers).  keep_gram: also return C and B (device, column-major).
    Returns {slot: dict(wout, batch_size, batches[, c, b])}."""
    import torch
    assert traininglength % timestep == 0 and discardlength % timestep == 0
    assert hourly.is_cuda and hourly.dtype == torch.float64 and tuple(hourly.shape) == (traininglength, bank.capacity, bank.max_d)
    batch = chunk_batch_size(traininglength, discardlength, timestep)
    discard = discardlength // timestep
    cap = bank.capacity
    dev = hourly.device
    cs, bs = [None] * cap, [None] * cap
    rows_dev = [None] * cap
    for slot, sp in enumerate(specs):
        if sp is None:
            continue
        n_aug = sp["n"] + sp["n_model"]
        cs[slot] = train.fortran_zeros(n_aug, n_aug)
        bs[slot] = train.fortran_zeros(sp["n_out"], n_aug)
        rows_dev[slot] = torch.as_tensor(np.asarray(sp["target_rows"]), dtype=torch.long, device=dev)
    gen = torch.Generator(device=dev)
    nb_total = 0
    for i in range(timestep):
        clean = hourly[i::timestep].contiguous()                                    # [T_pass, cap, max_d]
        gen.manual_seed(int(seed) * 1000 + i)
        if noisy_of_pass is not None:
            noisy = noisy_of_pass(i, clean)
        else:
            noisy = clean + torch.randn(clean.shape, dtype=torch.float64, device=dev, generator=gen) * noisemag * clean
        models, targets = [None] * cap, [None] * cap
        for slot, sp in enumerate(specs):
            if sp is None:
                continue
            truth = clean[:, slot, :].index_select(1, rows_dev[slot]).contiguous()        # (n_out, T_pass) column-major = torch [T_pass, n_out]
            targets[slot] = truth
            if sp["n_model"]:
                if model_of_pass is not None:
                    models[slot] = model_of_pass(i, slot, truth)
                else:
                    models[slot] = (truth[:, :sp["n_model"]] + model_sigma * torch.randn((truth.shape[0], sp["n_model"]), dtype=torch.float64, device=dev,
                                                                                         generator=gen)).contiguous()
        nb = bank.train_pass(noisy, discard, batch, models, targets, cs, bs, stream=stream)
        nb_total += nb
        del noisy, clean, models, targets
        if progress is not None:
            progress(i, nb)
    out, classes = {}, {}
    for slot, sp in enumerate(specs):
        if sp is not None:
            classes.setdefault((sp["n"], sp["n_model"], sp["n_out"]), []).append(slot)
    for (n, n_model, n_out), slots in classes.items():
        wouts = train.fit_chunk_hybrid_batched([cs[s] for s in slots], [bs[s] for s in slots], n, n_model, n_out, beta_res, beta_model,
                                               prior_val, using_prior, stream=stream)
        torch.cuda.synchronize()
        for slot, wout in zip(slots, wouts):
            host = np.asfortranarray(wout.cpu().numpy().T)
            bank.set_wout(slot, host)
            out[slot] = dict(wout=host, batch_size=batch, batches=nb_total // timestep)
            if keep_gram:
                out[slot].update(c=cs[slot], b=bs[slot], wout_dev=wout)
    return out


def shard_plan(rank, world, number_of_regions, group=64):
    """The regions rank `rank` of `world` trains, in bank-sized groups: processor_decomposition (src/res_domain.f90:31-62) exactly as
    program main's training loop uses it (src/parallelmain.f90:82-128), cut into groups of `group` reservoirs that are resident
    (and factorised) together.  Training has no collective: the ranks' region sets are disjoint and cover all regions."""
    regions = [int(r) for r in domain.processor_decomposition_manual(rank, world, number_of_regions)]
    return [regions[i:i + group] for i in range(0, len(regions), group)]


def train_sharded(rank, world, number_of_regions, build, traininglength, discardlength, timestep, group=64, out_dir=None, trial="trial",
                  **ridge):
    """program main's training loop for one rank (src/parallelmain.f90:82-128): every region of shard_plan(rank, world, ...) is built
    (`build(region)` -> dict with n, d, n_model, n_out, rows, cols, vals, win, mean, std, out_stat and the train_reservoirs spec keys
    trainingdata / clean / imperfect_model / target_rows), trained in a bank of `group` slots and, when out_dir is given, written as
    worker_RRRR_level_LL_<trial>.nc (write_trained_res, src/mod_reservoir.f90:1703-1737).  No rank talks to another.
    Returns {region: W_out (n_out, n_aug), Fortran order}."""
    from . import weights
    from .reservoir import ReservoirBank
    out = {}
    for regions in shard_plan(rank, world, number_of_regions, group):
        built = [build(r) for r in regions]
        bank = ReservoirBank(len(regions), max_d=max(b["d"] for b in built), max_n_model=max(max(b["n_model"], 1) for b in built),
                             max_n_out=max(b["n_out"] for b in built))
        specs = []
        for slot, b in enumerate(built):
            bank.load(slot, b["n"], b["d"], b["n_model"], b["n_out"], b["rows"], b["cols"], b["vals"], b["win"],
                      np.zeros((b["n_out"], b["n"] + b["n_model"])), b["mean"], b["std"], b.get("out_stat"))
            specs.append({k: b[k] for k in ("n", "n_model", "n_out", "trainingdata", "clean", "imperfect_model", "target_rows") if k in b})
        res = train_reservoirs(bank, specs, traininglength, discardlength, timestep, **ridge)
        for slot, (region, b) in enumerate(zip(regions, built)):
            out[region] = res[slot]["wout"]
            if out_dir is not None:
                import os
                weights.write_trained_res(os.path.join(out_dir, weights.trained_res_filename(region, trial)), b["win"], res[slot]["wout"],
                                          b["rows"], b["cols"], b["vals"], b["mean"], b["std"])
        bank.close()
    return out
