"""GPU: fp64-MFMA Gram accumulation and the LU ridge solve (through the C-ABI) against the CPU oracle.

Tolerances: Gram matrices 1e-12 relative to their max-abs (pin R^T R / R^T Y to 1e-12); W_out entrywise
1e-8 when well conditioned, otherwise by the backward error of the regularised system (cond * eps bounds the entries)."""
import os

import numpy as np
import pytest
import torch

from speedy_ml_amd import train

pytestmark = pytest.mark.gpu


def to_dev(a):       # numpy (r, c) -> column-major device buffer, torch shape [c, r]
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).T)).cuda()


def to_host(t):
    return t.cpu().numpy().T


@pytest.mark.parametrize("n,n_model,n_out,m", [(600, 12, 16, 98), (333, 7, 5, 7), (128, 0, 8, 16), (1000, 132, 136, 40), (61, 3, 2, 130)])
def test_accumulate_matches_oracle(oracle, n, n_model, n_out, m):
    rng = np.random.default_rng(n + m)
    n_aug = n + n_model
    c = train.fortran_zeros(n_aug, n_aug)
    b = train.fortran_zeros(n_out, n_aug)
    co = np.zeros((n_aug, n_aug), order="F")
    bo = np.zeros((n_out, n_aug), order="F")
    for _ in range(2):                                # accumulates across batches (C and B persist, Appendix D.4)
        states = rng.standard_normal((n, m))
        model = rng.standard_normal((n_model, m))
        y = rng.standard_normal((n_out, m))
        train.chunking_matmul(to_dev(states), to_dev(model) if n_model else None, to_dev(y), c, b)
        oracle.chunking_matmul(states, model if n_model else np.zeros((0, m)), y, co, bo)
    cg, bg = to_host(c), to_host(b)
    low = np.tril_indices(n_aug)
    assert np.max(np.abs(cg[low] - co[low])) <= 1e-12 * np.max(np.abs(co))
    assert np.max(np.abs(bg - bo)) <= 1e-12 * np.max(np.abs(bo))
    # tiles strictly above the diagonal tiles are left untouched until the mirror pass
    train.symmetrize(c)
    cs = to_host(c)
    assert np.array_equal(cs, cs.T) and np.array_equal(cs[low], cg[low])


def test_fit_well_conditioned(oracle):
    rng = np.random.default_rng(21)
    n, n_model, n_out, m = 300, 12, 16, 200
    n_aug = n + n_model
    c = train.fortran_zeros(n_aug, n_aug)
    b = train.fortran_zeros(n_out, n_aug)
    co = np.zeros((n_aug, n_aug), order="F")
    bo = np.zeros((n_out, n_aug), order="F")
    for _ in range(3):
        states, model, y = rng.standard_normal((n, m)), rng.standard_normal((n_model, m)), rng.standard_normal((n_out, m))
        train.chunking_matmul(to_dev(states), to_dev(model), to_dev(y), c, b)
        oracle.chunking_matmul(states, model, y, co, bo)
    for using_prior, prior_val in ((True, 0.0), (True, 0.3), (False, 0.0)):
        wg = to_host(train.fit_chunk_hybrid(c, b, n, n_model, n_out, 1e-3, 1.0, prior_val, using_prior))
        info, wo = oracle.fit_chunk_hybrid(n, n_model, n_out, 1e-3, 1.0, prior_val, using_prior, co, bo)
        assert info == 0
        assert np.max(np.abs(wg - wo)) <= 1e-8 * np.max(np.abs(wo)), (using_prior, prior_val)


def test_fit_ill_conditioned_by_residual_and_pivoting(oracle):
    rng = np.random.default_rng(22)
    n, n_model, n_out, m = 500, 12, 16, 98            # rank 98 Gram matrix + 1e-6 ridge: cond ~ 1e9
    n_aug = n + n_model
    states, model, y = rng.standard_normal((n, m)), rng.standard_normal((n_model, m)), rng.standard_normal((n_out, m))
    c = train.fortran_zeros(n_aug, n_aug)
    b = train.fortran_zeros(n_out, n_aug)
    train.chunking_matmul(to_dev(states), to_dev(model), to_dev(y), c, b)
    wg = to_host(train.fit_chunk_hybrid(c, b, n, n_model, n_out, 1e-3, 1.0, 0.0, True))
    cs, bs = to_host(c), to_host(b)
    reg = np.diag(np.r_[np.full(n_model, 1.0), np.full(n, 1e-6)])
    resid = (cs + reg).T @ wg.T - bs.T
    assert np.linalg.norm(resid) / np.linalg.norm(bs) < 1e-9
    # predictions agree with the oracle's solution although the entries only agree to cond*eps
    co = np.zeros((n_aug, n_aug), order="F")
    bo = np.zeros((n_out, n_aug), order="F")
    oracle.chunking_matmul(states, model, y, co, bo)
    _, wo = oracle.fit_chunk_hybrid(n, n_model, n_out, 1e-3, 1.0, 0.0, True, co, bo)
    aug = np.vstack([model, states])
    assert np.max(np.abs(wg @ aug - wo @ aug)) <= 1e-6 * np.max(np.abs(wo @ aug))


def test_fit_general_matrix_needs_pivoting():
    """dgesv semantics: a non-symmetric-looking system whose leading entry is tiny must still solve accurately.
    (C is symmetrised by the fit, so build a symmetric indefinite matrix with a zero leading diagonal.)"""
    rng = np.random.default_rng(23)
    n, n_out = 97, 3
    a = rng.standard_normal((n, n))
    a = a + a.T
    a[0, 0] = 0.0
    bmat = rng.standard_normal((n_out, n))
    wg = to_host(train.fit_chunk_hybrid(to_dev(a), to_dev(bmat), n, 0, n_out, 0.0, 0.0, 0.0, False))
    want = np.linalg.solve(a.T, bmat.T).T
    assert np.max(np.abs(wg - want)) <= 1e-9 * np.max(np.abs(want))


@pytest.mark.parametrize("n,n_out", [(700, 130), (385, 17), (128, 64)])
def test_fit_indefinite_systems_against_lapack(n, n_out):
    """Symmetric indefinite systems (every panel interchanges rows, displaced rows travel through the composite permutation) with
    full, ragged and multi-workgroup right-hand-side blocks: the blocked back substitution's near / step launches at sizes the host
    LAPACK solves in milliseconds."""
    rng = np.random.default_rng(1000 + n)
    a = rng.standard_normal((n, n))
    a = a + a.T
    a[np.arange(0, n, 7), np.arange(0, n, 7)] = 0.0
    bmat = rng.standard_normal((n_out, n))
    wg = to_host(train.fit_chunk_hybrid(to_dev(a), to_dev(bmat), n, 0, n_out, 0.0, 0.0, 0.0, False))
    want = np.linalg.solve(a.T, bmat.T).T
    resid = a.T @ wg.T - bmat.T
    eta = np.linalg.norm(resid) / (np.linalg.norm(a) * np.linalg.norm(wg) + np.linalg.norm(bmat))
    assert eta <= 1e-15, eta
    assert np.max(np.abs(wg - want)) <= 1e-8 * np.max(np.abs(want))


def test_singular_matrix_fails_loudly():
    from speedy_ml_amd._lib import SmlError
    n, n_out = 64, 2
    c = train.fortran_zeros(n, n)
    b = train.fortran_zeros(n_out, n)
    with pytest.raises(SmlError):
        train.fit_chunk_hybrid(c, b, n, 0, n_out, 0.0, 0.0, 0.0, False)


@pytest.mark.parametrize("m", [98, 2920])
def test_full_size_gram_against_torch_fp64(m):
    """BASELINE config 4 shape: n=5760, n_model=132, n_out=136; m=98 (shipped batch size, general kernel) and m=2920 (the
    40-year configuration, LDS-DMA kernel)."""
    torch.manual_seed(5)
    n, n_model, n_out = 5760, 132, 136
    n_aug = n + n_model
    states = torch.randn((m, n), dtype=torch.float64, device="cuda")
    model = torch.randn((m, n_model), dtype=torch.float64, device="cuda")
    y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
    c = train.fortran_zeros(n_aug, n_aug)
    b = train.fortran_zeros(n_out, n_aug)
    train.chunking_matmul(states, model, y, c, b)
    train.symmetrize(c)
    aug = torch.cat([model, states], dim=1)           # [m, n_aug]
    cref = aug.T @ aug                                # symmetric, so layout does not matter
    bref = aug.T @ y                                  # torch [n_aug, n_out] == column-major (n_out, n_aug)
    assert float((c - cref).abs().max()) <= 1e-12 * float(cref.abs().max())
    assert float((b - bref).abs().max()) <= 1e-12 * float(bref.abs().max())


@pytest.mark.parametrize("n,n_model,n_out,m", [
    (256, 0, 8, 512),        # the smallest shape the 256 x 128 kernel takes: one SYRK row tile, no model block
    (640, 132, 136, 520),    # n mod 256 = 128: a short first row tile; m mod 8 = 0
    (700, 132, 136, 1029),   # n not a multiple of 128; 5 left-over columns of m go through the general kernel
    (1500, 4, 2, 776),       # tiny skinny blocks, ragged everything
    (254, 2, 2, 600),        # n < 256: stays on the 128 x 128 kernel
])
def test_long_gram_shapes_and_determinism(n, n_model, n_out, m):
    """The one-launch Gram update (256 x 128 tiles, all products in one work list, K-split tail) on ragged shapes against torch
    fp64, accumulating twice into the same C and B; a second run from zero must give the same bits (the tail is reduced in a fixed
    order and every output element is owned by one work item)."""
    torch.manual_seed(n + m)
    n_aug = n + n_model
    states = torch.randn((m, n), dtype=torch.float64, device="cuda")
    model = torch.randn((m, max(n_model, 1)), dtype=torch.float64, device="cuda")[:, :n_model].contiguous()
    y = torch.randn((m, n_out), dtype=torch.float64, device="cuda")
    runs = []
    for _ in range(2):
        c = train.fortran_zeros(n_aug, n_aug)
        b = train.fortran_zeros(n_out, n_aug)
        train.chunking_matmul(states, model if n_model else None, y, c, b)
        train.chunking_matmul(states, model if n_model else None, y, c, b)
        train.symmetrize(c)
        runs.append((c.clone(), b.clone()))
    aug = torch.cat([model, states], dim=1) if n_model else states
    cref = 2.0 * (aug.T @ aug)
    bref = 2.0 * (aug.T @ y)
    c, b = runs[0]
    assert float((c - cref).abs().max()) <= 1e-12 * float(cref.abs().max())
    assert float((b - bref).abs().max()) <= 1e-12 * float(bref.abs().max())
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("n", [1500, 5760])
def test_single_solve_stream_layouts_agree_bitwise(tmp_path, n):
    """The pivoted LU (SML_FIT_SOLVER=lu): a single ridge solve confines its panel chain to reserved CUs and sends the interchanges /
    U12 of the far columns to the trailing stream; batches (and SML_LU_CONFINE=0) keep everything on one chain.  Same arithmetic,
    different stream ordering: the weights must agree bit for bit (a missing dependency between the two streams shows up here), for a
    system with a ragged last panel and right-hand sides that straddle it (n = 1500) and at the full size of config 4 (n_aug = 5892)."""
    import subprocess, sys
    outs = []
    for confine in ("1", "0"):
        f = tmp_path / f"w{confine}.npy"
        env = dict(os.environ, SML_LU_CONFINE=confine, SML_FIT_SOLVER="lu")
        subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_fit_dump.py"), str(f), str(n)], check=True, env=env, timeout=600)
        outs.append(np.load(f))
    assert np.all(np.isfinite(outs[0]))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("n", [1500, 5760])
def test_cholesky_fused_and_split_panel_forms_agree_bitwise(tmp_path, n):
    """The Cholesky (default solver): a single solve factors a panel and solves its U12 rows in ONE launch (k_chol_panel), batches use
    a factor-only launch and the MFMA triangular solve with the exported quad inverses.  Both forms must give the same bits (this is
    what makes a batched fit equal to single fits), at a ragged size and at the full size of config 4."""
    import subprocess, sys
    outs = []
    for fused in ("1", "0"):
        f = tmp_path / f"c{fused}.npy"
        env = dict(os.environ, SML_CHOL_FUSED=fused, SML_FIT_SOLVER="chol")
        subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_fit_dump.py"), str(f), str(n)], check=True, env=env, timeout=600)
        outs.append(np.load(f))
    assert np.all(np.isfinite(outs[0]))
    assert np.array_equal(outs[0], outs[1])


def test_device_training_pass_matches_oracle(oracle):
    """K7 on the device (reservoir recurrence + batch flushes into the MFMA Gram update) for two reservoirs of different
    size in one bank, against the oracle's restatement of reservoir_layer_chunking_hybrid."""
    from speedy_ml_amd.reservoir import ReservoirBank
    from speedy_ml_amd.synth import make_reservoir
    rs = [make_reservoir(n=120, d=12, n_model=4, n_out=6, seed=8), make_reservoir(n=256, d=8, n_model=2, n_out=3, seed=9)]
    bank = ReservoirBank(3, max_d=12, max_n_model=4, max_n_out=6)
    for i, r in enumerate(rs):
        bank.load(i, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)
    rng = np.random.default_rng(12)
    T, discard, batch = 4 + 3 * 5, 4, 5
    noisy = rng.standard_normal((T, 3, 12))
    dev_in = torch.from_numpy(noisy).cuda()
    models, targets, cs, bs, want = [], [], [], [], []
    for i, r in enumerate(rs):
        model, targ = rng.standard_normal((r.n_model, T)), rng.standard_normal((r.n_out, T))
        models.append(to_dev(model)); targets.append(to_dev(targ))
        cs.append(train.fortran_zeros(r.n_aug, r.n_aug)); bs.append(train.fortran_zeros(r.n_out, r.n_aug))
        co, bo = np.zeros((r.n_aug, r.n_aug), order="F"), np.zeros((r.n_out, r.n_aug), order="F")
        nb = oracle.train_states(r.n, r.d, r.rows, r.cols, r.vals, r.win, 1.0, np.asfortranarray(noisy[:, i, :r.d].T), discard,
                                 batch, model, targ, co, bo)
        want.append((nb, co, bo))
    nb = bank.train_pass(dev_in, discard, batch, models, targets, cs, bs)
    for i, r in enumerate(rs):
        nbo, co, bo = want[i]
        assert nb == nbo == 3          # flushes at i = 4, 9, 14 (i+1 divisible by the batch size)
        cg, bg = to_host(cs[i]), to_host(bs[i])
        low = np.tril_indices(r.n_aug)
        assert np.max(np.abs(cg[low] - co[low])) <= 1e-12 * np.max(np.abs(co))
        assert np.max(np.abs(bg - bo)) <= 1e-12 * np.max(np.abs(bo))
    # end to end: fit W_out from the device-accumulated matrices and load it back into the bank
    wout = train.fit_chunk_hybrid(cs[0], bs[0], rs[0].n, rs[0].n_model, rs[0].n_out, 1e-3, 1.0, 0.0, True)
    wh = to_host(wout)
    bank.set_wout(0, wh)
    r = rs[0]
    x0 = rng.standard_normal(r.n) * 0.3
    bank.set_state(0, x0)
    bank.set_feedback(0, r.feedback)
    bank.set_local_model(0, r.local_model)
    bank.predict()
    torch.cuda.synchronize()
    xw, ow = oracle.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, np.asfortranarray(wh), 1.0, r.feedback,
                                r.local_model, x0)
    assert np.max(np.abs(bank.get_state(0) - xw)) <= 1e-13
    assert np.max(np.abs(bank.get_outvec(0) - ow)) <= 1e-11 * np.max(np.abs(ow))


def test_device_training_pass_ml_variant(oracle):
    """reservoir_layer_chunking_ml (src/mod_reservoir.f90:963-1065): no physics-model rows, and the step after a batch flush
    multiplies A by the column whose even entries were squared in place while the leak term keeps the state (quirk Q6).
    A leak rate below one exercises both halves of that step; the hybrid variant on the same data must differ."""
    from speedy_ml_amd.reservoir import ReservoirBank
    from speedy_ml_amd.synth import make_reservoir
    r = make_reservoir(n=192, d=12, n_model=0, n_out=5, seed=18)
    leak = 0.6
    rng = np.random.default_rng(14)
    T, discard, batch = 3 + 4 * 6, 3, 6
    noisy = rng.standard_normal((T, 2, 12))
    dev_in = torch.from_numpy(noisy).cuda()
    targ = rng.standard_normal((r.n_out, T))
    model = np.zeros((0, T))
    results = {}
    for ml in (True, False):
        bank = ReservoirBank(2, max_d=12, max_n_model=1, max_n_out=5)
        bank.load(1, r.n, r.d, 0, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None, leakage=leak)
        cs, bs = [None, train.fortran_zeros(r.n, r.n)], [None, train.fortran_zeros(r.n_out, r.n)]
        nb = bank.train_pass(dev_in, discard, batch, [None, None], [None, to_dev(targ)], cs, bs, ml_variant=ml)
        co, bo = np.zeros((r.n, r.n), order="F"), np.zeros((r.n_out, r.n), order="F")
        nbo = oracle.train_states(r.n, r.d, r.rows, r.cols, r.vals, r.win, leak, np.asfortranarray(noisy[:, 1, :r.d].T), discard, batch,
                                  model, targ, co, bo, ml_variant=ml)
        assert nb == nbo == 4
        cg, bg = to_host(cs[1]), to_host(bs[1])
        low = np.tril_indices(r.n)
        assert np.max(np.abs(cg[low] - co[low])) <= 1e-12 * np.max(np.abs(co)), ml
        assert np.max(np.abs(bg - bo)) <= 1e-12 * np.max(np.abs(bo)), ml
        results[ml] = bg
    assert np.max(np.abs(results[True] - results[False])) > 1e-6 * np.max(np.abs(results[False]))


@pytest.mark.parametrize("n", [200, 600])
def test_batched_fit_is_one_arithmetic_for_any_count(oracle, n):
    """sml_train_fit_batched gives the same bits whether systems are handed over one at a time or all together (more systems than the
    workspace holds at once: scratch is reused in order) -- that is what lets a training queue pick its group size freely.  sml_train_fit
    (one system, latency form: the back substitution's far-row update is fused into the solve launch instead of running as a matrix-core
    product) solves the same system to the same backward error; its bits agree with the batched form where the back substitution
    has a single step (n_aug <= 256) and differ by rounding otherwise."""
    rng = np.random.default_rng(31)
    n_model, n_out, m = 8, 6, 150 if n == 200 else 900
    n_aug = n + n_model
    cs, bs = [], []
    for i in range(11):
        states, model, y = rng.standard_normal((n, m)), rng.standard_normal((n_model, m)), rng.standard_normal((n_out, m))
        c, b = train.fortran_zeros(n_aug, n_aug), train.fortran_zeros(n_out, n_aug)
        for _ in range(2):
            train.chunking_matmul(to_dev(states), to_dev(model), to_dev(y), c, b)
        cs.append(c); bs.append(b)
    single = [to_host(train.fit_chunk_hybrid(c.clone(), b, n, n_model, n_out)) for c, b in zip(cs, bs)]
    one_by_one = [to_host(train.fit_chunk_hybrid_batched([c.clone()], [b], n, n_model, n_out)[0]) for c, b in zip(cs, bs)]
    batched = [to_host(w) for w in train.fit_chunk_hybrid_batched(cs, bs, n, n_model, n_out)]
    for c, b, w1, w2, w3 in zip(cs, bs, single, one_by_one, batched):
        assert np.array_equal(w2, w3)
        if n_aug <= 256:
            assert np.array_equal(w1, w3)
        a = to_host(c) + np.diag(np.r_[np.full(n_model, 1.0), np.full(n, 1e-6)])
        bh = to_host(b)
        eta = lambda w: np.linalg.norm(w @ a - bh) / (np.linalg.norm(a) * np.linalg.norm(w) + np.linalg.norm(bh))
        assert eta(w1) <= 1e-15 and eta(w3) <= 1e-15, (eta(w1), eta(w3))
        assert np.max(np.abs(w1 - w3)) <= 1e-9 * np.max(np.abs(w3))


def test_full_size_ridge_fit_of_a_driven_reservoir():
    """BASELINE config 4 at full size: the 5892 x 5892 system with 136 right-hand sides that fit_chunk_hybrid hands to dgesv
    (src/mod_reservoir.f90:1235-1334, src/mod_linalg.f90:109-151), built from the Gram matrices of a DRIVEN region-954 reservoir:
    12 batches of 98 columns, so C has rank <= 1176 << 5892 and the ridge (beta_res^2 = 1e-6, beta_model^2 = 1) alone makes it
    regular (condition number ~1e9..1e10).  Checked against LAPACK's dgesv itself (numpy.linalg.solve: the routine the
    reference calls) through (i) the normwise backward error of the device solution, (ii) the agreement of the two W_out in what
    they predict on the training columns -- ||(W - W_ref) aug||_F / ||W_ref aug||_F evaluated through the Gram matrix -- and (iii)
    the entries of W_out to cond * eps."""
    from speedy_ml_amd.reservoir import ReservoirBank
    from speedy_ml_amd.synth import make_reservoir
    n, d, n_model, n_out = 5760, 576, 132, 136
    n_aug = n + n_model
    r = make_reservoir(n=n, d=d, n_model=n_model, n_out=n_out, seed=20240954)
    bank = ReservoirBank(1)
    bank.load(0, n, d, n_model, n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)
    rng = np.random.default_rng(954)
    batch, discard = 98, 40
    T = discard + 12 * batch
    drive = np.zeros((T, d))
    e = rng.standard_normal((T, d)) * 0.3
    for t in range(1, T):
        drive[t] = 0.95 * drive[t - 1] + e[t]
    noisy = np.zeros((T, 1, 576))
    noisy[:, 0, :d] = drive
    targ = drive[:, :n_out].T + 0.05 * rng.standard_normal((n_out, T))
    model = targ[:n_model] * 0.9 + 0.1 * rng.standard_normal((n_model, T))
    cs, bs = [train.fortran_zeros(n_aug, n_aug)], [train.fortran_zeros(n_out, n_aug)]
    nb = bank.train_pass(torch.from_numpy(noisy).cuda(), discard, batch, [to_dev(model)], [to_dev(targ)], cs, bs)
    assert nb == 12                                   # steps i = 1 .. T - discard - 1 flush whenever (i + 1) % 98 == 0
    beta_res, beta_model = 1e-3, 1.0
    wg = to_host(train.fit_chunk_hybrid(cs[0], bs[0], n, n_model, n_out, beta_res, beta_model, 0.0, True))
    c, b = to_host(cs[0]), to_host(bs[0])            # the fit mirrored C's lower triangle into the upper one
    assert np.array_equal(c, c.T)
    a = c + np.diag(np.r_[np.full(n_model, beta_model ** 2), np.full(n, beta_res ** 2)])
    want = np.linalg.solve(a.T, b.T).T                # dgesv
    resid = a.T @ wg.T - b.T
    eta = np.linalg.norm(resid) / (np.linalg.norm(a) * np.linalg.norm(wg) + np.linalg.norm(b))
    resid_ref = a.T @ want.T - b.T
    eta_ref = np.linalg.norm(resid_ref) / (np.linalg.norm(a) * np.linalg.norm(want) + np.linalg.norm(b))
    assert eta <= 1e-15 + 4.0 * eta_ref, (eta, eta_ref)
    dw = wg - want
    pred = np.sqrt(max(np.trace(dw @ c @ dw.T), 0.0) / np.trace(want @ c @ want.T))
    print(f"backward error {eta:.3e} (LAPACK on the host: {eta_ref:.3e}), prediction difference {pred:.3e}")
    assert pred <= 1e-9, pred
    assert np.max(np.abs(dw)) <= 1e-5 * np.max(np.abs(want)), np.max(np.abs(dw)) / np.max(np.abs(want))


def _select_solver(which):
    from speedy_ml_amd import _lib
    return _lib.lib().sml_train_select_solver(which)


@pytest.mark.parametrize("n,n_model,n_out,nsys", [(5760, 132, 136, 1), (1000, 12, 200, 3), (130, 2, 5, 2)])
def test_cholesky_is_the_solver_of_spd_ridge_systems(n, n_model, n_out, nsys):
    """The blocked Cholesky (solver 2: no LU behind it, so a failure cannot hide) against the pivoted LU (solver 1 = dgesv's
    algorithm) on ridge systems of a driven-reservoir-like Gram matrix: the full 5892 x 5892 system, one with more right-hand sides
    than one back-substitution group holds (200 > 136), one barely larger than a 128-row block; single and batched.  Both solve
    the same regularised system: compared by normwise backward error and by what the two W_out predict."""
    rng = np.random.default_rng(n + n_out)
    n_aug, m = n + n_model, min(2 * n, 1500)
    prev = _select_solver(-1)
    try:
        cs, bs = [], []
        for _ in range(nsys):
            states = np.tanh(rng.standard_normal((n, m)) * (0.995 ** np.arange(n))[:, None] * 2.0)
            model, y = rng.standard_normal((n_model, m)), rng.standard_normal((n_out, m))
            c, b = train.fortran_zeros(n_aug, n_aug), train.fortran_zeros(n_out, n_aug)
            train.chunking_matmul(to_dev(states), to_dev(model), to_dev(y), c, b)
            cs.append(c); bs.append(b)
        _select_solver(2)
        wc = [to_host(w) for w in train.fit_chunk_hybrid_batched(cs, bs, n, n_model, n_out)] if nsys > 1 else \
             [to_host(train.fit_chunk_hybrid(cs[0], bs[0], n, n_model, n_out))]
        _select_solver(1)
        wl = [to_host(w) for w in train.fit_chunk_hybrid_batched(cs, bs, n, n_model, n_out)] if nsys > 1 else \
             [to_host(train.fit_chunk_hybrid(cs[0], bs[0], n, n_model, n_out))]
        for c, b, w1, w2 in zip(cs, bs, wc, wl):
            ch, bh = to_host(c), to_host(b)
            a = ch + np.diag(np.r_[np.full(n_model, 1.0), np.full(n, 1e-6)])
            eta = lambda w: np.linalg.norm(w @ a - bh) / (np.linalg.norm(a) * np.linalg.norm(w) + np.linalg.norm(bh))
            assert np.isfinite(w1).all()
            assert eta(w1) <= 1e-15 + 4.0 * eta(w2), (eta(w1), eta(w2))
            dw = w1 - w2
            assert np.sqrt(max(np.trace(dw @ ch @ dw.T), 0.0) / np.trace(w2 @ ch @ w2.T)) <= 1e-9
    finally:
        _select_solver(prev)


def test_cholesky_only_rejects_an_indefinite_system_and_auto_falls_back():
    from speedy_ml_amd._lib import SmlError
    rng = np.random.default_rng(5)
    n, n_model, n_out = 300, 4, 6
    n_aug = n + n_model
    s = rng.standard_normal((n_aug, n_aug))
    sym = s + s.T                                       # symmetric, indefinite
    c, b = to_dev(sym), to_dev(rng.standard_normal((n_out, n_aug)))
    prev = _select_solver(-1)
    try:
        _select_solver(2)
        with pytest.raises(SmlError, match="not positive definite"):
            train.fit_chunk_hybrid(c.clone(), b, n, n_model, n_out, 1e-3, 1.0, 0.0, True)
        _select_solver(0)                               # auto: the pivoted LU takes over
        w = to_host(train.fit_chunk_hybrid(c.clone(), b, n, n_model, n_out, 1e-3, 1.0, 0.0, True))
        a = sym + np.diag(np.r_[np.full(n_model, 1.0), np.full(n, 1e-6)])
        want = np.linalg.solve(a.T, to_host(b).T).T
        assert np.max(np.abs(w - want)) <= 1e-8 * np.max(np.abs(want))
    finally:
        _select_solver(prev)


def test_lu_of_a_system_taller_than_the_register_leaf():
    """The pivoted LU has no size limit: panels taller than the 7168 rows its register-resident leaf holds go through k_lu_leaf_tall
    (the same pivot rule and arithmetic on the panel in memory).  A 7400-row symmetric INDEFINITE system (so the default solver's
    Cholesky gives up and the LU takes over) with 140 right-hand sides (two groups of the back substitution), against LAPACK."""
    rng = np.random.default_rng(17)
    n, n_model, n_out = 7396, 4, 140
    n_aug = n + n_model
    s = rng.standard_normal((n_aug, 64))
    sym = s @ s.T - 0.5 * np.diag(rng.uniform(1.0, 2.0, n_aug)) * 64          # symmetric with eigenvalues of both signs
    bm = rng.standard_normal((n_out, n_aug))
    c, b = to_dev(sym), to_dev(bm)
    prev = _select_solver(-1)
    try:
        _select_solver(0)
        w = to_host(train.fit_chunk_hybrid(c, b, n, n_model, n_out, 1e-3, 1.0, 0.0, True))
    finally:
        _select_solver(prev)
    a = sym + np.diag(np.r_[np.full(n_model, 1.0), np.full(n, 1e-6)])
    resid = w @ a.T - bm                                                       # W (C + reg) = B   (C symmetric)
    berr = np.linalg.norm(resid) / (np.linalg.norm(a) * np.linalg.norm(w) + np.linalg.norm(bm))
    assert np.all(np.isfinite(w)) and berr <= 1e-14, berr
