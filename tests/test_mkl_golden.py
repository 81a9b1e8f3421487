"""The oracle's restatements of the third-party arithmetic (MKL sparse COO product with duplicate entries, DGEMM of chunking_matmul,
dgesv of the ridge fit) and the HIP path, against known answers computed by the REAL Intel MKL in the build container
(tests/golden/make_mkl_golden.py -> tests/golden/mkl_golden.npz; inputs regenerated from seeds by tests/golden/mkl_inputs.py).
This pins the library semantics at src/mod_reservoir.f90:1444,1695 and src/mod_linalg.f90:145; the Fortran around those calls stays
unpinned."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import mkl_inputs  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "mkl_golden.npz"))


def test_fixture_comes_from_mkl(golden):
    assert "Math Kernel Library" in str(golden["mkl_version"])
    rows, cols, _, _ = mkl_inputs.spmv_inputs()
    pairs = rows.astype(np.int64) * 100000 + cols
    assert len(pairs) - len(np.unique(pairs)) >= mkl_inputs.NDUP          # the duplicates are really there


def test_oracle_coo_product_matches_mkl_sparse_d_mv(oracle, golden):
    rows, cols, vals, x = mkl_inputs.spmv_inputs()
    y = oracle.coo_mv(mkl_inputs.N, rows, cols, vals, x)
    ref = golden["spmv_y"]
    assert np.max(np.abs(y - ref)) <= 4e-16 * np.max(np.abs(ref)) * 8          # a handful of terms per row: summation order only
    # duplicates accumulate (they are not overwritten, not dropped)
    dense = np.zeros(mkl_inputs.N)
    np.add.at(dense, rows - 1, vals * x[cols - 1])
    assert np.max(np.abs(dense - ref)) <= 1e-15 * np.max(np.abs(ref)) * 8


def test_oracle_chunking_matmul_matches_mkl_dgemm(oracle, golden):
    model, states, y = mkl_inputs.gemm_inputs()
    n_aug = model.shape[0] + states.shape[0]
    c, b = np.zeros((n_aug, n_aug), order="F"), np.zeros((y.shape[0], n_aug), order="F")
    oracle.chunking_matmul(states, model, y, c, b)
    assert np.max(np.abs(c - golden["gemm_c"])) <= 1e-13 * np.max(np.abs(golden["gemm_c"]))
    assert np.max(np.abs(b - golden["gemm_b"])) <= 1e-13 * np.max(np.abs(golden["gemm_b"]))


def _ridge_checks(w, golden):
    n, n_model, n_out, c, b, beta_res, beta_model = mkl_inputs.gesv_inputs()
    reg = np.r_[np.full(n_model, beta_model ** 2), np.full(n, beta_res ** 2)]
    a = c + np.diag(reg)
    ref = golden["gesv_wout"]
    # both solve  W (C + reg) = B : compare by backward error against MKL's own, and entrywise to cond * eps
    berr = lambda z: np.linalg.norm(z @ a - b) / (np.linalg.norm(a, 2) * np.linalg.norm(z) + np.linalg.norm(b))
    assert berr(ref) <= 1e-15
    assert berr(w) <= 4 * max(berr(ref), 2e-17)
    cond = np.linalg.cond(a)
    assert np.max(np.abs(w - ref)) <= 50 * cond * 1.1e-16 * np.max(np.abs(ref))
    return berr(w), berr(ref), cond


def test_oracle_ridge_fit_matches_mkl_dgesv(oracle, golden):
    n, n_model, n_out, c, b, beta_res, beta_model = mkl_inputs.gesv_inputs()
    info, w = oracle.fit_chunk_hybrid(n, n_model, n_out, beta_res, beta_model, 0.0, True, c.copy(order="F"), b.copy(order="F"))
    assert info == 0
    print(_ridge_checks(w, golden))


@pytest.mark.gpu
def test_hip_update_matches_mkl_sparse_d_mv(golden):
    """the bank's SELL-64 product of the config-2 adjacency with duplicate entries against MKL's y = A x: x_new = tanh(A x) with
    W_in = 0, so atanh recovers the product"""
    import torch
    from speedy_ml_amd.reservoir import ReservoirBank
    rows, cols, vals, x = mkl_inputs.spmv_inputs()
    n, d = mkl_inputs.N, 4
    bank = ReservoirBank(1, max_d=d, max_n_model=1, max_n_out=2)
    win = np.zeros((n, d), order="F")
    bank.load(0, n, d, 0, 2, rows, cols, vals, win, np.zeros((2, n), order="F"), np.zeros(36), np.ones(36), None)
    bank.set_state(0, x)
    bank.set_feedback(0, np.zeros(d))
    bank.advance()
    torch.cuda.synchronize()
    got = bank.get_state(0)
    ref = np.tanh(golden["spmv_y"])
    assert np.max(np.abs(got - ref)) <= 1e-14


@pytest.mark.gpu
def test_hip_gram_and_ridge_match_mkl(golden):
    import torch
    from speedy_ml_amd import train
    model, states, y = mkl_inputs.gemm_inputs()
    n_aug = model.shape[0] + states.shape[0]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()              # column-major (r, m) = torch [m, r]
    c, b = train.fortran_zeros(n_aug, n_aug), train.fortran_zeros(y.shape[0], n_aug)
    train.chunking_matmul(dev(states), dev(model), dev(y), c, b)
    torch.cuda.synchronize()
    cg = c.cpu().numpy().T
    cg = np.tril(cg) + np.tril(cg, -1).T                                             # (tiles above the diagonal are not computed)
    assert np.max(np.abs(cg - golden["gemm_c"])) <= 1e-13 * np.max(np.abs(golden["gemm_c"]))
    assert np.max(np.abs(b.cpu().numpy().T - golden["gemm_b"])) <= 1e-13 * np.max(np.abs(golden["gemm_b"]))
    n, n_model, n_out, cm, bm, beta_res, beta_model = mkl_inputs.gesv_inputs()
    cd = torch.from_numpy(np.ascontiguousarray(cm.T)).cuda()
    bd = torch.from_numpy(np.ascontiguousarray(bm.T)).cuda()
    w = train.fit_chunk_hybrid(cd, bd, n, n_model, n_out, beta_res=beta_res, beta_model=beta_model, prior_val=0.0, using_prior=True)
    torch.cuda.synchronize()
    berr, berr_ref, cond = _ridge_checks(w.cpu().numpy().T, golden)
    print(f"ridge 1200: backward error {berr:.2e} (MKL dgesv {berr_ref:.2e}), cond {cond:.2e}")
