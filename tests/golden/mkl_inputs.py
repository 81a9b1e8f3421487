"""Seeded inputs of the MKL known-answer fixture (tests/golden/mkl_golden.npz): shared by the generator (make_mkl_golden.py, build
container only: it calls the real MKL) and by the tests that compare the oracle and the HIP path with the fixture's outputs.

    spmv : the BASELINE config-2 adjacency (n = 5760, k = 33177, 1-based COO, unsorted) with 400 injected duplicate (row, col) pairs
           and a state vector -- MKL_SPARSE_D_MV of predict (src/mod_reservoir.f90:1444; handle from mkl_sparse_d_create_coo,
           src/mod_linalg.f90:17)
    gemm : one chunking_matmul batch (src/mod_reservoir.f90:1645-1701) at a reduced width: aug = [model(132, m) ; states(128, m)], m = 98,
           targets (136, m) -- DGEMM('N','N', n, n, m, 1, aug, n, transpose(aug), m, 0, temp, n) and matmul(targets, transpose(aug))
    gesv : a 1200 x 1200 ridge system of a driven reservoir-like Gram matrix with 24 right-hand sides -- dgesv through mldivide
           (src/mod_linalg.f90:109-151) exactly as fit_chunk_hybrid calls it (:1297-1313): a_trans = transpose(C + reg), b_trans =
           transpose(B)
"""
import numpy as np

N, K, NDUP = 5760, 33177, 400


def spmv_inputs():
    rng = np.random.default_rng(20240954)
    rows = rng.integers(1, N + 1, K).astype(np.int32)
    cols = rng.integers(1, N + 1, K).astype(np.int32)
    vals = rng.random(K) * 0.1
    # duplicates: entries K-NDUP.. repeat the (row, col) of earlier entries with their own values
    src = rng.integers(0, K - NDUP, NDUP)
    rows[K - NDUP:] = rows[src]
    cols[K - NDUP:] = cols[src]
    x = rng.standard_normal(N) * 0.5
    return rows, cols, vals, x


def gemm_inputs():
    rng = np.random.default_rng(77)
    n_model, n_states, n_out, m = 132, 128, 136, 98
    model = np.asfortranarray(rng.standard_normal((n_model, m)))
    states = np.asfortranarray(np.tanh(rng.standard_normal((n_states, m))))
    y = np.asfortranarray(rng.standard_normal((n_out, m)))
    return model, states, y


def gesv_inputs():
    rng = np.random.default_rng(1200)
    n_model, n, n_out = 132, 1068, 24
    n_aug = n + n_model
    # a Gram matrix with a decaying spectrum, as a driven reservoir's is: columns of decreasing scale
    a = rng.standard_normal((n_aug, 3 * n_aug)) * (0.99 ** np.arange(n_aug))[:, None]
    c = np.asfortranarray(a @ a.T)
    c = np.asfortranarray(0.5 * (c + c.T))
    b = np.asfortranarray(rng.standard_normal((n_out, n_aug)))
    beta_res, beta_model = 1e-3, 1.0
    return n, n_model, n_out, c, b, beta_res, beta_model
