"""CPU: gen_res / makesparse / spectral radius (host entry points of the C-ABI) -- structural properties of the
reference's construction (src/mod_linalg.f90:180-218) and the rescaling (src/mod_reservoir.f90:196-198)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from speedy_ml_amd.reservoir import gen_res


def test_makesparse_structure_and_radius():
    n, k = 5760, 33177                     # interior + SST class
    rows, cols, vals, eigs = gen_res(n, k, 0.7, 20240954)
    # every row / column index appears floor(k/n) or floor(k/n)+1 times (concatenated permutations)
    for idx in (rows, cols):
        assert idx.min() == 1 and idx.max() == n
        cnt = np.bincount(idx - 1, minlength=n)
        assert set(np.unique(cnt)) <= {k // n, k // n + 1}
        assert (cnt == k // n + 1).sum() == k % n
    # each full block of n entries is a permutation
    assert sorted(rows[:n]) == list(range(1, n + 1)) and sorted(cols[n:2 * n]) == list(range(1, n + 1))
    assert 0.0 <= vals.min() and eigs > 0
    # after rescaling the spectral radius is `radius` (independent check with scipy/ARPACK)
    A = sp.coo_matrix((vals, (rows - 1, cols - 1)), shape=(n, n)).tocsr()
    lam = spla.eigs(A, k=1, which="LM", return_eigenvectors=False)[0]
    assert abs(abs(lam) - 0.7) < 1e-8
    # different seeds give different matrices, same seed the same one
    r2, c2, v2, _ = gen_res(n, k, 0.7, 20240954)
    assert np.array_equal(rows, r2) and np.array_equal(vals, v2)
    r3, _, _, _ = gen_res(n, k, 0.7, 1)
    assert not np.array_equal(rows, r3)


def test_small_k_branch():
    # k <= n: one partial permutation each (src/mod_linalg.f90:208-212)
    import ctypes as C
    from speedy_ml_amd import _lib
    rows, cols, vals = np.zeros(40, dtype=np.int32), np.zeros(40, dtype=np.int32), np.zeros(40)
    _lib.check(_lib.lib().sml_makesparse(100, 40, C.c_uint64(7), _lib.ip(rows), _lib.ip(cols), _lib.dp(vals)))
    assert len(set(rows)) == 40 and len(set(cols)) == 40 and rows.min() >= 1 and rows.max() <= 100
    # a nilpotent pattern has spectral radius 0: gen_res must refuse to divide by it
    import pytest
    with pytest.raises(_lib.SmlError):
        gen_res(100, 3, 0.9, 7)
