"""CPU: gen_res / makesparse / spectral radius (host entry points of the C-ABI) -- structural properties of the
reference's construction (src/mod_linalg.f90:180-218) and the rescaling (src/mod_reservoir.f90:196-198)."""
import numpy as np
import pytest
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


def _makesparse_numpy(n, k, draws):
    """independent evaluation of makesparse + shuffle (src/mod_linalg.f90:180-218, src/mod_utilities.f90:1569-1596) in numpy"""
    it = iter(draws)
    vals = np.array([next(it) for _ in range(k)])

    def shuffle(returnsize):
        choices = list(range(1, n + 1))
        out = []
        for chosen in range(n):
            this = int(next(it) * (n - chosen) + 1)
            tmp = choices[this - 1]
            out.append(tmp)
            choices[this - 1] = choices[n - chosen - 1]
            choices[n - chosen - 1] = tmp
        return out[:returnsize]
    rows, cols = [], []
    if k > n:
        for _ in range(k // n):
            rows += shuffle(n)
            cols += shuffle(n)
        if k % n:
            rows += shuffle(k % n)
            cols += shuffle(k % n)
    else:
        rows, cols = shuffle(k), shuffle(k)
    return np.array(rows, dtype=np.int32), np.array(cols, dtype=np.int32), vals


@pytest.mark.parametrize("n,k", [(5760, 33177), (300, 1000), (100, 40), (64, 64), (50, 100)])
def test_makesparse_index_construction_bit_exact_for_supplied_draws(oracle, n, k):
    """makesparse's index construction on SUPPLIED uniform deviates (the reference's RANDOM_NUMBER stream is an input):
    product (sml_makesparse_from_draws) = oracle (oracle/genres_oracle.c) = an independent numpy evaluation, bit for bit -- rows, cols
    and vals, at the config-2 shape (n = 5760, k = 33177: 5 full permutations + 4377), with k a multiple of n, and with k <= n."""
    import ctypes as C
    from speedy_ml_amd import _lib
    L = _lib.lib()
    L.sml_makesparse_draws.restype = C.c_long
    nd = L.sml_makesparse_draws(n, k)
    calls = 2 * (k // n) + (2 if k % n else 0) if k > n else 2
    assert nd == k + calls * n
    rng = np.random.default_rng(n * 7 + k)
    draws = rng.random(nd)
    draws[k] = 0.0                                   # the edges of [0, 1): first and (almost) last choice
    draws[k + 1] = np.nextafter(1.0, 0.0)
    rows, cols, vals = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32), np.zeros(k)
    _lib.check(L.sml_makesparse_from_draws(n, k, _lib.dp(draws), C.c_long(nd), _lib.ip(rows), _lib.ip(cols), _lib.dp(vals)))
    ro, co, vo = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32), np.zeros(k)
    used = oracle.lib.go_makesparse(n, k, draws.ctypes.data_as(C.POINTER(C.c_double)), ro.ctypes.data_as(C.POINTER(C.c_int32)),
                                    co.ctypes.data_as(C.POINTER(C.c_int32)), vo.ctypes.data_as(C.POINTER(C.c_double)))
    assert used == nd
    assert np.array_equal(rows, ro) and np.array_equal(cols, co) and np.array_equal(vals, vo)
    if n <= 300:
        rn, cn, vn = _makesparse_numpy(n, k, draws)
        assert np.array_equal(ro, rn) and np.array_equal(co, cn) and np.array_equal(vo, vn)
    # too few deviates: refused, not read past the end
    with pytest.raises(_lib.SmlError):
        _lib.check(L.sml_makesparse_from_draws(n, k, _lib.dp(draws), C.c_long(nd - 1), _lib.ip(rows), _lib.ip(cols), _lib.dp(vals)))
    # a deviate outside [0, 1) would index past the shuffle's choice list: refused (1.0, a negative value, NaN), wherever it sits
    for pos, bad in ((k + 3, 1.0), (nd - 1, -0.25), (0, float("nan"))):
        spoiled = draws.copy()
        spoiled[pos] = bad
        with pytest.raises(_lib.SmlError, match="not in"):
            _lib.check(L.sml_makesparse_from_draws(n, k, _lib.dp(spoiled), C.c_long(nd), _lib.ip(rows), _lib.ip(cols), _lib.dp(vals)))


def test_spectral_radius_of_the_config2_matrix_against_arpack():
    """lambda_max of the n = 5760, k = 33177 matrix (BASELINE config 2) to 1e-8 against scipy.sparse.linalg.eigs (ARPACK 'LM', the
    routine sparse_eigen drives, src/mod_linalg.f90:351,405); quirk Q4 is not reproduced (it reads uninitialised memory)."""
    import ctypes as C
    from speedy_ml_amd import _lib
    n, k = 5760, 33177
    rows, cols, vals = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32), np.zeros(k)
    _lib.check(_lib.lib().sml_makesparse(n, k, C.c_uint64(20240954), _lib.ip(rows), _lib.ip(cols), _lib.dp(vals)))
    lam, it = C.c_double(), C.c_int()
    _lib.check(_lib.lib().sml_spectral_radius(n, k, _lib.ip(rows), _lib.ip(cols), _lib.dp(vals), C.c_double(1e-13), 2000, C.byref(lam), C.byref(it)))
    A = sp.coo_matrix((vals, (rows - 1, cols - 1)), shape=(n, n)).tocsr()
    ref = spla.eigs(A, k=4, ncv=20, which="LM", return_eigenvectors=False, maxiter=300)          # nev = 4, ncv = 20, maxitr = 300 as :246-260
    top = ref[np.argmax(np.abs(ref))]
    assert abs(top.imag) < 1e-10 and abs(lam.value - top.real) <= 1e-8 * abs(top.real), (lam.value, top)
    assert it.value < 2000
