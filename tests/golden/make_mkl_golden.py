#!/usr/bin/env python3
"""Known answers of the third-party arithmetic the reference's reservoir path leans on, from the REAL library: Intel MKL as installed in
the build container (/opt/conda/lib/libmkl_rt.so.1), called through ctypes with the reference's own call shapes --

    mkl_sparse_d_create_coo + mkl_sparse_d_mv   src/mod_linalg.f90:17, src/mod_reservoir.f90:1444   (1-based COO, duplicates present)
    dgemm_                                        src/mod_reservoir.f90:1695 (DGEMM('N','N',..., aug, n, transpose(aug), m, ...))
    dgesv_                                        src/mod_linalg.f90:145 through mldivide / fit_chunk_hybrid :1297-1313

Inputs: tests/golden/mkl_inputs.py (seeded).  Output: tests/golden/mkl_golden.npz (outputs only + the MKL version string).
This pins the LIBRARY semantics the oracle restates (oracle/reservoir_oracle.c: duplicates accumulate in a COO product, DGEMM, LU with
partial pivoting); it does not pin the Fortran around the calls, which cannot be built here.
Run: MKL_NUM_THREADS=1 python3 tests/golden/make_mkl_golden.py   (MKL is not on the GPU box; the fixture travels instead)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mkl_inputs  # noqa: E402

os.environ.setdefault("MKL_NUM_THREADS", "1")
mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so.1", mode=C.RTLD_GLOBAL)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
P = lambda a: a.ctypes.data_as(dp)
I = lambda a: a.ctypes.data_as(ip)


class MatrixDescr(C.Structure):           # struct matrix_descr { sparse_matrix_type_t type; sparse_fill_mode_t mode; sparse_diag_type_t diag; }
    _fields_ = [("type", C.c_int), ("mode", C.c_int), ("diag", C.c_int)]


SPARSE_INDEX_BASE_ONE, SPARSE_OPERATION_NON_TRANSPOSE, SPARSE_MATRIX_TYPE_GENERAL = 1, 10, 20


def version():
    buf = C.create_string_buffer(256)
    mkl.mkl_get_version_string(buf, 256)
    return buf.value.decode().strip()


def spmv():
    rows, cols, vals, x = mkl_inputs.spmv_inputs()
    n, k = mkl_inputs.N, mkl_inputs.K
    handle = C.c_void_p()
    mkl.mkl_sparse_d_create_coo.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, dp]
    st = mkl.mkl_sparse_d_create_coo(C.byref(handle), SPARSE_INDEX_BASE_ONE, n, n, k, I(rows), I(cols), P(vals))
    assert st == 0, st
    y = np.full(n, 7.0)                    # beta = 0: whatever y held must not matter
    mkl.mkl_sparse_d_mv.argtypes = [C.c_int, C.c_double, C.c_void_p, MatrixDescr, dp, C.c_double, dp]
    st = mkl.mkl_sparse_d_mv(SPARSE_OPERATION_NON_TRANSPOSE, 1.0, handle, MatrixDescr(SPARSE_MATRIX_TYPE_GENERAL, 0, 0), P(x), 0.0, P(y))
    assert st == 0, st
    mkl.mkl_sparse_destroy.argtypes = [C.c_void_p]
    mkl.mkl_sparse_destroy(handle)
    return y


def gemm():
    model, states, y = mkl_inputs.gemm_inputs()
    aug = np.asfortranarray(np.vstack([model, states]))                 # aug(1:132,:) = model, aug(133:,:) = states
    n, m = aug.shape
    augt = np.asfortranarray(aug.T.copy())                              # transpose(aug): an (m, n) array, leading dimension m
    temp = np.zeros((n, n), order="F")
    one, zero = C.c_double(1.0), C.c_double(0.0)
    ci = lambda v: C.byref(C.c_int(v))
    mkl.dgemm_(C.c_char_p(b"N"), C.c_char_p(b"N"), ci(n), ci(n), ci(m), C.byref(one), P(aug), ci(n), P(augt), ci(m), C.byref(zero), P(temp), ci(n))
    # B's update is the compiler's matmul(targets, transpose(aug)) in the reference; DGEMM('N','T') gives the library's answer for it
    no = y.shape[0]
    tb = np.zeros((no, n), order="F")
    mkl.dgemm_(C.c_char_p(b"N"), C.c_char_p(b"T"), ci(no), ci(n), ci(m), C.byref(one), P(y), ci(no), P(aug), ci(n), C.byref(zero), P(tb), ci(no))
    return temp, tb


def gesv():
    n, n_model, n_out, c, b, beta_res, beta_model = mkl_inputs.gesv_inputs()
    n_aug = n + n_model
    a = c.copy(order="F")
    a[np.arange(n_model), np.arange(n_model)] += beta_model ** 2       # using_prior: the betas enter squared (src/mod_reservoir.f90:1275-1282)
    a[np.arange(n_model, n_aug), np.arange(n_model, n_aug)] += beta_res ** 2
    a_trans = np.asfortranarray(a.T.copy())
    b_trans = np.asfortranarray(b.T.copy())                             # prior_val = 0: nothing added
    ipiv = np.zeros(n_aug, dtype=np.int32)
    info = C.c_int(-1)
    ci = lambda v: C.byref(C.c_int(v))
    mkl.dgesv_(ci(n_aug), ci(n_out), P(a_trans), ci(n_aug), I(ipiv), P(b_trans), ci(n_aug), C.byref(info))
    assert info.value == 0, info.value
    return np.asfortranarray(b_trans.T.copy()), ipiv                    # wout = transpose(b_trans)


if __name__ == "__main__":
    y = spmv()
    temp, tb = gemm()
    wout, ipiv = gesv()
    out = os.path.join(HERE, "mkl_golden.npz")
    np.savez_compressed(out, spmv_y=y, gemm_c=temp, gemm_b=tb, gesv_wout=wout, gesv_ipiv=ipiv, mkl_version=np.array(version()))
    print("wrote", out, os.path.getsize(out), "bytes;", version())
    print("ipiv != identity at", int(np.sum(ipiv != np.arange(1, len(ipiv) + 1))), "of", len(ipiv), "rows")
