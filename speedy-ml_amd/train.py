"""Host-side mirror of the reference's chunked ridge-regression training (src/mod_reservoir.f90:1561-1701, 1235-1334)
over the C-ABI.  Names follow the reference: chunking_matmul accumulates states_x_states_aug / states_x_trainingdata_aug,
fit_chunk_hybrid solves for W_out.  All arrays are column-major fp64 device buffers (torch tensors created with the
transposed shape, i.e. a Fortran (r, c) matrix is a torch tensor of shape [c, r])."""
import ctypes as C

from . import _lib
from ._lib import check, dp, vp


def fortran_zeros(rows, cols, device="cuda"):
    """Device buffer for a column-major (rows, cols) matrix: torch shape [cols, rows]."""
    import torch
    return torch.zeros((cols, rows), dtype=torch.float64, device=device)


def chunking_matmul(states, model, y, c, b, stream=None):
    """C += aug aug^T (tiles on/below the diagonal), B += Y aug^T with aug = [model ; states].
    states: [m, n], model: [m, n_model] or None, y: [m, n_out], c: [n_aug, n_aug], b: [n_aug, n_out] (torch shapes)."""
    m, n = states.shape
    n_model = 0 if model is None else model.shape[1]
    n_out = y.shape[1]
    assert y.shape[0] == m and tuple(c.shape) == (n + n_model, n + n_model) and tuple(b.shape) == (n + n_model, n_out)
    check(_lib.lib().sml_train_accumulate(dp(states.data_ptr()), dp(model.data_ptr()) if n_model else None,
                                          dp(y.data_ptr()), n, n_model, n_out, m, dp(c.data_ptr()), dp(b.data_ptr()), vp(stream)))


def symmetrize(c, stream=None):
    check(_lib.lib().sml_train_symmetrize(dp(c.data_ptr()), c.shape[0], vp(stream)))


def fit_chunk_hybrid(c, b, n, n_model, n_out, beta_res=0.001, beta_model=1.0, prior_val=0.0, using_prior=True, stream=None):
    """Returns W_out as a device buffer of the column-major (n_out, n_aug) matrix (torch shape [n_aug, n_out])."""
    wout = fortran_zeros(n_out, n + n_model, device=c.device)
    check(_lib.lib().sml_train_fit(dp(c.data_ptr()), dp(b.data_ptr()), n, n_model, n_out, C.c_double(beta_res),
                                   C.c_double(beta_model), C.c_double(prior_val), int(using_prior), dp(wout.data_ptr()), vp(stream)))
    return wout


def fit_chunk_hybrid_batched(cs, bs, n, n_model, n_out, beta_res=0.001, beta_model=1.0, prior_val=0.0, using_prior=True, stream=None):
    """fit_chunk_hybrid for several reservoirs of one size class at once (up to 8 LU factorisations in flight)."""
    count = len(cs)
    wouts = [fortran_zeros(n_out, n + n_model, device=cs[0].device) for _ in range(count)]
    tc = (C.c_void_p * count)(*[t.data_ptr() for t in cs])
    tb = (C.c_void_p * count)(*[t.data_ptr() for t in bs])
    tw = (C.c_void_p * count)(*[t.data_ptr() for t in wouts])
    check(_lib.lib().sml_train_fit_batched(count, tc, tb, n, n_model, n_out, C.c_double(beta_res), C.c_double(beta_model),
                                           C.c_double(prior_val), int(using_prior), tw, vp(stream)))
    return wouts


def release_workspace():
    """Free the ridge solver's device scratch and streams (sml_train_release_workspace); the next fit allocates them again."""
    check(_lib.lib().sml_train_release_workspace())
