"""Host-side mirror of the reference's `mod_reservoir` prediction API over the C-ABI.

`ReservoirBank` holds every local reservoir of a rank resident in HBM; `predict` / `synchronize` keep the
reference's names (src/mod_reservoir.f90:1354-1489) but act on all resident reservoirs per call -- the
per-region loop of program main (src/parallelmain.f90:226-251) becomes one batched launch pair.
No CPU fallback: everything here runs through libspeedyml_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, ip, vp


class ReservoirBank:
    def __init__(self, capacity, max_d=576, max_n_model=132, max_n_out=136):
        self.capacity, self.max_d, self.max_n_model, self.max_n_out = capacity, max_d, max_n_model, max_n_out
        h = C.c_void_p()
        check(_lib.lib().sml_bank_create(capacity, max_d, max_n_model, max_n_out, C.byref(h)))
        self._h = h
        self.shapes = {}

    def close(self):
        if self._h:
            _lib.lib().sml_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- loading (trained_reservoir_prediction, src/mod_reservoir.f90:1783-1886) ----
    def load(self, slot, n, d, n_model, n_out, rows, cols, vals, win, wout, mean, std, out_stat_idx, leakage=1.0):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        win = np.asfortranarray(win, dtype=np.float64)
        wout = np.asfortranarray(wout, dtype=np.float64)
        assert win.shape == (n, d) and wout.shape == (n_out, n_model + n)
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        std = np.ascontiguousarray(std, dtype=np.float64)
        osi = None if out_stat_idx is None else np.ascontiguousarray(out_stat_idx, dtype=np.int32)
        check(_lib.lib().sml_bank_load(self._h, slot, n, d, len(vals), n_model, n_out, ip(rows), ip(cols), dp(vals),
                                       dp(win), dp(wout), C.c_double(leakage), dp(mean), dp(std), len(mean), ip(osi)))
        self.shapes[slot] = (n, d, n_model, n_out)

    def load_sparse_win(self, slot, n, d, n_model, n_out, rows, cols, vals, win_rows, win_cols, win_vals, wout,
                        mean, std, out_stat_idx, leakage=1.0):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        wr = np.ascontiguousarray(win_rows, dtype=np.int32)
        wc = np.ascontiguousarray(win_cols, dtype=np.int32)
        wv = np.ascontiguousarray(win_vals, dtype=np.float64)
        wout = np.asfortranarray(wout, dtype=np.float64)
        assert wout.shape == (n_out, n_model + n)
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        std = np.ascontiguousarray(std, dtype=np.float64)
        osi = None if out_stat_idx is None else np.ascontiguousarray(out_stat_idx, dtype=np.int32)
        check(_lib.lib().sml_bank_load_sparse_win(self._h, slot, n, d, len(vals), n_model, n_out, ip(rows), ip(cols),
                                                  dp(vals), len(wv), ip(wr), ip(wc), dp(wv), dp(wout),
                                                  C.c_double(leakage), dp(mean), dp(std), len(mean), ip(osi)))
        self.shapes[slot] = (n, d, n_model, n_out)

    def set_wout(self, slot, wout):
        check(_lib.lib().sml_bank_set_wout(self._h, slot, dp(np.asfortranarray(wout, dtype=np.float64))))

    # ---- host <-> device state ----
    def set_state(self, slot, x):
        check(_lib.lib().sml_bank_set_state(self._h, slot, dp(np.ascontiguousarray(x, dtype=np.float64))))

    def get_state(self, slot):
        x = np.zeros(self.shapes[slot][0])
        check(_lib.lib().sml_bank_get_state(self._h, slot, dp(x)))
        return x

    def set_feedback(self, slot, u):
        check(_lib.lib().sml_bank_set_feedback(self._h, slot, dp(np.ascontiguousarray(u, dtype=np.float64))))

    def set_local_model(self, slot, lm):
        check(_lib.lib().sml_bank_set_local_model(self._h, slot, dp(np.ascontiguousarray(lm, dtype=np.float64))))

    def get_outvec(self, slot):
        out = np.zeros(self.shapes[slot][3])
        check(_lib.lib().sml_bank_get_outvec(self._h, slot, dp(out)))
        return out

    @property
    def feedback_ptr(self):
        return _lib.lib().sml_bank_feedback_dev(self._h)

    @property
    def local_model_ptr(self):
        return _lib.lib().sml_bank_local_model_dev(self._h)

    @property
    def outvec_ptr(self):
        return _lib.lib().sml_bank_outvec_dev(self._h)

    # ---- the hot path ----
    def predict(self, raw=False, stream=None):
        """predict (src/mod_reservoir.f90:1418-1489) for every resident reservoir."""
        check(_lib.lib().sml_bank_predict_all(self._h, 1 if raw else 0, vp(stream)))

    def predict_one(self, slot, x, local_model):
        """Reference-shaped single call: returns (x_new, outvec); x and outvec live on the host."""
        x = np.array(x, dtype=np.float64).copy()
        out = np.zeros(self.shapes[slot][3])
        lm = None if local_model is None else np.ascontiguousarray(local_model, dtype=np.float64)
        check(_lib.lib().sml_bank_predict_one(self._h, slot, dp(x), dp(lm), dp(out)))
        return x, out

    def advance(self, stream=None):
        check(_lib.lib().sml_bank_advance_all(self._h, vp(stream)))

    def readout_part(self, part, raw=False, stream=None, persistent=False, drain=False):
        """The readout of predict in two column blocks: part 1 = reservoir-state columns (needs only the advanced state),
        part 2 = physics-model columns + un-standardisation -> outvec.  advance(); readout_part(1); readout_part(2) == predict()."""
        check(_lib.lib().sml_bank_readout_part(self._h, int(part), (1 if raw else 0) | (8 if persistent else 0) | (16 if drain else 0), vp(stream)))

    def outvec_contribs(self, stream=None):
        """predict's split readout (outvec_component_contribs, src/mod_reservoir.f90:1458-1461) for every slot: call after predict()"""
        check(_lib.lib().sml_bank_outvec_contribs(self._h, vp(stream)))

    def get_contribs(self, slot):
        """(v_p, v_ml) of one slot: the physics-model block and the reservoir-state block of the readout, standardised"""
        v_p, v_ml = np.zeros(self.shapes[slot][3]), np.zeros(self.shapes[slot][3])
        check(_lib.lib().sml_bank_get_contribs(self._h, slot, dp(v_p), dp(v_ml)))
        return v_p, v_ml

    def synchronize(self, inputs_dev_ptr, length, stream=None):
        """synchronize (src/mod_reservoir.f90:1354-1381); inputs: device [length][capacity][max_d]."""
        check(_lib.lib().sml_bank_synchronize_all(self._h, dp(int(inputs_dev_ptr)), length, vp(stream)))

    def compact(self):
        """True when the predict kernels read the compact (float) copies of W_out and of the operator's values: every loaded reservoir's
        weights are exactly floats, as after loading a reference weights file (sml_bank_storage)."""
        c = C.c_int()
        check(_lib.lib().sml_bank_storage(self._h, C.byref(c)))
        return bool(c.value)

    def use_compact(self, allow):
        """allow=False: keep to the 8-byte copies of the weights whatever they are (sml_bank_use_compact)"""
        check(_lib.lib().sml_bank_use_compact(self._h, 1 if allow else 0))

    def algorithmic_bytes(self):
        u, r = C.c_uint64(), C.c_uint64()
        check(_lib.lib().sml_bank_algorithmic_bytes(self._h, C.byref(u), C.byref(r)))
        return u.value, r.value

    def readout_part_bytes(self, part):
        r = C.c_uint64()
        check(_lib.lib().sml_bank_readout_part_bytes(self._h, int(part), C.byref(r)))
        return r.value

    def train_pass(self, noisy_inputs, discard, batch, models, targets, cs, bs, stream=None, ml_variant=False):
        """reservoir_layer_chunking_hybrid (src/mod_reservoir.f90:1067-1175; ml_variant: reservoir_layer_chunking_ml :963-1065,
        whose step after a batch flush feeds the squared column into A x, quirk Q6) for every loaded slot.
        noisy_inputs: device tensor [T, capacity, max_d]; models/targets/cs/bs: per-slot lists of device tensors
        (column-major buffers, see speedy_ml_amd.train) or None for slots to skip.  Returns #batches flushed."""
        T = noisy_inputs.shape[0]
        assert tuple(noisy_inputs.shape[1:]) == (self.capacity, self.max_d) and noisy_inputs.is_contiguous()

        def table(lst):
            arr = (C.c_void_p * self.capacity)()
            for i in range(self.capacity):
                t = lst[i] if i < len(lst) else None
                arr[i] = None if t is None else t.data_ptr()
            return arr
        tm, tt, tc, tb = table(models), table(targets), table(cs), table(bs)
        return check(_lib.lib().sml_bank_train_pass(self._h, dp(noisy_inputs.data_ptr()), T, discard, batch,
                                                    tm, tt, tc, tb, 1 if ml_variant else 0, vp(stream)))


def gen_res(n, k, radius, seed):
    """gen_res (src/mod_reservoir.f90:182-212): makesparse + spectral-radius rescale. Returns rows, cols, vals, eigs."""
    rows, cols = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32)
    vals = np.zeros(k)
    eigs = C.c_double()
    check(_lib.lib().sml_gen_res(n, k, C.c_double(radius), C.c_uint64(seed), ip(rows), ip(cols), dp(vals), C.byref(eigs)))
    return rows, cols, vals, eigs.value
