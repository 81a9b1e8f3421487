"""Host-side mirror of SPEEDY's time stepping: the adiabatic core and, with attach_physics, the column physics of grtend
(src/dyn_step.f90, dyn_grtend.f90, dyn_sptend.f90, dyn_implic.f90,
dyn_geop.f90, ini_indyns.f90, ini_impint.f90, ini_stepone.f90, dyn_stloop.f90) over the C-ABI.

Same names and argument meaning as the Fortran subroutines (impint, step, stepone, grtend); the model state lives on the
device as one float64 tensor state[2, 33, 32, 62]: time level (Fortran's last index of vor/div/t/ps), then the fields
vor(8) | div(8) | t(8) | tr(8) | ps, each (nx, mx2) = a Fortran complex (mx,nx) array.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, vp

KX, NX, MX, MX2, NSTATE = 8, 32, 31, 62, 33
F_VOR, F_DIV, F_T, F_TR, F_PS = 0, 8, 16, 24, 32
# src/mod_tsteps.f90:19,84-93
NSTEPS_PER_DAY = 96
DELT = 86400.0 / NSTEPS_PER_DAY
ROB, WIL, ALPH = 0.05, 0.53, 0.5
TABLES = {1: ("hsg", 9), 2: ("dhs", 8), 3: ("fsg", 8), 4: ("dhsr", 8), 5: ("fsgr", 8), 6: ("coriol", 48), 7: ("xgeop1", 8),
          8: ("xgeop2", 8), 9: ("dmp", 992), 10: ("dmpd", 992), 11: ("dmps", 992), 12: ("dmp1", 992), 13: ("dmp1d", 992),
          14: ("dmp1s", 992), 15: ("tcorv", 8), 16: ("qcorv", 8), 17: ("tref", 8), 18: ("tref1", 8), 19: ("tref2", 8),
          20: ("tref3", 8), 21: ("xc", 64), 22: ("xd", 64), 23: ("xj", 64 * 61), 24: ("dhsx", 8), 25: ("elz", 992), 26: ("alph", 1)}


def _chk(t, shape):
    assert t.is_cuda and t.element_size() == 8 and t.dtype.is_floating_point and t.is_contiguous(), "need a contiguous CUDA float64 tensor"
    assert tuple(t.shape) == shape, (tuple(t.shape), shape)
    return dp(t.data_ptr())


class Dynamics:
    def __init__(self, spectral):
        self.sp = spectral          # keeps the spectral handle alive
        h = C.c_void_p()
        check(_lib.lib().sml_dyn_create(spectral._h, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            _lib.lib().sml_dyn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self, which):
        out = np.zeros(TABLES[which][1])
        check(_lib.lib().sml_dyn_get_table(self._h, which, dp(out), out.size))
        return out

    def impint(self, dt, alph=ALPH):
        check(_lib.lib().sml_dyn_impint(self._h, C.c_double(dt), C.c_double(alph)))

    def set_boundary(self, phis, tcorh, qcorh, stream=None):
        check(_lib.lib().sml_dyn_set_boundary(self._h, _chk(phis, (NX, MX2)), _chk(tcorh, (NX, MX2)), _chk(qcorh, (NX, MX2)), vp(stream)))

    def boundary_ptr(self):
        """device address of the handle's own [3][32][62] = phis | tcorh | qcorh (Physics.fordate writes the last two in place)"""
        return int(_lib.lib().sml_dyn_boundary_dev(self._h))

    def boundary(self):
        """host copy (3, 32, 62) of phis | tcorh | qcorh as the time steps read them"""
        import torch
        from ._lib import device_view
        torch.cuda.synchronize()
        return device_view(self.boundary_ptr(), (3, NX, MX2)).cpu().numpy()

    def attach_physics(self, physics, nstrad=3):
        """grtend's physics call (src/dyn_grtend.f90:222-225): every later time step adds the column-physics tendencies of time
        level 1 to the grid-point tendencies.  nstrad: short-wave radiation every nstrad-th step (src/dyn_stloop.f90:39).
        None detaches."""
        self._physics = physics          # keeps the handle alive
        check(_lib.lib().sml_dyn_attach_physics(self._h, physics._h if physics is not None else None, int(nstrad)))

    @staticmethod
    def select_physics_form(fused):
        """3 (default): grtend's grid-point part and the physics as one launch of three wavefronts per 64 columns; True / 1: the
        two-wavefront launch of rounds 2-3; False / 0: two launches; 2: the one-wavefront fused launch (same bits in all four; 2 is the
        regression subject of the round-1 repeatability failure)"""
        check(_lib.lib().sml_dyn_select_physics_form(int(fused) if fused in (2, 3) else 1 if fused else 0))

    def physics_diag(self, on):
        """keep (default) or skip the physics' 2-D diagnostics during time steps"""
        check(_lib.lib().sml_dyn_physics_diag(self._h, 1 if on else 0))

    def set_range_guard(self, safe):
        """iogrid(30)'s range guard on the first time step's grids of every window that starts with stepone; safe: int32 device
        tensor holding 1 (cleared on violation), or None"""
        self._guard = safe
        check(_lib.lib().sml_dyn_set_range_guard(self._h, _lib.ip(safe.data_ptr()) if safe is not None else None))

    def set_lradsw(self, flag):
        """the module flag lradsw (src/mod_lflags.f90:22) seen by step()/grtend() and by the stepone part of the next window"""
        check(_lib.lib().sml_dyn_set_lradsw(self._h, 1 if flag else 0))

    def grtend(self, state, j2, out=None, stream=None):
        import torch
        out = torch.empty((NSTATE, NX, MX2), dtype=torch.float64, device=state.device) if out is None else out
        check(_lib.lib().sml_dyn_grtend(self._h, _chk(state, (2, NSTATE, NX, MX2)), int(j2), _chk(out, (NSTATE, NX, MX2)), vp(stream)))
        return out

    def spectral_step(self, state, tend, j1, j2, dt, alph=ALPH, rob=ROB, wil=WIL, stream=None):
        check(_lib.lib().sml_dyn_spectral_step(self._h, _chk(state, (2, NSTATE, NX, MX2)), _chk(tend, (NSTATE, NX, MX2)), int(j1), int(j2),
                                               C.c_double(dt), C.c_double(alph), C.c_double(rob), C.c_double(wil), vp(stream)))
        return tend

    def step(self, state, j1, j2, dt, alph=ALPH, rob=ROB, wil=WIL, stream=None):
        check(_lib.lib().sml_dyn_step(self._h, _chk(state, (2, NSTATE, NX, MX2)), int(j1), int(j2), C.c_double(dt), C.c_double(alph),
                                      C.c_double(rob), C.c_double(wil), vp(stream)))

    def window(self, state, nsteps, start=True, delt=DELT, alph=ALPH, rob=ROB, wil=WIL, stream=None):
        """stepone (when start) + nsteps leapfrog steps, enqueued by ONE call (src/ini_stepone.f90, src/dyn_stloop.f90:28-43)."""
        check(_lib.lib().sml_dyn_window(self._h, _chk(state, (2, NSTATE, NX, MX2)), 1 if start else 0, int(nsteps), C.c_double(delt),
                                        C.c_double(alph), C.c_double(rob), C.c_double(wil), vp(stream)))
