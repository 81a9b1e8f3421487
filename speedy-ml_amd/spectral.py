"""Host-side mirror of the reference's spectral-transform externals (src/spe_spectral.f90) over the C-ABI.

Same names as the Fortran subroutines (grid, spec, vdspec, uvspec, vds, grad, lap, invlap, trunct); arrays are
torch CUDA float64 tensors shaped [nf, 32, 62] (spectral, = Fortran (mx2,nx) per field) and [nf, 48, 96] (grid,
= (ix,il)).  Every call is batched over nf fields.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, vp

REARTH = 6.371e6          # src/mod_dyncon1.f90:13
MX2, NX, IX, IL = 62, 32, 96, 48
TABLES = {1: ("sia", 24), 2: ("coa", 24), 3: ("wt", 24), 4: ("wght", 24), 5: ("cosg", 48), 6: ("cosgr", 48),
          7: ("cosgr2", 48), 8: ("el2", 992), 9: ("elm2", 992), 10: ("el4", 992), 11: ("trfilt", 992), 12: ("nsh2", 32),
          13: ("epsi", 1023), 14: ("repsi", 1023), 15: ("consq", 31), 16: ("gradx", 31), 17: ("gradym", 992),
          18: ("gradyp", 992), 19: ("uvdx", 992), 20: ("uvdym", 992), 21: ("uvdyp", 992), 22: ("vddym", 992),
          23: ("vddyp", 992), 24: ("cpol", 62 * 32 * 24), 26: ("sqrhlf", 1)}


class Spectral:
    def __init__(self, a=REARTH):
        h = C.c_void_p()
        check(_lib.lib().sml_spectral_create(C.c_double(a), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            _lib.lib().sml_spectral_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self, which):
        out = np.zeros(TABLES[which][1])
        check(_lib.lib().sml_spectral_get_table(self._h, which, dp(out), out.size))
        return out

    @staticmethod
    def _chk(t, inner):
        assert t.is_cuda and t.dtype.is_floating_point and t.element_size() == 8 and t.is_contiguous()
        assert tuple(t.shape[-2:]) == inner, (tuple(t.shape), inner)
        return t.shape[0] if t.dim() == 3 else 1

    def _new(self, like, nf, inner):
        import torch
        return torch.empty((nf,) + inner, dtype=torch.float64, device=like.device)

    def grid(self, vorm, kcos, out=None, stream=None):
        nf = self._chk(vorm, (NX, MX2))
        out = self._new(vorm, nf, (IL, IX)) if out is None else out
        check(_lib.lib().sml_spectral_grid(self._h, dp(vorm.data_ptr()), dp(out.data_ptr()), nf, int(kcos), vp(stream)))
        return out

    def spec(self, vorg, out=None, stream=None):
        nf = self._chk(vorg, (IL, IX))
        out = self._new(vorg, nf, (NX, MX2)) if out is None else out
        check(_lib.lib().sml_spectral_spec(self._h, dp(vorg.data_ptr()), dp(out.data_ptr()), nf, vp(stream)))
        return out

    def grid_mixed(self, vorm, kcos_flags, out=None, stream=None):
        """One launch for fields with different kcos (int32 device tensor of 1|2 per field)."""
        nf = self._chk(vorm, (NX, MX2))
        assert kcos_flags.numel() == nf and kcos_flags.is_cuda and kcos_flags.element_size() == 4
        out = self._new(vorm, nf, (IL, IX)) if out is None else out
        check(_lib.lib().sml_spectral_grid_mixed(self._h, dp(vorm.data_ptr()), dp(out.data_ptr()), nf,
                                                 _lib.ip(kcos_flags.data_ptr()), vp(stream)))
        return out

    def grid_derived(self, base, desc, out=None, stream=None, aux=None):
        """Inverse transforms of derived fields in one launch: desc is an int32 device tensor [nf, 4] of
        (type, src0, src1, kcos) rows indexing the fields of `base` ([., 32, 62]); type 0 plain, 1|2 ucos|vcos of
        uvspec(src0, src1), 3|4 d/dx|d/dy of grad(src0), 7 geopotential of level src1 from the 8 temperature levels at src0
        (src/dyn_geop.f90; needs aux = xgeop1(8) | xgeop2(8) | corf(8) | phis(32x62) as one float64 device vector)."""
        self._chk(base, (NX, MX2))
        nf = desc.shape[0]
        assert desc.is_cuda and desc.element_size() == 4 and desc.is_contiguous() and tuple(desc.shape) == (nf, 4)
        out = self._new(base, nf, (IL, IX)) if out is None else out
        if aux is None:
            check(_lib.lib().sml_spectral_grid_derived(self._h, dp(base.data_ptr()), _lib.ip(desc.data_ptr()), dp(out.data_ptr()), nf, vp(stream)))
        else:
            assert aux.is_cuda and aux.element_size() == 8 and aux.is_contiguous() and aux.numel() == 24 + NX * MX2
            check(_lib.lib().sml_spectral_grid_derived_aux(self._h, dp(base.data_ptr()), _lib.ip(desc.data_ptr()), dp(aux.data_ptr()),
                                                           dp(out.data_ptr()), nf, vp(stream)))
        return out

    def spec_post(self, spec_in, desc, out, stream=None):
        """Output fields from transformed fields in one launch: desc int32 [nf_out, 4] = (type, src0, src1, truncate); type 0
        field src0, 5|6 vor|div of vds(src0, src1)."""
        self._chk(spec_in, (NX, MX2))
        nf = desc.shape[0]
        assert desc.is_cuda and desc.element_size() == 4 and desc.is_contiguous() and tuple(desc.shape) == (nf, 4)
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (nf, NX, MX2)
        check(_lib.lib().sml_spectral_spec_post(self._h, dp(spec_in.data_ptr()), _lib.ip(desc.data_ptr()), dp(out.data_ptr()), nf, vp(stream)))
        return out

    def spec_post_split(self, spec_in, desc, out, out2, stream=None):
        """spec_post with the last out2.shape[0] output fields written to a second array (what the hybrid engine uses to put fordate's
        two spectra into the boundary arrays from iogrid(30)'s launch)."""
        self._chk(spec_in, (NX, MX2))
        nf, nf2 = out.shape[0], out2.shape[0]
        assert desc.is_cuda and desc.element_size() == 4 and desc.is_contiguous() and tuple(desc.shape) == (nf + nf2, 4)
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (nf, NX, MX2)
        assert out2.is_cuda and out2.is_contiguous() and tuple(out2.shape) == (nf2, NX, MX2)
        check(_lib.lib().sml_spectral_spec_post_split(self._h, dp(spec_in.data_ptr()), _lib.ip(desc.data_ptr()), dp(out.data_ptr()), nf, dp(out2.data_ptr()), nf2,
                                                      vp(stream)))
        return out, out2

    def spec_mixed(self, vorg, scale_flags, out=None, stream=None):
        """One launch for fields with different forward pre-scaling (0 none, 1 *cosgr, 2 *cosgr2 per field)."""
        nf = self._chk(vorg, (IL, IX))
        assert scale_flags.numel() == nf and scale_flags.is_cuda and scale_flags.element_size() == 4
        out = self._new(vorg, nf, (NX, MX2)) if out is None else out
        check(_lib.lib().sml_spectral_spec_mixed(self._h, dp(vorg.data_ptr()), dp(out.data_ptr()), nf,
                                                 _lib.ip(scale_flags.data_ptr()), vp(stream)))
        return out

    def vdspec(self, ug, vg, kcos, out=None, stream=None):
        nf = self._chk(ug, (IL, IX))
        assert self._chk(vg, (IL, IX)) == nf
        vor, div = out if out is not None else (self._new(ug, nf, (NX, MX2)), self._new(ug, nf, (NX, MX2)))
        check(_lib.lib().sml_spectral_vdspec(self._h, dp(ug.data_ptr()), dp(vg.data_ptr()), dp(vor.data_ptr()),
                                             dp(div.data_ptr()), nf, int(kcos), vp(stream)))
        return vor, div

    def _two(self, fn, a, b, out, stream):
        nf = self._chk(a, (NX, MX2))
        assert self._chk(b, (NX, MX2)) == nf
        o1, o2 = out if out is not None else (self._new(a, nf, (NX, MX2)), self._new(a, nf, (NX, MX2)))
        check(getattr(_lib.lib(), fn)(self._h, dp(a.data_ptr()), dp(b.data_ptr()), dp(o1.data_ptr()), dp(o2.data_ptr()), nf, vp(stream)))
        return o1, o2

    def uvspec(self, vorm, divm, out=None, stream=None):
        return self._two("sml_spectral_uvspec", vorm, divm, out, stream)

    def vds(self, ucosm, vcosm, out=None, stream=None):
        return self._two("sml_spectral_vds", ucosm, vcosm, out, stream)

    def grad(self, psi, out=None, stream=None):
        nf = self._chk(psi, (NX, MX2))
        o1, o2 = out if out is not None else (self._new(psi, nf, (NX, MX2)), self._new(psi, nf, (NX, MX2)))
        check(_lib.lib().sml_spectral_grad(self._h, dp(psi.data_ptr()), dp(o1.data_ptr()), dp(o2.data_ptr()), nf, vp(stream)))
        return o1, o2

    def lap(self, strm, out=None, stream=None):
        nf = self._chk(strm, (NX, MX2))
        out = self._new(strm, nf, (NX, MX2)) if out is None else out
        check(_lib.lib().sml_spectral_lap(self._h, dp(strm.data_ptr()), dp(out.data_ptr()), nf, vp(stream)))
        return out

    def invlap(self, vorm, out=None, stream=None):
        nf = self._chk(vorm, (NX, MX2))
        out = self._new(vorm, nf, (NX, MX2)) if out is None else out
        check(_lib.lib().sml_spectral_invlap(self._h, dp(vorm.data_ptr()), dp(out.data_ptr()), nf, vp(stream)))
        return out

    def trunct(self, vor, stream=None):
        nf = self._chk(vor, (NX, MX2))
        check(_lib.lib().sml_spectral_trunct(self._h, dp(vor.data_ptr()), nf, vp(stream)))
        return vor
