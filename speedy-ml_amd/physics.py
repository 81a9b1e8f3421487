"""Host-side mirror of SPEEDY's column physics (src/phy_phypar.f90 grid-point part and the phy_*.f90 parametrisations) over the
C-ABI.  Grids are float64 CUDA tensors [nf, 48, 96]; surface fields are host numpy arrays (48, 96).  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, vp

HSG = np.array([0.000, 0.050, 0.140, 0.260, 0.420, 0.600, 0.770, 0.900, 1.000])      # src/ini_indyns.f90:38-41
NSTRAD = 3                                                                               # src/mod_tsteps.f90:65
DIAG = {n: i for i, n in enumerate(("precnv", "precls", "cbmf", "ts", "tskin", "ssrd", "slrd", "olr", "shf", "evap", "ustr", "vstr",
                                    "cloudc", "clstr", "tsr", "ssr", "slr", "hfluxn_land", "hfluxn_sea", "t0", "q0", "iptop", "icltop"))}


class Physics:
    def __init__(self, rlat, hsg=HSG):
        """rlat: the 48 Gaussian latitudes in radians, south to north (radang of src/ini_indyns.f90:72-80)"""
        h = C.c_void_p()
        hsg, rlat = np.ascontiguousarray(hsg, dtype=np.float64), np.ascontiguousarray(rlat, dtype=np.float64)
        check(_lib.lib().sml_phys_create(dp(hsg), dp(rlat), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            _lib.lib().sml_phys_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_surface(self, fmask, phis0, tland, tsea, swav, alb_l, alb_s, albsfc, snowc):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (fmask, phis0, tland, tsea, swav, alb_l, alb_s, albsfc, snowc)]
        assert all(x.shape == (48, 96) for x in a)
        check(_lib.lib().sml_phys_set_surface(self._h, *[dp(x) for x in a]))

    def set_sst(self, tsea_dev, stream=None):
        assert tsea_dev.is_cuda and tsea_dev.numel() == 4608 and tsea_dev.is_contiguous()
        check(_lib.lib().sml_phys_set_sst_dev(self._h, dp(tsea_dev.data_ptr()), vp(stream)))

    def bind_sst(self, tsea_dev):
        """read the sea temperature in place from a persistent device tensor (None: back to the handle's copy)"""
        if tsea_dev is not None:
            assert tsea_dev.is_cuda and tsea_dev.numel() == 4608 and tsea_dev.is_contiguous() and tsea_dev.element_size() == 8
            self._sst_keep = tsea_dev
        check(_lib.lib().sml_phys_bind_sst_dev(self._h, dp(tsea_dev.data_ptr()) if tsea_dev is not None else None))

    def set_fordate_fields(self, fmask_s, alb0=None, snowd_am=None, sice_am=None):
        """what fordate reads besides the surface fields (src/ini_fordate.f90): the sea fraction fmask_s and -- optionally, all or
        none -- the bare-land albedo, snow depth and sea-ice fraction from which it recomputes snowc / alb_l / alb_s / albsfc"""
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (fmask_s, alb0, snowd_am, sice_am)]
        assert all(x is None or x.shape == (48, 96) for x in a)
        self._fordate_keep = a
        check(_lib.lib().sml_phys_set_fordate_fields(self._h, *[dp(x) for x in a]))

    def update_surface(self, tland=None, swav=None, snowd_am=None, sice_am=None):
        """the coupler's daily output between two windows; (48, 96) arrays, None = unchanged"""
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (tland, swav, snowd_am, sice_am)]
        assert all(x is None or x.shape == (48, 96) for x in a)
        check(_lib.lib().sml_phys_update_surface(self._h, *[dp(x) for x in a]))

    def fordate(self, spectral, corh_spec, stream=None):
        """fordate(0) on the device: albedos (when their inputs were given), then tcorh | qcorh into corh_spec: a [2, 32, 62] CUDA
        tensor or a raw device address (Dynamics.boundary_ptr() + 32 * 62 * 8)"""
        if hasattr(corh_spec, "data_ptr"):
            assert corh_spec.is_cuda and corh_spec.is_contiguous() and tuple(corh_spec.shape) == (2, 32, 62) and corh_spec.element_size() == 8
            corh_spec = corh_spec.data_ptr()
        check(_lib.lib().sml_phys_fordate(self._h, spectral._h, dp(int(corh_spec)), vp(stream)))

    SURFACE = {n: i for i, n in enumerate(("fmask", "phis0", "tland", "tsea", "swav", "alb_l", "alb_s", "albsfc", "snowc", "forog", "corh_t", "corh_q"))}

    def surface(self, name):
        out = np.zeros((48, 96))
        check(_lib.lib().sml_phys_get_surface(self._h, self.SURFACE[name], dp(out)))
        return out

    def sol_oz(self, tyear, stream=None, asynchronous=False):
        """sol_oz(tyear) of the forcing day; asynchronous: enqueued on `stream` (the values travel as kernel arguments) instead of a blocking copy"""
        if asynchronous:
            check(_lib.lib().sml_phys_sol_oz_async(self._h, C.c_double(tyear), vp(stream)))
        else:
            check(_lib.lib().sml_phys_sol_oz(self._h, C.c_double(tyear)))

    def tables(self):
        z, f, l = np.zeros((6, 48)), np.zeros((301, 4)), np.zeros((9, 9))
        check(_lib.lib().sml_phys_get_tables(self._h, dp(z), dp(f), dp(l)))
        return z, f, l

    def tendencies(self, grids, lradsw, tend, off=(0, 8, 16, 24), accumulate=False, stream=None):
        assert grids.is_cuda and grids.is_contiguous() and tuple(grids.shape) == (41, 48, 96)
        assert tend.is_cuda and tend.is_contiguous() and tend.shape[0] >= max(off) + 8
        check(_lib.lib().sml_phys_tendencies(self._h, dp(grids.data_ptr()), 1 if lradsw else 0, dp(tend.data_ptr()), *[int(o) for o in off],
                                             1 if accumulate else 0, vp(stream)))
        return tend

    def tendencies_sfcwind(self, grids, lradsw, tend, off=(0, 8, 16, 24), accumulate=False, want_diag=True, stream=None):
        """The same from the 27 grids the parametrisations read: ug1(:,kx) vg1(:,kx) tg1(8) qg1(8) phig1(8) pslg1"""
        assert grids.is_cuda and grids.is_contiguous() and tuple(grids.shape) == (27, 48, 96)
        assert tend.is_cuda and tend.is_contiguous() and tend.shape[0] >= max(off) + 8
        check(_lib.lib().sml_phys_tendencies_sfcwind(self._h, dp(grids.data_ptr()), 1 if lradsw else 0, dp(tend.data_ptr()),
                                                     *[int(o) for o in off], 1 if accumulate else 0, 1 if want_diag else 0, vp(stream)))
        return tend

    def diag(self, name):
        out = np.zeros((48, 96))
        check(_lib.lib().sml_phys_diag(self._h, DIAG[name], dp(out)))
        return out
