"""Host-side mirror of the slab-ocean coupling in sendrecievegrid (src/mpires.f90:286-330, 470-484, 756-790) and of the sizing in
initialize_slab_ocean_model (src/mod_slab_ocean_reservoir.f90:9-133), over the C-ABI.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ResSizes, check, dp, ip, vp

TIMESTEP_SLAB = 168            # hours between slab predictions (model_parameters%timestep_slab, src/mod_reservoir.f90:60)
SLAB_M, SLAB_DEG = 4000, 6     # src/mod_slab_ocean_reservoir.f90:27-29
SLAB_SIGMA, SLAB_RADIUS = 0.6, 0.9


def slab_sizes(g, m=SLAB_M, deg=SLAB_DEG, local_predictvars=4):
    out = ResSizes()
    check(_lib.lib().sml_slab_sizes(C.byref(g), m, deg, local_predictvars, C.byref(out)))
    return out


class SlabCoupler:
    def __init__(self, atmo_bank, slab_bank, number_of_regions, regions, sea_of_slot, atmo_sst_input_of_slot, timestep=6):
        self.ring = TIMESTEP_SLAB // timestep - 1
        self.every = TIMESTEP_SLAB // timestep
        r = np.ascontiguousarray(regions, dtype=np.int32)
        sea = np.ascontiguousarray(sea_of_slot, dtype=np.int32)
        sst = np.ascontiguousarray(atmo_sst_input_of_slot, dtype=np.int32)
        h = C.c_void_p()
        check(_lib.lib().sml_slab_create(atmo_bank._h, slab_bank._h, number_of_regions, ip(r), len(r), ip(sea), ip(sst), self.ring, C.byref(h)))
        self._h = h
        self._keep = (atmo_bank, slab_bank)

    def close(self):
        if self._h:
            _lib.lib().sml_slab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def due(self, timestep, timestep_hours=6):
        """mod(t*timestep, timestep_slab) == 0 (src/parallelmain.f90:238)"""
        return (timestep * timestep_hours) % TIMESTEP_SLAB == 0

    def scatter_sst(self, all_slab_out, sea_of_region, g, stream=None):
        assert all_slab_out.is_cuda and all_slab_out.is_contiguous() and sea_of_region.element_size() == 4
        check(_lib.lib().sml_slab_scatter_sst(self._h, dp(all_slab_out.data_ptr()), all_slab_out.shape[1], ip(sea_of_region.data_ptr()),
                                              dp(g.data_ptr()), vp(stream)))

    def update_inputs(self, timestep, stream=None):
        check(_lib.lib().sml_slab_update_inputs(self._h, int(timestep), vp(stream)))


def predict_slab(slab_bank, stream=None):
    """predict_slab (src/mod_slab_ocean_reservoir.f90:1268-1316) for every slot of a slab bank loaded with n_model = n_out: the raw
    output is fed back as the next local_model, the bank's outvec holds the un-standardised SST / OHTC."""
    check(_lib.lib().sml_slab_predict_hybrid(slab_bank._h, vp(stream)))
