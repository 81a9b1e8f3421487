"""Host-side mirror of the device-resident part of mpires::sendrecievegrid (src/mpires.f90:218-804) and of the
iogrid 30/31 hand-off (src/ppo_iogrid.f90:497-601) over the C-ABI.  Torch tensors are used as device buffers only."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, ip, vp


class Exchange:
    def __init__(self, bank, number_of_regions, region_of_slot, sst_input_of_slot, overlap=1, precip_bool=True):
        self.bank = bank
        self.nreg = number_of_regions
        ros = np.ascontiguousarray(region_of_slot, dtype=np.int32)
        sst = np.ascontiguousarray(sst_input_of_slot, dtype=np.int32)
        h = C.c_void_p()
        check(_lib.lib().sml_exchange_create(bank._h, number_of_regions, ip(ros), len(ros), overlap, int(precip_bool), ip(sst), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            _lib.lib().sml_exchange_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pack_outvec(self, all_outvec, stream=None):
        check(_lib.lib().sml_exchange_pack_outvec(self._h, dp(all_outvec.data_ptr()), vp(stream)))

    def scatter(self, all_outvec, g, base_sst=None, sea_mask=None, stream=None):
        check(_lib.lib().sml_exchange_scatter(self._h, dp(all_outvec.data_ptr()), dp(g.data_ptr()),
                                              dp(base_sst.data_ptr()) if base_sst is not None else None,
                                              ip(sea_mask.data_ptr()) if sea_mask is not None else None, vp(stream)))

    def gather(self, g, f=None, stream=None):
        """g -> feedback of every slot, f -> local_model of every slot; either may be None to skip that half."""
        check(_lib.lib().sml_exchange_gather(self._h, dp(g.data_ptr()) if g is not None else None,
                                             dp(f.data_ptr()) if f is not None else None, vp(stream)))


def handoff_to_fields(g, fields, stream=None):
    check(_lib.lib().sml_handoff_to_fields(dp(g.data_ptr()), dp(fields.data_ptr()), vp(stream)))


def handoff_from_fields(fields, f, stream=None):
    check(_lib.lib().sml_handoff_from_fields(dp(fields.data_ptr()), dp(f.data_ptr()), vp(stream)))


def handoff_check(fields, safe, stream=None):
    check(_lib.lib().sml_handoff_check(dp(fields.data_ptr()), ip(safe.data_ptr()), vp(stream)))
