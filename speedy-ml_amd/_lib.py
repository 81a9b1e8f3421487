"""ctypes loader for csrc/libspeedyml_hip.so (the C-ABI declared in include/speedyml_hip.h).

There is no CPU fallback: if the shared library is missing or a HIP call fails, the caller gets an
exception carrying sml_last_error().
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SML_LIB_PATH: a diagnostic build of the same library, e.g. csrc/libspeedyml_hip_span.so for profiles/micro/window_span.py)
LIB_PATH = os.environ.get("SML_LIB_PATH") or os.path.join(_HERE, "csrc", "libspeedyml_hip.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
_lib = None


class SmlError(RuntimeError):
    pass


class Region(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "res_xstart", "res_xend", "res_ystart", "res_yend", "resxchunk", "resychunk",
        "res_zstart", "res_zend", "reszchunk",
        "input_xstart", "input_xend", "input_ystart", "input_yend", "inputxchunk", "inputychunk",
        "input_zstart", "input_zend", "inputzchunk",
        "pole", "periodicboundary", "top", "bottom",
        "tdata_xstart", "tdata_xend", "tdata_ystart", "tdata_yend", "tdata_zstart", "tdata_zend")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ResSizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "chunk_size", "chunk_size_prediction", "chunk_size_speedy", "locality",
        "nodes_per_input", "n", "k", "reservoir_numinputs",
        "atmo3d_start", "atmo3d_end", "logp_start", "logp_end", "precip_start", "precip_end",
        "sst_start", "sst_end", "tisr_start", "tisr_end")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/speedyml_hip.h declares (checked by tests/test_cabi_symbols.py)
EXPORTS = [
    "sml_last_error", "sml_version", "sml_device_count", "sml_set_device", "sml_device_synchronize",
    "sml_stream_create_cu_mask", "sml_stream_destroy", "sml_dev_alloc", "sml_dev_free", "sml_dev_zero", "sml_dev_upload", "sml_dev_download",
    "sml_domain_decompose", "sml_domain_region_owner", "sml_domain_region", "sml_domain_sizes", "sml_domain_out_map", "sml_domain_in_map", "sml_domain_message_sizes", "sml_domain_target_map", "sml_find_closest_divisor", "sml_calendar_date", "sml_hours_into_year", "sml_tisr_index",
    "sml_bank_create", "sml_bank_destroy", "sml_bank_load", "sml_bank_load_sparse_win", "sml_bank_set_wout",
    "sml_bank_set_state", "sml_bank_get_state", "sml_bank_set_feedback", "sml_bank_set_local_model",
    "sml_bank_get_outvec", "sml_bank_feedback_dev", "sml_bank_local_model_dev", "sml_bank_outvec_dev",
    "sml_bank_predict_all", "sml_bank_predict_one", "sml_bank_synchronize_all", "sml_bank_synchronize_one", "sml_bank_advance_all", "sml_bank_readout_part", "sml_bank_outvec_contribs", "sml_bank_get_contribs",
    "sml_bank_algorithmic_bytes", "sml_bank_storage", "sml_bank_use_compact", "sml_bank_readout_part_bytes", "sml_bank_timing", "sml_bank_timing_collect",
    "sml_exchange_create", "sml_exchange_destroy", "sml_exchange_scatter", "sml_exchange_gather",
    "sml_comm_unique_id", "sml_comm_create", "sml_comm_bootstrap", "sml_comm_destroy", "sml_comm_allgather_outvec", "sml_comm_unpack_regions",
    "sml_hybrid_create", "sml_hybrid_destroy", "sml_hybrid_set_state", "sml_hybrid_get_state", "sml_hybrid_set_base_sst", "sml_hybrid_set_orography",
    "sml_hybrid_set_tisr_table", "sml_hybrid_get_phis0", "sml_hybrid_update_surface", "sml_hybrid_set_fordate_fields", "sml_hybrid_attach_physics", "sml_hybrid_initial_inputs", "sml_hybrid_exchange_and_speedy", "sml_hybrid_safe", "sml_hybrid_timing", "sml_hybrid_timing_collect",
    "sml_hybrid_g_dev", "sml_hybrid_f_dev", "sml_hybrid_attach_slab", "sml_hybrid_set_comm", "sml_hybrid_restart", "sml_hybrid_slab_due", "sml_hybrid_step", "sml_hybrid_step_predict", "sml_hybrid_step_finish",
    "sml_slab_sizes", "sml_slab_create", "sml_slab_destroy", "sml_slab_scatter_sst", "sml_slab_predict_hybrid", "sml_slab_update_inputs",
    "sml_exchange_pack_outvec", "sml_handoff_to_fields", "sml_handoff_from_fields", "sml_handoff_check",
    "sml_spectral_create", "sml_spectral_destroy", "sml_spectral_get_table", "sml_spectral_grid",
    "sml_spectral_spec", "sml_spectral_grid_mixed", "sml_spectral_grid_derived", "sml_spectral_grid_derived_aux", "sml_spectral_spec_post", "sml_spectral_spec_post_split", "sml_spectral_spec_mixed", "sml_spectral_vdspec", "sml_spectral_uvspec", "sml_spectral_vds", "sml_spectral_grad",
    "sml_spectral_lap", "sml_spectral_invlap", "sml_spectral_trunct",
    "parmtr_", "inifft_", "grid_", "spec_", "vdspec_", "uvspec_", "vds_", "grad_", "lap_", "invlap_", "trunct_",
    "sml_dyn_create", "sml_dyn_destroy", "sml_dyn_impint", "sml_dyn_get_table", "sml_dyn_set_boundary", "sml_dyn_boundary_dev", "sml_dyn_state_dev",
    "sml_dyn_set_state_host", "sml_dyn_get_state_host", "sml_dyn_set_boundary_host", "sml_dyn_grtend",
    "sml_dyn_spectral_step", "sml_dyn_step", "sml_dyn_window", "sml_dyn_attach_physics", "sml_dyn_set_lradsw", "sml_dyn_set_range_guard", "sml_dyn_physics_diag", "sml_dyn_select_physics_form", "sml_dyn_select_window_form",
    "sml_phys_create", "sml_phys_destroy", "sml_phys_set_surface", "sml_phys_set_sst_dev", "sml_phys_bind_sst_dev", "sml_phys_sol_oz", "sml_phys_sol_oz_async", "sml_phys_get_tables",
    "sml_phys_tendencies", "sml_phys_tendencies_sfcwind", "sml_phys_diag", "sml_phys_set_fordate_fields", "sml_phys_update_surface", "sml_phys_fordate", "sml_phys_get_surface",
    "sml_makesparse", "sml_makesparse_draws", "sml_makesparse_from_draws", "sml_spectral_radius", "sml_gen_res", "sml_bank_train_pass",
    "sml_train_accumulate", "sml_train_symmetrize", "sml_train_fit", "sml_train_fit_batched", "sml_train_select_solver", "sml_train_release_workspace",
]


def lib():
    """Load (once) and return the C-ABI library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmlError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; "
                       "there is no CPU fallback for the product path")
    L = C.CDLL(LIB_PATH)
    L.sml_last_error.restype = C.c_char_p
    for name in ("sml_bank_feedback_dev", "sml_bank_local_model_dev", "sml_bank_outvec_dev", "sml_hybrid_g_dev", "sml_hybrid_f_dev", "sml_dyn_boundary_dev"):
        getattr(L, name).restype = C.c_void_p
        getattr(L, name).argtypes = [C.c_void_p]
    _lib = L
    # (CU-masked streams -- the ridge solver's, sml_stream_create_cu_mask's -- are released by an exit handler the LIBRARY registers
    # at the first such stream, csrc/bank.hip, so that non-Python hosts are covered too; nothing to do here.)
    return L


def check(rc):
    if rc < 0:
        raise SmlError(f"libspeedyml_hip: status {rc}: {lib().sml_last_error().decode()}")
    return rc


def dp(a):
    """double* of a contiguous float64 numpy array, or a raw device address (int)."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.cast(C.c_void_p(int(a)), c_dp)
    assert a.dtype == np.float64 and (a.flags.c_contiguous or a.flags.f_contiguous)
    return a.ctypes.data_as(c_dp)


def ip(a):
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.cast(C.c_void_p(int(a)), c_ip)
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_ip)


def vp(stream):
    """hipStream_t as void* from None | int | torch.cuda.Stream."""
    if stream is None:
        return C.c_void_p(0)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


def device_view(ptr, shape):
    """torch view (no copy) of device memory owned by the C-ABI library."""
    import torch

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(np.prod(shape)),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda").view(*shape)
