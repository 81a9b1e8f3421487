"""speedy-ml_amd: MI355X-native (gfx950) implementation of the SPEEDY-ML hybrid-step hot path.

The directory name carries a hyphen (it mirrors the reference's name), so it is loaded under the
import name ``speedy_ml_amd`` by ``__graft_entry__.load_package()``.

Layout:
  csrc/      hand-written HIP kernels + the C-ABI shared library (include/speedyml_hip.h)
  _lib.py    ctypes loader for csrc/libspeedyml_hip.so -- raises loudly when the library is missing
  reservoir.py / domain.py / spectral.py / hybrid.py   host-side mirrors of the reference's module API
  fortran/   iso_c_binding module + drop-in wrappers for the reference's Fortran driver
  synth.py   seeded synthetic inputs (no ERA5 in the image)
Importing the package never touches the GPU; the shared library is loaded on first use.
"""
__all__ = ["synth"]
