"""CPU: the C-ABI library loads without a GPU and exports every symbol include/speedyml_hip.h declares;
host-only entry points work; device entry points fail loudly (no silent CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from speedy_ml_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "speedyml_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|long|void|double|char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 60
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/speedyml_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names


F77_EXTERNALS = {"parmtr_", "inifft_", "grid_", "spec_", "vdspec_", "uvspec_", "vds_", "grad_", "lap_", "invlap_", "trunct_"}


def test_nothing_but_the_abi_is_exported():
    """nm -D: the dynamic symbol table holds the sml_* entry points and the eleven F77 externals of the reference's spectral files,
    nothing else (csrc/exports.map) -- no C++ helper, template instantiation or kernel stub leaks into the ABI"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    stray = sorted(n for n in names if not n.startswith("sml_") and n not in F77_EXTERNALS)
    assert not stray, stray
    assert F77_EXTERNALS <= names
    # (a handful of sml_*_debug_stamps profiling aids are exported without a declaration in include/; everything declared is there)
    assert set(declared_symbols()) <= names


def test_host_only_entry_points():
    L = _lib.lib()
    assert L.sml_version() >= 100
    assert L.sml_device_count() >= 0
    buf = np.zeros(200, dtype=np.int32)
    assert L.sml_domain_decompose(3, 8, 1152, _lib.ip(buf), 200) == 144 and buf[0] == 432 and buf[143] == 575


def test_device_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speedy_ml_amd.reservoir import ReservoirBank
    from speedy_ml_amd.spectral import Spectral
    with pytest.raises(_lib.SmlError):
        ReservoirBank(4)
    with pytest.raises(_lib.SmlError):
        Spectral()


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libspeedyml_hip.so")
    with pytest.raises(_lib.SmlError):
        _lib.lib()
