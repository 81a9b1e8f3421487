"""GPU: the Fortran host side (iso_c_binding module + drop-in mklsparse/synchronize/predict + F77 spectral externals)
against the reference's own statements written out in Fortran, compiled with amdflang and run on the MI355X."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "speedy-ml_amd", "fortran")


def test_fortran_driver_parity():
    exe = os.path.join(FDIR, "test_driver")
    if not os.path.exists(exe):
        assert shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang"), "no prebuilt driver and no amdflang"
        subprocess.check_call(["make", "-C", FDIR])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "FORTRAN HOST PARITY OK" in p.stdout
