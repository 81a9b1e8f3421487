"""CPU (build container only): the source-level drop-in boundary of SURVEY 8b.  The reference's own driver, src/parallelmain.f90, is
compiled IN PLACE -- unchanged, never copied -- against the drop-in modules of speedy-ml_amd/fortran/ (mpires, mod_reservoir,
mod_slab_ocean_reservoir, speedy_res_interface, resdomain, mod_utilities, mod_calendar) and linked with them and libspeedyml_hip.so:
every name it imports (src/parallelmain.f90:6-12), every argument list it calls and every derived-type field it touches must resolve.
The image has no Fortran MPI, so the test supplies a module named `mpi` with the two calls the driver makes itself
(fortran/test_support_mpi.f90); the NetCDF / ERA5 readers are the synthetic stand-ins of fortran/test_support.f90.
Nothing built from reference source is committed or sent to the GPU box (api_build/ is git-ignored, the two files are in
.gpurunignore)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "speedy-ml_amd", "fortran")
REF_MAIN = "/root/reference/src/parallelmain.f90"
FLANG = "/opt/rocm/bin/amdflang"

needs_reference = pytest.mark.skipif(not (os.path.exists(REF_MAIN) and os.path.exists(FLANG)),
                                     reason="needs /root/reference (build container) and amdflang")


@needs_reference
def test_reference_program_main_compiles_and_links_against_the_dropin():
    p = subprocess.run(["make", "-C", FDIR, "reference_main"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    exe = os.path.join(FDIR, "api_build", "reference_main")
    assert os.path.exists(exe)
    # no unresolved symbol other than what the shared libraries provide at load time
    undefined = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    mangled = [l.split()[-1] for l in undefined.splitlines() if "_QM" in l]          # flang-mangled module procedures / variables
    assert mangled == [], f"unresolved module entities: {mangled}"
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libspeedyml_hip.so" in ldd and "not found" not in ldd


@needs_reference
def test_every_name_the_driver_imports_is_exported():
    """the `use ..., only :` lists of the reference's driver against the module files the drop-in build produced"""
    subprocess.check_call(["make", "-C", FDIR, "api"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(REF_MAIN).read()
    uses = re.findall(r"^\s*use\s+(\w+)\s*,\s*only\s*:\s*(.*)$", text, flags=re.M | re.I)
    assert len(uses) >= 6
    checked = 0
    for module, names in uses:
        mod = os.path.join(FDIR, "api_build", module.lower() + ".mod")
        assert os.path.exists(mod), f"the driver uses module {module}: no drop-in"
        body = open(mod, errors="replace").read().lower()
        for name in [n.strip().lower() for n in names.split(",") if n.strip()]:
            assert re.search(r"\b" + re.escape(name) + r"\b", body), f"{module} does not export {name}"
            checked += 1
    assert checked >= 36          # 8 + 8 + 8 + 1 + 3 + 8 names in the six `only` lists


@needs_reference
def test_speedy_res_interface_exports_what_the_reference_tree_imports():
    """every `use speedy_res_interface, only : ...` of the reference tree (program main, dyn_stloop, ppo_iogrid, mod_reservoir,
    mod_slab_ocean_reservoir) against the drop-in's module: startspeedy, getspeedyvariable, write_restart_new,
    truncate_letkf_code_version, read_era, read_model_states (+ read_era_netcdf_opened)"""
    import glob
    subprocess.check_call(["make", "-C", FDIR, "api"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    body = open(os.path.join(FDIR, "api_build", "speedy_res_interface.mod"), errors="replace").read().lower()
    wanted = {"read_era_netcdf_opened"}
    for f in glob.glob("/root/reference/src/*.f90"):
        text = open(f, errors="replace").read()
        for names in re.findall(r"^\s*use\s+speedy_res_interface\s*,\s*only\s*:\s*((?:.*&\s*\n)*.*)$", text, flags=re.M | re.I):
            wanted |= {n.strip().lower() for n in names.replace("&", " ").replace("\n", " ").split(",") if n.strip()}
    assert {"startspeedy", "getspeedyvariable", "write_restart_new", "truncate_letkf_code_version", "read_era", "read_model_states"} <= wanted
    for name in sorted(wanted):
        assert re.search(r"\b" + re.escape(name) + r"\b", body), f"speedy_res_interface does not export {name}"
