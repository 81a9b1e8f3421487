"""fordate(0) -- the forcing set-up the reference's hybrid repeats at the start of every 6-hour window (src/ini_agcm_init.f90:86 ->
src/ini_fordate.f90): surface albedos, tcorh = spec(gamlat phis0) and qcorh = spec(refrh1 (q_sat(tref, 1) - q_sat(tsfc, psfc))) from the
land temperature and the (ML-predicted) SST.

The checker is the reference's OWN routine: oracle/build_ref.sh compiles src/ini_fordate.f90 in place into oracle/_ref/libref_phy.so;
tests/golden/fordate_golden.npz holds its inputs and outputs (tests/golden/make_fordate_golden.py).
  CPU: the fixture still equals the compiled reference (when it is present) and a plain numpy reading of the 25 lines + the oracle's
       spec -- so a change of either shows up without a GPU.
  GPU: sml_phys_fordate (one grid-point kernel + one two-field spec) against the fixture and the compiled reference: albedos bit for
       bit, the spectral corrections <= 1e-12 of their maxima (device libm exp / pow differ from the host's in the last bits)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_fordate_golden import fordate_inputs  # noqa: E402
from make_physics_golden import HSG, TYEAR, gaussian_latitudes  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fordate_golden.npz")
KEYS = ("phis0", "fmask_l", "fmask_s", "stl_am", "sst_am", "alb0", "snowd_am", "sice_am")


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def test_fixture_inputs_are_the_seeded_inputs():
    gold, inp = np.load(GOLD), fordate_inputs()
    for k in KEYS:
        assert np.array_equal(gold["in_" + k], inp[k]), k


def test_fixture_equals_the_compiled_reference():
    from _oracle import RefPhys
    if not RefPhys.available():
        pytest.skip("oracle/_ref/libref_phy.so not present")
    gold = np.load(GOLD)
    out = RefPhys(HSG, gaussian_latitudes()).fordate(TYEAR, **{k: gold["in_" + k] for k in KEYS})
    for k, v in out.items():
        assert np.array_equal(v, gold[k]), k


def test_numpy_reading_of_fordate_matches_the_reference(oracle):
    """src/ini_fordate.f90:54-61,72-113 read line by line in numpy, transforms by the oracle's spec (itself pinned to the compiled
    spe_spectral.f90): guards the constants and the statement order the device kernel follows"""
    g = np.load(GOLD)
    i = {k: g["in_" + k] for k in KEYS}
    snowc = np.minimum(1.0, i["snowd_am"] / 60.0)
    alb_l = i["alb0"] + snowc * (0.60 - i["alb0"])
    alb_s = 0.07 + i["sice_am"] * (0.60 - 0.07)
    albsfc = alb_s + i["fmask_l"] * (alb_l - alb_s)
    for name, v in (("snowc", snowc), ("alb_l", alb_l), ("alb_s", alb_s), ("albsfc", albsfc)):
        assert np.array_equal(v.ravel(), g[name]), name
    gamlat = 6.0 / (1000.0 * 9.81)
    corh = gamlat * i["phis0"]
    assert rel(oracle.spec(corh.T), g["tcorh"]) <= 1e-13
    tsfc = i["fmask_l"] * i["stl_am"] + i["fmask_s"] * i["sst_am"]
    tref = tsfc + corh
    psfc = (tsfc / tref) ** (1.0 / (287.0 * gamlat))

    def qsat(ta, p):
        e = np.where(ta >= 273.16, 6.108e-3 * np.exp(17.269 * (ta - 273.16) / (ta - 35.86)), 6.108e-3 * np.exp(21.875 * (ta - 273.16) / (ta - 7.66)))
        return 622.0 * e / (p - 0.378 * e)
    corq = 0.7 * (qsat(tref, 1.0) - qsat(tsfc, psfc))
    assert rel(oracle.spec(corq.T), g["qcorh"]) <= 1e-12
    # the reference does not truncate either spectrum: the total-wavenumber-31 diagonal that trunct would clear is populated
    trfilt = np.repeat(oracle.table(11), 2, axis=0)
    assert np.max(np.abs(g["qcorh"][trfilt == 0])) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("with_albedo", [True, False])
def test_device_fordate_against_the_reference(with_albedo):
    import torch
    from _oracle import RefPhys
    from speedy_ml_amd.dynamics import Dynamics
    from speedy_ml_amd.physics import Physics
    from speedy_ml_amd.spectral import Spectral
    g = np.load(GOLD)
    i = {k: g["in_" + k] for k in KEYS}
    want = {k: g[k] for k in ("tcorh", "qcorh", "snowc", "alb_l", "alb_s", "albsfc")}
    if RefPhys.available():                                    # the reference itself, not only its recorded outputs
        live = RefPhys(HSG, gaussian_latitudes()).fordate(TYEAR, **i)
        assert all(np.array_equal(live[k], want[k]) for k in want)
    sp = Spectral()
    dyn = Dynamics(sp)
    ph = Physics(gaussian_latitudes())
    given = dict(alb_l=np.full((48, 96), 0.2), alb_s=np.full((48, 96), 0.07), albsfc=np.full((48, 96), 0.1), snowc=np.zeros((48, 96)))
    ph.set_surface(i["fmask_l"], i["phis0"], i["stl_am"], np.zeros((48, 96)), np.full((48, 96), 0.5), given["alb_l"], given["alb_s"], given["albsfc"],
                   given["snowc"])
    sst = torch.from_numpy(i["sst_am"].copy()).cuda()
    ph.bind_sst(sst)                                           # sst_am read in place, as the hybrid binds G's SST segment
    if with_albedo:
        ph.set_fordate_fields(i["fmask_s"], i["alb0"], i["snowd_am"], i["sice_am"])
    else:
        ph.set_fordate_fields(i["fmask_s"])
    z = torch.zeros((32, 62), dtype=torch.float64, device="cuda")
    dyn.set_boundary(z + 1.0, z + 2.0, z + 3.0)
    ph.fordate(sp, dyn.boundary_ptr() + 32 * 62 * 8)           # straight into the time steps' tcorh | qcorh
    bc = dyn.boundary()
    assert np.all(bc[0] == 1.0)                                # phis untouched
    for k, name in ((1, "tcorh"), (2, "qcorh")):
        assert rel(bc[k].T, want[name]) <= 1e-12, (name, rel(bc[k].T, want[name]))
    assert rel(ph.surface("corh_t").ravel(), (6.0 / (1000.0 * 9.81) * i["phis0"]).ravel()) == 0.0
    for name in ("snowc", "alb_l", "alb_s", "albsfc"):
        got = ph.surface(name).ravel()
        assert np.array_equal(got, want[name] if with_albedo else given[name].ravel()), name
    # the coupler's daily output enters between windows: a new land temperature moves qcorh, new snow the albedos
    if with_albedo:
        ph.update_surface(tland=i["stl_am"] + 2.0, snowd_am=i["snowd_am"] * 0.5)
        ph.fordate(sp, dyn.boundary_ptr() + 32 * 62 * 8)
        bcu = dyn.boundary()
        assert not np.array_equal(bcu[2], bc[2]) and np.array_equal(bcu[1], bc[1])
        assert np.array_equal(ph.surface("snowc").ravel(), np.minimum(1.0, i["snowd_am"] * 0.5 / 60.0).ravel())
        if RefPhys.available():
            upd = RefPhys(HSG, gaussian_latitudes()).fordate(TYEAR, **dict(i, stl_am=i["stl_am"] + 2.0, snowd_am=i["snowd_am"] * 0.5))
            assert rel(bcu[2].T, upd["qcorh"]) <= 1e-12 and np.array_equal(ph.surface("albsfc").ravel(), upd["albsfc"])
        ph.update_surface(tland=i["stl_am"], snowd_am=i["snowd_am"])
        ph.fordate(sp, dyn.boundary_ptr() + 32 * 62 * 8)
        assert np.array_equal(dyn.boundary()[2], bc[2])
    else:
        with pytest.raises(Exception):
            ph.update_surface(snowd_am=i["snowd_am"])          # no albedo inputs were given: nothing the snow depth could enter
    # a second call after the SST changed follows it (the hybrid's per-window recomputation)
    sst += 1.5
    ph.fordate(sp, dyn.boundary_ptr() + 32 * 62 * 8)
    bc2 = dyn.boundary()
    assert np.array_equal(bc2[1], bc[1]) and not np.array_equal(bc2[2], bc[2])
    if RefPhys.available():
        again = RefPhys(HSG, gaussian_latitudes()).fordate(TYEAR, **dict(i, sst_am=i["sst_am"] + 1.5))
        assert rel(bc2[2].T, again["qcorh"]) <= 1e-12
