"""GPU: SPEEDY's column physics on the device (sml_phys_*, one kernel for the grid-point sequence of phypar) against fixtures
generated from the COMPILED REFERENCE (tests/golden/physics_golden.npz <- oracle/_ref/libref_phy.so, the reference's own
phy_*.f90) and, when that library travelled with the snapshot, against the reference itself on all 4608 columns.

Tolerance: 1e-11 of each field's max-abs.  The kernel keeps the reference's operation order; the differences are the device
libm's exp / sqrt / log (1-2 ulp) amplified by the transmissivity products.  Integer outputs (convection top, cloud top) must
be identical."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_physics_golden import HSG, KEEP, TYEAR, gaussian_latitudes, physics_inputs, run_reference  # noqa: E402

from speedy_ml_amd.physics import Physics  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "physics_golden.npz")
TOL = 1e-11
DIAGS = ("precnv", "precls", "cbmf", "ts", "tskin", "ssrd", "slrd", "olr", "shf", "evap", "ustr", "vstr", "slr", "hfluxn_land", "hfluxn_sea",
         "t0", "q0", "iptop")
SW_DIAGS = ("cloudc", "clstr", "tsr", "ssr", "icltop")


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def device_run(inp):
    ph = Physics(gaussian_latitudes())
    g = lambda a: np.asarray(a).reshape(48, 96)
    ph.set_surface(g(inp["fmask"]), g(inp["phis0"]), g(inp["tland"]), g(inp["tsea"]), g(inp["swav"]), g(inp["alb_l"]), g(inp["alb_s"]),
                   g(inp["albsfc"]), g(inp["snowc"]))
    ph.sol_oz(TYEAR)
    out = {}
    for tag, lradsw, dt in (("sw", True, 0.0), ("nosw", False, 1.5)):
        grids = np.zeros((41, 4608))
        for i, k in enumerate(("ug", "vg", "tg", "qg", "phig")):
            grids[8 * i:8 * i + 8] = (inp[k] + (dt if k == "tg" else 0.0)).T
        grids[40] = inp["pslg"]
        tend = np.concatenate([inp[k].T for k in ("utend", "vtend", "ttend", "qtend")])
        dg = torch.from_numpy(grids.reshape(41, 48, 96)).cuda()
        dt_ = torch.from_numpy(tend.reshape(32, 48, 96).copy()).cuda()
        ph.tendencies(dg, lradsw, dt_, accumulate=True)
        res = dt_.cpu().numpy().reshape(32, 4608)
        d = dict(utend=res[0:8].T, vtend=res[8:16].T, ttend=res[16:24].T, qtend=res[24:32].T)
        for name in DIAGS + (SW_DIAGS if lradsw else ()):
            d[name] = ph.diag(name).ravel()
        out[tag] = d
    return ph, out


def compare(got, want, cols):
    for tag in ("sw", "nosw"):
        for name in ("utend", "vtend", "ttend", "qtend") + DIAGS + (SW_DIAGS if tag == "sw" else ()):
            a, b = np.asarray(got[tag][name])[cols], np.asarray(want(tag, name))
            if name in ("iptop", "icltop"):
                assert np.array_equal(a, b), (tag, name, int((a != b).sum()))
            else:
                assert rel(a, b) < TOL, (tag, name, rel(a, b))


def test_physics_matches_reference_fixture():
    gold = np.load(GOLD)
    inp = physics_inputs()
    ph, got = device_run(inp)
    zonal, fband, lev = ph.tables()
    for i, k in enumerate(("fsol", "ozone", "ozupp", "zenit", "stratz")):
        assert rel(zonal[i], gold["zonal_" + k]) < 1e-13, k
    assert rel(fband, gold["fband"]) < 1e-15
    assert rel(lev[0, 1:], gold["sig"]) < 1e-15 and rel(lev[4, 1:], gold["grdsig"]) < 1e-15 and rel(lev[6, 1:], gold["wvi"][:, 1]) < 1e-14
    compare(got, lambda tag, name: gold[f"{tag}_{name}"], KEEP)
    # the inputs exercise the branches: deep convection, condensation, clouds, columns below psmin, land and sea
    sw = got["sw"]
    assert (sw["cbmf"] > 0).sum() > 200 and (sw["precls"] > 0).sum() > 200 and (sw["cloudc"] > 0).sum() > 500
    assert (sw["iptop"] == 9).sum() > 200 and 0 < (np.exp(inp["pslg"]) < 0.8).sum()


def test_physics_matches_compiled_reference_all_columns():
    from _oracle import RefPhys
    if not RefPhys.available():
        pytest.skip("oracle/_ref/libref_phy.so not present")
    inp = physics_inputs(seed=11)
    want = run_reference(inp, RefPhys(HSG, gaussian_latitudes()))
    _, got = device_run(inp)
    compare(got, lambda tag, name: want[tag][name], slice(None))


def test_physics_overwrite_mode_gives_the_physics_tendencies_alone():
    inp = physics_inputs(seed=5)
    zero = {k: np.zeros_like(inp[k]) for k in ("utend", "vtend", "ttend", "qtend")}
    _, with_dyn = device_run(inp)
    _, phys_only = device_run({**inp, **zero})
    for k in ("utend", "vtend", "ttend", "qtend"):
        a = with_dyn["sw"][k] - inp[k]
        assert rel(a, phys_only["sw"][k]) < 1e-9          # (dyn + phys) - dyn == phys up to the rounding of the larger sum


def test_sfcwind_entry_equals_full_entry_and_diag_switch():
    """sml_phys_tendencies_sfcwind reads only the lowest-level winds: same bits as the 41-grid entry; want_diag = 0 leaves the
    diagnostics of the previous call untouched."""
    inp = physics_inputs(seed=13)
    ph = Physics(gaussian_latitudes())
    g = lambda a: np.asarray(a).reshape(48, 96)
    ph.set_surface(*[g(inp[k]) for k in ("fmask", "phis0", "tland", "tsea", "swav", "alb_l", "alb_s", "albsfc", "snowc")])
    ph.sol_oz(TYEAR)
    grids = np.zeros((41, 4608))
    for i, k in enumerate(("ug", "vg", "tg", "qg", "phig")):
        grids[8 * i:8 * i + 8] = inp[k].T
    grids[40] = inp["pslg"]
    tend0 = np.concatenate([inp[k].T for k in ("utend", "vtend", "ttend", "qtend")]).reshape(32, 48, 96)
    full = torch.from_numpy(tend0.copy()).cuda()
    ph.tendencies(torch.from_numpy(grids.reshape(41, 48, 96)).cuda(), True, full, accumulate=True)
    olr = ph.diag("olr").copy()
    g27 = np.concatenate([grids[7:8], grids[15:16], grids[16:41]]).reshape(27, 48, 96)
    part = torch.from_numpy(tend0.copy()).cuda()
    ph.tendencies_sfcwind(torch.from_numpy(g27.copy()).cuda(), True, part, accumulate=True, want_diag=True)
    assert torch.equal(full, part)
    assert np.array_equal(ph.diag("olr"), olr)
    # a warmer column set without diagnostics: tendencies change, the stored diagnostics do not
    g27[2:10] += 2.0
    ph.tendencies_sfcwind(torch.from_numpy(g27.copy()).cuda(), False, part, accumulate=False, want_diag=False)
    assert np.array_equal(ph.diag("olr"), olr)
    ph.tendencies_sfcwind(torch.from_numpy(g27.copy()).cuda(), False, part, accumulate=False, want_diag=True)
    assert not np.array_equal(ph.diag("olr"), olr)
    # accumulate = 0 overwrites: no wind tendency above the lowest level
    assert float(part[0:7].abs().max()) == 0.0 and float(part[8:15].abs().max()) == 0.0 and float(part[7].abs().max()) > 0.0


def test_attach_detach_and_shortwave_flag():
    """sml_dyn_attach_physics / sml_dyn_set_lradsw: detached = the adiabatic step (bit for bit), the short-wave flag changes the step"""
    from make_physics_golden import coupled_inputs
    from speedy_ml_amd.dynamics import Dynamics
    _, st, phis, surf = coupled_inputs(seed=2)
    got_phys, dyn, ph = device_window(st, phis, surf, 0)

    def one_step(dyn):
        state = np.zeros((2, 33, 32, 62))
        for j in range(2):
            for off, k in ((0, "vor"), (8, "div"), (16, "t"), (24, "tr")):
                state[j, off:off + 8] = st[k][..., j].transpose(2, 1, 0)
            state[j, 32] = st["ps"][..., j].T
        d = torch.from_numpy(state).cuda()
        dyn.impint(1800.0)
        dyn.step(d, 2, 2, 1800.0)
        return d

    dyn.set_lradsw(True)
    a = one_step(dyn)
    dyn.set_lradsw(False)
    b = one_step(dyn)                       # no short-wave call: everything it would have produced was kept from the previous step
    assert torch.equal(a, b)                # (transmissivities, stratospheric terms, heating, surface flux), so same state -> same bits
    ph2 = Physics(gaussian_latitudes())     # a handle that never ran the short-wave scheme has nothing to fall back on
    g = lambda x: np.asarray(x).reshape(48, 96)
    ph2.set_surface(*[g(surf[k]) for k in ("fmask", "phis0", "tland", "tsea", "swav", "alb_l", "alb_s", "albsfc", "snowc")])
    ph2.sol_oz(TYEAR)
    dyn.attach_physics(ph2)
    dyn.set_lradsw(False)
    c = one_step(dyn)
    assert not torch.equal(a, c)
    dyn.attach_physics(None)
    dry = one_step(dyn)
    fresh = Dynamics(dyn.sp)
    dev2 = lambda x: torch.from_numpy(np.ascontiguousarray(np.asarray(x).T)).cuda()
    fresh.set_boundary(dev2(phis), dev2(np.zeros((62, 32))), dev2(np.zeros((62, 32))))
    assert torch.equal(dry, one_step(fresh)) and not torch.equal(dry, a)


# ---- the physics inside the time step (src/dyn_grtend.f90:222-225): stepone + leapfrog steps with the parametrisations attached ----
WTOL = 1e-10        # north_star: fields within 1e-10 relative after the window


def device_window(st, phis, surf, nsteps):
    from make_physics_golden import TYEAR as TY
    from speedy_ml_amd.dynamics import Dynamics
    from speedy_ml_amd.spectral import Spectral
    sp = Spectral()
    dyn = Dynamics(sp)
    ph = Physics(gaussian_latitudes())
    g = lambda a: np.asarray(a).reshape(48, 96)
    ph.set_surface(*[g(surf[k]) for k in ("fmask", "phis0", "tland", "tsea", "swav", "alb_l", "alb_s", "albsfc", "snowc")])
    ph.sol_oz(TY)
    dyn.attach_physics(ph)
    dev2 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).T)).cuda()
    zero = np.zeros((62, 32))
    dyn.set_boundary(dev2(phis), dev2(zero), dev2(zero))
    state = np.zeros((2, 33, 32, 62))
    for j in range(2):
        for off, k in ((0, "vor"), (8, "div"), (16, "t"), (24, "tr")):
            state[j, off:off + 8] = st[k][..., j].transpose(2, 1, 0)
        state[j, 32] = st["ps"][..., j].T
    dstate = torch.from_numpy(state).cuda()
    dyn.window(dstate, nsteps, start=True)
    torch.cuda.synchronize()
    got = dstate.cpu().numpy()
    out = {k: np.stack([got[j, off:off + 8].transpose(2, 1, 0) for j in range(2)], axis=-1) for off, k in ((0, "vor"), (8, "div"), (16, "t"), (24, "tr"))}
    out["ps"] = np.stack([got[j, 32].T for j in range(2)], axis=-1)
    return out, dyn, ph


def test_fused_forms_agree_bit_for_bit():
    """The four forms of grtend's grid-point part + phypar -- two launches (k_gridtend, k_physics), one two-wavefront launch, one
    one-wavefront launch and the three-wavefront launch (the default since round 4) -- leave identical bits after stepone + 7 steps
    (short-wave and other steps), and each fused form repeats itself.  The one-wavefront form is the shape that was non-repeatable at -O3
    in round 1."""
    from make_physics_golden import coupled_inputs
    from speedy_ml_amd.dynamics import Dynamics
    _, st, phis, surf = coupled_inputs(seed=4)
    try:
        Dynamics.select_physics_form(False)
        two, _, ph_two = device_window(st, phis, surf, 7)
        diag_two = {k: ph_two.diag(k).copy() for k in ("olr", "precnv", "iptop", "ssrd", "shf")}
        Dynamics.select_physics_form(2)
        single, _, ph_single = device_window(st, phis, surf, 7)
        single_again, _, _ = device_window(st, phis, surf, 7)
        diag_single = {k: ph_single.diag(k).copy() for k in diag_two}
        Dynamics.select_physics_form(True)
        one, _, ph_one = device_window(st, phis, surf, 7)
        again, _, _ = device_window(st, phis, surf, 7)
        Dynamics.select_physics_form(3)
        three, _, ph_three = device_window(st, phis, surf, 7)
        three_again, _, _ = device_window(st, phis, surf, 7)
        diag_three = {k: ph_three.diag(k).copy() for k in diag_two}
    finally:
        Dynamics.select_physics_form(3)
    for k in ("vor", "div", "t", "tr", "ps"):
        assert np.all(np.isfinite(one[k]))
        assert np.array_equal(one[k], two[k]), k
        assert np.array_equal(one[k], again[k]), k
        assert np.array_equal(single[k], two[k]), k
        assert np.array_equal(single[k], single_again[k]), k
        assert np.array_equal(three[k], two[k]), k
        assert np.array_equal(three[k], three_again[k]), k
    for k, v in diag_two.items():
        assert np.array_equal(ph_one.diag(k), v), k
        assert np.array_equal(diag_single[k], v), k
        assert np.array_equal(diag_three[k], v), k


def test_window_with_physics_matches_reference_fixture():
    from make_physics_golden import WINDOW_KEEP, WINDOW_STEPS, coupled_inputs
    gold = np.load(GOLD)
    _, st, phis, surf = coupled_inputs()
    got, _, _ = device_window(st, phis, surf, WINDOW_STEPS)
    for k in ("vor", "div", "t", "tr", "ps"):
        a, b = got[k].reshape(-1, order="F")[WINDOW_KEEP], gold["window_" + k]
        assert rel(a, b) < WTOL, (k, rel(a, b))


def test_window_with_physics_matches_oracle_plus_compiled_reference():
    from _oracle import RefPhys
    from make_physics_golden import coupled_inputs, run_coupled_reference
    if not RefPhys.available():
        pytest.skip("oracle/_ref/libref_phy.so not present")
    o, st, phis, surf = coupled_inputs(seed=9)
    want = run_coupled_reference(o, st, phis, surf, RefPhys(HSG, gaussian_latitudes()), nsteps=7)
    got, dyn, _ = device_window(st, phis, surf, 7)
    for k in ("vor", "div", "t", "tr", "ps"):
        assert rel(got[k], want[k]) < WTOL, (k, rel(got[k], want[k]))
    # the physics did something: the adiabatic window ends somewhere else
    from _oracle import DynOracle
    do = DynOracle(o)
    dry = dict(st)
    zero = np.zeros((62, 32))
    for j1, j2, dt in [(1, 1, 450.0), (1, 2, 900.0)] + [(2, 2, 1800.0)] * 7:
        do.impint(dt, 0.5)
        dry = do.step_dry(j1, j2, dt, 0.5, 0.05, 0.53, dry, phis, zero, zero)
    assert rel(dry["t"], want["t"]) > 1e-6


def test_sol_oz_enqueued_equals_the_blocking_upload():
    """sml_phys_sol_oz_async (the hybrid engine's once-a-day call: the zonal solar / ozone fields travel as kernel arguments in stream
    order) leaves the same table as sml_phys_sol_oz (host computation + blocking copy)."""
    import torch
    from speedy_ml_amd.physics import Physics
    from make_physics_golden import gaussian_latitudes
    ph = Physics(gaussian_latitudes())
    for ty in (0.0, 0.31, 0.77):
        ph.sol_oz(ty)
        want = ph.tables()[0].copy()
        ph.sol_oz(0.5)                                             # something else in between
        ph.sol_oz(ty, stream=torch.cuda.current_stream(), asynchronous=True)
        torch.cuda.synchronize()
        assert np.array_equal(ph.tables()[0], want), ty
