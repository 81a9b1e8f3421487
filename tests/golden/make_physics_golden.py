"""Generates tests/golden/physics_golden.npz from the COMPILED REFERENCE (oracle/_ref/libref_phy.so: the reference's own phy_*.f90
compiled in place by oracle/build_ref.sh).  Run in the build container only:

    python tests/golden/make_physics_golden.py

The fixture holds data only: the reference's outputs (every 8th column) for the seeded column set of physics_inputs() through the
grid-point sequence of phypar (src/phy_phypar.f90:80-230), a short-wave step followed by a non-short-wave step, plus the
set-up tables (sol_oz fields, fband, sigma-level constants).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

HSG = np.array([0.000, 0.050, 0.140, 0.260, 0.420, 0.600, 0.770, 0.900, 1.000])
TYEAR = 0.37
KEEP = slice(None, None, 8)


def gaussian_latitudes():
    """radang of src/ini_indyns.f90:72-80 from the reference's Gauss nodes (the spectral fixture holds sia)"""
    sia = np.load(os.path.join(os.path.dirname(__file__), "spectral_golden.npz"))["tab_sia"].ravel()
    return np.concatenate([-np.arcsin(sia), np.arcsin(sia)[::-1]])


def physics_inputs(seed=7):
    """4608 columns (ngp, nlev) with enough spread to exercise every branch: deep convection, large-scale condensation, stable and
    unstable boundary layers, land / sea / coast, snow, polar night."""
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import synth
    rng = np.random.default_rng(seed)
    g4, logp, _, sst = synth.synthetic_state(seed)
    sig = 0.5 * (HSG[1:] + HSG[:-1])
    T = g4[..., 0].reshape(8, -1).T.copy()
    u = g4[..., 1].reshape(8, -1).T.copy()
    v = g4[..., 2].reshape(8, -1).T.copy()
    q = g4[..., 3].reshape(8, -1).T.copy()
    lat = np.repeat(np.linspace(-87.159, 87.159, 48), 96)
    # moist, conditionally unstable tropics; dry subtropics; supersaturated patches
    trop = np.exp(-(lat / 25.0) ** 2)
    T[:, 7] += 6.0 * trop * rng.random(4608)
    T[:, 6] += 3.0 * trop * rng.random(4608)
    q *= (0.4 + 1.4 * rng.random((4608, 1))) * (1.0 + 0.5 * trop[:, None])
    q[:, 0:2] = np.maximum(q[:, 0:2], 1e-3)
    q[rng.random(4608) < 0.02, 5] = -0.01                                  # a few negative humidities (clamped by phypar)
    pslg = logp.ravel() + 0.05 * rng.standard_normal(4608) - 0.25 * (rng.random(4608) < 0.1)      # some columns below psmin
    phis0 = np.maximum(0.0, synth.synthetic_orography().ravel())
    rgas = 2.0 / 7.0 * 1004.0
    xg1 = rgas * np.log(HSG[1:] / sig)
    xg2 = np.zeros(8)
    xg2[1:] = rgas * np.log(sig[1:] / HSG[1:-1])
    phi = np.zeros((4608, 8))
    phi[:, 7] = phis0 + xg1[7] * T[:, 7]
    for k in range(6, -1, -1):
        phi[:, k] = phi[:, k + 1] + xg2[k + 1] * T[:, k + 1] + xg1[k] * T[:, k]
    sea = synth.land_mask().ravel().astype(float)
    fmask = np.clip(1.0 - sea + 0.3 * (rng.random(4608) - 0.5) * (rng.random(4608) < 0.3), 0.0, 1.0)
    tland = T[:, 7] + 4.0 * rng.standard_normal(4608)
    tsea = sst.ravel() + rng.standard_normal(4608)
    swav = rng.random(4608)
    snowc = np.where(np.abs(lat) > 55, rng.random(4608), 0.0)
    alb_l = 0.15 + 0.45 * snowc
    alb_s = 0.07 + 0.5 * (np.abs(lat) > 70)
    albsfc = alb_s + fmask * (alb_l - alb_s)
    utend, vtend = 1e-4 * rng.standard_normal((4608, 8)), 1e-4 * rng.standard_normal((4608, 8))
    ttend, qtend = 1e-4 * rng.standard_normal((4608, 8)), 1e-6 * rng.standard_normal((4608, 8))
    return dict(ug=u, vg=v, tg=T, qg=q, phig=phi, pslg=pslg, fmask=fmask, phis0=phis0, tland=tland, tsea=tsea, swav=swav, snowc=snowc,
                alb_l=alb_l, alb_s=alb_s, albsfc=albsfc, utend=utend, vtend=vtend, ttend=ttend, qtend=qtend)


def run_reference(inp, ref):
    ref.set_surface(inp["phis0"], inp["alb_l"], inp["alb_s"], inp["albsfc"], inp["snowc"])
    ref.sol_oz(TYEAR)
    out = {}
    args = [inp[k] for k in ("ug", "vg", "tg", "qg", "phig", "pslg", "fmask", "phis0", "tland", "tsea", "swav")]
    for tag, lradsw, dt in (("sw", True, 0.0), ("nosw", False, 1.5)):
        a = list(args)
        a[2] = inp["tg"] + dt                        # the second step sees a slightly different temperature
        u, v, t, q, diag = ref.phypar(*a, lradsw, inp["utend"], inp["vtend"], inp["ttend"], inp["qtend"])
        out[tag] = dict(utend=u, vtend=v, ttend=t, qtend=q, **diag)
    return out


WINDOW_STEPS = 4          # leapfrog steps after stepone in the coupled fixture: short-wave flags 1 1 | 1 0 0 1
WINDOW_KEEP = slice(None, None, 4)


def coupled_inputs(seed=5):
    """Spectral start state (both time levels equal, as the hybrid hand-off leaves it), boundary fields and surface fields of the
    dynamics + physics window fixture.  Returns (state dict of (62,32,8,2)/(62,32,2), phis (62,32), surface dict)."""
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import synth
    from _oracle import Oracle, oracle_iogrid30
    o = Oracle()
    g4, logp, _, _ = synth.synthetic_state(seed)
    lvl = oracle_iogrid30(o, g4, logp)
    st = {k: np.stack([lvl[k], lvl[k]], axis=-1) for k in ("vor", "div", "t", "tr", "ps")}
    phis = o.trunct(o.spec(synth.synthetic_orography().T))
    surf = physics_inputs(seed)
    return o, st, phis, {k: surf[k] for k in ("fmask", "phis0", "tland", "tsea", "swav", "snowc", "alb_l", "alb_s", "albsfc")}


def run_coupled_reference(o, st, phis, surf, ref, nsteps=WINDOW_STEPS, delt=900.0, tyear=TYEAR, tcorh=None, qcorh=None):
    """stepone + nsteps leapfrog steps (src/ini_stepone.f90, src/dyn_stloop.f90:28-43) with the oracle's dynamics and, in grtend's
    physics slot (src/dyn_grtend.f90:222-225), the compiled reference parametrisations.  tcorh / qcorh: fordate's diffusion
    corrections (62,32), zero when not given (the committed window fixture was generated without them)."""
    from _oracle import DynOracle
    ref.set_surface(surf["phis0"], surf["alb_l"], surf["alb_s"], surf["albsfc"], surf["snowc"])
    ref.sol_oz(tyear)
    do = DynOracle(o)
    zero = np.zeros((62, 32))
    tcorh, qcorh = zero if tcorh is None else tcorh, zero if qcorh is None else qcorh
    flag = {"lradsw": True}

    def hook(ug, vg, tg, qg, phig, pslg, ut, vt, tt, qt):
        return ref.phypar(ug, vg, tg, qg, phig, pslg, surf["fmask"], surf["phis0"], surf["tland"], surf["tsea"], surf["swav"],
                          flag["lradsw"], ut, vt, tt, qt)[:4]

    sched = [(1, 1, 0.5 * delt, 0.5 * delt, True), (1, 2, delt, delt, True)]
    sched += [(2, 2, 2 * delt, 2 * delt, (i + 1) % 3 == 1) for i in range(nsteps)]
    for j1, j2, dt_imp, dt, sw in sched:
        do.impint(dt_imp, 0.5)
        flag["lradsw"] = sw
        st = do.step(j1, j2, dt, 0.5, 0.05, 0.53, st, phis, tcorh, qcorh, phys=hook)
    return st


def main():
    from _oracle import RefPhys
    ref = RefPhys(HSG, gaussian_latitudes())
    inp = physics_inputs()
    res = run_reference(inp, ref)
    f = ref.fields()
    out = {"fband": ref.fband, "sig": ref.sig, "grdsig": ref.grdsig, "grdscp": ref.grdscp, "wvi": ref.wvi}
    for k in ("fsol", "ozone", "ozupp", "zenit", "stratz"):
        out["zonal_" + k] = f[k][::96]
    out["forog"] = f["forog"]
    for tag, d in res.items():
        for k, v in d.items():
            out[f"{tag}_{k}"] = np.asarray(v)[KEEP]
    o, st, phis, surf = coupled_inputs()
    end = run_coupled_reference(o, st, phis, surf, RefPhys(HSG, gaussian_latitudes()))
    for k, v in end.items():
        out["window_" + k] = np.asarray(v).reshape(-1, order="F")[WINDOW_KEEP]
    path = os.path.join(os.path.dirname(__file__), "physics_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;",
          "convecting columns", int((res["sw"]["iptop"] < 9).sum()), "precls>0", int((res["sw"]["precls"] > 0).sum()),
          "cloudc>0", int((res["sw"]["cloudc"] > 0).sum()), "psa<psmin", int((np.exp(inp["pslg"]) < 0.8).sum()))


if __name__ == "__main__":
    main()
