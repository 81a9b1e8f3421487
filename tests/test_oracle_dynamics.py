"""CPU: the dynamics oracle (oracle/dynamics_oracle.c) against the compiled reference's outputs.

Pinned by tests/golden/dynamics_golden.npz (generated from oracle/_ref/libref_dyn.so by tests/golden/make_dynamics_golden.py)
and, when the compiled reference is present in this container, directly against it on further seeded cases.  grtend cannot be
run in the reference without the column physics (src/dyn_grtend.f90:222-225 -> phypar), so do_grtend_dry is checked against an
independent vectorised numpy evaluation of the same equations plus physical invariants.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from _oracle import DYN_TABLES, IL, IX, KX, MX2, NX, S2, S3, DynOracle, Oracle, RefDyn  # noqa: E402
from make_dynamics_golden import seeded_state  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dynamics_golden.npz")
AKAP = 2.0 / 7.0
RGAS = AKAP * 1004.0


@pytest.fixture(scope="module")
def dyn():
    return DynOracle(Oracle())


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def test_tables_match_reference(dyn, gold):
    for dt in (450.0, 900.0, 1800.0):
        dyn.impint(dt, 0.5)
        for w, (name, _) in DYN_TABLES.items():
            assert rel(dyn.table(w), gold["tab%d_%s" % (int(dt), name)]) < 1e-14, (dt, name)


def test_spectral_routines_match_reference(dyn, gold):
    dt, alph, rob, wil = gold["params"]
    dyn.impint(dt, alph)
    st = seeded_state(gold["trfilt"])
    for jj in (1, 2):
        assert rel(dyn.geop(st["t"][..., jj - 1], st["phis"]), gold["geop%d" % jj]) < 1e-15
    for j4 in (1, 2):
        d, t, p, _ = dyn.sptend(st["div"][..., j4 - 1], st["t"][..., j4 - 1], st["ps"][..., j4 - 1], st["phis"],
                                st["tend3"], st["tend3b"], st["tend2"])
        assert rel(d, gold["sptend%d_divdt" % j4]) < 1e-14
        assert rel(t, gold["sptend%d_tdt" % j4]) < 1e-14
        assert rel(p, gold["sptend%d_psdt" % j4]) < 1e-14
    d, t, p = dyn.implic(st["tend3"], st["tend3b"], st["tend2"])
    assert rel(d, gold["implic_divdt"]) < 1e-14 and rel(t, gold["implic_tdt"]) < 1e-14 and rel(p, gold["implic_psdt"]) < 1e-14
    d, t, p, _ = dyn.sptend(st["div"][..., 0], st["t"][..., 0], st["ps"][..., 0], st["phis"], st["tend3"], st["tend3b"], st["tend2"])
    d, t, p = dyn.implic(d, t, p)
    assert rel(d, gold["chain_divdt"]) < 1e-14 and rel(t, gold["chain_tdt"]) < 1e-14 and rel(p, gold["chain_psdt"]) < 1e-14
    for which in (1, 2, 3):
        for nlev in (8, 1):
            assert rel(dyn.hordif(nlev, st["vor"][..., 0], st["tend3"], which), gold["hordif%d_%d" % (which, nlev)]) < 1e-15
    for j1, eps in ((1, 0.0), (2, rob)):
        f, g = dyn.timint(j1, dt, eps, wil, 8, st["t"], st["tend3b"])
        assert rel(f, gold["timint%d_t" % j1]) < 1e-15 and rel(g, gold["timint%d_tdt" % j1]) < 1e-15
        f, _ = dyn.timint(j1, dt, eps, wil, 1, st["ps"].reshape(MX2, NX, 1, 2), st["tend2"].reshape(MX2, NX, 1))
        assert rel(f, gold["timint%d_ps" % j1]) < 1e-15


@pytest.mark.skipif(not RefDyn.available(), reason="compiled reference (oracle/_ref/libref_dyn.so) not built")
def test_against_compiled_reference_more_seeds(dyn, gold):
    ref = RefDyn()
    for seed, dt, alph in ((1, 450.0, 0.5), (2, 900.0, 0.5), (3, 1800.0, 1.0)):
        dyn.impint(dt, alph)
        ref.impint(dt, alph)
        for w in DYN_TABLES:
            assert rel(dyn.table(w), ref.table(w)) < 1e-14
        st = seeded_state(gold["trfilt"], seed)
        ref.set_state(st["vor"], st["div"], st["t"], st["ps"], st["tr"], st["phis"], st["tcorh"], st["qcorh"])
        assert rel(dyn.geop(st["t"][..., 0], st["phis"]), ref.geop(1)) < 1e-15
        a = ref.sptend(st["tend3"], st["tend3b"], st["tend2"], 1)
        b = dyn.sptend(st["div"][..., 0], st["t"][..., 0], st["ps"][..., 0], st["phis"], st["tend3"], st["tend3b"], st["tend2"])
        for x, y in zip(a, b):
            assert rel(y, x) < 1e-14
        a2, b2 = ref.implic(*a), dyn.implic(*b[:3])
        for x, y in zip(a2, b2):
            assert rel(y, x) < 1e-14
        fa, _ = ref.timint(2, 2 * dt, 0.05, 0.53, 8, st["vor"], a2[0])
        fb, _ = dyn.timint(2, 2 * dt, 0.05, 0.53, 8, st["vor"], b2[0])
        assert rel(fb, fa) < 1e-15


# ------------------------------------------------------------------ grtend (adiabatic): independent numpy evaluation
def grtend_numpy(o, tb, vor, div, t, tr, ps):
    """src/dyn_grtend.f90 without phypar, written with whole-array numpy and the Oracle's (pinned) transforms."""
    dhs, dhsr, fsgr, tref, tref3, coriol = tb["dhs"], tb["dhsr"], tb["fsgr"], tb["tref"], tb["tref3"], tb["coriol"]
    G = lambda s, kc: o.grid(s, kc)                                   # (96,48)
    ug, vg, tg, vorg, divg, trg = (np.zeros((IX, IL, KX)) for _ in range(6))
    for k in range(KX):
        vorg[..., k] = G(vor[..., k], 1) + coriol[None, :]
        divg[..., k] = G(div[..., k], 1)
        tg[..., k] = G(t[..., k], 1)
        trg[..., k] = G(tr[..., k], 1)
        uc, vc = o.uvspec(vor[..., k], div[..., k])
        ug[..., k], vg[..., k] = G(uc, 2), G(vc, 2)
    umean, vmean, dmean = (ug * dhs).sum(-1), (vg * dhs).sum(-1), (divg * dhs).sum(-1)
    dx, dy = o.grad(ps)
    px, py = G(dx, 2), G(dy, 2)
    psdt = o.spec(-umean * px - vmean * py)
    psdt[0:2, 0] = 0.0
    puv = (ug - umean[..., None]) * px[..., None] + (vg - vmean[..., None]) * py[..., None]
    sigdt, sigm = np.zeros((IX, IL, KX + 1)), np.zeros((IX, IL, KX + 1))
    for k in range(KX):
        sigdt[..., k + 1] = sigdt[..., k] - dhs[k] * (puv[..., k] + divg[..., k] - dmean)
        sigm[..., k + 1] = sigm[..., k] - dhs[k] * puv[..., k]
    sigdt[..., KX] = 0.0        # the Fortran loop stops at kx-1; the closed column sums to ~0 anyway
    sigm[..., KX] = 0.0
    tgg = tg - tref
    px, py = RGAS * px, RGAS * py

    def vert(f):                # (temp(k+1)+temp(k))*dhsr(k) with temp(k) = sigdt(k)*(f(k)-f(k-1)), zero at the ends
        tmp = np.zeros((IX, IL, KX + 1))
        tmp[..., 1:KX] = sigdt[..., 1:KX] * (f[..., 1:] - f[..., :-1])
        return tmp

    tu, tv = vert(ug), vert(vg)
    utend = vg * vorg - tgg * px[..., None] - (tu[..., 1:] + tu[..., :-1]) * dhsr
    vtend = -ug * vorg - tgg * py[..., None] - (tv[..., 1:] + tv[..., :-1]) * dhsr
    tt = vert(tgg)
    tt[..., 1:KX] += sigm[..., 1:KX] * (tref[1:] - tref[:-1])
    ttend = (tgg * divg - (tt[..., 1:] + tt[..., :-1]) * dhsr + fsgr * tgg * (sigdt[..., 1:] + sigdt[..., :-1])
             + tref3 * (sigm[..., 1:] + sigm[..., :-1]) + AKAP * (tg * puv - tgg * dmean[..., None]))
    tq = vert(trg)
    tq[..., 1:3] = 0.0
    trtend = trg * divg - (tq[..., 1:] + tq[..., :-1]) * dhsr
    vordt, divdt, tdt, trdt = (np.zeros(S3) for _ in range(4))
    for k in range(KX):
        vordt[..., k], divdt[..., k] = o.vdspec(utend[..., k], vtend[..., k], 2)
        divdt[..., k] -= o.lap(o.spec(0.5 * (ug[..., k] ** 2 + vg[..., k] ** 2)))
        _, d = o.vdspec(-ug[..., k] * tgg[..., k], -vg[..., k] * tgg[..., k], 2)
        tdt[..., k] = d + o.spec(ttend[..., k])
        _, d = o.vdspec(-ug[..., k] * trg[..., k], -vg[..., k] * trg[..., k], 2)
        trdt[..., k] = d + o.spec(trtend[..., k])
    return vordt, divdt, tdt, psdt, trdt


def test_grtend_dry_matches_numpy_evaluation(dyn, gold):
    dyn.impint(1800.0, 0.5)
    st = seeded_state(gold["trfilt"], 5)
    tb = dyn.tables()
    args = [st[k][..., 1] for k in ("vor", "div", "t", "tr", "ps")]
    got = dyn.grtend_dry(*args)
    want = grtend_numpy(dyn.o, tb, *args)
    for name, a, b in zip(("vordt", "divdt", "tdt", "psdt", "trdt"), got, want):
        assert rel(a, b) < 1e-11, name


def test_grtend_dry_state_of_rest_has_no_tendency(dyn):
    dyn.impint(1800.0, 0.5)
    tref = dyn.table(17)
    t = np.zeros(S3)
    t[0, 0, :] = tref * np.sqrt(2.0)          # horizontally uniform T(k): the (0,0) mode carries sqrt(2) * mean
    z3, z2 = np.zeros(S3), np.zeros(S2)
    vordt, divdt, tdt, psdt, trdt = dyn.grtend_dry(z3, z3, t, z3, z2)
    for a in (divdt, tdt, psdt, trdt):
        assert np.max(np.abs(a)) < 1e-12
    # the only vorticity source at rest is -d/dlambda(0) - ... = 0 as well (f*v = 0)
    assert np.max(np.abs(vordt)) < 1e-18


def test_step_dry_leapfrog_bookkeeping(dyn, gold):
    """dt <= 0 leaves the state untouched (dyn_step.f90:110); a forward step with j1=1 sets F(2)=Fnew and keeps F(1)."""
    dyn.impint(450.0, 0.5)
    st = seeded_state(gold["trfilt"], 7)
    state = {k: st[k] for k in ("vor", "div", "t", "tr", "ps")}
    same = dyn.step_dry(1, 1, 0.0, 0.5, 0.05, 0.53, state, st["phis"], st["tcorh"], st["qcorh"])
    for k in state:
        assert np.array_equal(same[k], state[k])
    new = dyn.step_dry(1, 1, 450.0, 0.5, 0.05, 0.53, state, st["phis"], st["tcorh"], st["qcorh"])
    for k in state:
        assert np.array_equal(new[k][..., 0], state[k][..., 0])          # eps = 0: F(1) unchanged
        assert not np.array_equal(new[k][..., 1], state[k][..., 1])
    assert new["ps"][0, 0, 1] == state["ps"][0, 0, 0]                       # global-mean log(ps) is conserved
