"""GPU: the batched HIP spectral transforms (through the C-ABI) against (a) the committed fixtures generated from the
COMPILED REFERENCE (tests/golden/spectral_golden.npz) and (b) the CPU oracle on seeded random batches.
Tolerance: 1e-12 of the field's max-abs (north_star: 1e-10 relative); the pointwise operators are bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from speedy_ml_amd.spectral import IL, IX, MX2, NX, Spectral

pytestmark = pytest.mark.gpu
TOL = 1e-12


def dev(a):
    """numpy Fortran-shaped (..., d0, d1) field(s) -> torch [nf][d1][d0]"""
    a = np.asarray(a)
    if a.ndim == 2:
        a = a[None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 1))).cuda()


def host(t):
    return t.cpu().numpy().transpose(0, 2, 1)


def close(a, b, tol=TOL):
    return np.max(np.abs(a - b)) <= tol * max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def sp():
    return Spectral()


def test_tables_match_compiled_reference(sp, golden_spectral):
    from speedy_ml_amd.spectral import TABLES
    for w, (name, _) in TABLES.items():
        got, want = sp.table(w), golden_spectral["tab_" + name].ravel(order="F")
        if name in ("nsh2", "trfilt"):
            assert np.array_equal(got, want), name
        else:
            assert close(got, want, 1e-12), name


def test_transforms_against_golden(sp, golden_spectral):
    g = golden_spectral
    s, x = dev(g["in_spec"]), dev(g["in_grid"])
    assert close(host(sp.grid(s, 1)), g["grid_k1"])
    assert close(host(sp.grid(s, 2)), g["grid_k2"])
    assert close(host(sp.spec(x)), g["spec"])
    for kc in (1, 2):
        u = dev(np.stack([g["in_grid"][0], g["in_grid"][2]]))
        v = dev(np.stack([g["in_grid"][2], g["in_grid"][1]]))
        vor, div = sp.vdspec(u, v, kc)
        assert close(host(vor), g[f"vdspec_vor_k{kc}"], 1e-11) and close(host(div), g[f"vdspec_div_k{kc}"], 1e-11)


def test_operators_bit_exact_against_golden(sp, golden_spectral):
    g = golden_spectral
    s = g["in_spec"]
    a, b = dev(np.stack([s[0], s[1]])), dev(np.stack([s[1], s[0]]))
    u, v = sp.uvspec(a, b)
    assert np.array_equal(host(u), g["uvspec_u"]) and np.array_equal(host(v), g["uvspec_v"])
    vo, di = sp.vds(a, b)
    assert np.array_equal(host(vo), g["vds_vor"]) and np.array_equal(host(di), g["vds_div"])
    gx, gy = sp.grad(dev(s))
    assert np.array_equal(host(gx), g["grad_x"]) and np.array_equal(host(gy), g["grad_y"])
    assert np.array_equal(host(sp.lap(dev(s))), g["lap"])
    assert np.array_equal(host(sp.invlap(dev(s))), g["invlap"])
    assert np.array_equal(host(sp.trunct(dev(s + 1.0))), g["trunct"])


def test_large_batch_against_oracle(sp, oracle):
    """One launch for a whole transform set (73 forward / 91 inverse per SPEEDY step)."""
    rng = np.random.default_rng(31)
    nsh2 = oracle.table(12).astype(int)
    specs = rng.standard_normal((91, MX2, NX))
    for n in range(NX):
        specs[:, nsh2[n]:, n] = 0
    specs[:, 1, :] = 0
    out = host(sp.grid(dev(specs), 2))
    for i in (0, 17, 45, 90):
        assert close(out[i], oracle.grid(specs[i], 2))
    grids = rng.standard_normal((73, IX, IL)) * 30.0
    out = host(sp.spec(dev(grids)))
    for i in (0, 9, 36, 72):
        assert close(out[i], oracle.spec(grids[i]))
    # legendre analysis keeps row n=32 and everything above the triangular cut at exactly zero
    assert np.all(out[:, :, NX - 1] == 0.0)
    for n in range(NX):
        assert np.all(out[:, nsh2[n]:, n] == 0.0)


def test_roundtrip_property_full_batch(sp, oracle):
    """Size-independent property: spec(grid(s)) == s for triangularly truncated s."""
    rng = np.random.default_rng(33)
    s = rng.standard_normal((164, MX2, NX))
    for n in range(NX):
        for c in range(MX2):
            if c // 2 + n > 30:
                s[:, c, n] = 0
    s[:, 1, :] = 0
    back = host(sp.spec(sp.grid(dev(s), 1)))
    assert np.max(np.abs(back - s)) < 1e-12 * np.max(np.abs(s))
    # linearity
    a, b = dev(s[:80]), dev(s[80:160])
    lhs = host(sp.grid(a * 2.0 + b, 1))
    rhs = 2.0 * host(sp.grid(a, 1)) + host(sp.grid(b, 1))
    assert np.max(np.abs(lhs - rhs)) < 1e-12 * np.max(np.abs(rhs))


def test_empty_batch_and_bad_args(sp):
    from speedy_ml_amd._lib import SmlError
    e = torch.empty((0, NX, MX2), dtype=torch.float64, device="cuda")
    assert sp.grid(e, 1).shape == (0, IL, IX)
    with pytest.raises(SmlError):
        sp.grid(torch.zeros((1, NX, MX2), dtype=torch.float64, device="cuda"), 3)


def test_f77_dropin_symbols(oracle, golden_spectral):
    """grid_/spec_/vdspec_/... with the reference's external F77 calling convention (host arrays by reference)."""
    from speedy_ml_amd import _lib
    L = _lib.lib()
    a = C.c_double(6.371e6)
    L.parmtr_(C.byref(a))
    L.inifft_()
    g = golden_spectral
    vorm = np.asfortranarray(g["in_spec"][0])
    vorg = np.zeros((IX, IL), order="F")
    k = C.c_int(2)
    L.grid_(_lib.dp(vorm), _lib.dp(vorg), C.byref(k))
    assert close(vorg, g["grid_k2"][0])
    back = np.zeros((MX2, NX), order="F")
    x = np.asfortranarray(g["in_grid"][1])
    L.spec_(_lib.dp(x), _lib.dp(back))
    assert close(back, g["spec"][1])
    vor, div = np.zeros((MX2, NX), order="F"), np.zeros((MX2, NX), order="F")
    u, v = np.asfortranarray(g["in_grid"][0]), np.asfortranarray(g["in_grid"][2])
    L.vdspec_(_lib.dp(u), _lib.dp(v), _lib.dp(vor), _lib.dp(div), C.byref(k))
    assert close(vor, g["vdspec_vor_k2"][0], 1e-11) and close(div, g["vdspec_div_k2"][0], 1e-11)
    uc, vc = np.zeros((MX2, NX), order="F"), np.zeros((MX2, NX), order="F")
    s0, s1 = np.asfortranarray(g["in_spec"][0]), np.asfortranarray(g["in_spec"][1])
    L.uvspec_(_lib.dp(s0), _lib.dp(s1), _lib.dp(uc), _lib.dp(vc))
    assert np.array_equal(uc, g["uvspec_u"][0]) and np.array_equal(vc, g["uvspec_v"][0])
    L.vds_(_lib.dp(s0), _lib.dp(s1), _lib.dp(uc), _lib.dp(vc))
    assert np.array_equal(uc, g["vds_vor"][0])
    L.grad_(_lib.dp(s0), _lib.dp(uc), _lib.dp(vc))
    assert np.array_equal(uc, g["grad_x"][0]) and np.array_equal(vc, g["grad_y"][0])
    L.lap_(_lib.dp(s0), _lib.dp(uc))
    assert np.array_equal(uc, g["lap"][0])
    L.invlap_(_lib.dp(s0), _lib.dp(uc))
    assert np.array_equal(uc, g["invlap"][0])
    t = np.asfortranarray(g["in_spec"][0] + 1.0)
    L.trunct_(_lib.dp(t))
    assert np.array_equal(t, g["trunct"][0])


def test_mixed_flag_launches_match_separate_launches(sp):
    """A whole transform set in one launch (per-field kcos / pre-scale flags) == the separate launches, bit for bit."""
    rng = np.random.default_rng(41)
    s = torch.from_numpy(rng.standard_normal((10, NX, MX2))).cuda()
    flags = torch.tensor([1, 2, 2, 1, 1, 2, 1, 2, 2, 1], dtype=torch.int32, device="cuda")
    mixed = sp.grid_mixed(s, flags)
    k1, k2 = sp.grid(s, 1), sp.grid(s, 2)
    for i, f in enumerate(flags.tolist()):
        assert torch.equal(mixed[i], (k1 if f == 1 else k2)[i])
    g = torch.from_numpy(rng.standard_normal((6, IL, IX))).cuda()
    sc = torch.tensor([0, 1, 2, 1, 0, 2], dtype=torch.int32, device="cuda")
    mixed = sp.spec_mixed(g, sc)
    plain = sp.spec(g)
    u = torch.from_numpy(np.zeros((6, IL, IX))).cuda()
    for i, f in enumerate(sc.tolist()):
        if f == 0:
            assert torch.equal(mixed[i], plain[i])
    # scale 1 == the specx/specy halves of vdspec(.,.,2): feed (g, 0) and compare through vds linearity: vds(a, 0)
    zero = torch.zeros_like(g)
    vor, div = sp.vdspec(g, zero, 2)
    a = sp.spec_mixed(g, torch.ones(6, dtype=torch.int32, device="cuda"))
    vor2, div2 = sp.vds(a, sp.spec(zero))
    assert torch.equal(vor, vor2) and torch.equal(div, div2)


def test_grid_derived_equals_separate_operators(sp):
    """uvspec / grad folded into the inverse transform's staging give the same bits as uvspec / grad followed by grid."""
    rng = np.random.default_rng(11)
    base = torch.from_numpy(rng.standard_normal((6, NX, MX2))).cuda()
    sp.trunct(base)
    rows = [(0, 2, 2, 1), (1, 0, 1, 2), (2, 0, 1, 2), (3, 4, 4, 2), (4, 4, 4, 2), (0, 5, 5, 2), (2, 3, 2, 1), (1, 5, 0, 2)]
    desc = torch.tensor(rows, dtype=torch.int32, device="cuda")
    got = sp.grid_derived(base, desc)
    for f, (typ, a, b, kcos) in enumerate(rows):
        if typ == 0:
            src = base[a:a + 1]
        elif typ in (1, 2):
            u, v = sp.uvspec(base[a:a + 1].contiguous(), base[b:b + 1].contiguous())
            src = u if typ == 1 else v
        else:
            dx, dy = sp.grad(base[a:a + 1].contiguous())
            src = dx if typ == 3 else dy
        want = sp.grid(src.contiguous(), kcos)
        assert torch.equal(got[f], want[0]), (f, typ)


def test_grid_derived_geopotential_matches_geop_then_grid(sp, oracle):
    """type 7 of sml_spectral_grid_derived_aux: phypar's grid(phi1) with geop (src/dyn_geop.f90:19-35) folded into the staging,
    against the oracle's geop followed by its grid.  Spectral geopotential is formed with the reference's operation order (bit
    exact); the transform's DFT sums in another order: 1e-12 of the field's max."""
    from _oracle import DynOracle
    from speedy_ml_amd.dynamics import Dynamics
    do = DynOracle(oracle)
    dyn = Dynamics(sp)
    rng = np.random.default_rng(21)
    mask = np.repeat(oracle.table(11), 2, axis=0)                     # trfilt on (62, 32)
    t = rng.standard_normal((62, 32, 8)) * mask[..., None]
    t[0, 0] += 250.0 * np.sqrt(2.0)
    t[1, :] = 0.0                                                     # zonal means are real
    phis = rng.standard_normal((62, 32)) * 500.0 * mask
    phis[1, :] = 0.0
    want_spec = do.geop(t, phis)                                      # (62, 32, 8)
    xg1, xg2, hsg, fsg = dyn.table(7), dyn.table(8), dyn.table(1), dyn.table(3)
    corf = np.zeros(8)
    for k in range(1, 7):
        corf[k] = xg1[k] * 0.5 * np.log(hsg[k + 1] / fsg[k]) / np.log(fsg[k + 1] / fsg[k - 1])
    aux = torch.from_numpy(np.concatenate([xg1, xg2, corf, phis.T.ravel()])).cuda()
    base = torch.from_numpy(np.ascontiguousarray(t.transpose(2, 1, 0))).cuda()        # [8][32][62]
    desc = torch.tensor([(7, 0, k, 1) for k in range(8)], dtype=torch.int32, device="cuda")
    got = sp.grid_derived(base, desc, aux=aux).cpu().numpy()
    for k in range(8):
        want = oracle.grid(want_spec[..., k], 1).T
        assert np.max(np.abs(got[k] - want)) <= 1e-12 * np.max(np.abs(want)), k
    # the spectral geopotential itself, bit for bit: transform back is not needed -- level 7 has no correction and a single term
    back = sp.spec(torch.from_numpy(got[7:8].copy()).cuda()).cpu().numpy()[0].T
    assert np.max(np.abs(back - want_spec[..., 7] * mask)) <= 1e-11 * np.max(np.abs(want_spec[..., 7]))


def test_spec_post_equals_separate_operators(sp):
    """vds + trunct by descriptor give the same bits as vds followed by trunct."""
    rng = np.random.default_rng(12)
    raw = torch.from_numpy(rng.standard_normal((5, NX, MX2))).cuda()
    rows = [(5, 0, 1, 1), (6, 0, 1, 1), (0, 2, 2, 1), (6, 3, 4, 0), (0, 4, 4, 0), (5, 3, 4, 0)]
    desc = torch.tensor(rows, dtype=torch.int32, device="cuda")
    out = torch.zeros((len(rows), NX, MX2), dtype=torch.float64, device="cuda")
    sp.spec_post(raw, desc, out)
    for f, (typ, a, b, tr) in enumerate(rows):
        if typ == 0:
            want = raw[a:a + 1].clone()
        else:
            vor, div = sp.vds(raw[a:a + 1].contiguous(), raw[b:b + 1].contiguous())
            want = (vor if typ == 5 else div).clone()
        if tr:
            sp.trunct(want)
        assert torch.equal(out[f], want[0]), (f, typ)
    # the same rows with the last two written to a second array (the hybrid engine's iogrid(30) + fordate launch)
    out_a = torch.zeros((len(rows) - 2, NX, MX2), dtype=torch.float64, device="cuda")
    out_b = torch.full((2, NX, MX2), 7.0, dtype=torch.float64, device="cuda")
    sp.spec_post_split(raw, desc, out_a, out_b)
    assert torch.equal(out_a, out[:-2]) and torch.equal(out_b, out[-2:])
