"""CPU: pins the spectral oracle (oracle/spectral_oracle.c) against the compiled reference.

Two legs:  (1) committed fixtures tests/golden/spectral_golden.npz (generated from oracle/_ref by
tests/golden/make_spectral_golden.py) -- runs everywhere;  (2) live comparison against
oracle/_ref/libref_spectral.so when it is present (this container, or the GPU box via the travelling .so).
Tolerance: 1e-12 of the field's max-abs (SPEC_TOL) -- north_star allows 1e-10 relative.
"""
import numpy as np
import pytest

from _oracle import IL, IX, MX2, NX, TABLES, RefSpectral

SPEC_TOL = 1e-12


def close(a, b, tol=SPEC_TOL):
    scale = max(np.max(np.abs(b)), 1e-300)
    return np.max(np.abs(a - b)) <= tol * scale


@pytest.mark.parametrize("which", sorted(TABLES))
def test_tables_match_reference(oracle, golden_spectral, which):
    name, _ = TABLES[which]
    got, want = oracle.table(which), golden_spectral["tab_" + name]
    if name in ("nsh2", "trfilt"):
        assert np.array_equal(got, want)
    else:
        assert close(got, want, 1e-12), name   # 1-ulp libm differences in gaussl amplify through 1-sia**2


def test_nsh2_triangular_cut(oracle):
    nsh2 = oracle.table(12).astype(int)
    assert list(nsh2[:4]) == [62, 62, 60, 58] and nsh2[-1] == 2  # SURVEY Appendix B probe


def test_grid_spec_against_golden(oracle, golden_spectral):
    g = golden_spectral
    for i, s in enumerate(g["in_spec"]):
        assert close(oracle.grid(s, 1), g["grid_k1"][i])
        assert close(oracle.grid(s, 2), g["grid_k2"][i])
        assert close(oracle.gridy(s), g["gridy"][i])
    for i, x in enumerate(g["in_grid"]):
        assert close(oracle.spec(x), g["spec"][i])
        assert close(oracle.specx(x), g["specx"][i])
        assert close(oracle.specy(g["specx"][i]), g["specy"][i])


def test_vdspec_against_golden(oracle, golden_spectral):
    g = golden_spectral
    pairs = [(g["in_grid"][0], g["in_grid"][2]), (g["in_grid"][2], g["in_grid"][1])]
    for kc in (1, 2):
        for i, (u, v) in enumerate(pairs):
            vor, div = oracle.vdspec(u, v, kc)
            assert close(vor, g[f"vdspec_vor_k{kc}"][i], 1e-11)
            assert close(div, g[f"vdspec_div_k{kc}"][i], 1e-11)


def test_operators_bit_exact(oracle, golden_spectral):
    g = golden_spectral
    s = g["in_spec"]
    for i, (a, b) in enumerate([(s[0], s[1]), (s[1], s[0])]):
        u, v = oracle.uvspec(a, b)
        assert np.array_equal(u, g["uvspec_u"][i]) and np.array_equal(v, g["uvspec_v"][i])
        vo, di = oracle.vds(a, b)
        assert np.array_equal(vo, g["vds_vor"][i]) and np.array_equal(di, g["vds_div"][i])
    for i, a in enumerate(s):
        gx, gy = oracle.grad(a)
        assert np.array_equal(gx, g["grad_x"][i]) and np.array_equal(gy, g["grad_y"][i])
        assert np.array_equal(oracle.lap(a), g["lap"][i])
        assert np.array_equal(oracle.invlap(a), g["invlap"][i])
        assert np.array_equal(oracle.trunct(a + 1.0), g["trunct"][i])


def test_fftpack_semantics(oracle, golden_spectral):
    g = golden_spectral
    for i, r in enumerate(g["in_r96"]):
        assert close(oracle.rfftf(r), g["rfftf"][i])
        assert close(oracle.rfftb(r), g["rfftb"][i])
        # rfftb(rfftf(x)) = n*x  (FFTPACK is unnormalised)
        assert close(oracle.rfftb(oracle.rfftf(r)), 96.0 * r)


def test_single_mode_convention(oracle):
    # (Re,Im)=(1,1/2) at m=1 synthesises to 2cos(theta) - sin(theta); specx returns (1,1/2)
    varm = np.zeros((MX2, IL))
    varm[2, :] = 1.0
    varm[3, :] = 0.5
    vg = oracle.gridx(varm, 1)
    th = 2 * np.pi * np.arange(IX) / IX
    assert np.allclose(vg[:, 0], 2 * np.cos(th) - np.sin(th), atol=1e-13)
    back = oracle.specx(vg)
    assert np.allclose(back[2, :], 1.0, atol=1e-14) and np.allclose(back[3, :], 0.5, atol=1e-14)


def test_roundtrip_idempotent(oracle, golden_spectral):
    # spec(grid(s)) == s for a triangularly truncated state (reference probe: idempotent to 2.4e-13)
    s = golden_spectral["in_spec"][1].copy()
    nsh2 = oracle.table(12).astype(int)
    for n in range(NX):
        for c in range(MX2):
            if c // 2 + n > 30:
                s[c, n] = 0.0
    back = oracle.spec(oracle.grid(s, 1))
    assert np.max(np.abs(back - s)) < 1e-12 * np.max(np.abs(s))


def test_gauss_weights_sum(oracle):
    assert abs(oracle.table(3).sum() - 1.0) < 1e-13
    assert abs(oracle.table(1)[0] - 0.998771) < 1e-6   # sia(1), SURVEY Appendix C


@pytest.mark.skipif(not RefSpectral.available(), reason="oracle/_ref not built (needs /root/reference + amdflang)")
def test_live_against_compiled_reference(oracle):
    ref = RefSpectral()
    rng = np.random.default_rng(7)
    nsh2 = oracle.table(12).astype(int)
    for _ in range(3):
        v = rng.standard_normal((MX2, NX))
        for n in range(NX):
            v[nsh2[n]:, n] = 0
        v[1, :] = 0
        assert close(oracle.grid(v, 2), ref.grid(v, 2))
        x = rng.standard_normal((IX, IL)) * 50
        assert close(oracle.spec(x), ref.spec(x))
        y = rng.standard_normal((IX, IL))
        for kc in (1, 2):
            a, b = oracle.vdspec(x, y, kc)
            ra, rb = ref.vdspec(x, y, kc)
            assert close(a, ra, 1e-11) and close(b, rb, 1e-11)
