"""CPU: checks the resdomain oracle (oracle/domain_oracle.c) against the reference-run facts of SURVEY.md
Appendix A and the reference's own known answer (tests/mod_unit_test.f90:63-96).  Integers: bit-exact."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FACTS = json.load(open(os.path.join(HERE, "golden", "survey_appendix_a.json")))


def tagged_grids():
    g4 = np.zeros((8, 48, 96, 4))  # C order [z][y][x][v] == Fortran (v,x,y,z)
    z, y, x, v = np.meshgrid(np.arange(1, 9), np.arange(1, 49), np.arange(1, 97), np.arange(1, 5), indexing="ij")
    g4[...] = v * 1e6 + x * 1e4 + y * 1e2 + z
    yy, xx = np.meshgrid(np.arange(1, 49), np.arange(1, 97), indexing="ij")
    g2 = (7e6 + xx * 1e4 + yy * 1e2).astype(float)
    gp = (8e6 + xx * 1e4 + yy * 1e2).astype(float)
    return g4.ravel(), g2.ravel(), gp.ravel()


def test_reference_known_answer(oracle):
    f = FACTS["getxyresextent_288_145"]
    xs, xe, ys, ye, xc, yc = oracle.getxyresextent(288, 145)
    assert (xs, xe, xc, yc) == (f["xs"], f["xe"], f["xchunk"], f["ychunk"])   # what the reference test pins
    assert (ys, ye) == (f["ys"], f["ye"])                                      # what the shipped code yields


def test_regions_match_survey(oracle):
    for r, f in FACTS["regions_1152"].items():
        g = oracle.initializedomain(1152, int(r))
        assert [g.res_xstart, g.res_xend, g.res_ystart, g.res_yend] == f["res"]
        assert [g.input_xstart, g.input_xend, g.input_ystart, g.input_yend] == f["input"]
        assert bool(g.pole) == f["pole"] and bool(g.periodicboundary) == f["periodic"]
        if "tdata" in f:
            assert [g.tdata_xstart, g.tdata_xend, g.tdata_ystart, g.tdata_yend] == f["tdata"]
        assert (g.res_zstart, g.res_zend, g.reszchunk, g.inputzchunk, g.bottom, g.top) == (1, 8, 8, 8, 1, 1)


def test_corner_rule_and_counts(oracle):
    polar = periodic = 0
    seen = np.zeros((48, 96), dtype=int)
    for r in range(1152):
        g = oracle.initializedomain(1152, r)
        assert g.res_xstart == (r // 24) * 2 + 1 and g.res_ystart == (r % 24) * 2 + 1
        assert g.resxchunk == 2 and g.resychunk == 2
        seen[g.res_ystart - 1:g.res_yend, g.res_xstart - 1:g.res_xend] += 1
        polar += g.pole
        periodic += g.periodicboundary
    assert np.all(seen == 1)                        # res patches tile the globe exactly once
    assert polar == FACTS["counts_1152"]["polar"] and periodic == FACTS["counts_1152"]["periodic"]


def test_size_classes(oracle):
    sc = FACTS["size_classes"]
    gi, gp = oracle.initializedomain(1152, 954), oracle.initializedomain(1152, 0)
    for g, sst, key in ((gi, 1, "interior_sst"), (gi, 0, "interior_land"), (gp, 1, "polar_sst"), (gp, 0, "polar_land")):
        s = oracle.allocate_sizes(g, sst_input=sst)
        assert (s.reservoir_numinputs, s.n, s.k) == (sc[key]["d"], sc[key]["n"], sc[key]["k"]), key
    s = oracle.allocate_sizes(gi, sst_input=1)
    assert s.chunk_size_prediction == 136 and s.chunk_size_speedy == 132 and s.n + s.chunk_size_speedy == 5892
    lay = FACTS["input_layout_interior"]
    assert [s.atmo3d_start, s.atmo3d_end] == lay["atmo3d"] and [s.logp_start, s.logp_end] == lay["logp"]
    assert [s.precip_start, s.precip_end] == lay["precip"] and [s.sst_start, s.sst_end] == lay["sst"]
    assert [s.tisr_start, s.tisr_end] == lay["tisr"]


def test_tile_input_region0_layout(oracle):
    g4, g2, gp = tagged_grids()
    f = FACTS["tile_input_region0"]
    g = oracle.initializedomain(1152, 0)
    n = 4 * g.inputxchunk * g.inputychunk * 8 + 2 * g.inputxchunk * g.inputychunk
    v = oracle.tile_input(1152, 0, g4, g2, gp, n)

    def tag(var, x, y, z):
        return var * 1e6 + x * 1e4 + y * 1e2 + z
    for (lo, key) in ((0, "v1_4"), (4, "v5_8"), (48, "v49_52")):
        x, y, z = f[key]
        assert list(v[lo:lo + 4]) == [tag(k, x, y, z) for k in (1, 2, 3, 4)]
    ls = f["logp_start"] - 1
    assert list(v[ls:ls + 5]) == [7e6 + x * 1e4 + y * 1e2 for x, y in f["logp_order"]]
    ps = f["precip_start"] - 1
    assert v[ps] == 8e6 + 96 * 1e4 + 1 * 1e2 and n == ps + 12


def test_scatter_region954_layout(oracle):
    f = FACTS["scatter_region954"]
    g4, g2, gp = np.zeros(4 * 96 * 48 * 8), np.zeros(96 * 48), np.zeros(96 * 48)
    oracle.scatter_res(1152, 954, np.arange(1, 137, dtype=float), g4, g2, gp)
    G4 = g4.reshape(8, 48, 96, 4)

    def at(var, x, y, z):
        return G4[z - 1, y - 1, x - 1, var - 1]
    x, y, z = f["1_4"]
    assert [at(k, x, y, z) for k in (1, 2, 3, 4)] == [1, 2, 3, 4]
    x, y, z = f["5_8"]
    assert [at(k, x, y, z) for k in (1, 2, 3, 4)] == [5, 6, 7, 8]
    assert at(1, 79, 38, 1) == 9 and at(1, 79, 37, 2) == 17 and at(4, 80, 38, 8) == 128
    assert [g2.reshape(48, 96)[yy - 1, xx - 1] for xx, yy in f["logp_129_132"]] == [129, 130, 131, 132]
    assert [gp.reshape(48, 96)[yy - 1, xx - 1] for xx, yy in f["precip_133_136"]] == [133, 134, 135, 136]
    # inverse tiler returns the first 132 entries
    back = oracle.tile_res(1152, 954, g4, g2, 132)
    assert list(back) == list(range(1, 133))


def test_scatter_then_tile_roundtrip_all_regions(oracle):
    rng = np.random.default_rng(3)
    g4, g2, gp = np.zeros(4 * 96 * 48 * 8), np.zeros(96 * 48), np.zeros(96 * 48)
    vecs = rng.standard_normal((1152, 136))
    for r in range(1152):
        oracle.scatter_res(1152, r, vecs[r], g4, g2, gp)
    for r in (0, 1, 23, 24, 500, 954, 1151):
        assert np.array_equal(oracle.tile_res(1152, r, g4, g2, 132), vecs[r][:132])
        g = oracle.initializedomain(1152, r)
        s = oracle.allocate_sizes(g)
        inp = oracle.tile_input(1152, r, g4, g2, gp, s.precip_end)
        # the res patch sits inside the input patch at tdata indices
        loc = inp[:s.atmo3d_end].reshape(8, g.inputychunk, g.inputxchunk, 4)
        sub = loc[:, g.tdata_ystart - 1:g.tdata_yend, g.tdata_xstart - 1:g.tdata_xend, :]
        assert np.array_equal(sub.ravel(), vecs[r][:128])


def test_processor_decomposition(oracle):
    f = FACTS["processor_decomposition"]
    for p in range(8):
        idx = oracle.processor_decomposition(p, 8, 1152)
        assert list(idx) == list(range(144 * p, 144 * p + 144))
    idx = oracle.processor_decomposition(f["7_ranks_1152"]["rank"], 7, 1152)
    assert len(idx) == f["7_ranks_1152"]["count"] and idx[-1] == f["7_ranks_1152"]["last"]
    # every region is owned exactly once, also with a remainder
    for nprocs in (1, 2, 4, 5, 7, 8):
        owned = np.concatenate([oracle.processor_decomposition(p, nprocs, 1152) for p in range(nprocs)])
        assert sorted(owned) == list(range(1152))


def test_standardize_roundtrip(oracle):
    rng = np.random.default_rng(5)
    mean, std = rng.uniform(-1, 1, 36), rng.uniform(0.5, 2, 36)
    g = oracle.initializedomain(1152, 954)
    v = rng.standard_normal(136)
    un = oracle.unstandardize_res(g, mean, std, v)
    # element 0 is T level 1 -> slot 0 ; element 1 is u level 1 -> slot 8 ; logp slot 32 ; precip slot 34
    assert un[0] == v[0] * std[0] + mean[0] and un[1] == v[1] * std[8] + mean[8]
    assert un[16] == v[16] * std[1] + mean[1]
    assert un[128] == v[128] * std[32] + mean[32] and un[132] == v[132] * std[34] + mean[34]
    st = oracle.standardize_res(g, mean, std, un[:132])
    assert np.allclose(st, v[:132], rtol=0, atol=1e-14)
    s = oracle.allocate_sizes(g)
    u = rng.standard_normal(576)
    su = oracle.standardize_input(g, s, mean, std, u)
    assert su[0] == (u[0] - mean[0]) / std[0] and su[1] == (u[1] - mean[8]) / std[8]
    assert su[512] == (u[512] - mean[32]) / std[32]
    assert np.array_equal(su[528:], u[528:])   # precip/sst/tisr untouched by standardize_state_vec_input


def test_radius_by_lat(oracle):
    assert oracle.radius_by_lat(-87.159, -83.479) == 0.7
    assert oracle.radius_by_lat(-1.856, 1.856) == (0.7 - 0.3) / 45.0 + 0.3   # quirk Q5: constant, not a ramp
