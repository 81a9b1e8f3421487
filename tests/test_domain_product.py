"""CPU: the product's closed-form resdomain index maps (speedy-ml_amd/csrc/domain.cpp, through the C-ABI)
against the oracle's slice-and-reshape restatement of src/res_domain.f90.  Integers: bit-exact."""
import numpy as np
import pytest

from speedy_ml_amd import domain

REGION_COUNTS = (1152, 288, 4608, 72)


@pytest.mark.parametrize("nreg", REGION_COUNTS)
def test_region_extents_all_regions(oracle, nreg):
    for r in range(nreg):
        for ov in ((1, 2) if nreg != 4608 else (1,)):
            want = oracle.initializedomain(nreg, r, overlap=ov).asdict()
            got = domain.initializedomain(nreg, r, overlap=ov).asdict()
            for k, v in got.items():
                assert v == want[k], (nreg, r, ov, k)


def test_vertical_localisation_extents(oracle):
    for nvl in (1, 2, 4, 8):
        for vl in range(1, nvl + 1):
            for vo in (0, 1, 2):
                want = oracle.initializedomain(1152, 954, 1, nvl, vl, vo).asdict()
                got = domain.initializedomain(1152, 954, 1, nvl, vl, vo).asdict()
                for k, v in got.items():
                    assert v == want[k], (nvl, vl, vo, k)


def test_sizes_all_classes(oracle):
    for r in (0, 1, 23, 24, 500, 954, 1151):
        for sst in (0, 1):
            go, gp = oracle.initializedomain(1152, r), domain.initializedomain(1152, r)
            want = oracle.allocate_sizes(go, sst_input=sst).asdict()
            got = domain.allocate_res_sizes(gp, sst_bool_input=bool(sst)).asdict()
            for k, v in got.items():
                assert v == want[k], (r, sst, k)


@pytest.mark.parametrize("nprocs", (1, 2, 4, 5, 7, 8, 16))
def test_processor_decomposition(oracle, nprocs):
    for p in range(nprocs):
        assert np.array_equal(domain.processor_decomposition_manual(p, nprocs, 1152),
                              oracle.processor_decomposition(p, nprocs, 1152))


def test_out_map_equals_oracle_scatter(oracle):
    # scatter a tagged outvec through the oracle tiler; the product map must point at the same cells
    for r in (0, 1, 23, 24, 47, 48, 500, 954, 1128, 1151):
        gi, si = domain.out_map(1152, r)
        assert len(gi) == 136
        g4, g2, gp = np.full(147456, -1.0), np.full(4608, -1.0), np.full(4608, -1.0)
        oracle.scatter_res(1152, r, np.arange(136, dtype=float), g4, g2, gp)
        G = np.concatenate([g4, g2, gp])
        assert np.array_equal(G[gi], np.arange(136, dtype=float))
        assert (G >= 0).sum() == 136
        # first 132 entries == gather map of tile_4d_and_logp_full_grid_to_local_res_vec
        tag4, tag2 = np.arange(147456, dtype=float), 1e6 + np.arange(4608, dtype=float)
        want = oracle.tile_res(1152, r, tag4, tag2, 132)
        assert np.array_equal(np.concatenate([tag4, tag2])[gi[:132]], want)
        # mean/std slots: T,u,v,q x 8 levels, logp 32, precip 34
        assert list(si[:4]) == [0, 8, 16, 24] and si[16] == 1
        assert list(si[128:132]) == [32] * 4 and list(si[132:]) == [34] * 4


def test_in_map_equals_oracle_tiler(oracle):
    tag4 = np.arange(147456, dtype=float)
    tag2 = 1e6 + np.arange(4608, dtype=float)
    tagp = 2e6 + np.arange(4608, dtype=float)
    tags = 3e6 + np.arange(4608, dtype=float)
    tagt = 4e6 + np.arange(4608, dtype=float)
    G = np.concatenate([tag4, tag2, tagp, tags, tagt])
    assert G.size == domain.G_SIZE
    for r in range(1152):
        for sst in (True, False):
            gi, si = domain.in_map(1152, r, sst_bool_input=sst)
            g = oracle.initializedomain(1152, r)
            s = oracle.allocate_sizes(g, sst_input=int(sst))
            assert len(gi) == s.reservoir_numinputs
            want = oracle.tile_input(1152, r, tag4, tag2, tagp, s.precip_end)
            assert np.array_equal(G[gi[:s.precip_end]], want), r
            in2d = g.inputxchunk * g.inputychunk
            if sst:
                assert np.array_equal(G[gi[s.sst_start - 1:s.sst_end]], oracle.tile_input2d(1152, r, tags, in2d))
                assert np.all(si[s.sst_start - 1:s.sst_end] == 35)
            assert np.array_equal(G[gi[s.tisr_start - 1:s.tisr_end]], oracle.tile_input2d(1152, r, tagt, in2d))
            assert np.all(si[s.tisr_start - 1:s.tisr_end] == 33)
            assert np.all(si[s.logp_start - 1:s.logp_end] == 32) and np.all(si[s.precip_start - 1:s.precip_end] == 34)
            if r % 97 == 0:
                # statistics slots of the 3-d block follow standardize_state_vec_input (var-major, level-minor)
                mean, std = np.arange(36, dtype=float), np.ones(36)
                v = oracle.standardize_input(g, s, mean, std, np.zeros(s.reservoir_numinputs))
                assert np.array_equal(-v[:s.logp_end], si[:s.logp_end].astype(float))


def test_bad_arguments_fail_loudly():
    from speedy_ml_amd._lib import SmlError
    with pytest.raises(SmlError):
        domain.initializedomain(1152, 1152)
    with pytest.raises(SmlError):
        domain.processor_decomposition_manual(8, 8, 1152)


def test_message_sizes_and_batch_divisor(oracle):
    import numpy as np
    from speedy_ml_amd import _lib
    sizes = np.zeros(5, dtype=np.int32)
    # interior region: outvec 136, SPEEDY patch 132, input without sst/tisr 544, slab 8 / 32
    _lib.check(_lib.lib().sml_domain_message_sizes(1152, 954, 1, 1, 1, 0, 1, 1, _lib.ip(sizes)))
    assert list(sizes) == [136, 132, 544, 8, 32]
    _lib.check(_lib.lib().sml_domain_message_sizes(1152, 0, 1, 1, 1, 0, 1, 1, _lib.ip(sizes)))
    assert list(sizes) == [136, 132, 408, 8, 24]          # polar: 4x3 input patch
    for target, number in ((98, 1960), (7, 64), (2920, 58400), (13, 100), (97, 1960)):
        assert _lib.lib().sml_find_closest_divisor(target, number) == oracle.find_closest_divisor(target, number)
    assert _lib.lib().sml_find_closest_divisor(98, 1960) == 98


def test_calendar_and_tisr_index_match_oracle(oracle):
    """get_current_time_delta_hour / numof_hours_into_year / get_tisr_by_date (src/mod_calendar.f90, src/mpires.f90:1676-1708):
    integer bookkeeping, bit-exact, quirks included."""
    from speedy_ml_amd import domain
    rng = np.random.default_rng(5)
    hours = list(range(0, 24 * 800)) + list(range(0, 45 * 8760, 6))[::7] + [int(h) for h in rng.integers(0, 60 * 8760, 2000)]
    for h in hours:
        assert domain.calendar_date(h) == oracle.calendar_date(h), h
        assert domain.tisr_index(h) == oracle.tisr_index(h), h
        assert 1 <= domain.tisr_index(h) <= 8760
    # hand-evaluated from the Fortran: hour 0 is "day 0 of the year" -> 31 December of the previous year, slice 8760;
    # one year (8760 h) after 1 Jan 1981 the calendar reads 31 Dec 1981; the shipped configuration starts predictions
    # 12000 + 359*? hours in -- only the arithmetic is checked here
    assert domain.calendar_date(0) == (1980, 12, 31, 0) and domain.tisr_index(0) == 8760
    assert domain.calendar_date(24) == (1981, 1, 1, 0) and domain.tisr_index(24) == 1
    assert domain.calendar_date(25) == (1981, 1, 1, 1) and domain.tisr_index(25) == 1
    assert domain.calendar_date(24 * 32 + 5) == (1981, 2, 1, 5) and domain.tisr_index(24 * 32 + 5) == 31 * 24 + 5
    assert domain.calendar_date(8760) == (1981, 12, 31, 0)
    # 1984 is a leap year: three years in, leap_days is still 0; four years in it is 1
    assert domain.calendar_date(4 * 8760 + 24 * 60)[0] == 1985


def test_slab_sizes_match_survey():
    """initialize_slab_ocean_model (src/mod_slab_ocean_reservoir.f90:57-124); SURVEY 8a row 11: d=128, n=3968, k=23617, out=8."""
    from speedy_ml_amd import domain
    from speedy_ml_amd.slab import slab_sizes
    s = slab_sizes(domain.initializedomain(1152, 954))
    assert (s.reservoir_numinputs, s.n, s.k, s.chunk_size_prediction, s.chunk_size, s.chunk_size_speedy, s.nodes_per_input) == (128, 3968, 23617, 8, 8, 0, 31)
    assert (s.atmo3d_start, s.atmo3d_end, s.logp_start, s.logp_end, s.sst_start, s.sst_end, s.tisr_start, s.tisr_end) == (1, 64, 65, 80, 81, 96, 97, 112)
    p = slab_sizes(domain.initializedomain(1152, 0))          # polar region: 4 x 3 input patch
    assert (p.reservoir_numinputs, p.nodes_per_input, p.n, p.chunk_size_prediction) == (96, 42, 4032, 8)


@pytest.mark.parametrize("region", [0, 1, 23, 24, 954, 1127, 1151])
def test_target_map_matches_oracle_tiler(oracle, region):
    """tile_full_input_to_target_data (src/res_domain.f90:602-689): the product's index map applied to an index-tagged input vector
    gives what the oracle's restatement of the reshape/slice/reshape gives; polar, x-periodic and interior regions."""
    g = oracle.initializedomain(1152, region)
    s = oracle.allocate_sizes(g)
    rng = np.random.default_rng(region)
    u = np.asfortranarray(np.arange(s.reservoir_numinputs * 3, dtype=np.float64).reshape(3, -1).T + rng.random((s.reservoir_numinputs, 3)))
    want = oracle.tile_target(g, s, u, s.chunk_size_prediction)
    rows = domain.target_map(1152, region)
    assert rows.size == s.chunk_size_prediction == 136
    assert np.array_equal(u[rows, :], want)
    assert len(set(rows.tolist())) == rows.size and rows.max() < s.precip_end
    if region == 0:
        # SURVEY Appendix A (run of the reference): region 0 trains on tdata x 2-3, y 1-2 of its 4 x 3 input patch
        nx = g.inputxchunk
        first_level = rows[:16].reshape(2, 2, 4)           # (y, x, var)
        xs = (first_level[..., 0] // 4) % nx + 1
        ys = (first_level[..., 0] // 4) // nx % g.inputychunk + 1
        assert sorted(set(xs.ravel().tolist())) == [2, 3] and sorted(set(ys.ravel().tolist())) == [1, 2]
    # without precipitation the target vector stops after logp
    assert domain.target_map(1152, region, precip_bool=False).size == 132


def test_target_map_rejects_bad_arguments():
    """errors are loud: bad region / level / capacity come back as SML_ERR_ARG with a message (no silent clamping)"""
    from speedy_ml_amd import _lib
    buf = np.zeros(16, dtype=np.int32)
    for args in ((1152, 1152, 1, 1, 1, 0, 1), (1152, -1, 1, 1, 1, 0, 1), (1152, 0, 1, 8, 9, 0, 1)):
        rc = _lib.lib().sml_domain_target_map(*args, _lib.ip(buf), 4608 * 8)
        assert rc < 0
    rc = _lib.lib().sml_domain_target_map(1152, 954, 1, 1, 1, 0, 1, _lib.ip(buf), 16)        # 136 entries do not fit 16
    assert rc < 0
    with pytest.raises(Exception):
        domain.target_map(1152, 5000)


@pytest.mark.parametrize("nreg,nranks", [(1152, 1), (1152, 2), (1152, 5), (1152, 7), (1152, 8), (1152, 16), (288, 5), (72, 7), (13, 4)])
def test_region_owner_inverts_processor_decomposition(nreg, nranks):
    """sml_domain_region_owner (used by the ragged all-gather of the C-ABI, sml_comm_unpack_regions): rank p's i-th region is owned
    by (p, i), for even splits and for the remainder rule of src/res_domain.f90:53-60."""
    import ctypes as C
    from speedy_ml_amd import _lib
    seen = set()
    for p in range(nranks):
        for i, r in enumerate(domain.processor_decomposition_manual(p, nranks, nreg)):
            rk, sl = C.c_int(), C.c_int()
            _lib.check(_lib.lib().sml_domain_region_owner(nranks, nreg, int(r), C.byref(rk), C.byref(sl)))
            assert (rk.value, sl.value) == (p, i), (p, i, r)
            seen.add(int(r))
    assert seen == set(range(nreg))
