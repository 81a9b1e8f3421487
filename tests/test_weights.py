"""The reference's trained-weights files (src/mod_reservoir.f90:1703-1776, src/mod_io.f90:1274-1319,1434-1523,2938-3030): classic
NetCDF (CDF-1) with float32 real variables.  The writer is checked against the PUBLISHED classic-format layout with an independent
parser written here from the format specification (NetCDF classic format: magic, numrecs, dim_list, gatt_list, var_list, big-endian
data at `begin`), so that a file a netCDF-linked reference reads is what comes out; the reader is checked on those files."""
import os
import struct

import numpy as np
import pytest

from speedy_ml_amd import weights


def parse_cdf1(buf):
    """Independent parser of the classic header (netcdf classic format specification): returns (dims, vars)."""
    assert buf[:4] == b"CDF\x01"
    pos = 4

    def u32():
        nonlocal pos
        v = struct.unpack(">I", buf[pos:pos + 4])[0]
        pos += 4
        return v

    def name():
        nonlocal pos
        n = u32()
        s = buf[pos:pos + n].decode()
        pos += (n + 3) // 4 * 4
        return s

    def att_list():
        nonlocal pos
        tag, n = u32(), u32()
        assert tag in (0, 12)
        out = {}
        for _ in range(n):
            k = name()
            typ, cnt = u32(), u32()
            size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 4, 6: 8}[typ] * cnt
            out[k] = (typ, buf[pos:pos + size])
            pos += (size + 3) // 4 * 4
        return out

    numrecs = u32()
    assert numrecs == 0
    tag, ndims = u32(), u32()
    assert tag == 10
    dims = [(name(), u32()) for _ in range(ndims)]
    gatts = att_list()
    tag, nvars = u32(), u32()
    assert tag == 11
    out_vars = []
    for _ in range(nvars):
        vname = name()
        nd = u32()
        dimids = [u32() for _ in range(nd)]
        atts = att_list()
        typ, vsize, begin = u32(), u32(), u32()
        out_vars.append(dict(name=vname, dimids=dimids, atts=atts, type=typ, vsize=vsize, begin=begin))
    return dims, gatts, out_vars


@pytest.fixture
def sample():
    rng = np.random.default_rng(3)
    n, d, n_model, n_out, k = 96, 40, 12, 16, 500
    return dict(win=rng.uniform(-0.5, 0.5, (n, d)), wout=rng.standard_normal((n_out, n + n_model)) * 1e-2,
                rows=rng.integers(1, n + 1, k), cols=rng.integers(1, n + 1, k), vals=rng.uniform(0, 1, k),
                mean=rng.uniform(-1, 1, 36), std=rng.uniform(0.5, 2, 36))


def test_filenames():
    assert weights.trained_res_filename(954, "6000_20_20_20_sigma0.5_beta_res1.0 ") == "worker_0954_level_1_6000_20_20_20_sigma0.5_beta_res1.0.nc"
    assert weights.trained_res_filename(7, "trial", level_index=3) == "worker_0007_level_3_trial.nc"
    assert weights.trained_res_filename(1151, "trial", ocean=True) == "worker_1151_ocean_trial.nc"


def test_written_file_is_the_reference_layout(tmp_path, sample):
    path = str(tmp_path / weights.trained_res_filename(954, "t"))
    weights.write_trained_res(path, **sample)
    buf = open(path, "rb").read()
    dims, gatts, vs = parse_cdf1(buf)
    n, d = sample["win"].shape
    n_out, n_aug = sample["wout"].shape
    k = sample["vals"].size
    # dimensions in the order the reference's helper calls define them (x before y, variable by variable)
    assert dims == [("win_x", n), ("win_y", d), ("wout_x", n_out), ("wout_y", n_aug), ("rows_x", k), ("cols_x", k), ("vals_x", k),
                    ("mean_x", 36), ("std_x", 36)]
    assert gatts == {}
    assert [v["name"] for v in vs] == ["win", "wout", "rows", "cols", "vals", "mean", "std"]
    NC_INT, NC_FLOAT = 4, 5
    assert [v["type"] for v in vs] == [NC_FLOAT, NC_FLOAT, NC_INT, NC_INT, NC_FLOAT, NC_FLOAT, NC_FLOAT]
    # Fortran dims (x, y) are stored reversed: the file's slowest dimension is y
    assert vs[0]["dimids"] == [1, 0] and vs[1]["dimids"] == [3, 2] and vs[2]["dimids"] == [4]
    for v in vs:
        typ, raw = v["atts"]["units"]
        assert typ == 2 and raw == b"unitless"
    # data: big-endian float32 in Fortran element order of win(n, d) = C order of [d][n]
    raw = np.frombuffer(buf, dtype=">f4", count=n * d, offset=vs[0]["begin"]).reshape(d, n)
    assert np.array_equal(raw, sample["win"].T.astype(np.float32))
    raw = np.frombuffer(buf, dtype=">i4", count=k, offset=vs[3]["begin"])
    assert np.array_equal(raw, sample["cols"])


def test_roundtrip_carries_float32_rounding(tmp_path, sample):
    path = str(tmp_path / "w.nc")
    weights.write_trained_res(path, **sample)
    got = weights.read_trained_res(path)
    for k in ("win", "wout", "vals", "mean", "std"):
        want = sample[k].astype(np.float32).astype(np.float64)
        assert got[k].dtype == np.float64 and got[k].shape == sample[k].shape
        assert np.array_equal(got[k], want), k
        assert not np.array_equal(got[k], sample[k])           # the reference's NF90_REAL quirk is kept
    assert got["win"].flags.f_contiguous
    assert np.array_equal(got["rows"], sample["rows"]) and np.array_equal(got["cols"], sample["cols"])
    # writing again replaces the file (NF90_CLOBBER on the first helper call)
    weights.write_trained_res(path, **dict(sample, mean=sample["mean"] + 1))
    assert np.allclose(weights.read_trained_res(path)["mean"], sample["mean"] + 1, atol=1e-6)


def test_reader_errors(tmp_path, sample):
    from scipy.io import netcdf_file
    with pytest.raises(FileNotFoundError):
        weights.read_trained_res(str(tmp_path / "absent.nc"))
    path = str(tmp_path / "broken.nc")
    f = netcdf_file(path, "w", version=1)
    f.createDimension("win_x", 3)
    f.createVariable("win", "f", ("win_x",))[:] = np.zeros(3, dtype=np.float32)
    f.close()
    with pytest.raises(ValueError):
        weights.read_trained_res(path)                          # win has the wrong rank
    with pytest.raises(ValueError):
        weights.write_trained_res(path, **dict(sample, cols=sample["cols"][:-1]))


def test_controller_file_roundtrip(tmp_path):
    params = dict(num_hor_regions=1152, ml_only=False, num_vert_levels=1, atmo_timestep=6, ocean_timestep=168, ocean_model_bool=True,
                  train_on_sst_anomalies=False, precip_bool=True, precip_epsilon=0.001, full_predictvars=4, full_heightlevels=8,
                  vert_loc_overlap=0, overlap=1, regional_vary=True, using_prior=True, reservoir_nodes=6000, deg=6, radius=0.9,
                  beta_res=0.001, beta_model=1.0, sigma=0.5, leakage=1.0, prior_val=0.0)
    path = str(tmp_path / "trial_controller_file.txt")
    weights.write_controller_file(path, params)
    lines = open(path).read().splitlines()
    assert lines[0].strip() == "-" * 59 and lines[-1].strip() == "-" * 59 and len(lines) == len(weights.CONTROLLER_KEYS) + 2
    assert lines[1].strip() == "num_hor_regions: 1152" and lines[2].strip() == "ml_only: F"
    assert weights.read_controller_file(path) == params
