"""The reference's trained-weights files: `worker_RRRR_level_L_<trial>.nc` and `worker_RRRR_ocean_<trial>.nc`.

`write_trained_res` (src/mod_reservoir.f90:1703-1737) appends seven variables to one file through the `write_netcdf_*_non_met_data`
helpers of src/mod_io.f90 (:1274-1319, :1434-1523).  Those helpers create the file with `NF90_CLOBBER` only, i.e. the CLASSIC
NetCDF format (CDF-1, not NetCDF-4/HDF5), and define every real variable as `NF90_REAL`: the double-precision arrays are stored as
FLOAT32 (nf90_put_var converts), `rows`/`cols` as `NF90_INT`.  `read_trained_res` (src/mod_io.f90:2938-2983) reads them back into
real(dp) arrays, so a trained model that went through a file carries float32-rounded weights -- reproduced here, not "fixed".

    variable   Fortran shape              dims in definition order (Fortran)   in the file (C order)
    win        (n, reservoir_numinputs)   win_x, win_y                         [win_y][win_x]  float32
    wout       (n_out, n + n_model)       wout_x, wout_y                       [wout_y][wout_x] float32
    rows cols  (k)                        rows_x / cols_x                      int32
    vals       (k)                        vals_x                               float32
    mean std   (n_stat)                   mean_x / std_x                       float32
    every variable has the attribute units = "unitless".

The classic format is simple enough to be written directly (`_cdf1_bytes`, in the library's own layout: definition order, 4-byte
padding) and is read with `scipy.io.netcdf_file` (pure Python, in the image), which also serves as the independent check of the
writer; nothing here needs the NetCDF or HDF5 libraries.  Host-side only: the arrays go to the device through
`ReservoirBank.load`, exactly as the reference hands the arrays it read to `mklsparse`."""
import os

import numpy as np

REAL_VARS_2D = (("win", "win_x", "win_y"), ("wout", "wout_x", "wout_y"))
INT_VARS_1D = (("rows", "rows_x"), ("cols", "cols_x"))
REAL_VARS_1D = (("vals", "vals_x"), ("mean", "mean_x"), ("std", "std_x"))


def trained_res_filename(region, trial_name, level_index=1, ocean=False):
    """File name as read_trained_res / read_trained_ocean_res build it (src/mod_io.f90:2953-2958, :3000-3003): 0-based region
    number in (i0.4), vertical level index in (i0.1)."""
    if ocean:
        return "worker_%04d_ocean_%s.nc" % (region, trial_name.strip())
    return "worker_%04d_level_%d_%s.nc" % (region, level_index, trial_name.strip())


def _cdf1_bytes(dims, variables):
    """A classic-format (CDF-1) file image, laid out as the netCDF library does for fixed-size variables: header (magic, numrecs = 0,
    dim_list, absent gatt_list, var_list with one `units` attribute per variable), then each variable's big-endian data in definition
    order, every block padded to 4 bytes.  dims: [(name, length)]; variables: [(name, dimids, nc_type, array in file (C) order)]."""
    import struct
    NC_CHAR, NC_INT, NC_FLOAT, NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE = 2, 4, 5, 10, 11, 12

    def pad(b):
        return b + b"\0" * (-len(b) % 4)

    def name(sn):
        e = sn.encode()
        return struct.pack(">I", len(e)) + pad(e)

    units = b"unitless"
    vatt = struct.pack(">II", NC_ATTRIBUTE, 1) + name("units") + struct.pack(">II", NC_CHAR, len(units)) + pad(units)
    head = b"CDF\x01" + struct.pack(">I", 0) + struct.pack(">II", NC_DIMENSION, len(dims))
    for dn, dl in dims:
        head += name(dn) + struct.pack(">I", dl)
    head += struct.pack(">II", 0, 0)                            # no global attributes
    head += struct.pack(">II", NC_VARIABLE, len(variables))
    entries, blobs = [], []
    for vn, dimids, typ, arr in variables:
        data = pad(np.ascontiguousarray(arr, dtype=">i4" if typ == NC_INT else ">f4").tobytes())
        entries.append(name(vn) + struct.pack(">I", len(dimids)) + b"".join(struct.pack(">I", i) for i in dimids) + vatt + struct.pack(">I", typ))
        blobs.append(data)
    begin = len(head) + sum(len(e) + 8 for e in entries)        # + vsize and begin (4 bytes each in CDF-1)
    if begin + sum(len(b) for b in blobs) >= 2 ** 31:
        raise ValueError("too large for the classic NetCDF format the reference writes (32-bit offsets)")
    for e, b in zip(entries, blobs):
        head += e + struct.pack(">II", len(b), begin)
        begin += len(b)
    return head + b"".join(blobs)


def write_trained_res(path, win, wout, rows, cols, vals, mean, std):
    """write_trained_res (src/mod_reservoir.f90:1727-1736): one classic NetCDF file, dimensions and variables in the order the
    reference's helper calls define them, real data as float32.  win (n, d), wout (n_out, n_aug) in the reference's (Fortran) shapes;
    rows/cols 1-based.  An existing file is replaced (the first helper call creates it with NF90_CLOBBER)."""
    NC_INT, NC_FLOAT = 4, 5
    win, wout = np.asarray(win, dtype=np.float64), np.asarray(wout, dtype=np.float64)
    if win.ndim != 2 or wout.ndim != 2:
        raise ValueError("win and wout must be 2-D (reference shapes (n, d) and (n_out, n_aug))")
    rows, cols = np.asarray(rows).ravel(), np.asarray(cols).ravel()
    vals, mean, std = (np.asarray(a, dtype=np.float64).ravel() for a in (vals, mean, std))
    if not (rows.shape == cols.shape == vals.shape) or mean.shape != std.shape:
        raise ValueError("rows/cols/vals (and mean/std) must have equal lengths")
    arrays = {"win": win, "wout": wout, "rows": rows, "cols": cols, "vals": vals, "mean": mean, "std": std}
    dims, variables = [], []
    for vname, xdim, ydim in REAL_VARS_2D:                      # Fortran (xdim, ydim) == C [ydim][xdim]
        a = arrays[vname]
        dims += [(xdim, a.shape[0]), (ydim, a.shape[1])]
        variables.append((vname, [len(dims) - 1, len(dims) - 2], NC_FLOAT, a.T))
    for vname, xdim in INT_VARS_1D:
        dims.append((xdim, arrays[vname].size))
        variables.append((vname, [len(dims) - 1], NC_INT, arrays[vname]))
    for vname, xdim in REAL_VARS_1D:
        dims.append((xdim, arrays[vname].size))
        variables.append((vname, [len(dims) - 1], NC_FLOAT, arrays[vname]))
    image = _cdf1_bytes(dims, variables)
    with open(path, "wb") as f:
        f.write(image)


def read_trained_res(path):
    """read_trained_res / read_trained_ocean_res (src/mod_io.f90:2938-3030): dict of float64 / int32 arrays in the reference's
    shapes -- win (n, d) and wout (n_out, n_aug) Fortran-ordered, rows/cols 1-based.  Raises like nc_check does when the file or a
    variable is missing or has the wrong rank."""
    from scipy.io import netcdf_file
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    f = netcdf_file(path, "r", mmap=False)
    try:
        out = {}
        for name, rank in (("win", 2), ("wout", 2), ("rows", 1), ("cols", 1), ("vals", 1), ("mean", 1), ("std", 1)):
            if name not in f.variables:
                raise KeyError("%s: variable '%s' not found" % (path, name))
            a = np.array(f.variables[name][:])
            if a.ndim != rank:
                raise ValueError("%s: variable '%s' has %d dimensions, expected %d" % (path, name, a.ndim, rank))
            if name in ("rows", "cols"):
                out[name] = a.astype(np.int32)
            else:
                a = a.astype(np.float64)                       # nf90_get_var into real(dp): exact widening of the stored float32
                out[name] = np.asfortranarray(a.T) if rank == 2 else a
        if not (out["rows"].size == out["cols"].size == out["vals"].size):
            raise ValueError("%s: rows/cols/vals lengths differ" % path)
        if out["mean"].size != out["std"].size:
            raise ValueError("%s: mean/std lengths differ" % path)
        return out
    finally:
        f.close()


def load_trained_res(bank, slot, path, n_model, out_stat):
    """trained_reservoir_prediction's start (src/mod_reservoir.f90:1779-1800): read the weights file and hand the arrays to the
    device bank (the reference calls mklsparse next).  n_model = chunk_size_speedy (0 for the ML-only / ocean reservoirs)."""
    w = read_trained_res(path)
    n, d = w["win"].shape
    n_out, n_aug = w["wout"].shape
    if n_aug != n + n_model:
        raise ValueError("%s: wout has %d columns, expected n + n_model = %d" % (path, n_aug, n + n_model))
    bank.load(slot, n, d, n_model, n_out, w["rows"], w["cols"], w["vals"], w["win"], w["wout"], w["mean"], w["std"], out_stat)
    return w


CONTROLLER_KEYS = ("num_hor_regions", "ml_only", "num_vert_levels", "atmo_timestep", "ocean_timestep", "ocean_model_bool",
                   "train_on_sst_anomalies", "precip_bool", "precip_epsilon", "full_predictvars", "full_heightlevels", "num_vert_levels",
                   "vert_loc_overlap", "overlap", "regional_vary", "using_prior", "reservoir_nodes", "deg", "radius", "beta_res",
                   "beta_model", "sigma", "leakage", "prior_val")


def write_controller_file(path, params):
    """write_controller_file (src/mod_reservoir.f90:1739-1776): `<trial>_controller_file.txt`, one list-directed `key: value` line per
    parameter between two dashed lines.  Logicals as T / F like Fortran's list-directed output."""
    dash = " " + "-" * 59
    with open(path, "w") as f:
        f.write(dash + "\n")
        for k in CONTROLLER_KEYS:
            v = params[k]
            if isinstance(v, (bool, np.bool_)):
                v = "T" if v else "F"
            f.write(" %s: %s\n" % (k, v))
        f.write(dash + "\n")


def read_controller_file(path):
    """Parses a controller file (the reference only writes it; its post-processing scripts read it) into a dict: T/F -> bool, then
    int, then float."""
    out = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            k, v = line.split(":", 1)
            k, v = k.strip(), v.strip()
            if v in ("T", "F"):
                out[k] = v == "T"
                continue
            try:
                out[k] = int(v)
            except ValueError:
                try:
                    out[k] = float(v.replace("D", "E").replace("d", "e"))
                except ValueError:
                    out[k] = v
    return out
