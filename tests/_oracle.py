"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the compiled
reference spectral core (oracle/_ref/libref_spectral.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

IX, IY, IL, NX, MX, MX2, NXP, MXP = 96, 24, 48, 32, 31, 62, 33, 31
REARTH = 6.371e6  # src/mod_dyncon1.f90:13

TABLES = {  # which -> (name, shape in Fortran order)
    1: ("sia", (IY,)), 2: ("coa", (IY,)), 3: ("wt", (IY,)), 4: ("wght", (IY,)),
    5: ("cosg", (IL,)), 6: ("cosgr", (IL,)), 7: ("cosgr2", (IL,)),
    8: ("el2", (MX, NX)), 9: ("elm2", (MX, NX)), 10: ("el4", (MX, NX)), 11: ("trfilt", (MX, NX)),
    12: ("nsh2", (NX,)), 13: ("epsi", (MXP, NXP)), 14: ("repsi", (MXP, NXP)), 15: ("consq", (MXP,)),
    16: ("gradx", (MX,)), 17: ("gradym", (MX, NX)), 18: ("gradyp", (MX, NX)),
    19: ("uvdx", (MX, NX)), 20: ("uvdym", (MX, NX)), 21: ("uvdyp", (MX, NX)),
    22: ("vddym", (MX, NX)), 23: ("vddyp", (MX, NX)), 24: ("cpol", (MX2, NX, IY)), 26: ("sqrhlf", (1,)),
}


def _p(a):
    assert a.dtype == np.float64 and (a.flags.c_contiguous or a.flags.f_contiguous), "oracle call with a strided or non-float64 array"
    return a.ctypes.data_as(_dp)


def _pi(a):
    assert a.dtype == np.int32 and (a.flags.c_contiguous or a.flags.f_contiguous), "oracle call with a strided or non-int32 array"
    return a.ctypes.data_as(_ip)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in
            ("spectral_oracle.c", "domain_oracle.c", "reservoir_oracle.c", "dynamics_oracle.c", "sml_oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class RdGrid(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "res_xstart", "res_xend", "res_ystart", "res_yend", "resxchunk", "resychunk",
        "res_zstart", "res_zend", "reszchunk",
        "input_xstart", "input_xend", "input_ystart", "input_yend", "inputxchunk", "inputychunk",
        "input_zstart", "input_zend", "inputzchunk",
        "pole", "periodicboundary", "top", "bottom",
        "tdata_xstart", "tdata_xend", "tdata_ystart", "tdata_yend", "tdata_zstart", "tdata_zend",
        "overlap", "num_vert_levels", "vert_overlap", "number_of_regions")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RdSizes(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "logp_size_input", "sst_size_input", "precip_size_input", "tisr_size_input",
        "logp_size_res", "precip_size_res",
        "chunk_size", "chunk_size_prediction", "chunk_size_speedy", "locality",
        "nodes_per_input", "n", "k", "reservoir_numinputs",
        "atmo3d_start", "atmo3d_end", "logp_start", "logp_end", "precip_start", "precip_end",
        "sst_start", "sst_end", "tisr_start", "tisr_end")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Oracle:
    """Thin numpy-facing wrapper over liboracle.so."""

    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.so_tables_new.restype = C.c_void_p
        L.so_parmtr.argtypes = [C.c_void_p, C.c_double]
        L.so_get_table.argtypes = [C.c_void_p, C.c_int, _dp]
        for name, nargs in (("so_gridy", 2), ("so_specx", 2), ("so_specy", 2), ("so_spec", 2), ("so_lap", 2),
                            ("so_invlap", 2), ("so_trunct", 1), ("so_rfftf", 1), ("so_rfftb", 1),
                            ("so_grad", 3), ("so_uvspec", 4), ("so_vds", 4)):
            getattr(L, name).argtypes = [C.c_void_p] + [_dp] * nargs
        L.so_gridx.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.so_grid.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.so_vdspec.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.c_int]
        L.rd_get_radius_by_lat.restype = C.c_double
        L.rd_get_radius_by_lat.argtypes = [C.c_double, C.c_double]
        self.t = C.c_void_p(L.so_tables_new())
        L.so_parmtr(self.t, REARTH)

    # ---- spectral ----
    def table(self, which):
        name, shape = TABLES[which]
        out = np.zeros(int(np.prod(shape)))
        self.lib.so_get_table(self.t, which, _p(out))
        return out.reshape(shape, order="F")

    def tables(self):
        return {TABLES[w][0]: self.table(w) for w in TABLES}

    def _call(self, fn, ins, out_shapes, *extra):
        ins = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F")) for a in ins]
        outs = [np.zeros(int(np.prod(s))) for s in out_shapes]
        getattr(self.lib, fn)(self.t, *[_p(a) for a in ins], *[_p(o) for o in outs], *extra)
        outs = [o.reshape(s, order="F") for o, s in zip(outs, out_shapes)]
        return outs[0] if len(outs) == 1 else outs

    def grid(self, vorm, kcos):
        return self._call("so_grid", [vorm], [(IX, IL)], kcos)

    def spec(self, vorg):
        return self._call("so_spec", [vorg], [(MX2, NX)])

    def gridy(self, v):
        return self._call("so_gridy", [v], [(MX2, IL)])

    def gridx(self, varm, kcos):
        return self._call("so_gridx", [varm], [(IX, IL)], kcos)

    def specx(self, vorg):
        return self._call("so_specx", [vorg], [(MX2, IL)])

    def specy(self, varm):
        return self._call("so_specy", [varm], [(MX2, NX)])

    def vdspec(self, ug, vg, kcos):
        return self._call("so_vdspec", [ug, vg], [(MX2, NX), (MX2, NX)], kcos)

    def uvspec(self, vorm, divm):
        return self._call("so_uvspec", [vorm, divm], [(MX2, NX), (MX2, NX)])

    def vds(self, u, v):
        return self._call("so_vds", [u, v], [(MX2, NX), (MX2, NX)])

    def grad(self, psi):
        return self._call("so_grad", [psi], [(MX2, NX), (MX2, NX)])

    def lap(self, s):
        return self._call("so_lap", [s], [(MX2, NX)])

    def invlap(self, v):
        return self._call("so_invlap", [v], [(MX2, NX)])

    def trunct(self, v):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel(order="F")).copy()
        self.lib.so_trunct(self.t, _p(a))
        return a.reshape((MX2, NX), order="F")

    def rfftf(self, r):
        a = np.array(r, dtype=np.float64).copy()
        self.lib.so_rfftf(self.t, _p(a))
        return a

    def rfftb(self, r):
        a = np.array(r, dtype=np.float64).copy()
        self.lib.so_rfftb(self.t, _p(a))
        return a

    # ---- domain ----
    def processor_decomposition(self, proc, numprocs, nregions):
        buf = np.zeros(nregions // numprocs + 2, dtype=np.int32)
        n = self.lib.rd_processor_decomposition(proc, numprocs, nregions, _pi(buf))
        return buf[:n].copy()

    def getxyresextent(self, num_regions, region):
        v = [C.c_int() for _ in range(6)]
        self.lib.rd_getxyresextent(num_regions, region, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)  # xs, xe, ys, ye, xchunk, ychunk

    def initializedomain(self, num_regions, region, overlap=1, num_vert_levels=1, vert_level=1, vert_overlap=0):
        g = RdGrid()
        self.lib.rd_initializedomain(num_regions, region, overlap, num_vert_levels, vert_level, vert_overlap, C.byref(g))
        return g

    def allocate_sizes(self, g, m=6000, deg=6, local_predictvars=4, logp=1, precip=1, sst_input=1, tisr=1, ml_only=0):
        s = RdSizes()
        self.lib.rd_allocate_sizes(C.byref(g), m, deg, local_predictvars, logp, precip, sst_input, tisr, ml_only, C.byref(s))
        return s

    def tile_input(self, num_regions, region, grid4d, grid2d, precip, nout, overlap=1, nvl=1, vl=1, vo=0, precip_bool=1):
        out = np.zeros(nout)
        self.lib.rd_tile_input(num_regions, region, overlap, nvl, vl, vo, precip_bool,
                               _p(grid4d), _p(grid2d), _p(precip), _p(out))
        return out

    def tile_input2d(self, num_regions, region, grid2d, nout, overlap=1):
        out = np.zeros(nout)
        self.lib.rd_tile_input2d(num_regions, region, overlap, _p(grid2d), _p(out))
        return out

    def scatter_res(self, num_regions, region, statevec, grid4d, grid2d, precip, nvl=1, vl=1, precip_bool=1):
        sv = np.ascontiguousarray(statevec, dtype=np.float64)
        self.lib.rd_scatter_res(num_regions, nvl, region, vl, precip_bool, sv.size, _p(sv), _p(grid4d), _p(grid2d), _p(precip))

    def tile_target(self, g, s, statevec, nout, local_predictvars=4, logp=1, precip=1):
        """tile_full_input_to_target_data2d: statevec (numinputs, T) -> (chunk_size_prediction, T)"""
        sv = np.asfortranarray(statevec, dtype=np.float64)
        out = np.zeros((nout, sv.shape[1]), order="F")
        self.lib.rd_tile_target(C.byref(g), C.byref(s), local_predictvars, logp, precip, _p(sv), sv.shape[0], sv.shape[1], _p(out), nout)
        return out

    def tile_res(self, num_regions, region, grid4d, grid2d, nout, nvl=1, vl=1):
        out = np.zeros(nout)
        self.lib.rd_tile_res(num_regions, nvl, region, vl, _p(grid4d), _p(grid2d), _p(out))
        return out

    def standardize_input(self, g, s, mean, std, vec, local_predictvars=4, logp=1):
        v = np.array(vec, dtype=np.float64).copy()
        self.lib.rd_standardize_input(C.byref(g), C.byref(s), local_predictvars, logp, _p(mean), _p(std), _p(v))
        return v

    def standardize_res(self, g, mean, std, vec, local_predictvars=4, heightlevels_input=8, logp=1):
        v = np.array(vec, dtype=np.float64).copy()
        self.lib.rd_standardize_res(C.byref(g), local_predictvars, heightlevels_input, logp, _p(mean), _p(std), _p(v))
        return v

    def unstandardize_res(self, g, mean, std, vec, logp_idx=33, precip_idx=35, local_predictvars=4,
                          heightlevels_input=8, logp=1, precip=1):
        v = np.array(vec, dtype=np.float64).copy()
        self.lib.rd_unstandardize_res(C.byref(g), local_predictvars, heightlevels_input, logp, precip,
                                      logp_idx, precip_idx, _p(mean), _p(std), _p(v))
        return v

    def radius_by_lat(self, a, b):
        return self.lib.rd_get_radius_by_lat(a, b)

    # ---- reservoir ----
    def coo_mv(self, n, rows, cols, vals, x):
        y = np.zeros(n)
        self.lib.ro_coo_mv(n, len(vals), _pi(rows), _pi(cols), _p(vals), _p(x), _p(y))
        return y

    def synchronize(self, n, d, rows, cols, vals, win, leakage, inputs, x):
        """inputs: (d, length) Fortran-ordered; win: (n, d) Fortran-ordered; returns new x."""
        x = np.array(x, dtype=np.float64).copy()
        inp = np.asfortranarray(inputs)
        self.lib.ro_synchronize(n, d, len(vals), _pi(rows), _pi(cols), _p(vals), _p(np.asfortranarray(win)),
                                C.c_double(leakage), _p(inp), inp.shape[1], _p(x))
        return x

    def predict_raw(self, n, d, n_model, n_out, rows, cols, vals, win, wout, leakage, feedback, local_model, x):
        x = np.array(x, dtype=np.float64).copy()
        out = np.zeros(n_out)
        lm = np.zeros(1) if local_model is None else local_model
        self.lib.ro_predict_raw(n, d, len(vals), n_model, n_out, _pi(rows), _pi(cols), _p(vals),
                                _p(np.asfortranarray(win)), _p(np.asfortranarray(wout)), C.c_double(leakage),
                                _p(feedback), _p(lm), _p(x), _p(out))
        return x, out

    def chunking_matmul(self, states, model, y, c, b):
        n, m = states.shape
        self.lib.ro_chunking_matmul(n, model.shape[0], y.shape[0], m, _p(np.asfortranarray(states)),
                                    _p(np.asfortranarray(model)), _p(np.asfortranarray(y)), _p(c), _p(b))

    def fit_chunk_hybrid(self, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, c, b):
        wout = np.zeros((n_out, n + n_model), order="F")
        self.lib.ro_fit_chunk_hybrid.restype = C.c_int
        info = self.lib.ro_fit_chunk_hybrid(n, n_model, n_out, C.c_double(beta_res), C.c_double(beta_model),
                                            C.c_double(prior_val), int(using_prior), _p(c), _p(b), _p(wout))
        return info, wout

    def train_states(self, n, d, rows, cols, vals, win, leakage, noisy_inputs, discard, batch, model, targets, c, b, ml_variant=False):
        T = noisy_inputs.shape[1]
        return self.lib.ro_train_states(n, d, len(vals), _pi(rows), _pi(cols), _p(vals), _p(np.asfortranarray(win)),
                                        C.c_double(leakage), _p(np.asfortranarray(noisy_inputs)), T, discard, batch,
                                        model.shape[0], targets.shape[0], _p(np.asfortranarray(model)),
                                        _p(np.asfortranarray(targets)), _p(c), _p(b), 1 if ml_variant else 0)

    def slab_sst(self, base_sst, mask_gt0, sea_of_region, res_cell, all_slab_out):
        nreg = len(sea_of_region)
        out = np.zeros(4608)
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (mask_gt0, sea_of_region, res_cell)]
        so = np.ascontiguousarray(all_slab_out, dtype=np.float64)
        self.lib.rd_slab_sst(nreg, _p(np.ascontiguousarray(base_sst, dtype=np.float64)), _pi(a[0]), _pi(a[1]), _pi(a[2]), _p(so),
                             so.shape[1], _p(out))
        return out

    def slab_ring_update(self, timestep, idx, feedback_atmo, ring, feedback_slab):
        """ring: (R, nidx) C-contiguous float64, updated in place; feedback_slab[:nidx] overwritten in place"""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        fa = np.ascontiguousarray(feedback_atmo, dtype=np.float64)
        self.lib.rd_slab_ring_update(int(timestep), ring.shape[0], len(idx), _pi(idx), _p(fa), _p(ring), _p(feedback_slab))

    def calendar_date(self, hours_elapsed, startyear=1981):
        d = (C.c_int * 4)()
        self.lib.rd_calendar_date(int(startyear), int(hours_elapsed), d)
        return tuple(d)

    def tisr_index(self, hours_elapsed, startyear=1981):
        return self.lib.rd_tisr_index(int(startyear), int(hours_elapsed))

    def find_closest_divisor(self, approx, number):
        return self.lib.ro_find_closest_divisor(approx, number)


class RefSpectral:
    """The compiled reference spectral core (oracle/_ref/libref_spectral.so).  Optional."""

    PATH = os.path.join(ORACLE_DIR, "_ref", "libref_spectral.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)
        self.lib.ref_init.argtypes = [C.c_double]
        self.lib.ref_get_table.argtypes = [C.c_int, _dp, C.c_int]
        self.lib.ref_init(REARTH)

    def table(self, which):
        name, shape = TABLES[which]
        n = int(np.prod(shape))
        out = np.zeros(n)
        self.lib.ref_get_table(which, _p(out), n)
        return out.reshape(shape, order="F")

    def tables(self):
        return {TABLES[w][0]: self.table(w) for w in TABLES}

    def _call(self, fn, ins, out_shapes, *extra):
        ins = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F")).copy() for a in ins]
        outs = [np.zeros(int(np.prod(s))) for s in out_shapes]
        getattr(self.lib, fn)(*[_p(a) for a in ins], *[_p(o) for o in outs], *[C.c_int(e) for e in extra])
        outs = [o.reshape(s, order="F") for o, s in zip(outs, out_shapes)]
        return outs[0] if len(outs) == 1 else outs

    def grid(self, vorm, kcos):
        return self._call("ref_grid", [vorm], [(IX, IL)], kcos)

    def spec(self, vorg):
        return self._call("ref_spec", [vorg], [(MX2, NX)])

    def gridy(self, v):
        return self._call("ref_gridy", [v], [(MX2, IL)])

    def gridx(self, varm, kcos):
        return self._call("ref_gridx", [varm], [(IX, IL)], kcos)

    def specx(self, vorg):
        return self._call("ref_specx", [vorg], [(MX2, IL)])

    def specy(self, varm):
        return self._call("ref_specy", [varm], [(MX2, NX)])

    def vdspec(self, ug, vg, kcos):
        return self._call("ref_vdspec", [ug, vg], [(MX2, NX), (MX2, NX)], kcos)

    def uvspec(self, vorm, divm):
        return self._call("ref_uvspec", [vorm, divm], [(MX2, NX), (MX2, NX)])

    def vds(self, u, v):
        return self._call("ref_vds", [u, v], [(MX2, NX), (MX2, NX)])

    def grad(self, psi):
        return self._call("ref_grad", [psi], [(MX2, NX), (MX2, NX)])

    def lap(self, s):
        return self._call("ref_lap", [s], [(MX2, NX)])

    def invlap(self, v):
        return self._call("ref_invlap", [v], [(MX2, NX)])

    def trunct(self, v):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel(order="F")).copy()
        self.lib.ref_trunct(_p(a))
        return a.reshape((MX2, NX), order="F")

    def rfftf(self, r):
        a = np.array(r, dtype=np.float64).copy()
        self.lib.ref_rfftf(_p(a))
        return a

    def rfftb(self, r):
        a = np.array(r, dtype=np.float64).copy()
        self.lib.ref_rfftb(_p(a))
        return a


# ---------------------------------------------------------------- SPEEDY adiabatic core (dynamics_oracle.c / libref_dyn.so)
KX, KXP, LMAX = 8, 9, 61
DYN_TABLES = {  # which -> (name, Fortran shape); numbering of refd_get / do_get_table
    1: ("hsg", (KXP,)), 2: ("dhs", (KX,)), 3: ("fsg", (KX,)), 4: ("dhsr", (KX,)), 5: ("fsgr", (KX,)), 6: ("coriol", (IL,)),
    7: ("xgeop1", (KX,)), 8: ("xgeop2", (KX,)), 9: ("dmp", (MX, NX)), 10: ("dmpd", (MX, NX)), 11: ("dmps", (MX, NX)),
    12: ("dmp1", (MX, NX)), 13: ("dmp1d", (MX, NX)), 14: ("dmp1s", (MX, NX)), 15: ("tcorv", (KX,)), 16: ("qcorv", (KX,)),
    17: ("tref", (KX,)), 18: ("tref1", (KX,)), 19: ("tref2", (KX,)), 20: ("tref3", (KX,)), 21: ("xc", (KX, KX)),
    22: ("xd", (KX, KX)), 23: ("xj", (KX, KX, LMAX)), 24: ("dhsx", (KX,)), 25: ("elz", (MX, NX)), 26: ("alph", (1,)),
}
S3 = (MX2, NX, KX)          # one time level of a 3-D spectral array, interleaved re/im (Fortran order)
S2 = (MX2, NX)


def _flat(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F")).copy()


class DynOracle:
    """dynamics_oracle.c over an Oracle's spectral tables.  Arrays are numpy, Fortran-ordered, interleaved complex:
    3-D (62,32,8[,2]), 2-D (62,32[,2])."""

    def __init__(self, oracle=None):
        self.o = oracle or Oracle()
        L = self.lib = self.o.lib
        L.do_tables_new.restype = C.c_void_p
        L.do_indyns.argtypes = [C.c_void_p, C.c_void_p]
        L.do_impint.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.do_get_table.argtypes = [C.c_void_p, C.c_int, _dp]
        L.do_geop.argtypes = [C.c_void_p] + [_dp] * 3
        L.do_sptend.argtypes = [C.c_void_p, C.c_void_p] + [_dp] * 8
        L.do_implic.argtypes = [C.c_void_p] + [_dp] * 3
        L.do_hordif.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int]
        L.do_timint.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.do_grtend_dry.argtypes = [C.c_void_p, C.c_void_p] + [_dp] * 10
        L.do_step_dry.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 4 + [_dp] * 8
        L.do_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 4 + [_dp] * 8 + [C.c_void_p, C.c_void_p]
        self.d = C.c_void_p(L.do_tables_new())
        L.do_indyns(self.d, self.o.t)

    def impint(self, dt, alph):
        self.lib.do_impint(self.d, dt, alph)

    def table(self, which):
        name, shape = DYN_TABLES[which]
        out = np.zeros(int(np.prod(shape)))
        self.lib.do_get_table(self.d, which, _p(out))
        return out.reshape(shape, order="F")

    def tables(self):
        return {DYN_TABLES[w][0]: self.table(w) for w in DYN_TABLES}

    def geop(self, t, phis):
        t, phis, phi = _flat(t), _flat(phis), np.zeros(int(np.prod(S3)))
        self.lib.do_geop(self.d, _p(t), _p(phis), _p(phi))
        return phi.reshape(S3, order="F")

    def sptend(self, div, t, ps, phis, divdt, tdt, psdt):
        a = [_flat(x) for x in (div, t, ps, phis, divdt, tdt, psdt)]
        phi = np.zeros(int(np.prod(S3)))
        self.lib.do_sptend(self.d, self.o.t, *[_p(x) for x in a], _p(phi))
        return a[4].reshape(S3, order="F"), a[5].reshape(S3, order="F"), a[6].reshape(S2, order="F"), phi.reshape(S3, order="F")

    def implic(self, divdt, tdt, psdt):
        a = [_flat(x) for x in (divdt, tdt, psdt)]
        self.lib.do_implic(self.d, *[_p(x) for x in a])
        return a[0].reshape(S3, order="F"), a[1].reshape(S3, order="F"), a[2].reshape(S2, order="F")

    def hordif(self, nlev, field, fdt, which):
        f, g = _flat(field), _flat(fdt)
        self.lib.do_hordif(self.d, nlev, _p(f), _p(g), which)
        return g.reshape(np.shape(fdt), order="F")

    def timint(self, j1, dt, eps, wil, nlev, field, fdt):
        f, g = _flat(field), _flat(fdt)
        self.lib.do_timint(self.o.t, j1, dt, eps, wil, nlev, _p(f), _p(g))
        return f.reshape(np.shape(field), order="F"), g.reshape(np.shape(fdt), order="F")

    def grtend_dry(self, vor, div, t, tr, ps):
        a = [_flat(x) for x in (vor, div, t, tr, ps)]
        outs = [np.zeros(int(np.prod(S3))), np.zeros(int(np.prod(S3))), np.zeros(int(np.prod(S3))), np.zeros(int(np.prod(S2))),
                np.zeros(int(np.prod(S3)))]
        self.lib.do_grtend_dry(self.d, self.o.t, *[_p(x) for x in a], *[_p(x) for x in outs])
        shp = [S3, S3, S3, S2, S3]
        return [o.reshape(s, order="F") for o, s in zip(outs, shp)]      # vordt, divdt, tdt, psdt, trdt

    def step_dry(self, j1, j2, dt, alph, rob, wil, state, phis, tcorh, qcorh):
        """state: dict vor/div/t/tr (62,32,8,2), ps (62,32,2); returns a new dict."""
        keys = ("vor", "div", "t", "tr", "ps")
        a = [_flat(state[k]) for k in keys]
        c = [_flat(x) for x in (phis, tcorh, qcorh)]
        self.lib.do_step_dry(self.d, self.o.t, j1, j2, dt, alph, rob, wil, *[_p(x) for x in a], *[_p(x) for x in c])
        return {k: x.reshape(np.shape(state[k]), order="F") for k, x in zip(keys, a)}


    PHYS_FN = C.CFUNCTYPE(None, C.c_void_p, *([C.POINTER(C.c_double)] * 10))

    def step(self, j1, j2, dt, alph, rob, wil, state, phis, tcorh, qcorh, phys=None):
        """step() with the physics hook of grtend (src/dyn_grtend.f90:222-225).  phys(ug1, vg1, tg1, qg1, phig1, pslg1, utend,
        vtend, ttend, qtend) gets (ngp, nlev) Fortran-ordered arrays ((ngp,) for pslg1) of time level 1 and returns the four
        updated tendencies; None = adiabatic."""
        if phys is None:
            return self.step_dry(j1, j2, dt, alph, rob, wil, state, phis, tcorh, qcorh)
        ngp = 96 * 48

        def hook(ctx, *ptrs):
            arr = [np.ctypeslib.as_array(q, shape=(ngp if i == 5 else 8 * ngp,)) for i, q in enumerate(ptrs)]
            ins = [a.copy() if i == 5 else a.reshape((ngp, 8), order="F").copy(order="F") for i, a in enumerate(arr[:6])]
            tends = [a.reshape((ngp, 8), order="F") for a in arr[6:]]
            new = phys(*ins, *[t.copy(order="F") for t in tends])
            for t, n in zip(tends, new):
                t[...] = n

        keys = ("vor", "div", "t", "tr", "ps")
        a = [_flat(state[k]) for k in keys]
        c = [_flat(x) for x in (phis, tcorh, qcorh)]
        cb = self.PHYS_FN(hook)
        self.lib.do_step(self.d, self.o.t, j1, j2, dt, alph, rob, wil, *[_p(x) for x in a], *[_p(x) for x in c], C.cast(cb, C.c_void_p), None)
        return {k: x.reshape(np.shape(state[k]), order="F") for k, x in zip(keys, a)}


class RefDyn:
    """The compiled reference dynamics subset (oracle/_ref/libref_dyn.so: ini_indyns, ini_impint, spe_matinv, dyn_geop,
    dyn_sptend, dyn_implic, dyn_step compiled from /root/reference/src in place).  Optional."""

    PATH = os.path.join(ORACLE_DIR, "_ref", "libref_dyn.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        L = self.lib = C.CDLL(self.PATH)
        L.refd_impint.argtypes = [C.c_double, C.c_double]
        L.refd_get.argtypes = [C.c_int, _dp, C.c_int]
        L.refd_set_state.argtypes = [_dp] * 8
        L.refd_get_state.argtypes = [_dp] * 6
        L.refd_geop.argtypes = [C.c_int]
        L.refd_sptend.argtypes = [_dp, _dp, _dp, C.c_int]
        L.refd_implic.argtypes = [_dp, _dp, _dp]
        L.refd_hordif.argtypes = [C.c_int, _dp, _dp, C.c_int]
        L.refd_timint.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.refd_init()

    def impint(self, dt, alph):
        self.lib.refd_impint(dt, alph)

    def table(self, which):
        name, shape = DYN_TABLES[which]
        n = int(np.prod(shape))
        out = np.zeros(n)
        self.lib.refd_get(which, _p(out), n)
        return out.reshape(shape, order="F")

    def tables(self):
        return {DYN_TABLES[w][0]: self.table(w) for w in DYN_TABLES}

    def set_state(self, vor, div, t, ps, tr, phis, tcorh=None, qcorh=None):
        z = np.zeros(S2)
        a = [_flat(x) for x in (vor, div, t, ps, tr, phis, z if tcorh is None else tcorh, z if qcorh is None else qcorh)]
        self.lib.refd_set_state(*[_p(x) for x in a])

    def get_state(self):
        shp = [S3 + (2,), S3 + (2,), S3 + (2,), S2 + (2,), S3 + (2,), S3]
        outs = [np.zeros(int(np.prod(s))) for s in shp]
        self.lib.refd_get_state(*[_p(x) for x in outs])
        return dict(zip(("vor", "div", "t", "ps", "tr", "phi"), [o.reshape(s, order="F") for o, s in zip(outs, shp)]))

    def geop(self, jj):
        self.lib.refd_geop(jj)
        return self.get_state()["phi"]

    def sptend(self, divdt, tdt, psdt, j4):
        a = [_flat(x) for x in (divdt, tdt, psdt)]
        self.lib.refd_sptend(*[_p(x) for x in a], j4)
        return a[0].reshape(S3, order="F"), a[1].reshape(S3, order="F"), a[2].reshape(S2, order="F")

    def implic(self, divdt, tdt, psdt):
        a = [_flat(x) for x in (divdt, tdt, psdt)]
        self.lib.refd_implic(*[_p(x) for x in a])
        return a[0].reshape(S3, order="F"), a[1].reshape(S3, order="F"), a[2].reshape(S2, order="F")

    def hordif(self, nlev, field, fdt, which):
        f, g = _flat(field), _flat(fdt)
        self.lib.refd_hordif(nlev, _p(f), _p(g), which)
        return g.reshape(np.shape(fdt), order="F")

    def timint(self, j1, dt, eps, wil, nlev, field, fdt):
        f, g = _flat(field), _flat(fdt)
        self.lib.refd_timint(j1, dt, eps, wil, nlev, _p(f), _p(g))
        return f.reshape(np.shape(field), order="F"), g.reshape(np.shape(fdt), order="F")


# ---------------------------------------------------------------- the SPEEDY leg of one hybrid step, with the oracle
def oracle_iogrid30(o, g4, logp):
    """iogrid(30) (src/ppo_iogrid.f90:497-547): g4[z,y,x,v] (T,u,v,q), logp[y,x] -> spectral time level 1 as a dict of
    oracle-layout arrays vor/div/t/tr (62,32,8), ps (62,32).  real(4) rounding and the q<0 clamp as in the reference."""
    r4 = lambda a: a.astype(np.float32)
    T, u, v, q = (r4(g4[..., i]) for i in range(4))
    q = np.where(q < 0, np.float32(0), q)
    out = {k: np.zeros(S3) for k in ("vor", "div", "t", "tr")}
    for k in range(KX):
        f = lambda a: np.asarray(a[k], dtype=np.float64).T        # [48][96] -> Fortran (ix,il)
        vor, div = o.vdspec(f(u), f(v), 2)
        out["vor"][..., k], out["div"][..., k] = o.trunct(vor), o.trunct(div)
        out["t"][..., k], out["tr"][..., k] = o.trunct(o.spec(f(T))), o.trunct(o.spec(f(q)))
    out["ps"] = o.trunct(o.spec(np.asarray(r4(logp), dtype=np.float64).T))
    return out


def oracle_iogrid31(o, lvl):
    """iogrid(31) (src/ppo_iogrid.f90:579-601): spectral level-1 dict -> (F4[z,y,x,v], F2[y,x]); q floor as regrid applies it."""
    F4 = np.zeros((KX, IL, IX, 4))
    for k in range(KX):
        uc, vc = o.uvspec(lvl["vor"][..., k], lvl["div"][..., k])
        F4[k, :, :, 1] = o.grid(uc, 2).T
        F4[k, :, :, 2] = o.grid(vc, 2).T
        F4[k, :, :, 0] = o.grid(lvl["t"][..., k], 1).T
        F4[k, :, :, 3] = o.grid(lvl["tr"][..., k], 1).T
    F2 = o.grid(lvl["ps"], 1).T
    return F4, F2


def oracle_window(dyn, lvl1, phis, tcorh, qcorh, nsteps, delt=900.0, alph=0.5, rob=0.05, wil=0.53, phys=None, nstrad=3):
    """stepone (istart = 2) + nsteps leapfrog steps (src/ini_stepone.f90, src/dyn_stloop.f90:28-43) from a level-1 state
    (level 2 is whatever is passed: stepone's forward step overwrites it).  Returns the two-level state dict.
    phys: a PhysHook for grtend's physics slot (its .lradsw is the module flag: kept through stepone, then
    mod(istep, nstrad) == 1, src/dyn_stloop.f90:39); None = adiabatic."""
    two = lambda a: np.stack([a, a], axis=-1)
    cur = {k: two(lvl1[k]) for k in ("vor", "div", "t", "tr", "ps")}
    bc = (phis, tcorh, qcorh)
    dyn.impint(0.5 * delt, alph)
    cur = dyn.step(1, 1, 0.5 * delt, alph, rob, wil, cur, *bc, phys=phys)
    dyn.impint(delt, alph)
    cur = dyn.step(1, 2, delt, alph, rob, wil, cur, *bc, phys=phys)
    dyn.impint(2 * delt, alph)
    for i in range(nsteps):
        if phys is not None:
            phys.lradsw = (i + 1) % nstrad == 1
        cur = dyn.step(2, 2, 2 * delt, alph, rob, wil, cur, *bc, phys=phys)
    return cur


# ---------------------------------------------------------------- SPEEDY column physics: the compiled reference (libref_phy.so)
NGP = IX * IL


class RefPhys:
    """The reference's own parametrisation routines (oracle/_ref/libref_phy.so: phy_convmf, phy_lscond, phy_shtorh, phy_radiat,
    phy_suflux, phy_vdifsc, ini_inphys compiled in place) and, in `phypar`, the call sequence of the grid-point part of
    src/phy_phypar.f90:80-230 issued through them.  Arrays are (ngp, nlev) Fortran-ordered like the reference's."""

    PATH = os.path.join(ORACLE_DIR, "_ref", "libref_phy.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self, hsg, rlat):
        L = self.lib = C.CDLL(self.PATH)
        L.refp_sol_oz.argtypes = [C.c_double]
        L.refp_shtorh.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp]
        L.refp_radlw.argtypes = [C.c_int] + [_dp] * 7
        L.refp_init(_p(np.ascontiguousarray(hsg, dtype=np.float64)), _p(np.ascontiguousarray(rlat, dtype=np.float64)))
        self.tt_rsw = np.zeros((NGP, KX), order="F")
        self.ssrd = np.zeros(NGP)
        self.fields()

    def fields(self):
        names = ("fsol", "ozone", "ozupp", "zenit", "stratz", "forog")
        arrs = [np.zeros(NGP) for _ in names]
        fband = np.zeros((301, 4), order="F")
        sig, dsig, sigh, grdsig, grdscp = np.zeros(KX), np.zeros(KX), np.zeros(KX + 1), np.zeros(KX), np.zeros(KX)
        wvi = np.zeros((KX, 2), order="F")
        self.lib.refp_get_fields(*[_p(a) for a in arrs], _p(fband), _p(sig), _p(dsig), _p(sigh), _p(grdsig), _p(grdscp), _p(wvi))
        self.sig, self.dsig, self.sigh, self.grdsig, self.grdscp, self.wvi, self.fband = sig, dsig, sigh, grdsig, grdscp, wvi, fband
        return dict(zip(names, arrs))

    def set_surface(self, phis0, alb_l, alb_s, albsfc, snowc):
        self.lib.refp_set_surface(*[_p(np.ascontiguousarray(a, dtype=np.float64).ravel()) for a in (phis0, alb_l, alb_s, albsfc, snowc)])

    def sol_oz(self, tyear):
        self.lib.refp_sol_oz(tyear)

    def fordate(self, tyear, phis0, fmask_l, fmask_s, stl_am, sst_am, alb0, snowd_am, sice_am):
        """The reference's own fordate(0) (src/ini_fordate.f90, compiled in place) on the given module-variable values, each (48,96)
        or (4608,).  Returns dict(tcorh, qcorh: oracle-layout (62,32); snowc, alb_l, alb_s, albsfc: (4608,)).  Leaves the physics'
        albedo / sol_oz / sflset module state as fordate sets it."""
        a = [np.ascontiguousarray(x, dtype=np.float64).ravel().copy() for x in (phis0, fmask_l, fmask_s, stl_am, sst_am, alb0, snowd_am, sice_am)]
        tc, qc = np.zeros(2 * 31 * 32), np.zeros(2 * 31 * 32)
        alb = [np.zeros(NGP) for _ in range(4)]
        self.lib.refp_fordate.argtypes = [C.c_double] + [_dp] * 14
        self.lib.refp_fordate(float(tyear), *[_p(x) for x in a], _p(tc), _p(qc), *[_p(x) for x in alb])
        return dict(tcorh=tc.reshape((62, 32), order="F"), qcorh=qc.reshape((62, 32), order="F"), snowc=alb[0], alb_l=alb[1], alb_s=alb[2],
                    albsfc=alb[3])

    def radstate(self):
        tau2, stratc, qcloud = np.zeros((NGP, KX, 4), order="F"), np.zeros((NGP, 2), order="F"), np.zeros(NGP)
        self.lib.refp_get_radstate(_p(tau2), _p(stratc), _p(qcloud))
        return tau2, stratc, qcloud

    def phypar(self, ug, vg, tg, qg, phig, pslg, fmask, phis0, tland, tsea, swav, lradsw, utend, vtend, ttend, qtend):
        """Returns (utend, vtend, ttend, qtend, diag) after phypar's sections 1.2-4.2; inputs (ngp, nlev) / (ngp,)"""
        L = self.lib
        F = lambda a: np.asfortranarray(a, dtype=np.float64).copy(order="F")
        ug, vg, tg, qg, phig = F(ug), F(vg), F(tg), F(qg), F(phig)
        fmask, phis0, tland, tsea, swav = (np.ascontiguousarray(a, dtype=np.float64).ravel().copy() for a in (fmask, phis0, tland, tsea, swav))
        z2 = lambda: np.zeros((NGP, KX), order="F")
        psg = np.exp(np.asarray(pslg, dtype=np.float64).ravel())
        rps = 1.0 / psg
        qg = np.maximum(qg, 0.0)
        se = F(1004.0 * tg + phig)
        rh, qsat = z2(), z2()
        for k in range(KX):
            q, r, s = qg[:, k].copy(), np.zeros(NGP), np.zeros(NGP)
            L.refp_shtorh(1, _p(tg[:, k].copy()), _p(psg), float(self.sig[k]), _p(q), _p(r), _p(s))
            rh[:, k], qsat[:, k] = r, s
        iptop = np.zeros(NGP, dtype=np.int32)
        cbmf, precnv, tt_cnv, qt_cnv = np.zeros(NGP), np.zeros(NGP), z2(), z2()
        L.refp_convmf(_p(psg), _p(se), _p(qg), _p(qsat), _pi(iptop), _p(cbmf), _p(precnv), _p(tt_cnv), _p(qt_cnv))
        for k in range(1, KX):
            tt_cnv[:, k] = tt_cnv[:, k] * rps * self.grdscp[k]
            qt_cnv[:, k] = qt_cnv[:, k] * rps * self.grdsig[k]
        icnv = (KX - iptop).astype(np.int32)
        precls, tt_lsc, qt_lsc = np.zeros(NGP), z2(), z2()
        L.refp_lscond(_p(psg), _p(qg), _p(qsat), _pi(iptop), _p(precls), _p(tt_lsc), _p(qt_lsc))
        ttend = F(ttend) + tt_cnv + tt_lsc
        qtend = F(qtend) + qt_cnv + qt_lsc
        diag = {}
        if lradsw:
            gse = (se[:, KX - 2] - se[:, KX - 1]) / (phig[:, KX - 2] - phig[:, KX - 1])
            icltop = np.zeros(NGP, dtype=np.int32)
            cloudc, clstr = np.zeros(NGP), np.zeros(NGP)
            L.refp_cloud(_p(qg), _p(rh), _p(precnv), _p(precls), _pi(iptop), _p(gse), _p(fmask), _pi(icltop), _p(cloudc), _p(clstr))
            ssrd, ssr, tsr, tt_rsw = np.zeros(NGP), np.zeros(NGP), np.zeros(NGP), z2()
            L.refp_radsw(_p(psg), _p(qg), _pi(icltop), _p(cloudc), _p(clstr), _p(ssrd), _p(ssr), _p(tsr), _p(tt_rsw))
            for k in range(KX):
                tt_rsw[:, k] = tt_rsw[:, k] * rps * self.grdscp[k]
            self.tt_rsw, self.ssrd = tt_rsw, ssrd
            diag.update(cloudc=cloudc, clstr=clstr, tsr=tsr, ssr=ssr, icltop=icltop.astype(float))
        ts = np.zeros(NGP)
        slrd, slru3, slr, olr, tt_rlw = np.zeros(NGP), np.zeros(NGP), np.zeros(NGP), np.zeros(NGP), z2()
        L.refp_radlw(-1, _p(tg), _p(ts), _p(slrd), _p(slru3), _p(slr), _p(olr), _p(tt_rlw))
        z3 = lambda: np.zeros((NGP, 3), order="F")
        ustr, vstr, shf, evap, slru = z3(), z3(), z3(), z3(), z3()
        hfluxn = np.zeros((NGP, 2), order="F")
        tskin, u0, v0, t0, q0 = (np.zeros(NGP) for _ in range(5))
        psg_io = psg.copy()
        L.refp_suflux(_p(psg_io), _p(ug), _p(vg), _p(tg), _p(qg), _p(rh), _p(phig), _p(phis0), _p(fmask), _p(tland), _p(tsea), _p(swav),
                      _p(self.ssrd), _p(slrd), _p(ustr), _p(vstr), _p(shf), _p(evap), _p(slru), _p(hfluxn), _p(ts), _p(tskin), _p(u0), _p(v0),
                      _p(t0), _p(q0), 1)
        slru3 = slru[:, 2].copy()
        L.refp_radlw(1, _p(tg), _p(ts), _p(slrd), _p(slru3), _p(slr), _p(olr), _p(tt_rlw))
        for k in range(KX):
            tt_rlw[:, k] = tt_rlw[:, k] * rps * self.grdscp[k]
        ttend = ttend + self.tt_rsw + tt_rlw
        ut, vt, tt, qt = z2(), z2(), z2(), z2()
        L.refp_vdifsc(_p(ug), _p(vg), _p(se), _p(rh), _p(qg), _p(qsat), _p(phig), _pi(icnv), _p(ut), _p(vt), _p(tt), _p(qt))
        n = KX - 1
        ut[:, n] = ut[:, n] + ustr[:, 2] * rps * self.grdsig[n]
        vt[:, n] = vt[:, n] + vstr[:, 2] * rps * self.grdsig[n]
        tt[:, n] = tt[:, n] + shf[:, 2] * rps * self.grdscp[n]
        qt[:, n] = qt[:, n] + evap[:, 2] * rps * self.grdsig[n]
        utend, vtend = F(utend) + ut, F(vtend) + vt
        ttend, qtend = ttend + tt, qtend + qt
        diag.update(precnv=precnv, precls=precls, cbmf=cbmf, ts=ts, tskin=tskin, ssrd=self.ssrd.copy(), slrd=slrd, olr=olr, shf=shf[:, 2].copy(),
                    evap=evap[:, 2].copy(), ustr=ustr[:, 2].copy(), vstr=vstr[:, 2].copy(), slr=slr, hfluxn_land=hfluxn[:, 0].copy(),
                    hfluxn_sea=hfluxn[:, 1].copy(), t0=t0, q0=q0, iptop=iptop.astype(float))
        return utend, vtend, ttend, qtend, diag


class PhysHook:
    """grtend's physics slot (src/dyn_grtend.f90:222-225) filled with the compiled reference parametrisations: callable as
    DynOracle.step's `phys`.  surf: dict of (4608,) / (48,96) arrays fmask, phis0, tland, tsea, swav, alb_l, alb_s, albsfc, snowc."""

    def __init__(self, ref, surf, tyear, lradsw=True):
        self.ref, self.lradsw = ref, lradsw
        self.s = {k: np.asarray(v, dtype=np.float64).ravel() for k, v in surf.items()}
        ref.set_surface(self.s["phis0"], self.s["alb_l"], self.s["alb_s"], self.s["albsfc"], self.s["snowc"])
        ref.sol_oz(tyear)

    def __call__(self, ug, vg, tg, qg, phig, pslg, ut, vt, tt, qt):
        s = self.s
        return self.ref.phypar(ug, vg, tg, qg, phig, pslg, s["fmask"], s["phis0"], s["tland"], s["tsea"], s["swav"], self.lradsw,
                               ut, vt, tt, qt)[:4]


def oracle_rest_state(o, phi0_grid, hsg):
    """invars with istart = 0 (src/ini_invars.f90:27-111): the reference atmosphere at rest over the surface geopotential
    phi0_grid[48][96] (m2/s2) -- zero vorticity / divergence, 288 K at z = 0 with a 6 K/km lapse rate under a 216 K stratosphere,
    log(ps) in hydrostatic balance with it, tropospheric humidity from the reference relative humidity.  Returns (level-1 dict of
    oracle-layout arrays vor/div/t/tr (62,32,8), ps (62,32); phis (62,32)).  Complex (1.,0.)*sqrt(2) = the (re, im) pair of
    coefficient (1,1)."""
    gamma, hscale, hshum, refrh1, grav = 6.0, 7.5, 2.5, 0.7, 9.81
    rgas = 2.0 / 7.0 * 1004.0
    fsg = 0.5 * (np.asarray(hsg)[1:] + np.asarray(hsg)[:-1])
    gam1 = gamma / (1000.0 * grav)
    ccon = np.sqrt(2.0)
    phis = o.trunct(o.spec(np.asarray(phi0_grid, dtype=np.float64).T))
    phis0 = o.grid(phis, 1)                                   # Fortran (ix, il)
    out = {k: np.zeros(S3) for k in ("vor", "div", "t", "tr")}
    tref, ttop = 288.0, 216.0
    gam2, rgam = gam1 / tref, rgas * gam1
    rgamr = 1.0 / rgam
    surfs = -gam1 * phis
    surfs[0, 0] = ccon * tref - gam1 * phis[0, 0]
    out["t"][0, 0, 0] = out["t"][0, 0, 1] = ccon * ttop
    for k in range(2, KX):
        out["t"][..., k] = surfs * fsg[k] ** rgam
    rlog0 = np.log(1.013)
    surfg = rlog0 + rgamr * np.log(1.0 - gam2 * phis0)
    out["ps"] = o.trunct(o.spec(surfg))
    qref, qexp = refrh1 * 0.622 * 17.0, hscale / hshum
    qs = o.trunct(o.spec(qref * np.exp(qexp * surfg)))
    for k in range(2, KX):
        out["tr"][..., k] = qs * fsg[k] ** qexp
    return out, phis
