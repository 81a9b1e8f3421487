"""Generates tests/golden/dynamics_golden.npz from the COMPILED REFERENCE (oracle/_ref/libref_dyn.so, built in place from
/root/reference/src by oracle/build_ref.sh).  Run in the build container only:

    python tests/golden/make_dynamics_golden.py

The fixture holds data only: one seeded two-time-level spectral state and the reference's outputs for every spectral-space
routine of the SPEEDY step that is runnable without the column physics: the indyns and
impint tables, geop, sptend, implic, hordif and timint.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from _oracle import DYN_TABLES, MX2, NX, S2, S3, RefDyn, RefSpectral  # noqa: E402

DT, ALPH, ROB, WIL = 1800.0, 0.5, 0.05, 0.53


def seeded_state(trfilt, seed=20240954):
    """Two-time-level state with SPEEDY-like magnitudes; triangular truncation applied, Im(m=0) = 0."""
    rng = np.random.default_rng(seed)
    mask = np.repeat(trfilt, 2, axis=0)

    def spec(shape, scale):
        a = rng.standard_normal(shape) * scale
        a *= mask.reshape((MX2, NX) + (1,) * (a.ndim - 2))
        a[1] = 0.0
        return a

    st = dict(vor=spec(S3 + (2,), 1e-5), div=spec(S3 + (2,), 1e-6), t=spec(S3 + (2,), 5.0), tr=spec(S3 + (2,), 1e-3),
              ps=spec(S2 + (2,), 0.01), phis=spec(S2, 1000.0), tcorh=spec(S2, 1.0), qcorh=spec(S2, 1e-3))
    st["t"][0, 0] += 360.0
    st["tend3"] = spec(S3, 1e-9)
    st["tend3b"] = spec(S3, 1e-4)
    st["tend2"] = spec(S2, 1e-7)
    return st


def main():
    ref = RefDyn()
    trfilt = RefSpectral().table(11)
    out = {"trfilt": trfilt, "params": np.array([DT, ALPH, ROB, WIL])}
    for dt in (450.0, 900.0, DT):
        ref.impint(dt, ALPH)
        for w, (name, _) in DYN_TABLES.items():
            out["tab%d_%s" % (int(dt), name)] = ref.table(w)
    st = seeded_state(trfilt)
    ref.set_state(st["vor"], st["div"], st["t"], st["ps"], st["tr"], st["phis"], st["tcorh"], st["qcorh"])
    for jj in (1, 2):
        out["geop%d" % jj] = ref.geop(jj)
    for j4 in (1, 2):
        d, t, p = ref.sptend(st["tend3"], st["tend3b"], st["tend2"], j4)
        out["sptend%d_divdt" % j4], out["sptend%d_tdt" % j4], out["sptend%d_psdt" % j4] = d, t, p
    d, t, p = ref.implic(st["tend3"], st["tend3b"], st["tend2"])
    out["implic_divdt"], out["implic_tdt"], out["implic_psdt"] = d, t, p
    d, t, p = ref.implic(*ref.sptend(st["tend3"], st["tend3b"], st["tend2"], 1))      # as step() chains them (dyn_step.f90:52-56)
    out["chain_divdt"], out["chain_tdt"], out["chain_psdt"] = d, t, p
    for which in (1, 2, 3):
        for nlev in (8, 1):
            out["hordif%d_%d" % (which, nlev)] = ref.hordif(nlev, st["vor"][..., 0], st["tend3"], which)
    for j1, eps in ((1, 0.0), (2, ROB)):
        f, g = ref.timint(j1, DT, eps, WIL, 8, st["t"], st["tend3b"])
        out["timint%d_t" % j1], out["timint%d_tdt" % j1] = f, g
        f, g = ref.timint(j1, DT, eps, WIL, 1, st["ps"].reshape(MX2, NX, 1, 2), st["tend2"].reshape(MX2, NX, 1))
        out["timint%d_ps" % j1] = f
    path = os.path.join(os.path.dirname(__file__), "dynamics_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
