"""Generates tests/golden/fordate_golden.npz from the COMPILED REFERENCE (oracle/_ref/libref_phy.so, which holds the reference's own
src/ini_fordate.f90 compiled in place by oracle/build_ref.sh).  Run in the build container only:

    python tests/golden/make_fordate_golden.py

The fixture holds data only: the seeded inputs fordate(0) reads from its modules (fordate_inputs()) and what it leaves behind --
tcorh, qcorh (spectral, (62,32) oracle layout), snowc, alb_l, alb_s, albsfc.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

from make_physics_golden import HSG, TYEAR, gaussian_latitudes          # noqa: E402


def fordate_inputs(seed=11):
    """(48,96) fields with coasts (fractional masks thresholded as src/ini_inbcon.f90:55-65,148-157 does), mountains, snow at high
    latitudes and sea ice near the poles; stl_am / sst_am around the synthetic climate's surface temperatures."""
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import synth
    rng = np.random.default_rng(seed)
    g4, _, _, sst = synth.synthetic_state(seed)
    lat = np.repeat(np.linspace(-87.159, 87.159, 48)[:, None], 96, axis=1)
    sea = synth.land_mask().astype(float)
    fmask = np.clip(1.0 - sea + 0.6 * (rng.random((48, 96)) - 0.5) * (rng.random((48, 96)) < 0.35), 0.0, 1.0)
    thr = 0.1
    fmask_l = np.where(fmask >= thr, np.where(fmask > 1.0 - thr, 1.0, fmask), 0.0)
    fs = 1.0 - fmask
    fmask_s = np.where(fs >= thr, np.where(fs > 1.0 - thr, 1.0, fs), 0.0)
    phis0 = synth.synthetic_orography() + 20.0 * rng.standard_normal((48, 96))        # (negative values occur in the truncated orography too)
    stl_am = g4[7, :, :, 0] + 4.0 * rng.standard_normal((48, 96))
    sst_am = sst + rng.standard_normal((48, 96))
    alb0 = 0.12 + 0.2 * rng.random((48, 96))
    snowd_am = np.where(np.abs(lat) > 50, 150.0 * rng.random((48, 96)), 0.0)             # some above sd2sc = 60 (snowc saturates at 1)
    sice_am = np.where(np.abs(lat) > 65, rng.random((48, 96)), 0.0)
    return dict(phis0=phis0, fmask_l=fmask_l, fmask_s=fmask_s, stl_am=stl_am, sst_am=sst_am, alb0=alb0, snowd_am=snowd_am, sice_am=sice_am)


def main():
    from _oracle import RefPhys
    ref = RefPhys(HSG, gaussian_latitudes())
    inp = fordate_inputs()
    out = ref.fordate(TYEAR, **inp)
    path = os.path.join(os.path.dirname(__file__), "fordate_golden.npz")
    np.savez_compressed(path, **{"in_" + k: v for k, v in inp.items()}, **out)
    print("wrote", path, os.path.getsize(path), "bytes; |tcorh| max", np.abs(out["tcorh"]).max(), "|qcorh| max", np.abs(out["qcorh"]).max(),
          "snowc == 1 at", int((out["snowc"] == 1.0).sum()), "points")


if __name__ == "__main__":
    main()
