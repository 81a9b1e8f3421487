"""Generates tests/golden/spectral_golden.npz from the COMPILED REFERENCE (oracle/_ref/libref_spectral.so,
built in place from /root/reference/src by oracle/build_ref.sh).  Run in the build container only:

    python tests/golden/make_spectral_golden.py

The fixture holds data only: seeded inputs and the reference's outputs for every spectral entry point on
the hot path plus every parmtr/inifft table (row 15).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from _oracle import IL, IX, MX2, NX, TABLES, RefSpectral  # noqa: E402


def main():
    ref = RefSpectral()
    out = {}
    for w, (name, _) in TABLES.items():
        out["tab_" + name] = ref.table(w)
    nsh2 = ref.table(12).astype(int)
    rng = np.random.default_rng(20240954)

    def rand_spec(scale=1.0):
        v = rng.standard_normal((MX2, NX)) * scale
        for n in range(NX):
            v[nsh2[n]:, n] = 0.0          # above the triangular cut
        v[:, NX - 1] = 0.0                # row n=32 is never produced by specy
        v[1, :] = 0.0                     # Im of zonal wavenumber 0
        return v

    # three spectral states of different spectra: white, red (l^-2) and a single mode
    red = rand_spec()
    for n in range(NX):
        for c in range(MX2):
            l = c // 2 + n
            red[c, n] /= (1.0 + l) ** 2
    single = np.zeros((MX2, NX))
    single[2, 0] = 1.0
    single[3, 0] = 0.5                    # (Re,Im)=(1,1/2) at m=1 (SURVEY Appendix C convention check)
    specs = np.stack([rand_spec(), red, single])
    out["in_spec"] = specs
    out["grid_k1"] = np.stack([ref.grid(s, 1) for s in specs])
    out["grid_k2"] = np.stack([ref.grid(s, 2) for s in specs])
    out["gridy"] = np.stack([ref.gridy(s) for s in specs])

    grids = np.stack([rng.standard_normal((IX, IL)), out["grid_k1"][1] * 3.0 + 280.0,
                      10.0 * rng.standard_normal((IX, IL))])
    out["in_grid"] = grids
    out["spec"] = np.stack([ref.spec(g) for g in grids])
    out["specx"] = np.stack([ref.specx(g) for g in grids])
    out["specy"] = np.stack([ref.specy(x) for x in out["specx"]])
    for kc in (1, 2):
        vd = [ref.vdspec(grids[0], grids[2], kc), ref.vdspec(grids[2], grids[1], kc)]
        out[f"vdspec_vor_k{kc}"] = np.stack([a for a, _ in vd])
        out[f"vdspec_div_k{kc}"] = np.stack([b for _, b in vd])
    uv = [ref.uvspec(specs[0], specs[1]), ref.uvspec(specs[1], specs[0])]
    out["uvspec_u"] = np.stack([a for a, _ in uv])
    out["uvspec_v"] = np.stack([b for _, b in uv])
    vd = [ref.vds(specs[0], specs[1]), ref.vds(specs[1], specs[0])]
    out["vds_vor"] = np.stack([a for a, _ in vd])
    out["vds_div"] = np.stack([b for _, b in vd])
    gr = [ref.grad(s) for s in specs]
    out["grad_x"] = np.stack([a for a, _ in gr])
    out["grad_y"] = np.stack([b for _, b in gr])
    out["lap"] = np.stack([ref.lap(s) for s in specs])
    out["invlap"] = np.stack([ref.invlap(s) for s in specs])
    out["trunct"] = np.stack([ref.trunct(rng.standard_normal((MX2, NX)) * 0 + s + 1.0) for s in specs])
    r96 = rng.standard_normal((4, IX))
    out["in_r96"] = r96
    out["rfftf"] = np.stack([ref.rfftf(r) for r in r96])
    out["rfftb"] = np.stack([ref.rfftb(r) for r in r96])
    path = os.path.join(os.path.dirname(__file__), "spectral_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
