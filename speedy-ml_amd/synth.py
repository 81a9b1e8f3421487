"""Synthetic, ERA5-shaped inputs for the hybrid-step hot path.

There is no network and no ERA5/NetCDF in the image, so every test and the benchmark use these seeded
generators.  Pure numpy, no device code: the same arrays feed the HIP path and the CPU oracle.

Reservoir construction mirrors the reference's generators in *distribution* (not bit-for-bit: the
reference uses the Fortran compiler's RNG):
  * A  : k = int(deg/m * n * n) COO entries; rows and cols are concatenated random permutations of 1..n (1-based,
         unsorted, duplicate (row,col) pairs possible), vals ~ U(0,1) rescaled to spectral radius ~rho
                                                                    (mod_linalg.f90:180-218, mod_reservoir.f90:182-212)
  * Win: one nonzero per row, rows (i-1)q+1..iq of column i ~ sigma*U(-1,1), q = n/d   (mod_reservoir.f90:262-283)
  * Wout ~ N(0, 1e-2), mean ~ U(-1,1), std ~ U(0.5,2)
"""
from dataclasses import dataclass

import numpy as np

XGRID, YGRID, ZGRID = 96, 48, 8


@dataclass
class SynthReservoir:
    n: int
    d: int
    n_model: int
    n_out: int
    rows: np.ndarray      # int32 (k,) 1-based
    cols: np.ndarray      # int32 (k,) 1-based
    vals: np.ndarray      # f64 (k,)
    win: np.ndarray       # f64 (n, d) Fortran order
    wout: np.ndarray      # f64 (n_out, n_model + n) Fortran order
    mean: np.ndarray      # f64 (36,)
    std: np.ndarray       # f64 (36,)
    feedback: np.ndarray  # f64 (d,)
    local_model: np.ndarray  # f64 (n_model,)

    @property
    def k(self):
        return len(self.vals)

    @property
    def n_aug(self):
        return self.n + self.n_model


def make_reservoir(n=5760, d=576, n_model=132, n_out=136, seed=20240954, deg=6, m=6000, radius=0.7, sigma=0.5,
                   dense_win=True):
    rng = np.random.default_rng(seed)
    k = int((deg / float(m)) * n * n)
    # makesparse (src/mod_linalg.f90:180-218): rows and cols are concatenated random permutations of 1..n, so every
    # row/column of A carries floor(k/n) or floor(k/n)+1 entries
    def perm_list():
        blocks = [rng.permutation(n) + 1 for _ in range(k // n)]
        if k % n:
            blocks.append(rng.permutation(n)[:k % n] + 1)
        return np.concatenate(blocks).astype(np.int32) if blocks else np.zeros(0, dtype=np.int32)
    rows, cols = perm_list(), perm_list()
    vals = rng.random(k)
    # spectral radius of a non-negative random matrix ~ mean row sum
    lam = max(k / float(n) * 0.5, 1e-3)
    vals *= radius / lam
    q = n // d
    win = np.zeros((n, d), order="F") if dense_win else None
    wvals = sigma * rng.uniform(-1.0, 1.0, size=n)
    if dense_win:
        for i in range(d):
            win[i * q:(i + 1) * q, i] = wvals[i * q:(i + 1) * q]
    wout = np.asfortranarray(rng.standard_normal((n_out, n_model + n)) * 1e-2)
    mean = rng.uniform(-1.0, 1.0, 36)
    std = rng.uniform(0.5, 2.0, 36)
    feedback = rng.standard_normal(d)
    local_model = rng.standard_normal(n_model)
    r = SynthReservoir(n, d, n_model, n_out, rows, cols, vals, win, wout, mean, std, feedback, local_model)
    r.win_vals = wvals          # the structured nonzeros, for memory-lean construction at full scale
    r.win_q = q
    return r


def synthetic_state(seed=0):
    """ERA5-shaped global state: returns (grid4d[z,y,x,v], logp[y,x], precip[y,x], sst[y,x]).
    Variable order T,u,v,q (src/ppo_iogrid.f90:596-599)."""
    rng = np.random.default_rng(seed)
    lat = np.deg2rad(np.linspace(-87.159, 87.159, YGRID))[:, None]
    sig = np.array([0.025, 0.095, 0.20, 0.34, 0.51, 0.685, 0.835, 0.95])
    g4 = np.zeros((ZGRID, YGRID, XGRID, 4))
    smooth = rng.standard_normal((YGRID, XGRID))
    smooth = (smooth + np.roll(smooth, 1, 1) + np.roll(smooth, -1, 1)) / 3.0
    for z in range(ZGRID):
        g4[z, :, :, 0] = 288.0 * sig[z] ** 0.19 + 5.0 * np.sin(lat) * smooth
        g4[z, :, :, 1] = 10.0 * rng.standard_normal((YGRID, XGRID))
        g4[z, :, :, 2] = 10.0 * rng.standard_normal((YGRID, XGRID))
        g4[z, :, :, 3] = np.maximum(1e-6, 10.0 * sig[z] ** 3 * np.exp(-(np.rad2deg(lat) / 40.0) ** 2)) * np.ones((1, XGRID))
    logp = 0.05 * rng.standard_normal((YGRID, XGRID))
    precip = np.log1p(np.maximum(0.0, rng.exponential(1e-4, (YGRID, XGRID))) / 1e-3)
    sst = np.maximum(272.0, 300.0 - 30.0 * np.sin(lat) ** 2) * np.ones((1, XGRID))
    return g4, logp, precip, sst


def land_mask(seed=1):
    """Synthetic sea mask (1 = sea, SST input present): ~70 % sea in coherent blobs."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((YGRID, XGRID))
    for _ in range(6):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5.0
    return (f < np.quantile(f, 0.7)).astype(np.int32)
