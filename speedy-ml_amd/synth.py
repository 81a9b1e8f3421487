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
                   dense_win=True, passthrough=False, float32_weights=False):
    """float32_weights: round vals / win / wout / mean / std to float32 (kept as float64 arrays) -- what a reservoir read back from the
    reference's NetCDF weight files holds (NF90_REAL, src/mod_io.f90 via write_trained_res, src/mod_reservoir.f90:1727-1736); the bank then
    stores the compact copies (sml_bank_storage)."""
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
    if passthrough:
        # a "trained" hybrid whose readout leans on the physics-model rows: output = SPEEDY forecast at the region's own
        # points + a small reservoir correction, so a closed-loop run stays on physical states
        wout *= 1e-2
        wout[:, :n_model] = 0.0
        wout[np.arange(min(n_out, n_model)), np.arange(min(n_out, n_model))] = 1.0
    mean = rng.uniform(-1.0, 1.0, 36)
    std = rng.uniform(0.5, 2.0, 36)
    feedback = rng.standard_normal(d)
    local_model = rng.standard_normal(n_model)
    if float32_weights:
        f32 = lambda a: None if a is None else np.asarray(a, dtype=np.float32).astype(np.float64, order="F" if np.ndim(a) == 2 else "C")
        vals, win, wout, mean, std, wvals = f32(vals), f32(win), f32(wout), f32(mean), f32(std), f32(wvals)
        wout = np.asfortranarray(wout)
    r = SynthReservoir(n, d, n_model, n_out, rows, cols, vals, win, wout, mean, std, feedback, local_model)
    r.win_vals = wvals          # the structured nonzeros, for memory-lean construction at full scale
    r.win_q = q
    return r


SIGMA = np.array([0.025, 0.095, 0.20, 0.34, 0.51, 0.685, 0.835, 0.95])      # full sigma levels (src/ini_indyns.f90:38-41)
LAT_DEG = np.linspace(-87.159, 87.159, YGRID)


def reference_temperature():
    """SPEEDY's semi-implicit reference profile 288*max(0.2,sigma)**(R*gamma/g) (src/ini_impint.f90:57-62)."""
    rgam = (2.0 / 7.0 * 1004.0) * 6.0 / (1000.0 * 9.81)
    return 288.0 * np.maximum(0.2, SIGMA) ** rgam


def synthetic_state(seed=0):
    """ERA5-shaped global state: returns (grid4d[z,y,x,v], logp[y,x], precip[y,x], sst[y,x]).
    Variable order T,u,v,q (src/ppo_iogrid.f90:596-599); q in g/kg, logp = log(ps/p0).

    The fields are smooth and roughly balanced (mid-latitude westerly jets, a pole-to-equator temperature contrast on
    SPEEDY's reference lapse rate, planetary waves 3-5 with seeded phases): SPEEDY's adiabatic core integrates this state
    stably for days, so a benchmark step works on physical values rather than on overflowing garbage."""
    rng = np.random.default_rng(seed)
    lat = np.deg2rad(LAT_DEG)[:, None]
    lon = np.deg2rad(np.arange(XGRID) * 3.75)[None, :]
    ph = rng.uniform(0, 2 * np.pi, 8)
    c, sn = np.cos(lat), np.sin(lat)
    tref = reference_temperature()
    g4 = np.zeros((ZGRID, YGRID, XGRID, 4))
    for z in range(ZGRID):
        s = SIGMA[z]
        wave_t = 1.5 * np.cos(3 * lon + ph[0]) * c ** 2 + 0.8 * np.cos(5 * lon + ph[1]) * np.sin(2 * lat) ** 2
        g4[z, :, :, 0] = tref[z] + 18.0 * (c ** 2 - 2.0 / 3.0) * min(1.0, s + 0.3) + wave_t * (0.4 + s)
        jet = 28.0 * np.sin(2 * lat) ** 2 * (1.15 - s)
        g4[z, :, :, 1] = jet + 4.0 * np.cos(4 * lon + ph[2]) * c ** 2 * (1.1 - s)
        g4[z, :, :, 2] = 4.0 * np.sin(4 * lon + ph[3]) * c ** 2 * sn * (1.1 - s) + 1.5 * np.sin(2 * lon + ph[4]) * c ** 3
        g4[z, :, :, 3] = np.maximum(1e-6, 14.0 * s ** 3 * np.exp(-(np.rad2deg(lat) / 35.0) ** 2)
                                    * (1.0 + 0.1 * np.cos(2 * lon + ph[5]))) * np.ones((1, XGRID))
    logp = 0.012 * np.cos(2 * lon + ph[6]) * c ** 2 - 0.008 * np.cos(3 * lon + ph[7]) * np.sin(2 * lat) ** 2
    precip = np.log1p(np.maximum(0.0, rng.exponential(1e-4, (YGRID, XGRID))) / 1e-3)
    sst = np.maximum(272.0, 300.0 - 30.0 * sn ** 2) * np.ones((1, XGRID))
    return g4, logp, precip, sst


def synthetic_orography():
    """Smooth surface geopotential phis0 [m^2/s^2] on the (48, 96) grid: three broad mountain massifs up to ~2.5 km."""
    lat = np.deg2rad(LAT_DEG)[:, None]
    lon = np.deg2rad(np.arange(XGRID) * 3.75)[None, :]
    h = np.zeros((YGRID, XGRID))
    for (la0, lo0, amp, wid) in ((35.0, 90.0, 2500.0, 0.30), (-20.0, 290.0, 1500.0, 0.22), (40.0, 250.0, 1200.0, 0.25)):
        d2 = (lat - np.deg2rad(la0)) ** 2 + (np.cos(lat) * np.angle(np.exp(1j * (lon - np.deg2rad(lo0))))) ** 2
        h += amp * np.exp(-d2 / wid ** 2)
    return 9.81 * h


def climate_stats():
    """Per-variable standardisation statistics (mean(36), std(36)) of the synthetic climate, slot l = var*8 + level for
    T,u,v,q, then logp (32), tisr (33), precip (34), sst (35) (src/res_domain.f90:1270-1315)."""
    mean, std = np.zeros(36), np.ones(36)
    mean[0:8], std[0:8] = reference_temperature(), 10.0
    mean[8:16], std[8:16] = 12.0 * (1.15 - SIGMA), 10.0
    mean[16:24], std[16:24] = 0.0, 6.0
    mean[24:32], std[24:32] = 5.0 * SIGMA ** 3, 1e-3 + 4.0 * SIGMA ** 3
    mean[32], std[32] = 0.0, 0.02
    mean[33], std[33] = 1.2e6, 1.4e6
    mean[34], std[34] = 0.05, 0.1
    mean[35], std[35] = 288.0, 10.0
    return mean, std


def land_mask(seed=1):
    """Synthetic sea mask (1 = sea, SST input present): ~70 % sea in coherent blobs."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((YGRID, XGRID))
    for _ in range(6):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5.0
    return (f < np.quantile(f, 0.7)).astype(np.int32)


def ar1_series_device(length, width, phi=0.98, seed=0, chunk=1024, device="cuda"):
    """ERA5-shaped synthetic training inputs generated ON the device: `width` independent stationary AR(1)
    series of `length` hourly samples, x_t = phi x_{t-1} + sqrt(1 - phi^2) eps_t, unit variance (standardised inputs), as one float64
    tensor [length, width].  Time is the sequential dimension: a chunk of L steps is one lower-triangular Toeplitz product
    X = phi^(1..L) x_prev + T eps on the matrix cores, so 350 640 hours x 9216 columns take a few hundred launches, not 350 640."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    out = torch.empty((length, width), dtype=torch.float64, device=device)
    k = torch.arange(chunk, dtype=torch.float64, device=device)
    lag = k[:, None] - k[None, :]
    toep = torch.where(lag >= 0, phi ** lag.clamp(min=0), torch.zeros((), dtype=torch.float64, device=device)) * (1.0 - phi * phi) ** 0.5
    carry = phi ** (k + 1.0)
    prev = torch.randn((width,), dtype=torch.float64, device=device, generator=g)          # stationary start
    for t0 in range(0, length, chunk):
        n = min(chunk, length - t0)
        eps = torch.randn((n, width), dtype=torch.float64, device=device, generator=g)
        blk = toep[:n, :n] @ eps + carry[:n, None] * prev[None, :]
        out[t0:t0 + n] = blk
        prev = blk[n - 1].clone()
    return out
