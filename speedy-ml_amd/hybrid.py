"""Device-resident hybrid step: the MI355X form of one iteration of program main's `t` loop
(src/parallelmain.f90:207-272) for the regions owned by one rank.

    predict (all resident reservoirs)               src/parallelmain.f90:226-251 -> mod_reservoir.f90:1418
    exchange: [all-gather] / scatter + clamps       src/mpires.f90:309-490
    SPEEDY hand-off iogrid(30)/(31) transforms      src/ppo_iogrid.f90:497-601
    SPEEDY time-step transform schedule             src/dyn_grtend.f90:61-277, src/phy_phypar.f90:54-66
    gather + standardise next inputs                src/mpires.f90:580-775

Everything runs through libspeedyml_hip.so; torch is used only for device buffers, streams and
torch.distributed (RCCL).  SPEEDY's grid-point dynamics and column physics stay on the host in the reference and
are out of scope: the forecast handed back to the reservoirs is the state after the
hand-off transforms (single-precision rounding + triangular truncation), and the transform schedule of the 26
time steps of one 6-h window is replayed on the device-resident spectral state with the grid-point work left out
(DESIGN.md "What a bench step is").
"""
import numpy as np

from . import domain
from .exchange import Exchange, handoff_check, handoff_from_fields, handoff_to_fields
from .reservoir import ReservoirBank
from .spectral import IL, IX, MX2, NX, Spectral
from .synth import make_reservoir, synthetic_state

NREG = 1152
STEPS_PER_WINDOW = 26          # stepone (2 steps) + 24 leapfrog steps of one 6-h window


def region_classes(sea_mask):
    """(pole, sst_input) per region: a region takes SST input when its 2x2 res patch is mostly sea."""
    out = []
    for r in range(NREG):
        g = domain.initializedomain(NREG, r)
        patch = sea_mask[g.res_ystart - 1:g.res_yend, g.res_xstart - 1:g.res_xend]
        out.append((bool(g.pole), bool(patch.mean() >= 0.5)))
    return out


def build_bank(regions, classes, seed=20240000, n_override=None, verbose=False):
    """Load one synthetic trained reservoir per region into a ReservoirBank (slot i <-> regions[i]).

    One base reservoir is generated per size class and shared by the regions of the class (each slot still owns
    a private copy in HBM; throughput does not depend on the values); statistics are per region.
    n_override = nodes per input (reference: NINT(6000/d)) -> small reservoirs for parity tests."""
    bank = ReservoirBank(len(regions))
    base, sizes, keep = {}, {}, {}
    for slot, r in enumerate(regions):
        pole, sst = classes[r]
        g = domain.initializedomain(NREG, r)
        s = domain.allocate_res_sizes(g, sst_bool_input=sst)
        d = s.reservoir_numinputs
        n = s.n if n_override is None else n_override * d
        key = (n, d)
        if key not in base:
            b = make_reservoir(n=n, d=d, n_model=s.chunk_size_speedy, n_out=s.chunk_size_prediction,
                               seed=seed + len(base), dense_win=False)
            b.win_rows = np.arange(1, n + 1, dtype=np.int32)
            b.win_cols = (np.arange(n, dtype=np.int32) // b.win_q + 1).astype(np.int32)
            base[key] = b
            if verbose:
                print(f"class n={n} d={d} k={b.k}", flush=True)
        b = base[key]
        rng = np.random.default_rng(seed + 7919 * (r + 1))
        mean, std = rng.uniform(-1.0, 1.0, 36), rng.uniform(0.5, 2.0, 36)
        _, stat = domain.out_map(NREG, r)
        bank.load_sparse_win(slot, b.n, b.d, b.n_model, b.n_out, b.rows, b.cols, b.vals, b.win_rows, b.win_cols,
                             b.win_vals, b.wout, mean, std, stat)
        sizes[slot] = s
        keep[slot] = (b, mean, std, stat)
    bank.host_copies = keep
    return bank, sizes


def device_view(ptr, shape):
    """torch view (no copy) of device memory owned by the C-ABI library."""
    import torch

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(np.prod(shape)),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda").view(*shape)


def gather_outvec_slab(local_outvec, regions, all_out, even_split):
    """Region-ordered slab of every rank's outvecs on every rank (works on any torch device / backend).

    processor_decomposition gives each rank a contiguous block of regions; when the region count divides evenly the
    rank blocks are equal and one all_gather_into_tensor of the resident outvec buffer IS the region-ordered slab
    (no packing, no copy).  With a remainder (ranks 1..left_over own one extra region taken from the tail,
    src/res_domain.f90:53-60) the blocks are ragged: every rank writes its rows into a zeroed slab and the slabs
    are summed -- each row has exactly one owner, so the sum is exact."""
    import torch.distributed as dist
    if all_out.is_cuda and dist.get_backend() != "nccl":
        # rehearsal only (several ranks sharing one GPU over gloo): stage the slab through the host
        host = all_out.cpu()
        gather_outvec_slab(local_outvec.cpu(), regions, host, even_split)
        all_out.copy_(host)
        return all_out
    if even_split:
        dist.all_gather_into_tensor(all_out, local_outvec)
    else:
        import torch
        all_out.zero_()
        all_out[torch.as_tensor(regions, dtype=torch.long, device=all_out.device)] = local_outvec[:len(regions)]
        dist.all_reduce(all_out)
    return all_out


class HybridRank:
    """All state of one rank for the device-resident step loop."""

    def __init__(self, regions, classes, world=1, rank=0, sea_mask=None, mode="hybrid", seed=20240000, n_override=None,
                 replay_steps=STEPS_PER_WINDOW):
        import torch
        self.torch = torch
        self.regions, self.classes, self.world, self.rank, self.mode = list(regions), classes, world, rank, mode
        self.replay_steps = replay_steps
        self.bank, self.sizes = build_bank(self.regions, classes, seed=seed, n_override=n_override)
        cap = self.bank.capacity
        self.feedback = device_view(self.bank.feedback_ptr, (cap, self.bank.max_d))
        self.local_model = device_view(self.bank.local_model_ptr, (cap, self.bank.max_n_model))
        self.outvec = device_view(self.bank.outvec_ptr, (cap, self.bank.max_n_out))
        dev = "cuda"
        f64 = torch.float64
        self.G = torch.zeros(domain.G_SIZE, dtype=f64, device=dev)
        self.F = torch.zeros(domain.G_SIZE, dtype=f64, device=dev)
        g4, logp, precip, sst = synthetic_state(seed)
        self.base_sst = torch.from_numpy(np.ascontiguousarray(sst.ravel())).to(dev)
        self.G[domain.G4_OFF:domain.G2_OFF] = torch.from_numpy(g4.ravel()).to(dev)
        self.G[domain.G2_OFF:domain.GP_OFF] = torch.from_numpy(logp.ravel()).to(dev)
        self.G[domain.GP_OFF:domain.GS_OFF] = torch.from_numpy(precip.ravel()).to(dev)
        self.G[domain.GS_OFF:domain.GT_OFF] = self.base_sst
        # TISR table: one (96,48) slice per 6-h step of a 365-day year (full_tisr, src/mod_reservoir.f90:890-909)
        lat = np.deg2rad(np.linspace(-87.159, 87.159, 48))[None, :, None]
        lon = np.deg2rad(np.arange(96) * 3.75)[None, None, :]
        hours = (np.arange(1460) * 6.0)[:, None, None]
        decl = np.deg2rad(23.44) * np.sin(2 * np.pi * (hours / 24.0 - 80.0) / 365.0)
        cosz = np.sin(lat) * np.sin(decl) + np.cos(lat) * np.cos(decl) * np.cos(2 * np.pi * hours / 24.0 + lon - np.pi)
        self.tisr = torch.from_numpy(np.maximum(0.0, cosz) * 1361.0 * 3600.0).to(dev).contiguous()   # [1460][48][96]
        self.t = 0
        if mode == "hybrid":
            sst_flags = [int(classes[r][1]) for r in self.regions]
            self.ex = Exchange(self.bank, NREG, self.regions, sst_flags)
            self.sp = Spectral()
            self.all_out = torch.zeros((NREG, self.bank.max_n_out), dtype=f64, device=dev)
            self.fields = torch.zeros((33, IL, IX), dtype=f64, device=dev)
            self.fields_out = torch.zeros((33, IL, IX), dtype=f64, device=dev)
            self.spec_state = torch.zeros((33, NX, MX2), dtype=f64, device=dev)     # [t(8) | vor(8) | div(8) | q(8) | ps]
            self.uv = torch.zeros((16, NX, MX2), dtype=f64, device=dev)
            self.safe = torch.ones(1, dtype=torch.int32, device=dev)
            # scratch for the time-step transform schedule (91 inverse + 73 forward per step)
            self.sched_spec = torch.zeros((91, NX, MX2), dtype=f64, device=dev)
            self.sched_grid = torch.zeros((91, IL, IX), dtype=f64, device=dev)
            self.sched_out = torch.zeros((98, NX, MX2), dtype=f64, device=dev)
            self.sched_vds2 = torch.zeros((24, NX, MX2), dtype=f64, device=dev)
            self.sched_kcos = torch.tensor([1] * 57 + [2] * 34, dtype=torch.int32, device=dev)
            self.sched_scale = torch.tensor([1] * 48 + [0] * 25, dtype=torch.int32, device=dev)
            self.even_split = (NREG % world == 0)
            # first inputs: gather from the synthetic state (forecast = the same state) so feedback is realistic
            self.G[domain.GT_OFF:] = self.tisr[0].reshape(-1)
            self.ex.gather(self.G, self.G)
        else:
            rng = np.random.default_rng(seed + rank)
            self.feedback.copy_(torch.from_numpy(rng.standard_normal((cap, self.bank.max_d))))
            self.local_model.copy_(torch.from_numpy(rng.standard_normal((cap, self.bank.max_n_model))))
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ the step
    def exchange_outvec(self, stream):
        """All ranks end up with every region's outvec in region order (the MPI gather-to-root of
        src/mpires.f90:347-454 becomes one RCCL all-gather of the contiguous outvec slab)."""
        if self.world == 1:
            return self.outvec
        return gather_outvec_slab(self.outvec, self.regions, self.all_out, self.even_split)

    def handoff(self, stream):
        """iogrid(30) then iogrid(31) (src/ppo_iogrid.f90:497-601) on the device, 33 fields per launch."""
        sp, S = self.sp, self.spec_state
        handoff_to_fields(self.G, self.fields, stream)
        fT, fu, fv, fq_ps = self.fields[0:8], self.fields[8:16], self.fields[16:24], self.fields[24:33]
        sp.vdspec(fu, fv, 2, out=(S[8:16], S[16:24]), stream=stream)
        sp.spec(fT, out=S[0:8], stream=stream)
        sp.spec(fq_ps, out=S[24:33], stream=stream)
        sp.trunct(S, stream=stream)
        # back to grid point space: uvspec -> grid(.,2) for u,v ; grid(.,1) for t, q, ps
        sp.uvspec(S[8:16], S[16:24], out=(self.uv[0:8], self.uv[8:16]), stream=stream)
        sp.grid(self.uv, 2, out=self.fields_out[8:24], stream=stream)
        sp.grid(S[0:8], 1, out=self.fields_out[0:8], stream=stream)
        sp.grid(S[24:33], 1, out=self.fields_out[24:33], stream=stream)
        handoff_check(self.fields_out, self.safe, stream)
        handoff_from_fields(self.fields_out, self.F, stream)

    def speedy_transform_schedule(self, stream):
        """The 164 transforms of one SPEEDY time step (SURVEY Appendix C 'Schedule inside one step()') as THREE launches on
        the device-resident spectral state: all 91 inverse transforms (57 with kcos=1, 34 with kcos=2; per-field flags),
        all 73 forward transforms (48 pre-scaled by 1/cos = the specx halves of the 24 vdspec(.,.,2), 25 plain), then vds
        for the 24 (u,v) pairs."""
        sp = self.sp
        sp.grid_mixed(self.sched_spec, self.sched_kcos, out=self.sched_grid, stream=stream)
        sp.spec_mixed(self.sched_grid[0:73], self.sched_scale, out=self.sched_out[0:73], stream=stream)
        sp.vds(self.sched_out[0:24], self.sched_out[24:48], out=(self.sched_out[73:97], self.sched_vds2), stream=stream)

    def step(self, stream):
        self.bank.predict(stream=stream)
        if self.mode == "sweep":
            return
        allv = self.exchange_outvec(stream)
        self.ex.scatter(allv, self.G, base_sst=self.base_sst, stream=stream)
        self.handoff(stream)
        if self.replay_steps:
            # inputs of the replay: the hand-off's spectral state, tiled over the 91 schedule slots
            self.sched_spec[0:33].copy_(self.spec_state)
            self.sched_spec[33:66].copy_(self.spec_state)
            self.sched_spec[66:91].copy_(self.spec_state[0:25])
            for _ in range(self.replay_steps):
                self.speedy_transform_schedule(stream)
        self.t += 1
        self.G[domain.GT_OFF:].copy_(self.tisr[self.t % self.tisr.shape[0]].reshape(-1))
        self.ex.gather(self.G, self.F, stream=stream)

    # ------------------------------------------------------------------ measurement helpers
    def timing(self, on):
        from ._lib import check, lib
        check(lib().sml_bank_timing(self.bank._h, 1 if on else 0))

    def timing_collect(self):
        import ctypes as C
        from ._lib import check, lib
        um, rm, uc, rc = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        check(lib().sml_bank_timing_collect(self.bank._h, C.byref(um), C.byref(uc), C.byref(rm), C.byref(rc)))
        return {"update_ms": um.value, "update_launches": uc.value, "readout_ms": rm.value, "readout_launches": rc.value}

    def describe(self):
        if self.mode == "sweep":
            wl = "config3 sweep-only: batched predict of the rank's resident reservoirs"
        else:
            wl = ("BASELINE config 3: 1152-reservoir batched predict + region exchange (scatter, clamps, gather, standardise) "
                  "+ SPEEDY hand-off transforms (33 forward, 33 inverse) + transform schedule of %d SPEEDY time steps "
                  "(91 inverse + 73 forward each) on the device; host grid-point dynamics/column physics excluded "
                  "(SURVEY section 8: out of scope)" % self.replay_steps)
        return {"workload": wl, "regions_total": NREG, "regions_this_rank": len(self.regions),
                "transforms_per_step": 66 + 164 * self.replay_steps if self.mode == "hybrid" else 0,
                "parallelism": f"regions sharded by processor_decomposition over {self.world} rank(s); "
                               + ("one all-gather of the outvec slab per step" if self.world > 1 else "no collective")}

    def cpu_baseline(self, budget_s=12.0):
        """Reference-faithful CPU path (the oracle, 1 core) on a bounded sample of the same workload."""
        import os
        import sys
        import time
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "tests"))
        from _oracle import Oracle
        o = Oracle()
        r = make_reservoir(seed=20240954)          # interior + SST class, dense W_in as the reference stores it
        x = np.zeros(r.n)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s or n < 3:
            x, out = o.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, 1.0,
                                   r.feedback, r.local_model, x)
            n += 1
        per_predict = (time.perf_counter() - t0) / n
        sample = (f"{n} reference-faithful predict calls (COO SpMV, dense 26.5 MB W_in matmul, W_out GEMV) of one interior "
                  f"reservoir: {per_predict * 1e3:.3f} ms each, x1152 per step")
        total = per_predict * NREG
        if self.mode == "hybrid":
            from _oracle import RefSpectral
            rng = np.random.default_rng(1)
            v = rng.standard_normal((MX2, NX))
            gfield = rng.standard_normal((IX, IL))
            # the transform leg uses the COMPILED REFERENCE (oracle/_ref: the reference's own FFTPACK + Legendre code) when
            # its .so travelled with the snapshot, else the oracle's direct-DFT restatement (5x slower: flattering)
            eng = RefSpectral() if RefSpectral.available() else o
            which = "compiled reference spe_spectral.f90/FFTPACK (oracle/_ref)" if RefSpectral.available() else "oracle direct-DFT restatement"
            t1 = time.perf_counter()
            m = 0
            while time.perf_counter() - t1 < 3.0:
                eng.grid(v, 1)
                eng.spec(gfield)
                m += 1
            per_pair = (time.perf_counter() - t1) / m
            ntr = 66 + 164 * self.replay_steps
            total += per_pair * ntr / 2.0
            sample += (f"; {m} grid+spec pairs with the {which}: {per_pair * 1e6:.0f} us per pair (incl. ctypes call overhead), "
                       f"x{ntr // 2} pairs per step; exchange tilers not timed (small)")
        return {"value": 1.0 / total, "unit": "steps/s", "cores": 1, "kind": "port", "sample": sample}
