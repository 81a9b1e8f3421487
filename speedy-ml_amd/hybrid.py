"""Device-resident hybrid step: the MI355X form of one iteration of program main's `t` loop
(src/parallelmain.f90:207-272) for the regions owned by one rank.

    predict (all resident reservoirs)               src/parallelmain.f90:226-251 -> mod_reservoir.f90:1418
    exchange: [all-gather] / scatter + clamps       src/mpires.f90:309-490
    SPEEDY hand-off in, iogrid(30)                  src/ppo_iogrid.f90:497-577
    SPEEDY 6-hour window: stepone + 24 leapfrog     src/ini_stepone.f90, src/dyn_stloop.f90:28-43, src/dyn_step.f90
    SPEEDY hand-off out, iogrid(31)                 src/ppo_iogrid.f90:579-601
    gather + standardise next inputs                src/mpires.f90:580-775

Everything runs through libspeedyml_hip.so; torch is used only for device buffers, streams and
torch.distributed (RCCL).  The SPEEDY leg runs on the device (speedy-ml_amd/csrc/dynamics.hip, physics_dev.h): every spectral
transform, the grid-point tendencies with phypar's column physics (convection, condensation, clouds, radiation, surface fluxes,
vertical diffusion; physics=False gives the adiabatic core), the semi-implicit spectral step and the leapfrog of the 26 time
steps of a window (DESIGN.md "What a bench step is").
"""
import numpy as np

from . import domain
from ._lib import device_view
from .dynamics import DELT, F_DIV, F_PS, F_T, F_TR, F_VOR, NSTATE, Dynamics
from .exchange import Exchange, handoff_check, handoff_from_fields, handoff_to_fields
from .reservoir import ReservoirBank
from .spectral import IL, IX, MX2, NX, Spectral
from .synth import climate_stats, make_reservoir, synthetic_orography, synthetic_state

NREG = 1152
LEAPFROG_PER_WINDOW = 24       # nsteps/4 leapfrog steps of one 6-h window after stepone's two starters
GAMLAT = 6.0 / (1000.0 * 9.81)  # reference lapse rate gamma/(1000 g) of setgam (src/ini_fordate.f90:117-135)


def region_classes(sea_mask):
    """(pole, sst_input) per region: a region takes SST input when its 2x2 res patch is mostly sea."""
    out = []
    for r in range(NREG):
        g = domain.initializedomain(NREG, r)
        patch = sea_mask[g.res_ystart - 1:g.res_yend, g.res_xstart - 1:g.res_xend]
        out.append((bool(g.pole), bool(patch.mean() >= 0.5)))
    return out


def build_bank(regions, classes, seed=20240000, n_override=None, verbose=False, physical=True, ml_only=False, float32_weights=False):
    """Load one synthetic trained reservoir per region into a ReservoirBank (slot i <-> regions[i]).

    One base reservoir is generated per size class and shared by the regions of the class (each slot still owns
    a private copy in HBM; throughput does not depend on the values); statistics are per region.
    n_override = nodes per input (reference: NINT(6000/d)) -> small reservoirs for parity tests.
    physical: W_out passes the SPEEDY forecast through with a small reservoir correction and the statistics are those of
    the synthetic climate (+-5 % per region), so closed-loop runs stay on physical states; False = random W_out/statistics."""
    bank = ReservoirBank(len(regions))
    base, sizes, keep = {}, {}, {}
    # the seed of a size class is its position in the order of first appearance over ALL regions, so that a rank holding any subset
    # builds the reservoirs the single-rank run builds
    class_index = {}
    for r in range(NREG):
        sz = domain.allocate_res_sizes(domain.initializedomain(NREG, r), sst_bool_input=classes[r][1])
        class_index.setdefault(sz.reservoir_numinputs, len(class_index))
    for slot, r in enumerate(regions):
        pole, sst = classes[r]
        g = domain.initializedomain(NREG, r)
        s = domain.allocate_res_sizes(g, sst_bool_input=sst)
        d = s.reservoir_numinputs
        n = s.n if n_override is None else n_override * d
        f32 = bool(float32_weights(r)) if callable(float32_weights) else bool(float32_weights)      # (a callable: per region)
        key = (n, d, f32)
        if key not in base:
            # ml_only: chunk_size_speedy = 0 (predict_ml, src/mod_reservoir.f90:1491-1535): W_out acts on the reservoir state alone
            b = make_reservoir(n=n, d=d, n_model=0 if ml_only else s.chunk_size_speedy, n_out=s.chunk_size_prediction,
                               seed=seed + class_index[d], dense_win=False, passthrough=physical and not ml_only, float32_weights=f32)
            b.win_rows = np.arange(1, n + 1, dtype=np.int32)
            b.win_cols = (np.arange(n, dtype=np.int32) // b.win_q + 1).astype(np.int32)
            base[key] = b
            if verbose:
                print(f"class n={n} d={d} k={b.k}", flush=True)
        b = base[key]
        rng = np.random.default_rng(seed + 7919 * (r + 1))
        if physical:
            mean, std = climate_stats()
            mean, std = mean * rng.uniform(0.98, 1.02, 36), std * rng.uniform(0.95, 1.05, 36)
        else:
            mean, std = rng.uniform(-1.0, 1.0, 36), rng.uniform(0.5, 2.0, 36)
        _, stat = domain.out_map(NREG, r)
        bank.load_sparse_win(slot, b.n, b.d, b.n_model, b.n_out, b.rows, b.cols, b.vals, b.win_rows, b.win_cols,
                             b.win_vals, b.wout, mean, std, stat)
        sizes[slot] = s
        keep[slot] = (b, mean, std, stat)
    bank.host_copies = keep
    return bank, sizes


def agree_on_storage(banks):
    """Several ranks (torch.distributed initialised): a bank reads its compact (float) copies only if EVERY rank's bank can
    (ReservoirBank.compact) -- the compact readout sums in another association than the 8-byte one, and a run's result must not depend on
    how the regions are dealt to ranks.  Collective."""
    import torch
    import torch.distributed as dist
    for bank in banks:
        if bank is None:
            continue
        flag = torch.tensor([1 if bank.compact() else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()):
            bank.use_compact(False)


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
                 leapfrog_steps=LEAPFROG_PER_WINDOW, physical=True, pipeline=False, persistent_readout=True, drain_readout=True,
                 start_hours=12000 + 24 * 14, slab=False, physics=True, speedy_cus=0, float32_weights=False):
        import torch
        self.torch = torch
        self.regions, self.classes, self.world, self.rank, self.mode = list(regions), classes, world, rank, mode
        self.leapfrog_steps = leapfrog_steps
        self.pipeline = pipeline and mode == "hybrid"
        self._region_index = None
        self.persistent_readout, self.drain_readout = persistent_readout, drain_readout
        self.bank, self.sizes = build_bank(self.regions, classes, seed=seed, n_override=n_override, physical=physical,
                                           ml_only=mode == "ml_only", float32_weights=float32_weights)
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
        # TISR table: one (96,48) slice per hour of a 365-day year (full_tisr, src/mod_reservoir.f90:890-909); the slice of a
        # step is chosen by the hybrid's calendar exactly as get_tisr_by_date does (src/mpires.f90:1676-1708)
        lat = np.deg2rad(np.linspace(-87.159, 87.159, 48))[None, :, None]
        lon = np.deg2rad(np.arange(96) * 3.75)[None, None, :]
        hours = np.arange(8760, dtype=np.float64)[:, None, None]
        decl = np.deg2rad(23.44) * np.sin(2 * np.pi * (hours / 24.0 - 80.0) / 365.0)
        cosz = np.sin(lat) * np.sin(decl) + np.cos(lat) * np.cos(decl) * np.cos(2 * np.pi * hours / 24.0 + lon - np.pi)
        self.tisr = torch.from_numpy(np.maximum(0.0, cosz) * 1361.0 * 3600.0).to(dev).contiguous()   # [8760][48][96]
        # hours since 1 Jan 1981 00h at the first prediction step: traininglength + prediction marker + synclength of the
        # shipped configuration (src/mod_reservoir.f90:40-78: 12000-h training window given in hours, 14-day sync)
        self.start_hours = start_hours
        self.timestep_hours = 6
        self.t = 0
        self._safe_ring, self._safe_ev, self._safe_n, self._aborted = None, None, 0, False
        self.stop_on_unsafe = True            # (False only for load emulation: a rank's share of the regions without its peers' outvecs)
        self._phase_on, self._phase_log = False, []
        if mode == "ml_only":
            # the reference's ml_only run (src/parallelmain.f90:229-231, src/mpires.f90:566,588): predict_ml, the same exchange,
            # no SPEEDY window, no local_model
            sst_flags = [int(classes[r][1]) for r in self.regions]
            self.ex = Exchange(self.bank, NREG, self.regions, sst_flags)
            self.all_out = torch.zeros((NREG, self.bank.max_n_out), dtype=f64, device=dev)
            self.even_split = (NREG % world == 0)
            self.slab = self.phys = None
            self.leapfrog_steps = None
            self.G[domain.GT_OFF:] = self.tisr_slice(0).reshape(-1)
            self.ex.gather(self.G, None)
        elif mode == "hybrid":
            sst_flags = [int(classes[r][1]) for r in self.regions]
            self.ex = Exchange(self.bank, NREG, self.regions, sst_flags)
            self.sp = Spectral()
            self.all_out = torch.zeros((NREG, self.bank.max_n_out), dtype=f64, device=dev)
            self.fields = torch.zeros((33, IL, IX), dtype=f64, device=dev)
            self.fields_out = torch.zeros((33, IL, IX), dtype=f64, device=dev)
            # SPEEDY's prognostic state, both leapfrog levels: [2][vor(8) | div(8) | t(8) | q(8) | ps] (mod_dynvar.f90)
            self.state = torch.zeros((2, NSTATE, NX, MX2), dtype=f64, device=dev)
            # iogrid(31)'s inverse transforms in one launch: [t | u v (uvspec of vor, div) | q | ps] (see Spectral.grid_derived)
            rows = ([(0, F_T + k, F_T + k, 1) for k in range(8)] + [(1, F_VOR + k, F_DIV + k, 2) for k in range(8)]
                    + [(2, F_VOR + k, F_DIV + k, 2) for k in range(8)] + [(0, F_TR + k, F_TR + k, 1) for k in range(8)]
                    + [(0, F_PS, F_PS, 1)])
            self.out_desc = torch.tensor(rows, dtype=torch.int32, device=dev)
            # iogrid(30)'s forward side in two launches: all 33 fields [t u v q ps] transformed at once (u, v pre-scaled by 1/cos as
            # vdspec(.,.,2) does), then vds + trunct straight into time level 1 of the state [vor div t q ps]
            self.in_scale = torch.tensor([0] * 8 + [1] * 16 + [0] * 9, dtype=torch.int32, device=dev)
            rows = ([(5, 8 + k, 16 + k, 1) for k in range(8)] + [(6, 8 + k, 16 + k, 1) for k in range(8)]
                    + [(0, k, k, 1) for k in range(8)] + [(0, 24 + k, 24 + k, 1) for k in range(8)] + [(0, 32, 32, 1)])
            self.in_desc = torch.tensor(rows, dtype=torch.int32, device=dev)
            self.raw_spec = torch.zeros((33, NX, MX2), dtype=f64, device=dev)
            self.safe = torch.ones(1, dtype=torch.int32, device=dev)
            self.dyn = Dynamics(self.sp)
            # boundary fields as invars / fordate leave them (src/ini_invars.f90:31-34, src/ini_fordate.f90:72-86): phis = trunct(spec(phi0)),
            # phis0 = grid(phis) -- the truncated orography, also the physics' phis0 -- and tcorh = spec(gamlat phis0), not truncated.
            # The humidity correction qcorh needs the surface temperatures: zero for the adiabatic core; with the physics attached every
            # window recomputes tcorh and qcorh from the hybrid SST (Physics.fordate in speedy_leg)
            phi0 = torch.from_numpy(np.ascontiguousarray(synthetic_orography())).to(dev)
            self.phis = self.sp.spec(phi0[None])
            self.sp.trunct(self.phis)
            phis0 = self.sp.grid(self.phis, 1)
            self.phis0_grid = phis0[0].cpu().numpy()
            self.tcorh = self.sp.spec(torch.from_numpy(GAMLAT * self.phis0_grid).to(dev)[None])[0].contiguous()
            self.phis = self.phis[0].contiguous()
            self.dyn.set_boundary(self.phis, self.tcorh, torch.zeros((NX, MX2), dtype=f64, device=dev))
            self.dyn.set_range_guard(self.safe)
            self.phys = None
            if physics:
                self.init_physics(sea_mask, g4)
            self.even_split = (NREG % world == 0)
            self.slab = None
            if slab:
                self.init_slab(classes, sea_mask, seed, n_override, physical)
            if world > 1:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    agree_on_storage([self.bank, self.slab_bank if self.slab is not None else None])
            # first inputs: gather from the synthetic state (forecast = the same state) so feedback is realistic
            self.G[domain.GT_OFF:] = self.tisr_slice(0).reshape(-1)
            self.ex.gather(self.G, self.G)
            if self.pipeline:
                # Software pipeline (see step()): the reservoir advance and the state block of the readout of step t+1 run on a
                # side stream under the SPEEDY window of step t.  Prologue: advance + state block of the first step.
                self.side, self.main = torch.cuda.Stream(), None
                if speedy_cus:
                    # CU partition: the SPEEDY window (and the small exchange kernels) on the first `speedy_cus` compute units, the
                    # reservoir advance + readout on the rest, so that the HBM-streaming readout never shares a CU with the
                    # latency-bound window (sml_stream_create_cu_mask)
                    import ctypes as C
                    from ._lib import check, lib
                    ncu = torch.cuda.get_device_properties(0).multi_processor_count
                    assert 0 < speedy_cus < ncu
                    words = (ncu + 31) // 32

                    def masked(lo, hi):
                        m = (C.c_uint32 * words)()
                        for cu in range(lo, hi):
                            m[cu // 32] |= 1 << (cu % 32)
                        h = C.c_void_p()
                        check(lib().sml_stream_create_cu_mask(m, words, C.byref(h)))
                        return torch.cuda.ExternalStream(h.value), h
                    self.main, self._main_h = masked(0, speedy_cus)
                    self.side, self._side_h = masked(speedy_cus, ncu)
                self.ev_feedback, self.ev_partial = torch.cuda.Event(), torch.cuda.Event()
                self.bank.advance(stream=torch.cuda.current_stream())
                self.bank.readout_part(1, stream=torch.cuda.current_stream())
                self.ev_partial.record(torch.cuda.current_stream())
        else:
            rng = np.random.default_rng(seed + rank)
            self.feedback.copy_(torch.from_numpy(rng.standard_normal((cap, self.bank.max_d))))
            self.local_model.copy_(torch.from_numpy(rng.standard_normal((cap, self.bank.max_n_model))))
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ SPEEDY column physics inside every time step
    def init_physics(self, sea_mask, g4):
        """phypar in grtend (src/dyn_grtend.f90:222-225) with synthetic surface boundary fields: land fraction from the land-sea
        mask, the truncated orography of the dynamics (mod_surfcon's phis0), a land temperature from the lowest model level of the start
        state, mid-range soil wetness, snow-free albedos.  The sea temperature is the hybrid state's SST grid, refreshed every step
        (sst_am).  fordate's sea fraction is 1 - fmask_l (src/ini_inbcon.f90:55-65,148-157)."""
        from .physics import NSTRAD, Physics
        from .synth import land_mask
        sea = (land_mask() if sea_mask is None else np.asarray(sea_mask)).reshape(IL, IX).astype(np.float64)
        fmask = 1.0 - sea
        phis0 = self.phis0_grid
        tland = np.ascontiguousarray(g4[7, :, :, 0])
        sia = np.asarray(self.sp.table(1)).ravel()                       # sines of the 24 northern Gauss latitudes
        self.phys = Physics(np.concatenate([-np.arcsin(sia), np.arcsin(sia)[::-1]]))     # radang, src/ini_indyns.f90:72-80
        # albedos as fordate derives them (src/ini_fordate.f90:54-61) from a bare-land albedo of 0.2, no snow and no sea ice; every
        # window's fordate recomputes them on the device from the same three inputs
        alb0, snowd_am, sice_am = np.full((IL, IX), 0.2), np.zeros((IL, IX)), np.zeros((IL, IX))
        alb_l, alb_s = alb0 + 0.0 * (0.60 - alb0), 0.07 + sice_am * (0.60 - 0.07)
        self.surface = dict(fmask=fmask, phis0=phis0, tland=tland, tsea=self.base_sst.cpu().numpy().reshape(IL, IX), swav=np.full((IL, IX), 0.5),
                            alb_l=alb_l, alb_s=alb_s, albsfc=alb_s + fmask * (alb_l - alb_s), snowc=np.zeros((IL, IX)),
                            alb0=alb0, snowd_am=snowd_am, sice_am=sice_am)
        self.phys.set_surface(*[self.surface[k] for k in ("fmask", "phis0", "tland", "tsea", "swav", "alb_l", "alb_s", "albsfc", "snowc")])
        self.phys.bind_sst(self.G[domain.GS_OFF:domain.GT_OFF])      # sst_am = the hybrid state's SST grid, read in place
        self.phys.set_fordate_fields(1.0 - fmask, alb0, snowd_am, sice_am)
        self.phys_day = None
        self.update_forcing()
        self.dyn.attach_physics(self.phys, NSTRAD)       # diagnostics stay on: skipping their stores changes nothing measurable

    def update_forcing(self):
        """fordate's daily call of sol_oz(tyear), tyear = (day of the 365-day year - 0.5) / 365 (src/ini_fordate.f90:47-50 with
        src/mod_date.f90's tyear)"""
        _, month, iday, _ = domain.calendar_date(self.start_hours + self.t * self.timestep_hours)
        day = (0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334)[month - 1] + iday       # ndaycal(imonth, 2) + iday
        if day != self.phys_day:
            self.phys_day = day
            self.phys.sol_oz((day - 0.5) / 365.0)

    # ------------------------------------------------------------------ slab ocean (BASELINE config 5)
    def init_slab(self, classes, sea_mask, seed, n_override, physical):
        """One slab-ocean reservoir per region that predicts SST (src/mod_slab_ocean_reservoir.f90:9-133: inputs = 27-step mean of
        the lowest-level atmosphere inputs + logp + SST + TISR, plus an OHTC patch; outputs SST and OHTC of the 2x2 res patch)."""
        import torch
        from .slab import SLAB_RADIUS, SLAB_SIGMA, SlabCoupler, slab_sizes
        dev = "cuda"
        sea_slot = [int(classes[r][1]) for r in self.regions]
        self.sea_of_region = torch.tensor([int(c[1]) for c in classes], dtype=torch.int32, device=dev)
        # the SST kernel restores base_sst where sea_mask > 0 (src/mpires.f90:470-478): the reference's mask marks land
        land = (1 - np.asarray(sea_mask, dtype=np.int32)) if sea_mask is not None else np.zeros((48, 96), dtype=np.int32)
        self.sst_mask = torch.from_numpy(np.ascontiguousarray(land.reshape(-1))).to(dev)
        sizes = [slab_sizes(domain.initializedomain(NREG, r)) for r in self.regions]
        max_d = max(s.reservoir_numinputs for s in sizes)
        max_out = max(s.chunk_size_prediction for s in sizes)
        self.slab_bank = ReservoirBank(len(self.regions), max_d=max_d, max_n_model=1, max_n_out=max_out)
        base = {}
        class_index = {}                                        # (as in build_bank: first appearance over ALL regions with a slab model)
        for r in range(NREG):
            if classes[r][1]:
                class_index.setdefault(slab_sizes(domain.initializedomain(NREG, r)).reservoir_numinputs, len(class_index))
        for slot, r in enumerate(self.regions):
            if not sea_slot[slot]:
                continue
            s = sizes[slot]
            d = s.reservoir_numinputs
            n = s.n if n_override is None else n_override * d
            if (n, d) not in base:
                b = make_reservoir(n=n, d=d, n_model=0, n_out=s.chunk_size_prediction, seed=seed + 500 + class_index[d], deg=6, m=4000,
                                   radius=SLAB_RADIUS, sigma=SLAB_SIGMA, dense_win=False)
                if physical:
                    b.wout *= 1e-2          # small anomalies around the region's mean SST
                b.win_rows = np.arange(1, n + 1, dtype=np.int32)
                b.win_cols = (np.arange(n, dtype=np.int32) // b.win_q + 1).astype(np.int32)
                base[(n, d)] = b
            b = base[(n, d)]
            _, mean, std, _ = self.bank.host_copies[slot]
            stat = np.full(b.n_out, 35, dtype=np.int32)          # predict_slab_ml un-standardises every output with the SST statistics
            self.slab_bank.load_sparse_win(slot, b.n, b.d, 0, b.n_out, b.rows, b.cols, b.vals, b.win_rows, b.win_cols, b.win_vals,
                                           b.wout, mean, std, stat)
        self.slab_feedback = device_view(self.slab_bank.feedback_ptr, (self.slab_bank.capacity, max_d))
        self.slab_outvec = device_view(self.slab_bank.outvec_ptr, (self.slab_bank.capacity, max_out))
        self.slab = SlabCoupler(self.bank, self.slab_bank, NREG, self.regions, sea_slot, [int(classes[r][1]) for r in self.regions])
        # OHTC input patch: the exchange tiles an all-zero wholegrid_ohtc and standardises it with statistics slot 1
        # (src/mpires.f90:294-295,730-733; src/mod_slab_ocean_reservoir.f90:349)
        for slot in range(len(self.regions)):
            if sea_slot[slot]:
                _, mean, std, _ = self.bank.host_copies[slot]
                s = sizes[slot]
                self.slab_feedback[slot, s.tisr_end:s.reservoir_numinputs] = (0.0 - mean[0]) / std[0]
        # SST outputs before the first slab prediction: the base SST at the region's res cells (start_prediction_slab takes them
        # from the synchronisation data)
        self.all_slab_out = torch.zeros((NREG, max_out), dtype=torch.float64, device=dev)
        base_sst = self.base_sst.cpu().numpy()
        rows = np.zeros((NREG, max_out))
        for r in range(NREG):
            gmap, _ = domain.out_map(NREG, r)
            cells = np.asarray(gmap[128:132]) - domain.G2_OFF
            rows[r, :4] = base_sst[cells]
        self.all_slab_out.copy_(torch.from_numpy(rows))
        self.slab_outvec.copy_(self.all_slab_out[torch.as_tensor(self.regions, dtype=torch.long, device=dev)])

    def slab_predict_and_share(self, stream):
        """predict_slab_ml for every SST-predicting region of the rank, then every rank gets all regions' outputs"""
        self.slab_bank.predict(stream=stream)
        if self.world == 1:
            self.all_slab_out.copy_(self.slab_outvec)
        else:
            gather_outvec_slab(self.slab_outvec, self.regions, self.all_slab_out, self.even_split)

    def scatter_all(self, allv, stream):
        if self.slab is not None:
            self.slab.scatter_sst(self.all_slab_out, self.sea_of_region, self.G, stream=stream)
            self.ex.scatter(allv, self.G, base_sst=self.base_sst, sea_mask=self.sst_mask, stream=stream)
        else:
            self.ex.scatter(allv, self.G, base_sst=self.base_sst, stream=stream)

    # ------------------------------------------------------------------ the step
    def exchange_outvec(self, stream):
        """All ranks end up with every region's outvec in region order (the MPI gather-to-root of
        src/mpires.f90:347-454 becomes one RCCL all-gather of the contiguous outvec slab)."""
        if self.world == 1:
            if len(self.regions) == NREG:
                return self.outvec
            # a single rank holding a subset of the regions (development runs: what one rank of an N-GPU job computes): the rows
            # of the absent regions stay zero instead of being read past the end of the resident slab
            if self._region_index is None:
                self._region_index = self.torch.as_tensor(self.regions, dtype=self.torch.long, device=self.outvec.device)
            self.all_out.index_copy_(0, self._region_index, self.outvec[:len(self.regions)])
            return self.all_out
        return gather_outvec_slab(self.outvec, self.regions, self.all_out, self.even_split)

    def handoff_in(self, stream):
        """iogrid(30) (src/ppo_iogrid.f90:497-577) on the device: hybrid grid state -> spectral time level 1, then back to
        the grid for the physical-range guard.  33 fields per launch."""
        sp, S = self.sp, self.state[0]
        handoff_to_fields(self.G, self.fields, stream)
        sp.spec_mixed(self.fields, self.in_scale, out=self.raw_spec, stream=stream)
        sp.spec_post(self.raw_spec, self.in_desc, S, stream=stream)
        if self.leapfrog_steps is None:                 # no window follows: the guard needs its own inverse set
            self.to_grid(stream)
            handoff_check(self.fields_out, self.safe, stream)
        # else: the window's first time step transforms exactly these fields; the guard reads them there (Dynamics.set_range_guard)

    def to_grid(self, stream):
        """uvspec -> grid(.,2) for u,v ; grid(.,1) for t, q, ps  (src/ppo_iogrid.f90:549-561 and 582-593)"""
        self.sp.grid_derived(self.state[0], self.out_desc, out=self.fields_out, stream=stream)

    def handoff_out(self, stream):
        """iogrid(31) (src/ppo_iogrid.f90:579-601): spectral time level 1 -> the SPEEDY forecast grids F"""
        self.to_grid(stream)
        handoff_from_fields(self.fields_out, self.F, stream)

    def speedy_leg(self, stream):
        self.handoff_in(stream)
        if self.phys is not None:
            # fordate(0) of this window's agcm_init (src/ini_agcm_init.f90:86): tcorh, and qcorh from the SST just scattered into G,
            # written straight into the time steps' boundary fields; then its daily sol_oz
            self.phys.fordate(self.sp, self.dyn.boundary_ptr() + NX * MX2 * 8, stream)
            self.update_forcing()
        if self.leapfrog_steps is not None:
            # agcm_init -> stepone, then stloop's first 6-hour window (src/dyn_stloop.f90:24-95 with onehr_hybrid)
            self.dyn.window(self.state, self.leapfrog_steps, start=True, delt=DELT, stream=stream)
        self.handoff_out(stream)

    def tisr_slice(self, timestep):
        """get_tisr_by_date(timestep): the table slice for `timestep` steps after the start of the prediction"""
        return self.tisr[domain.tisr_index(self.start_hours + timestep * self.timestep_hours) - 1]

    def next_tisr(self):
        # sendrecievegrid(res, t, ...) fills the next inputs with get_tisr_by_date(timestep - 1) (src/mpires.f90:750)
        self.t += 1
        self.G[domain.GT_OFF:].copy_(self.tisr_slice(self.t - 1).reshape(-1))

    def step(self, stream):
        """One hybrid step.  `stream` must be the current torch stream (the TISR copy and the RCCL exchange are torch ops).

        Sequential form (pipeline=False), the reference's order of program main's loop body:
            predict -> exchange -> scatter -> SPEEDY leg -> gather(feedback, local_model)
        Pipelined form (pipeline=True; measured +6 % on MI355X, off by default): the reservoir state of step t+1 depends only
        on the feedback, i.e. on the hybrid state G(t),
        which is complete BEFORE the SPEEDY window of step t starts; only the 132 physics-model columns of W_out wait for the
        forecast.  So  advance(t+1) + W_out[:, state columns] x~(t+1)  (99 % of the step's HBM bytes) run on a side stream
        concurrently with the SPEEDY leg of step t (26 time steps of small latency-bound kernels), and the step closes with
        the small model-column block.  Every step still does exactly one advance, one full readout, one scatter, one SPEEDY
        window and one gather; results equal the sequential form up to the association of the readout's column sum."""
        if self.mode == "sweep":
            self.bank.predict(stream=stream)
            return True
        if self.mode == "ml_only":
            self.bank.predict(stream=stream)                     # predict_ml of every resident reservoir
            allv = self.exchange_outvec(stream)
            self.scatter_all(allv, stream)
            self.next_tisr()
            self.ex.gather(self.G, None, stream=stream)          # feedback only: there is no forecast to tile
            return True                                          # (no SPEEDY hand-off, hence no range guard, in the ML-only loop)
        if not self.pipeline:
            if self.stop_on_unsafe and self.aborted():                   # run_speedy == .false. ends the forecast loop
                return False                                             # (src/mpires.f90:744, src/parallelmain.f90:269-271)
            ph = self._phase_events(stream) if self._phase_on else None
            self.bank.predict(stream=stream)
            if self.slab is not None and self.slab.due(self.t + 1):      # mod(t*timestep, timestep_slab) == 0, parallelmain.f90:238
                self.slab_predict_and_share(stream)
            if ph: ph.mark("predict")
            allv = self.exchange_outvec(stream)
            if ph: ph.mark("allgather")
            self.scatter_all(allv, stream)
            if ph: ph.mark("scatter")
            self.speedy_leg(stream)
            if ph: ph.mark("speedy")
            self.next_tisr()
            self.ex.gather(self.G, self.F, stream=stream)
            if self.slab is not None:
                self.slab.update_inputs(self.t, stream=stream)
            if ph: ph.mark("gather")
            self._post_safe(stream)
            return True
        assert self.slab is None, "the pipelined schedule does not carry the slab-ocean coupling"
        if self.stop_on_unsafe and self.aborted():
            return False
        if self.main is not None:                               # CU-partitioned form: the whole main leg runs on the masked stream
            caller = stream
            self.main.wait_stream(caller)
            with self.torch.cuda.stream(self.main):
                self._pipelined_body(self.main)
            caller.wait_stream(self.main)
        else:
            self._pipelined_body(stream)
        self._post_safe(stream)
        return True

    def _pipelined_body(self, stream):
        stream.wait_event(self.ev_partial)                      # state block of this step's readout (side stream)
        self.bank.readout_part(2, stream=stream)                # + physics-model block -> outvec(t)
        allv = self.exchange_outvec(stream)
        self.ex.scatter(allv, self.G, base_sst=self.base_sst, stream=stream)
        self.next_tisr()
        self.ex.gather(self.G, None, stream=stream)             # feedback(t+1)
        self.ev_feedback.record(stream)
        self.side.wait_event(self.ev_feedback)
        self.bank.advance(stream=self.side)                     # x(t+1)
        self.bank.readout_part(1, stream=self.side, persistent=self.persistent_readout)
        self.ev_partial.record(self.side)
        self.speedy_leg(stream)                                 # forecast F(t), concurrent with the side stream
        if self.persistent_readout and self.drain_readout:
            # the SPEEDY window is over: a full-occupancy launch empties the work queue the bounded kernel is still pulling from
            self.bank.readout_part(1, stream=stream, drain=True)
        self.ex.gather(None, self.F, stream=stream)             # local_model(t+1)

    # ------------------------------------------------------------------ measurement helpers
    # ------------------------------------------------------------------ abort propagation (the range guard of iogrid(30))
    # The reference's root sets run_speedy = .false. when SPEEDY's input is unphysical, broadcasts it (src/mpires.f90:744) and every
    # rank leaves the forecast loop (src/parallelmain.f90:269-271).  Here every rank evaluates the guard itself (SPEEDY is
    # replicated and deterministic, so all ranks see the same flag: no broadcast).  The flag lives on the device; each step copies
    # it asynchronously into a pinned ring, and step() polls the COMPLETED copies without blocking -- the host stays ahead of the
    # GPU and the loop stops within the few steps that were already enqueued.
    # Every ring entry carries the index of the step it belongs to, so that a caller can truncate its output at the first
    # unphysical state (first_unsafe_step; the reference stops AT that step, this loop a few enqueued steps later).
    SAFE_RING = 8
    first_unsafe_step = None

    def _note_unsafe(self, k):
        self._aborted = True
        if self.first_unsafe_step is None or self._safe_step[k] < self.first_unsafe_step:
            self.first_unsafe_step = self._safe_step[k]

    def _post_safe(self, stream):
        if getattr(self, "safe", None) is None:
            return
        if self._safe_ring is None:
            self._safe_ring = self.torch.ones(self.SAFE_RING, dtype=self.torch.int32).pin_memory()
            self._safe_ev = [None] * self.SAFE_RING
            self._safe_step = [0] * self.SAFE_RING
        k = self._safe_n % self.SAFE_RING
        if self._safe_ev[k] is not None:
            self._safe_ev[k].synchronize()                               # 8 steps old
            if int(self._safe_ring[k]) == 0:
                self._note_unsafe(k)
        else:
            self._safe_ev[k] = self.torch.cuda.Event()
        self._safe_ring[k:k + 1].copy_(self.safe, non_blocking=True)
        self._safe_step[k] = self.t                                      # (1-based index of the step just enqueued)
        self._safe_ev[k].record(stream)
        self._safe_n += 1

    def aborted(self, wait=False):
        """True once the range guard has tripped in a step whose flag has reached the host (wait=True: in any enqueued step)."""
        if self._aborted or self._safe_ring is None:
            return self._aborted
        for k, ev in enumerate(self._safe_ev):
            if ev is None:
                continue
            if wait:
                ev.synchronize()
            if ev.query() and int(self._safe_ring[k]) == 0:
                self._note_unsafe(k)
        return self._aborted

    # ------------------------------------------------------------------ timing
    class _Phases:
        def __init__(self, torch, stream, sink):
            self.torch, self.stream, self.sink = torch, stream, sink
            self.last = torch.cuda.Event(enable_timing=True)
            self.last.record(stream)

        def mark(self, name):
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record(self.stream)
            self.sink.append((name, self.last, ev))
            self.last = ev

    def _phase_events(self, stream):
        return self._Phases(self.torch, stream, self._phase_log)

    def timing(self, on, phases=True):
        """HIP events around the bank's two predict kernels; phases: also around each phase of the step (five more event records per
        step on the step's stream -- each one a barrier packet the dependent launches queue behind)"""
        from ._lib import check, lib
        check(lib().sml_bank_timing(self.bank._h, (1 if phases else 2) if on else 0))      # without the phases: k_readout's event pair only
        self._phase_on = bool(on) and phases and self.mode == "hybrid" and not self.pipeline

    def timing_collect(self):
        """per-kernel totals of the bank (HIP events around k_update / k_readout) and, for the sequential hybrid step, the time of
        each phase of the step on this rank: predict, all-gather of the outvec slab, scatter + clamps, the SPEEDY leg, gather"""
        import ctypes as C
        from ._lib import check, lib
        um, rm, uc, rc = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        check(lib().sml_bank_timing_collect(self.bank._h, C.byref(um), C.byref(uc), C.byref(rm), C.byref(rc)))
        out = {"update_ms": um.value, "update_launches": uc.value, "readout_ms": rm.value, "readout_launches": rc.value}
        if self._phase_log:
            self.torch.cuda.synchronize()
            tot, cnt = {}, {}
            for name, a, b in self._phase_log:
                tot[name] = tot.get(name, 0.0) + a.elapsed_time(b)
                cnt[name] = cnt.get(name, 0) + 1
            out["phases_ms_per_step"] = {k: tot[k] / cnt[k] for k in tot}
            self._phase_log.clear()
        return out

    def describe(self):
        if self.mode == "sweep":
            wl = "config3 sweep-only: batched predict of the rank's resident reservoirs"
        elif self.mode == "ml_only":
            wl = "ML-only forecast loop: batched predict_ml of the rank's resident reservoirs + region exchange, no SPEEDY window"
        else:
            nst = 0 if self.leapfrog_steps is None else self.leapfrog_steps + 2
            wl = ("BASELINE config 3: 1152-reservoir batched predict + region exchange (scatter, clamps, gather, standardise) "
                  "+ SPEEDY hand-off iogrid(30)/(31) + %sone 6-hour SPEEDY window of %d time steps (stepone + leapfrog: "
                  "%d inverse + 73 forward transforms, grid-point tendencies, %ssemi-implicit spectral step each) on the device"
                  % ("fordate(0) (albedos, tcorh, qcorh from the hybrid SST) + " if self.phys is not None else "", nst, 77 if self.phys is not None else 50,
                     "column physics (convection, condensation, clouds, SW every 3rd step, LW, surface fluxes, vertical diffusion), "
                     if self.phys is not None else "no column physics, "))
        if self.mode == "hybrid" and self.slab is not None:
            wl += ("; + slab-ocean coupling (config 5): SST assembly from the slab reservoirs, 27-step input averaging ring, "
                   "predict_slab_ml of the SST-predicting regions every 28th step")
        return {"workload": wl, "regions_total": NREG, "regions_this_rank": len(self.regions),
                "transforms_per_step": ((99 if self.leapfrog_steps is None else 66) + (150 if self.phys is not None else 123)
                                        * (0 if self.leapfrog_steps is None else self.leapfrog_steps + 2)
                                        + (2 if self.phys is not None else 0)) if self.mode == "hybrid" else 0,      # (+ fordate's two)
                "parallelism": f"regions sharded by processor_decomposition over {self.world} rank(s); "
                               + ("one all-gather of the outvec slab per step" if self.world > 1 else "no collective")}


def make_comm(world, rank):
    """sml_comm for the native engine's rank exchange inside a torch.distributed job: RCCL (`nccl` backend) with rank 0's unique id
    handed round through torch.distributed -- what MPI_Bcast does in an MPI host; under any other backend (several ranks sharing one
    GPU: rehearsal, never numbers) the host-staged shared-memory transport of sml_comm_bootstrap.  Returns a c_void_p (destroy with
    sml_comm_destroy)."""
    import ctypes as C
    import os

    import torch.distributed as dist

    from ._lib import check, lib
    L = lib()
    comm = C.c_void_p()
    if dist.get_backend() == "nccl":
        ident = C.create_string_buffer(128)
        if rank == 0:
            check(L.sml_comm_unique_id(ident))
        box = [ident.raw]
        dist.broadcast_object_list(box, src=0)
        check(L.sml_comm_create(world, rank, box[0], C.byref(comm)))
    else:
        os.environ["SML_COMM_TRANSPORT"] = "shm"
        os.environ.setdefault("SML_COMM_NONCE", os.environ.get("MASTER_PORT", "0"))
        name = ("speedyml_bench_%s" % os.environ.get("MASTER_PORT", "0")).encode()
        check(L.sml_comm_bootstrap(world, rank, name, C.c_uint64(0), C.byref(comm)))
    return comm


class NativeEngine:
    """The native hybrid engine of the C-ABI (sml_hybrid_*, csrc/hybrid.hip) -- the device-resident body of mpires::sendrecievegrid that
    the Fortran drop-in drives (speedy-ml_amd/fortran/mpires.f90) -- over the banks, boundary fields, tables and start state of a
    HybridRank built for the same regions.  The Python object supplies the synthetic inputs; every step from here on is ONE C call,
    sml_hybrid_step: predict of the resident reservoirs (+ predict_slab_ml when due), the rank exchange (sml_comm all-gather when a
    communicator is given), scatter + clamps, iogrid(30), fordate, the 6-hour window, iogrid(31), TISR, gather + standardise."""

    def __init__(self, model, comm=None):
        """comm: an sml_comm (make_comm) -> the engine all-gathers the outvec slabs itself (RCCL through the C-ABI, as under a Fortran
        host); None with model.world > 1 -> the rank exchange is the host's: torch.distributed's all-gather between
        sml_hybrid_step_predict and sml_hybrid_step_finish (the same RCCL, through torch's communicator)."""
        import ctypes as C

        from ._lib import check, dp, ip, lib
        assert model.mode == "hybrid" and not model.pipeline
        assert comm is not None or model.world == 1 or model.slab is None, "the slab coupling across ranks needs the engine's own communicator"
        self.host_collective = comm is None and model.world > 1
        if self.host_collective:
            agree_on_storage([model.bank, model.slab_bank if model.slab is not None else None])      # (the engine does the same over its own communicator)
        self.model, self.L, self.C = model, lib(), C
        L, regions, classes = self.L, model.regions, model.classes
        self._h = h = C.c_void_p()
        ros = np.ascontiguousarray(regions, dtype=np.int32)
        sst = np.array([int(classes[r][1]) for r in regions], dtype=np.int32)
        check(L.sml_hybrid_create(model.bank._h, NREG, ip(ros), len(ros), 1, 1, ip(sst), C.byref(h)))
        check(L.sml_hybrid_set_state(h, dp(model.G.cpu().numpy().copy())))
        check(L.sml_hybrid_set_orography(h, dp(np.ascontiguousarray(synthetic_orography()))))
        check(L.sml_hybrid_set_tisr_table(h, dp(np.ascontiguousarray(model.tisr.cpu().numpy())), model.start_hours, model.timestep_hours))
        if model.phys is not None:
            from .physics import HSG, NSTRAD
            sia = np.asarray(model.sp.table(1)).ravel()
            radang = np.concatenate([-np.arcsin(sia), np.arcsin(sia)[::-1]])
            s = model.surface
            f = lambda k: dp(np.ascontiguousarray(s[k], dtype=np.float64))
            check(L.sml_hybrid_attach_physics(h, dp(HSG), dp(radang), f("fmask"), f("phis0"), f("tland"), f("swav"), f("alb_l"), f("alb_s"),
                                              f("albsfc"), f("snowc"), NSTRAD))
            check(L.sml_hybrid_set_fordate_fields(h, dp(np.ascontiguousarray(1.0 - s["fmask"])), f("alb0"), f("snowd_am"), f("sice_am")))
        if model.slab is not None:
            base = np.ascontiguousarray(model.base_sst.cpu().numpy())
            mask = np.ascontiguousarray(model.sst_mask.cpu().numpy(), dtype=np.int32)
            check(L.sml_hybrid_set_base_sst(h, dp(base), ip(mask)))
            sea_slot = np.array([int(classes[r][1]) for r in regions], dtype=np.int32)
            sea_reg = np.array([int(c[1]) for c in classes], dtype=np.int32)
            check(L.sml_hybrid_attach_slab(h, model.slab_bank._h, ip(sea_slot), ip(sea_reg), 168))
        if comm is not None:
            check(L.sml_hybrid_set_comm(h, comm))
        check(L.sml_hybrid_initial_inputs(h, None))
        self.leapfrog_steps = -1 if model.leapfrog_steps is None else model.leapfrog_steps
        self.stop_on_unsafe = True

    def close(self):
        if self._h:
            self.L.sml_hybrid_destroy(self._h)
            self._h = None

    def step(self, stream):
        from ._lib import check, dp, vp
        if not self.host_collective:
            check(self.L.sml_hybrid_step(self._h, self.leapfrog_steps, vp(stream)))
            return True
        m = self.model
        check(self.L.sml_hybrid_step_predict(self._h, vp(stream)))
        allv = gather_outvec_slab(m.outvec, m.regions, m.all_out, m.even_split)          # `stream` must be torch's current stream
        check(self.L.sml_hybrid_step_finish(self._h, dp(allv.data_ptr()), self.leapfrog_steps, vp(stream)))
        return True

    def safe(self):
        """run_speedy (src/mpires.f90:744); synchronises the device"""
        from ._lib import check
        v = self.C.c_int()
        check(self.L.sml_hybrid_safe(self._h, self.C.byref(v)))
        return v.value == 1

    def aborted(self, wait=False):
        return not self.safe()

    def state(self):
        from ._lib import check, dp
        g, f = np.zeros(domain.G_SIZE), np.zeros(domain.G_SIZE)
        check(self.L.sml_hybrid_get_state(self._h, dp(g), dp(f)))
        return g, f

    def timing(self, on, phases=True):
        from ._lib import check
        check(self.L.sml_bank_timing(self.model.bank._h, (1 if phases else 2) if on else 0))      # without the phases: k_readout's event pair only
        check(self.L.sml_hybrid_timing(self._h, 1 if (on and phases) else 0))

    def timing_collect(self):
        from ._lib import check, dp
        C = self.C
        um, rm, uc, rc = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        check(self.L.sml_bank_timing_collect(self.model.bank._h, C.byref(um), C.byref(uc), C.byref(rm), C.byref(rc)))
        out = {"update_ms": um.value, "update_launches": uc.value, "readout_ms": rm.value, "readout_launches": rc.value}
        ms, n = np.zeros(5), C.c_int()
        check(self.L.sml_hybrid_timing_collect(self._h, dp(ms), C.byref(n)))
        if n.value:
            out["phases_ms_per_step"] = dict(zip(("predict", "allgather", "scatter", "speedy", "gather"), (ms / n.value).tolist()))
        return out

    def describe(self):
        d = self.model.describe()
        d["host"] = "native engine: one sml_hybrid_step call per step (csrc/hybrid.hip), what the Fortran drop-in's sendrecievegrid drives"
        if self.model.world > 1:
            d["parallelism"] = d["parallelism"].replace("one all-gather", "one RCCL all-gather through torch.distributed (between sml_hybrid_step_predict and "
                                                        "sml_hybrid_step_finish)" if self.host_collective else "one sml_comm (RCCL) all-gather inside the engine")
        return d
