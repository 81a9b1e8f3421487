"""GPU: the device-resident hybrid step (predict -> scatter+clamps -> iogrid(30) -> SPEEDY 6-hour window -> iogrid(31) ->
gather+standardise) against the CPU oracle, stage by stage with identical inputs (the hybrid trajectory is chaotic,
parity is per step).

All 1152 regions take part (full index maps, polar / periodic / land classes); the reservoirs are small (n = d) so
that the oracle's 1152 predicts stay cheap -- full-size reservoirs are covered by tests/test_reservoir_gpu.py."""
import numpy as np
import pytest
import torch

from speedy_ml_amd import domain, hybrid, synth

pytestmark = pytest.mark.gpu
NREG = 1152


@pytest.fixture(scope="module", params=[False, True], ids=["adiabatic", "physics"])
def model(request):
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    m = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="hybrid", n_override=1, pipeline=False, physics=request.param)
    m.classes_ = classes
    return m


def oracle_speedy_leg(o, m, G, phys=None):
    """iogrid(30), stepone + 24 leapfrog steps, iogrid(31) with the oracle, from the device's hybrid state G.  With the column
    physics attached, grtend's physics slot is filled with the compiled reference parametrisations (oracle/_ref/libref_phy.so);
    `phys` continues a previous window's hook (short-wave flag and the short-wave scheme's leftovers carry over, as the reference's
    module variables do).  The diffusion corrections tcorh / qcorh come from the reference's own fordate(0) (src/ini_fordate.f90,
    compiled in place into the same library) run on the surface fields and the SST of G, as every window's agcm_init does; the
    adiabatic core has no surface temperatures: tcorh from the orography, qcorh zero.  Returns (F4, F2, phys)."""
    from _oracle import DynOracle, PhysHook, RefPhys, oracle_iogrid30, oracle_iogrid31, oracle_window
    if m.phys is not None:
        if not RefPhys.available():
            pytest.skip("oracle/_ref/libref_phy.so not present")
        tyear = (m.phys_day - 0.5) / 365.0
        if phys is None:
            from speedy_ml_amd.physics import HSG
            sia = np.asarray(m.sp.table(1)).ravel()
            surf = dict(m.surface, tsea=G[domain.GS_OFF:domain.GT_OFF])
            phys = PhysHook(RefPhys(HSG, np.concatenate([-np.arcsin(sia), np.arcsin(sia)[::-1]])), surf, tyear)
        else:
            phys.s["tsea"] = np.ascontiguousarray(G[domain.GS_OFF:domain.GT_OFF], dtype=np.float64)
        s = m.surface
        fd = phys.ref.fordate(tyear, s["phis0"], s["fmask"], 1.0 - s["fmask"], s["tland"], G[domain.GS_OFF:domain.GT_OFF], s["alb0"], s["snowd_am"],
                              s["sice_am"])      # (includes fordate's sol_oz(tyear) and sflset; leaves the albedos it computes in the physics' modules)
        for k in ("alb_l", "alb_s", "albsfc", "snowc"):
            assert np.array_equal(fd[k], np.asarray(s[k]).ravel()) and np.array_equal(m.phys.surface(k).ravel(), fd[k]), k
        tcorh, qcorh = fd["tcorh"], fd["qcorh"]
        assert np.max(np.abs(qcorh)) > 1e-3
    else:
        tcorh, qcorh = m.tcorh.cpu().numpy().T, np.zeros((62, 32))
    g4 = G[:domain.G2_OFF].reshape(8, 48, 96, 4)
    logp = G[domain.G2_OFF:domain.GP_OFF].reshape(48, 96)
    lvl = oracle_iogrid30(o, g4, logp)
    sp2 = lambda t: t.cpu().numpy().T
    bc = m.dyn.boundary()
    assert np.max(np.abs(bc[1].T - tcorh)) <= 1e-12 * np.max(np.abs(tcorh)) and np.max(np.abs(bc[2].T - qcorh)) <= 1e-12 * max(np.max(np.abs(qcorh)), 1e-3)
    cur = oracle_window(DynOracle(o), lvl, sp2(m.phis), tcorh, qcorh, m.leapfrog_steps, phys=phys)
    F4, F2 = oracle_iogrid31(o, {k: cur[k][..., 0] for k in cur})
    qv = F4[..., 3]
    qv[qv < 0.000001] = 0.000001
    return F4, F2, phys


def test_hybrid_step_stage_parity(model, oracle):
    m, o = model, oracle
    torch.cuda.synchronize()
    fb0 = m.feedback.cpu().numpy().copy()
    lm0 = m.local_model.cpu().numpy().copy()
    x0 = [m.bank.get_state(s) for s in range(NREG)]
    stream = torch.cuda.current_stream()
    m.step(stream)
    torch.cuda.synchronize()
    G = m.G.cpu().numpy()
    F = m.F.cpu().numpy()

    # ---- stage A: predict + un-standardise + scatter + clamps ----
    g4, g2, gp = np.zeros(147456), np.zeros(4608), np.zeros(4608)
    for s in range(NREG):
        b, mean, std, stat = m.bank.host_copies[s]
        win = np.zeros((b.n, b.d), order="F")
        win[np.arange(b.n), b.win_cols - 1] = b.win_vals
        xw, out = o.predict_raw(b.n, b.d, b.n_model, b.n_out, b.rows, b.cols, b.vals, win, b.wout, 1.0,
                                fb0[s, :b.d].copy(), lm0[s, :b.n_model].copy(), x0[s])
        g = o.initializedomain(NREG, s)
        out = o.unstandardize_res(g, mean, std, out)
        o.scatter_res(NREG, s, out, g4, g2, gp)
        if s % 97 == 0:
            assert np.max(np.abs(m.bank.get_state(s) - xw)) <= 1e-13
    G4 = g4.reshape(8, 48, 96, 4)
    G4[..., 3][G4[..., 3] < 0.000001] = 0.000001
    gp[gp < 0.00001] = 0.0
    sst = np.maximum(m.base_sst.cpu().numpy(), 272.0)
    want = np.concatenate([g4, g2, gp, sst])
    got = G[:domain.GT_OFF]
    scale = np.maximum(np.abs(want), 1.0)
    assert np.max(np.abs(got - want) / scale) <= 1e-11

    # ---- stage B: the SPEEDY leg (hand-off in, 26 adiabatic time steps, hand-off out), oracle fed with the device's G.
    # north_star tolerance: 1e-10 relative per field after the 6-hour window
    F4w, F2w, hook = oracle_speedy_leg(o, m, G)
    F4g = F[:domain.G2_OFF].reshape(8, 48, 96, 4)
    for var in range(4):
        sc = np.max(np.abs(F4w[..., var]))
        assert np.max(np.abs(F4g[..., var] - F4w[..., var])) <= 1e-10 * sc, (var, np.max(np.abs(F4g[..., var] - F4w[..., var])) / sc)
    assert np.max(np.abs(F[domain.G2_OFF:domain.GP_OFF].reshape(48, 96) - F2w)) <= 1e-10 * np.max(np.abs(F2w))
    # the synthetic climate is physical: the range guard of iogrid(30) must not trip, and the forecast differs from the input
    assert int(m.safe.item()) == 1
    assert np.max(np.abs(F4g[..., 1] - G[:domain.G2_OFF].reshape(8, 48, 96, 4)[..., 1])) > 0.5

    m._hook_after_first_window = hook
    # ---- stage C: next inputs, oracle tilers fed with the device's G and F: bit-exact ----
    fb1 = m.feedback.cpu().numpy()
    lm1 = m.local_model.cpu().numpy()
    Gg4 = np.ascontiguousarray(G[:domain.G2_OFF])
    Gg2 = np.ascontiguousarray(G[domain.G2_OFF:domain.GP_OFF])
    Ggp = np.ascontiguousarray(G[domain.GP_OFF:domain.GS_OFF])
    Ggs = np.ascontiguousarray(G[domain.GS_OFF:domain.GT_OFF])
    Ggt = np.ascontiguousarray(G[domain.GT_OFF:])
    assert np.array_equal(Ggt, m.tisr_slice(m.t - 1).cpu().numpy().ravel())
    assert m.t == 1 and domain.tisr_index(m.start_hours) == oracle.tisr_index(m.start_hours)
    Ff4 = np.ascontiguousarray(F[:domain.G2_OFF])
    Ff2 = np.ascontiguousarray(F[domain.G2_OFF:domain.GP_OFF])
    for s in range(NREG):
        b, mean, std, stat = m.bank.host_copies[s]
        sst_in = m.classes_[s][1]
        g = o.initializedomain(NREG, s)
        sz = o.allocate_sizes(g, sst_input=int(sst_in))
        u = np.zeros(sz.reservoir_numinputs)
        u[:sz.precip_end] = o.tile_input(NREG, s, Gg4, Gg2, Ggp, sz.precip_end)
        in2d = g.inputxchunk * g.inputychunk
        u = o.standardize_input(g, sz, mean, std, u)
        u[sz.precip_start - 1:sz.precip_end] = (u[sz.precip_start - 1:sz.precip_end] - mean[34]) / std[34]
        if sst_in:
            u[sz.sst_start - 1:sz.sst_end] = (o.tile_input2d(NREG, s, Ggs, in2d) - mean[35]) / std[35]
        u[sz.tisr_start - 1:sz.tisr_end] = (o.tile_input2d(NREG, s, Ggt, in2d) - mean[33]) / std[33]
        assert np.array_equal(fb1[s, :b.d], u), s
        lm = o.standardize_res(g, mean, std, o.tile_res(NREG, s, Ff4, Ff2, 132))
        assert np.array_equal(lm1[s, :132], lm), s


def test_second_window_carries_the_shortwave_state(model, oracle):
    """The second hybrid step's SPEEDY leg: stepone now runs with the short-wave flag as the first window's last leapfrog step left
    it (.false.: mod(24,3) /= 1) and with that window's last short-wave results (transmissivities, heating, surface flux) -- the
    reference's module variables -- on both sides.  Must follow test_hybrid_step_stage_parity (module-scoped model)."""
    m, o = model, oracle
    if m.t != 1 or not hasattr(m, "_hook_after_first_window"):
        pytest.skip("needs the state left by test_hybrid_step_stage_parity")
    m.step(torch.cuda.current_stream())
    torch.cuda.synchronize()
    G, F = m.G.cpu().numpy(), m.F.cpu().numpy()
    F4w, F2w, _ = oracle_speedy_leg(o, m, G, phys=m._hook_after_first_window)
    F4g = F[:domain.G2_OFF].reshape(8, 48, 96, 4)
    for var in range(4):
        sc = np.max(np.abs(F4w[..., var]))
        assert np.max(np.abs(F4g[..., var] - F4w[..., var])) <= 1e-10 * sc, (var, np.max(np.abs(F4g[..., var] - F4w[..., var])) / sc)
    assert np.max(np.abs(F[domain.G2_OFF:domain.GP_OFF].reshape(48, 96) - F2w)) <= 1e-10 * np.max(np.abs(F2w))


def test_safety_guard_trips(model):
    """Abort path: |u|<=150, |v|<=120, 160<=T<=330, -6<=q<=30 (src/ppo_iogrid.f90:563-577)."""
    from speedy_ml_amd.exchange import handoff_check
    ok_fields = torch.zeros((33, 48, 96), dtype=torch.float64, device="cuda")
    ok_fields[0:8] = 280.0
    ok_fields[8:24] = 10.0
    ok_fields[24:32] = 5.0
    for (f, val, expect) in ((None, None, 1), (3, 400.0, 0), (3, 150.0, 0), (9, -151.0, 0), (17, 120.5, 0), (25, 31.0, 0),
                             (25, -5.9, 1), (9, 150.0, 1), (32, 1e9, 1), (5, float("nan"), 0)):
        fields = ok_fields.clone()
        if f is not None:
            fields[f, 10, 10] = val
        safe = torch.ones(1, dtype=torch.int32, device="cuda")
        handoff_check(fields, safe)
        assert int(safe.item()) == expect, (f, val)


@pytest.mark.parametrize("physics", [True, False])
def test_window_range_guard_trips_on_unphysical_wind(physics):
    """The guard of iogrid(30) inside the window (src/ppo_iogrid.f90:563-577 on the grids of the truncated state): with the physics attached
    the first time step's grid-point kernel checks the T, q, u, v it has just loaded, without it a launch of its own behind that step does;
    a physical state passes both, a jet of several hundred m/s trips both."""
    sea = synth.land_mask()
    m = hybrid.HybridRank(list(range(hybrid.NREG)), hybrid.region_classes(sea), sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2, physics=physics)
    stream = torch.cuda.current_stream()
    assert m.step(stream) is True
    torch.cuda.synchronize()
    assert int(m.safe.item()) == 1
    m.dyn.window(m.state, 1, start=True, stream=stream)
    torch.cuda.synchronize()
    assert int(m.safe.item()) == 1
    m.state[0, 2] *= 40.0                                          # vorticity of the third level, time level 1
    m.dyn.window(m.state, 1, start=True, stream=stream)
    torch.cuda.synchronize()
    assert int(m.safe.item()) == 0


def test_hybrid_closed_loop_stays_physical(model):
    """Two days of closed-loop hybrid steps: states stay finite and inside iogrid(30)'s physical-range guard."""
    m = model
    stream = torch.cuda.current_stream()
    for _ in range(8):
        m.step(stream)
    torch.cuda.synchronize()
    assert torch.isfinite(m.G).all() and torch.isfinite(m.F[:domain.GP_OFF]).all()
    assert torch.isfinite(m.feedback).all() and torch.isfinite(m.outvec).all()
    assert int(m.safe.item()) == 1
    T = m.F[:domain.G2_OFF].reshape(8, 48, 96, 4)[..., 0]
    assert 180.0 < float(T.min()) and float(T.max()) < 320.0


def test_unsafe_state_stops_the_forecast_loop():
    """Abort propagation (src/mpires.f90:744 broadcast of run_speedy, src/parallelmain.f90:269-271 exit): once the range guard of
    iogrid(30) has tripped, step() stops stepping -- without a host synchronisation per step, so within the few steps already
    enqueued -- and says so."""
    sea = synth.land_mask()
    m = hybrid.HybridRank(list(range(hybrid.NREG)), hybrid.region_classes(sea), sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2)
    stream = torch.cuda.current_stream()
    assert m.step(stream) is True and not m.aborted(wait=True)
    m.safe.zero_()                                                # what the guard does on an unphysical state (tested above); it never sets it back
    done = 0
    for _ in range(3 * m.SAFE_RING):
        if not m.step(stream):
            break
        done += 1
    assert m.aborted(wait=True)
    assert done <= m.SAFE_RING + 1, done                          # stopped within the ring's lag
    t_before = m.t
    assert m.step(stream) is False and m.t == t_before            # and stays stopped


def test_pipelined_step_equals_sequential_step():
    """The software-pipelined schedule (advance + state block of the readout on a side stream under the SPEEDY window) must
    reproduce the sequential schedule: same G, F, feedback, local_model and reservoir states after several steps.  The only
    arithmetic difference is the association of the readout's column sum (two partial sums instead of one).

    Run with the adiabatic window: the column physics has discrete switches (convection top, cloud top, stability classes), so
    a 1e-13 re-association difference can flip one column and show up as 1e-3 K a few steps later -- a property of the
    parametrisations, not of the schedule under test."""
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(NREG))
    seq = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, pipeline=False, leapfrog_steps=4, physics=False)
    pip = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, pipeline=True, leapfrog_steps=4, physics=False)
    stream = torch.cuda.current_stream()
    for _ in range(4):
        seq.step(stream)
        pip.step(stream)
    torch.cuda.synchronize()
    for name in ("G", "F", "feedback", "local_model", "outvec"):
        a, b = getattr(seq, name), getattr(pip, name)
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-11 * scale, name
    # the pipelined bank is one advance ahead (its next step's state block is already in flight)
    seq.bank.advance(stream=stream)
    torch.cuda.synchronize()
    for s in (0, 17, 1151):
        assert np.max(np.abs(seq.bank.get_state(s) - pip.bank.get_state(s))) <= 1e-11


def test_pipelined_step_with_physics_teacher_forced():
    """The pipelined schedule WITH the column physics, pinned per window: the two schedules differ only in the association of the
    readout's column sum (1e-13 in the outvec, hence in the state handed to SPEEDY), which a discrete switch of the parametrisations
    can turn into 1e-3 K a few steps later.  So each step the pipelined run's hybrid state is (a) compared with the sequential run's
    at the moment it is handed to SPEEDY (1e-11) and then (b) set to it: from identical inputs the window with physics -- the same
    kernels on the same stream order -- must give the same bits, and so must the next local_model gathered from its forecast."""
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(NREG))
    seq = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, pipeline=False, leapfrog_steps=6, physics=True)
    pip = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, pipeline=True, leapfrog_steps=6, physics=True)
    stream = torch.cuda.current_stream()
    handed = []
    seq_leg, pip_leg = seq.speedy_leg, pip.speedy_leg

    def seq_hook(st):
        handed.append(seq.G.clone())
        seq_leg(st)

    def pip_hook(st):
        want = handed[-1]
        part = slice(0, domain.GT_OFF)                          # (the TISR slice is advanced after the window in the sequential order)
        scale = float(want[part].abs().max())
        assert float((pip.G[part] - want[part]).abs().max()) <= 1e-11 * scale
        pip.G[part].copy_(want[part])
        pip_leg(st)

    seq.speedy_leg, pip.speedy_leg = seq_hook, pip_hook
    for _ in range(3):
        seq.step(stream)
        pip.step(stream)
        torch.cuda.synchronize()
        assert torch.equal(seq.F[:domain.GP_OFF], pip.F[:domain.GP_OFF])
        assert torch.equal(seq.local_model, pip.local_model)
    assert int(seq.safe.item()) == 1 and int(pip.safe.item()) == 1


def test_single_gather_launch_equals_the_two_halves(model):
    """sml_exchange_gather with both sources (one launch) against the feedback-only and the local_model-only calls"""
    m = model
    rng = np.random.default_rng(8)
    G = torch.from_numpy(rng.standard_normal(domain.G_SIZE)).cuda()
    F = torch.from_numpy(rng.standard_normal(domain.G_SIZE)).cuda()
    m.ex.gather(G, F)
    torch.cuda.synchronize()
    fb, lm = m.feedback.clone(), m.local_model.clone()
    m.feedback.zero_(); m.local_model.zero_()
    m.ex.gather(G, None)
    m.ex.gather(None, F)
    torch.cuda.synchronize()
    assert torch.equal(fb, m.feedback) and torch.equal(lm, m.local_model)
    assert float(fb.abs().max()) > 0 and float(lm.abs().max()) > 0


def test_cu_masked_stream_runs_the_bank():
    """sml_stream_create_cu_mask: a stream restricted to the first 64 compute units computes the same predict as the default stream"""
    import ctypes as C
    from speedy_ml_amd import _lib
    from speedy_ml_amd.reservoir import ReservoirBank
    r = synth.make_reservoir(n=640, d=64, n_model=12, n_out=16, seed=5)
    bank = ReservoirBank(1, max_d=64, max_n_model=12, max_n_out=16)
    bank.load(0, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)
    x0 = np.random.default_rng(1).standard_normal(r.n) * 0.2
    words = (C.c_uint32 * 8)(0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    _lib.check(_lib.lib().sml_stream_create_cu_mask(words, 8, C.byref(h)))
    try:
        outs = []
        for stream in (None, torch.cuda.ExternalStream(h.value)):
            bank.set_state(0, x0); bank.set_feedback(0, r.feedback); bank.set_local_model(0, r.local_model)
            bank.predict(stream=stream)
            torch.cuda.synchronize()
            outs.append((bank.get_state(0).copy(), bank.get_outvec(0).copy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    finally:
        _lib.check(_lib.lib().sml_stream_destroy(h))


def test_ml_only_step_stage_parity(oracle):
    """The reference's ml_only loop (src/parallelmain.f90:229-231, src/mpires.f90:566,588): predict_ml for every region, the same
    scatter + clamps, no SPEEDY window, feedback gathered from the assembled state.  Two steps against the oracle."""
    o = oracle
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    m = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="ml_only", n_override=1, physical=False)
    stream = torch.cuda.current_stream()
    for step in range(2):
        torch.cuda.synchronize()
        fb0 = m.feedback.cpu().numpy().copy()
        x0 = [m.bank.get_state(s) for s in range(NREG)]
        m.step(stream)
        torch.cuda.synchronize()
        G = m.G.cpu().numpy()
        g4, g2, gp = np.zeros(147456), np.zeros(4608), np.zeros(4608)
        for s in range(NREG):
            b, mean, std, stat = m.bank.host_copies[s]
            assert b.n_model == 0
            win = np.zeros((b.n, b.d), order="F")
            win[np.arange(b.n), b.win_cols - 1] = b.win_vals
            xw, out = o.predict_raw(b.n, b.d, 0, b.n_out, b.rows, b.cols, b.vals, win, b.wout, 1.0, fb0[s, :b.d].copy(), None, x0[s])
            g = o.initializedomain(NREG, s)
            out = o.unstandardize_res(g, mean, std, out)
            o.scatter_res(NREG, s, out, g4, g2, gp)
            if s % 131 == 0:
                assert np.max(np.abs(m.bank.get_state(s) - xw)) <= 1e-13
        G4 = g4.reshape(8, 48, 96, 4)
        G4[..., 3][G4[..., 3] < 0.000001] = 0.000001
        gp[gp < 0.00001] = 0.0
        sst = np.maximum(m.base_sst.cpu().numpy(), 272.0)
        want = np.concatenate([g4, g2, gp, sst])
        got = G[:domain.GT_OFF]
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= 1e-11, step
        # next feedback: oracle tilers on the device's G, bit-exact
        fb1 = m.feedback.cpu().numpy()
        Gg4, Gg2 = np.ascontiguousarray(G[:domain.G2_OFF]), np.ascontiguousarray(G[domain.G2_OFF:domain.GP_OFF])
        Ggp, Ggs = np.ascontiguousarray(G[domain.GP_OFF:domain.GS_OFF]), np.ascontiguousarray(G[domain.GS_OFF:domain.GT_OFF])
        Ggt = np.ascontiguousarray(G[domain.GT_OFF:])
        for s in range(0, NREG, 7):
            b, mean, std, stat = m.bank.host_copies[s]
            sst_in = classes[s][1]
            g = o.initializedomain(NREG, s)
            sz = o.allocate_sizes(g, sst_input=int(sst_in))
            u = np.zeros(sz.reservoir_numinputs)
            u[:sz.precip_end] = o.tile_input(NREG, s, Gg4, Gg2, Ggp, sz.precip_end)
            in2d = g.inputxchunk * g.inputychunk
            u = o.standardize_input(g, sz, mean, std, u)
            u[sz.precip_start - 1:sz.precip_end] = (u[sz.precip_start - 1:sz.precip_end] - mean[34]) / std[34]
            if sst_in:
                u[sz.sst_start - 1:sz.sst_end] = (o.tile_input2d(NREG, s, Ggs, in2d) - mean[35]) / std[35]
            u[sz.tisr_start - 1:sz.tisr_end] = (o.tile_input2d(NREG, s, Ggt, in2d) - mean[33]) / std[33]
            assert np.array_equal(fb1[s, :b.d], u), (step, s)


def test_full_size_bank_first_step_sampled_against_oracle():
    """BASELINE config 3 at full size: all 1152 reservoirs with N_res ~ 6000 in one bank (the bench's bank), one hybrid step.
    Sampled slots of every size class (poles, interior sea and land, the wrap-around columns, first and last region) against the
    oracle's predict at full size: new reservoir state, un-standardised outvec, and that region's footprint in the scattered
    hybrid state G."""
    from _oracle import Oracle
    o = Oracle()
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    m = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="hybrid", n_override=None, pipeline=False, physics=True)
    torch.cuda.synchronize()
    seen, sample = set(), [0, NREG - 1]
    for r in range(NREG):                               # one region per (n, d) class, plus a region at each end of a latitude row
        b = m.bank.host_copies[r][0]
        if (b.n, b.d) not in seen:
            seen.add((b.n, b.d))
            sample.append(r)
    sample += [47, 48, 95, 576, 577, 1103]
    sample = sorted(set(sample))
    assert len(seen) >= 3
    fb0 = m.feedback.cpu().numpy().copy()
    lm0 = m.local_model.cpu().numpy().copy()
    x0 = {s: m.bank.get_state(s) for s in sample}
    m.step(torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert int(m.safe.item()) == 1
    for s in sample:
        b, mean, std, stat = m.bank.host_copies[s]
        assert b.n >= 5760
        win = np.zeros((b.n, b.d), order="F")
        win[np.arange(b.n), b.win_cols - 1] = b.win_vals
        xw, out = o.predict_raw(b.n, b.d, b.n_model, b.n_out, b.rows, b.cols, b.vals, win, b.wout, 1.0,
                                fb0[s, :b.d].copy(), lm0[s, :b.n_model].copy(), x0[s])
        g = o.initializedomain(NREG, s)
        out = o.unstandardize_res(g, mean, std, out)
        assert np.max(np.abs(m.bank.get_state(s) - xw)) <= 1e-13, s
        got = m.bank.get_outvec(s)[:b.n_out]
        assert np.max(np.abs(got - out) / np.maximum(np.abs(out), 1.0)) <= 1e-11, s
