"""GPU: the native hybrid engine of the C-ABI (sml_hybrid_*: the device-resident body of mpires::sendrecievegrid for non-Python hosts,
csrc/hybrid.hip) against the Python-driven step of speedy-ml_amd/hybrid.py, which is itself pinned stage by stage to the oracle in
tests/test_hybrid_gpu.py.  Same kernels, same order: the global state, the forecast and every reservoir's next inputs must agree
bit for bit after two steps (full physics, 24-step windows, all 1152 regions)."""
import ctypes as C

import numpy as np
import pytest
import torch

from speedy_ml_amd import _lib, domain, hybrid, synth
from speedy_ml_amd.physics import NSTRAD

pytestmark = pytest.mark.gpu
HSG = np.array([0.000, 0.050, 0.140, 0.260, 0.420, 0.600, 0.770, 0.900, 1.000])


@pytest.mark.parametrize("float32_weights", [False, True])
def test_native_engine_equals_python_step(float32_weights):
    """float32_weights: every weight exactly a float, as after the reference's NetCDF weight files -- both hosts' banks then read their
    compact copies (k_readout32, k_update<float values>), and must still agree bit for bit"""
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(hybrid.NREG))
    ref = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, float32_weights=float32_weights)
    eng = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, float32_weights=float32_weights)      # supplies an identical bank and start state
    assert ref.bank.compact() == float32_weights and eng.bank.compact() == float32_weights
    L, check = _lib.lib(), _lib.check
    h = C.c_void_p()
    ros = np.arange(hybrid.NREG, dtype=np.int32)
    sst = np.array([int(classes[r][1]) for r in regions], dtype=np.int32)
    check(L.sml_hybrid_create(eng.bank._h, hybrid.NREG, _lib.ip(ros), hybrid.NREG, 1, 1, _lib.ip(sst), C.byref(h)))
    g0 = eng.G.cpu().numpy().copy()
    check(L.sml_hybrid_set_state(h, _lib.dp(g0)))
    check(L.sml_hybrid_set_orography(h, _lib.dp(np.ascontiguousarray(synth.synthetic_orography()))))
    check(L.sml_hybrid_set_tisr_table(h, _lib.dp(np.ascontiguousarray(eng.tisr.cpu().numpy())), eng.start_hours, eng.timestep_hours))
    sia = np.asarray(eng.sp.table(1)).ravel()
    radang = np.concatenate([-np.arcsin(sia), np.arcsin(sia)[::-1]])
    s = eng.surface
    f = lambda k: _lib.dp(np.ascontiguousarray(s[k], dtype=np.float64))
    check(L.sml_hybrid_attach_physics(h, _lib.dp(HSG), _lib.dp(radang), f("fmask"), f("phis0"), f("tland"), f("swav"), f("alb_l"), f("alb_s"),
                                      f("albsfc"), f("snowc"), NSTRAD))
    phis0 = np.zeros((48, 96))
    check(L.sml_hybrid_get_phis0(h, _lib.dp(phis0)))
    assert np.array_equal(phis0, s["phis0"])                   # mod_surfcon's truncated orography, the same on both hosts
    # (fordate's albedo inputs are left out here: the engine then keeps the albedos it was given, which are what fordate derives)
    check(L.sml_hybrid_initial_inputs(h, None))
    torch.cuda.synchronize()
    assert torch.equal(eng.feedback, ref.feedback) and torch.equal(eng.local_model, ref.local_model)
    stream = torch.cuda.current_stream()
    for _ in range(2):
        ref.step(stream)
        eng.bank.predict(stream=stream)
        check(L.sml_hybrid_exchange_and_speedy(h, None, hybrid.LEAPFROG_PER_WINDOW, _lib.vp(stream)))
    torch.cuda.synchronize()
    g, fc = np.zeros(domain.G_SIZE), np.zeros(domain.G_SIZE)
    check(L.sml_hybrid_get_state(h, _lib.dp(g), _lib.dp(fc)))
    assert np.array_equal(g, ref.G.cpu().numpy())
    assert np.array_equal(fc[:domain.GP_OFF], ref.F.cpu().numpy()[:domain.GP_OFF])
    assert torch.equal(eng.feedback, ref.feedback) and torch.equal(eng.local_model, ref.local_model) and torch.equal(eng.outvec, ref.outvec)
    safe = C.c_int()
    check(L.sml_hybrid_safe(h, C.byref(safe)))
    assert safe.value == 1 and not np.array_equal(g, g0)
    check(L.sml_hybrid_destroy(h))


def make_engine(model, regions, classes, comm=None):
    """sml_hybrid_* over the banks and the start state of a HybridRank built for the same regions (the Python object only supplies
    identical reservoirs, surface fields and tables; from here on the native engine steps them)."""
    L, check = _lib.lib(), _lib.check
    h = C.c_void_p()
    ros = np.ascontiguousarray(regions, dtype=np.int32)
    sst = np.array([int(classes[r][1]) for r in regions], dtype=np.int32)
    check(L.sml_hybrid_create(model.bank._h, hybrid.NREG, _lib.ip(ros), len(ros), 1, 1, _lib.ip(sst), C.byref(h)))
    check(L.sml_hybrid_set_state(h, _lib.dp(model.G.cpu().numpy().copy())))
    check(L.sml_hybrid_set_orography(h, _lib.dp(np.ascontiguousarray(synth.synthetic_orography()))))
    check(L.sml_hybrid_set_tisr_table(h, _lib.dp(np.ascontiguousarray(model.tisr.cpu().numpy())), model.start_hours, model.timestep_hours))
    sia = np.asarray(model.sp.table(1)).ravel()
    radang = np.concatenate([-np.arcsin(sia), np.arcsin(sia)[::-1]])
    s = model.surface
    f = lambda k: _lib.dp(np.ascontiguousarray(s[k], dtype=np.float64))
    check(L.sml_hybrid_attach_physics(h, _lib.dp(HSG), _lib.dp(radang), f("fmask"), f("phis0"), f("tland"), f("swav"), f("alb_l"), f("alb_s"),
                                      f("albsfc"), f("snowc"), NSTRAD))
    fs = np.ascontiguousarray(1.0 - s["fmask"])
    check(L.sml_hybrid_set_fordate_fields(h, _lib.dp(fs), f("alb0"), f("snowd_am"), f("sice_am")))
    if model.slab is not None:
        base = np.ascontiguousarray(model.base_sst.cpu().numpy())
        mask = np.ascontiguousarray(model.sst_mask.cpu().numpy(), dtype=np.int32)
        check(L.sml_hybrid_set_base_sst(h, _lib.dp(base), _lib.ip(mask)))
        sea_slot = np.array([int(classes[r][1]) for r in regions], dtype=np.int32)
        sea_reg = np.array([int(c[1]) for c in classes], dtype=np.int32)
        check(L.sml_hybrid_attach_slab(h, model.slab_bank._h, _lib.ip(sea_slot), _lib.ip(sea_reg), 168))
    if comm is not None:
        check(L.sml_hybrid_set_comm(h, comm))
    check(L.sml_hybrid_initial_inputs(h, None))
    return h


def test_fused_handoff_equals_separate_launches(monkeypatch):
    """The engine brackets the SPEEDY window with two fused launches (k_ingest: rows + scatter + clamps + SST + real(4) fields + fordate's
    grid-point work; k_egress: F, TISR slice, both gathers) and transforms fordate's two fields with iogrid(30)'s 33; with
    SML_HYBRID_FUSED_HANDOFF=0 it issues the stand-alone kernels instead (the launches the Python host issues).  Two engines, one of each
    kind, over the slab configuration: every state and every reservoir input identical after each of 4 steps."""
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(hybrid.NREG))
    ma = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, slab=True)
    mb = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, slab=True)
    L, check = _lib.lib(), _lib.check
    monkeypatch.delenv("SML_HYBRID_FUSED_HANDOFF", raising=False)
    ha = make_engine(ma, regions, classes)
    monkeypatch.setenv("SML_HYBRID_FUSED_HANDOFF", "0")
    hb = make_engine(mb, regions, classes)
    monkeypatch.delenv("SML_HYBRID_FUSED_HANDOFF")
    stream = torch.cuda.current_stream()
    for t in range(4):
        check(L.sml_hybrid_step(ha, hybrid.LEAPFROG_PER_WINDOW, _lib.vp(stream)))
        check(L.sml_hybrid_step(hb, hybrid.LEAPFROG_PER_WINDOW, _lib.vp(stream)))
        torch.cuda.synchronize()
        ga, fa, gb, fb = (np.zeros(domain.G_SIZE) for _ in range(4))
        check(L.sml_hybrid_get_state(ha, _lib.dp(ga), _lib.dp(fa)))
        check(L.sml_hybrid_get_state(hb, _lib.dp(gb), _lib.dp(fb)))
        assert np.array_equal(ga, gb) and np.array_equal(fa, fb), t
        assert torch.equal(ma.feedback, mb.feedback) and torch.equal(ma.local_model, mb.local_model) and torch.equal(ma.outvec, mb.outvec), t
        assert torch.equal(ma.slab_feedback, mb.slab_feedback), t
    for h in (ha, hb):
        safe = C.c_int()
        check(L.sml_hybrid_safe(h, C.byref(safe)))
        assert safe.value == 1
        check(L.sml_hybrid_destroy(h))


def test_native_engine_with_slab_equals_python_over_30_steps():
    """config 5 in the native engine: sml_hybrid_attach_slab + sml_hybrid_step (predict, predict_slab_ml on the 28th step, SST
    assembly, mask / floor, the averaging ring of the slab inputs) against HybridRank(slab=True), bit for bit, over 30 steps -- the slab
    reservoirs fire once (step 28) and their SST reaches the atmosphere reservoirs' inputs and SPEEDY's sea temperature after it."""
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(hybrid.NREG))
    ref = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, slab=True)
    eng = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1, slab=True)
    L, check = _lib.lib(), _lib.check
    h = make_engine(eng, regions, classes)
    torch.cuda.synchronize()
    assert torch.equal(eng.feedback, ref.feedback) and torch.equal(eng.local_model, ref.local_model)
    stream = torch.cuda.current_stream()
    fired = 0
    sst_before = ref.G[domain.GS_OFF:domain.GT_OFF].clone()
    for t in range(30):
        fired += L.sml_hybrid_slab_due(h)
        assert ref.step(stream) is True
        check(L.sml_hybrid_step(h, hybrid.LEAPFROG_PER_WINDOW, _lib.vp(stream)))
        if t in (0, 26, 27, 28, 29):
            torch.cuda.synchronize()
            g, fc = np.zeros(domain.G_SIZE), np.zeros(domain.G_SIZE)
            check(L.sml_hybrid_get_state(h, _lib.dp(g), _lib.dp(fc)))
            assert np.array_equal(g, ref.G.cpu().numpy()), t
            assert np.array_equal(fc[:domain.GP_OFF], ref.F.cpu().numpy()[:domain.GP_OFF]), t
            assert torch.equal(eng.feedback, ref.feedback) and torch.equal(eng.local_model, ref.local_model) and torch.equal(eng.outvec, ref.outvec), t
            assert torch.equal(eng.slab_feedback, ref.slab_feedback) and torch.equal(eng.slab_outvec, ref.slab_outvec), t
    assert fired == 1
    assert not torch.equal(ref.G[domain.GS_OFF:domain.GT_OFF], sst_before), "the slab reservoirs' SST never reached the hybrid state"
    safe = C.c_int()
    check(L.sml_hybrid_safe(h, C.byref(safe)))
    assert safe.value == 1
    check(L.sml_hybrid_destroy(h))


def test_engine_restart_starts_a_new_forecast():
    """sml_hybrid_restart (program main's prediction_num loop, src/parallelmain.f90:206): the step counter -- hence the TISR slice
    get_tisr_by_date picks -- and the range guard start over.  (The column physics' carried fields -- the short-wave heating of the last
    short-wave step and the like -- live on across forecasts, as the reference's module variables do inside one process, so a repeated
    forecast is not expected to repeat bit for bit.)"""
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(hybrid.NREG))
    eng = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1)
    L, check = _lib.lib(), _lib.check
    h = make_engine(eng, regions, classes)
    g0 = eng.G.cpu().numpy().copy()
    stream = torch.cuda.current_stream()
    table = eng.tisr.cpu().numpy().reshape(8760, -1)
    g, safe = np.zeros(domain.G_SIZE), C.c_int()
    for _ in range(3):
        check(L.sml_hybrid_step(h, 2, _lib.vp(stream)))
    check(L.sml_hybrid_get_state(h, _lib.dp(g), None))
    assert np.array_equal(g[domain.GT_OFF:], table[domain.tisr_index(eng.start_hours + 2 * 6) - 1])          # get_tisr_by_date(timestep - 1), step 3
    # an unphysical state trips iogrid(30)'s guard ...
    bad = g0.copy()
    bad[domain.G4_OFF:domain.G4_OFF + 4 * 96] = 1000.0
    check(L.sml_hybrid_set_state(h, _lib.dp(bad)))
    eng.outvec.fill_(1.0e4)
    check(L.sml_hybrid_exchange_and_speedy(h, None, 2, _lib.vp(stream)))
    check(L.sml_hybrid_safe(h, C.byref(safe)))
    assert safe.value == 0
    # ... and a new forecast, 336 hours later in the calendar, starts clean
    later = eng.start_hours + 336
    check(L.sml_hybrid_restart(h, later))
    check(L.sml_hybrid_set_state(h, _lib.dp(g0)))
    check(L.sml_hybrid_initial_inputs(h, None))
    check(L.sml_hybrid_step(h, 2, _lib.vp(stream)))
    check(L.sml_hybrid_safe(h, C.byref(safe)))
    assert safe.value == 1
    check(L.sml_hybrid_get_state(h, _lib.dp(g), None))
    assert np.array_equal(g[domain.GT_OFF:], table[domain.tisr_index(later) - 1])
    assert not np.array_equal(table[domain.tisr_index(later) - 1], table[domain.tisr_index(eng.start_hours + 2 * 6) - 1])
    check(L.sml_hybrid_destroy(h))
