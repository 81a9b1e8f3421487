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


def test_native_engine_equals_python_step():
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = list(range(hybrid.NREG))
    ref = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1)
    eng = hybrid.HybridRank(regions, classes, sea_mask=sea, mode="hybrid", n_override=1)      # supplies an identical bank and start state
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
