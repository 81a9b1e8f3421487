"""GPU: the slab-ocean coupling (BASELINE config 5) of the device-resident hybrid step against the oracle, step by step:
SST assembly (slab outputs / 272 K / mask / floor), the 27-column input averaging ring, and predict_slab_ml every 28th step.
Integer maps and the ring arithmetic are bit-exact; the slab prediction follows the reservoir tolerances."""
import numpy as np
import pytest
import torch

from speedy_ml_amd import domain, hybrid, synth
from speedy_ml_amd.slab import slab_sizes

pytestmark = pytest.mark.gpu
NREG = 1152


def test_slab_coupling_matches_oracle(oracle):
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    m = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=1, slab=True)
    o = oracle
    sea_reg = np.array([int(c[1]) for c in classes], dtype=np.int32)
    assert 200 < sea_reg.sum() < 1100
    res_cell = np.zeros((NREG, 4), dtype=np.int32)
    for r in range(NREG):
        gmap, _ = domain.out_map(NREG, r)
        res_cell[r] = np.asarray(gmap[128:132]) - domain.G2_OFF
    assert sorted(res_cell.ravel().tolist()) == list(range(4608))          # the res patches tile the globe
    base = m.base_sst.cpu().numpy()
    mask = m.sst_mask.cpu().numpy()
    # per-slot atmo_training_data_idx (src/mod_slab_ocean_reservoir.f90:364-378) from the atmosphere reservoir's segment offsets
    idx = {}
    for s in range(NREG):
        if not sea_reg[s]:
            continue
        g = domain.initializedomain(NREG, s)
        a = domain.allocate_res_sizes(g, sst_bool_input=True)
        in2d = g.inputxchunk * g.inputychunk
        idx[s] = np.array(list(range(a.atmo3d_end - 4 * in2d, a.logp_end)) + list(range(a.sst_start - 1, a.sst_end))
                          + list(range(a.tisr_start - 1, a.tisr_end)), dtype=np.int32)
        sl = slab_sizes(g)
        assert len(idx[s]) + in2d == sl.reservoir_numinputs
    rings = {s: np.zeros((27, len(idx[s]))) for s in idx}
    slab_fb = m.slab_feedback.cpu().numpy().copy()
    stream = torch.cuda.current_stream()
    probe = [s for s in idx][::37]
    for step in range(1, 31):
        x_before = {s: m.slab_bank.get_state(s) for s in probe} if step == 28 else None
        fb_before = m.slab_feedback.cpu().numpy().copy() if step == 28 else None
        m.step(stream)
        torch.cuda.synchronize()
        assert m.t == step
        G = m.G.cpu().numpy()
        # SST grid: assembled from the slab outputs that were current during this step
        want = o.slab_sst(base, mask, sea_reg, res_cell.ravel(), m.all_slab_out.cpu().numpy())
        assert np.array_equal(G[domain.GS_OFF:domain.GT_OFF], want), step
        # input averaging ring: bit-exact
        fa = m.feedback.cpu().numpy()
        got = m.slab_feedback.cpu().numpy()
        for s in idx:
            o.slab_ring_update(step, idx[s], fa[s], rings[s], slab_fb[s])
            assert np.array_equal(got[s, :m.slab_bank.shapes[s][1]], slab_fb[s, :m.slab_bank.shapes[s][1]]), (step, s)
        if step == 28:
            # predict_slab_ml ran in this step, BEFORE the exchange: state advanced with the averaged inputs of step 27
            for s in probe:
                b_n, b_d, _, b_out = m.slab_bank.shapes[s]
                assert not np.array_equal(m.slab_bank.get_state(s), x_before[s])
                outv = m.all_slab_out[s].cpu().numpy()[:b_out]
                _, mean, std, _ = m.bank.host_copies[s]
                assert np.all(np.abs(outv - mean[35]) < 5 * std[35])          # un-standardised with the SST statistics
        else:
            # between slab predictions the slab state and outputs do not move
            pass
    # land regions never get a slab reservoir and always write 272 K before the mask restores the base SST
    assert not any(s in m.slab_bank.shapes for s in range(NREG) if not sea_reg[s])


def test_slab_predict_matches_oracle(oracle):
    """predict_slab_ml (src/mod_slab_ocean_reservoir.f90:1318-1363) = predict with no physics-model rows and every output
    un-standardised with the SST statistics, at the full slab size (n = 3968, d = 128, k = 23617, 8 outputs)."""
    from speedy_ml_amd.reservoir import ReservoirBank
    g = domain.initializedomain(NREG, 954)
    s = slab_sizes(g)
    assert (s.reservoir_numinputs, s.n, s.k, s.chunk_size_prediction, s.chunk_size_speedy) == (128, 3968, 23617, 8, 0)
    r = synth.make_reservoir(n=s.n, d=s.reservoir_numinputs, n_model=0, n_out=8, seed=77, deg=6, m=4000, radius=0.9, sigma=0.6)
    assert r.k == s.k
    bank = ReservoirBank(2, max_d=128, max_n_model=1, max_n_out=8)
    stat = np.full(8, 35, dtype=np.int32)
    bank.load(1, r.n, r.d, 0, 8, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, stat)
    rng = np.random.default_rng(2)
    x0 = rng.standard_normal(r.n) * 0.3
    bank.set_state(1, x0)
    bank.set_feedback(1, r.feedback)
    bank.predict()
    torch.cuda.synchronize()
    xw, ow = oracle.predict_raw(r.n, r.d, 0, 8, r.rows, r.cols, r.vals, r.win, r.wout, 1.0, r.feedback, None, x0)
    ow = ow * r.std[35] + r.mean[35]
    assert np.max(np.abs(bank.get_state(1) - xw)) <= 1e-13
    assert np.max(np.abs(bank.get_outvec(1) - ow)) <= 1e-11 * np.max(np.abs(ow))
