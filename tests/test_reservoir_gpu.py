"""GPU: the batched HIP reservoir path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Tolerances (fp64): reservoir state |dx| <= 1e-13 absolute (|x| <= 1; device tanh vs libm differ by <= 2 ulp),
outvec <= 1e-11 of max|outvec| (north_star: fields within 1e-10 relative)."""
import numpy as np
import pytest
import torch

from speedy_ml_amd import domain
from speedy_ml_amd.reservoir import ReservoirBank
from speedy_ml_amd.synth import make_reservoir

pytestmark = pytest.mark.gpu
X_TOL, OUT_TOL = 1e-13, 1e-11


def load(bank, slot, r, stat=None):
    bank.load(slot, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, stat)


def oracle_predict(oracle, r, x0, stat=None, leakage=1.0):
    x1, out = oracle.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, leakage,
                                 r.feedback, r.local_model if r.n_model else None, x0)
    if stat is not None:
        out = np.array([o * r.std[s] + r.mean[s] if s >= 0 else o for o, s in zip(out, stat)])
    return x1, out


def test_small_mixed_bank(oracle):
    shapes = [(600, 60, 12, 16), (640, 64, 0, 8), (330, 30, 7, 17), (1280, 40, 12, 35), (64, 8, 2, 1)]
    rs = [make_reservoir(n=n, d=d, n_model=m, n_out=o, seed=100 + i) for i, (n, d, m, o) in enumerate(shapes)]
    bank = ReservoirBank(8, max_d=64, max_n_model=12, max_n_out=35)
    rng = np.random.default_rng(0)
    x0 = [rng.standard_normal(r.n) * 0.3 for r in rs]
    for i, r in enumerate(rs):
        stat = (np.arange(r.n_out) % 36).astype(np.int32)
        stat[::5] = -1
        r.stat = stat
        load(bank, i + 1, r, stat)                 # slot 0 and the tail stay empty on purpose
        bank.set_state(i + 1, x0[i])
        bank.set_feedback(i + 1, r.feedback)
        if r.n_model:
            bank.set_local_model(i + 1, r.local_model)
    bank.predict()
    torch.cuda.synchronize()
    for i, r in enumerate(rs):
        xw, ow = oracle_predict(oracle, r, x0[i], r.stat)
        assert np.max(np.abs(bank.get_state(i + 1) - xw)) <= X_TOL
        assert np.max(np.abs(bank.get_outvec(i + 1) - ow)) <= OUT_TOL * max(1.0, np.max(np.abs(ow)))
    # second step continues from the device-resident state
    bank.predict(raw=True)
    for i, r in enumerate(rs):
        x1, _ = oracle_predict(oracle, r, x0[i])
        x2, o2 = oracle_predict(oracle, r, x1)
        assert np.max(np.abs(bank.get_state(i + 1) - x2)) <= 2 * X_TOL
        assert np.max(np.abs(bank.get_outvec(i + 1) - o2)) <= OUT_TOL * max(1.0, np.max(np.abs(o2)))


def test_config2_single_reservoir_full_size(oracle):
    """BASELINE config 2: region 954 (interior, SST input): n=5760, d=576, k=33177, W_out 136x5892."""
    g = domain.initializedomain(1152, 954)
    s = domain.allocate_res_sizes(g)
    assert (s.n, s.reservoir_numinputs, s.k) == (5760, 576, 33177)
    r = make_reservoir(n=s.n, d=s.reservoir_numinputs, n_model=s.chunk_size_speedy, n_out=s.chunk_size_prediction, seed=20240954)
    assert r.k == s.k
    _, stat = domain.out_map(1152, 954)
    bank = ReservoirBank(1)
    load(bank, 0, r, stat)
    # 55 synchronisation steps with N(0,1) inputs, then predict
    rng = np.random.default_rng(5)
    inputs = rng.standard_normal((r.d, 55))
    dev_in = torch.zeros((55, 1, 576), dtype=torch.float64, device="cuda")
    dev_in[:, 0, :] = torch.from_numpy(np.ascontiguousarray(inputs.T)).cuda()
    bank.synchronize(dev_in.data_ptr(), 55)
    xs = oracle.synchronize(r.n, r.d, r.rows, r.cols, r.vals, r.win, 1.0, inputs, np.zeros(r.n))
    assert np.max(np.abs(bank.get_state(0) - xs)) <= 55 * X_TOL
    bank.set_state(0, xs)          # teacher-force identical state (parity is per step)
    bank.set_feedback(0, r.feedback)
    bank.set_local_model(0, r.local_model)
    bank.predict()
    xw, ow = oracle_predict(oracle, r, xs, stat)
    assert np.max(np.abs(bank.get_state(0) - xw)) <= X_TOL
    assert np.max(np.abs(bank.get_outvec(0) - ow)) <= OUT_TOL * np.max(np.abs(ow))
    # reference-shaped per-call entry (x in/out on the host)
    x1, o1 = bank.predict_one(0, xs, r.local_model)
    assert np.max(np.abs(x1 - xw)) <= X_TOL and np.max(np.abs(o1 - ow)) <= OUT_TOL * np.max(np.abs(ow))
    assert np.array_equal(bank.get_state(0), x1)


def test_split_readout_contributions(oracle):
    """outvec_component_contribs (src/mod_reservoir.f90:1458-1461): v_p = wout(:, 1:132) local_model and v_ml = wout(:, 133:) x~ of every
    slot after a predict, left standardised as the reference leaves them; v_p + v_ml is the readout before un-standardisation.  Full-size
    region 954 beside two small reservoirs."""
    g = domain.initializedomain(1152, 954)
    s = domain.allocate_res_sizes(g)
    _, stat = domain.out_map(1152, 954)
    rs = [make_reservoir(n=s.n, d=s.reservoir_numinputs, n_model=132, n_out=136, seed=20240954),
          make_reservoir(n=1152, d=576, n_model=132, n_out=136, seed=2), make_reservoir(n=560, d=560, n_model=132, n_out=136, seed=3)]
    bank = ReservoirBank(len(rs))
    rng = np.random.default_rng(8)
    x0 = [rng.standard_normal(r.n) * 0.2 for r in rs]
    for i, r in enumerate(rs):
        load(bank, i, r, stat)
        bank.set_state(i, x0[i])
        bank.set_feedback(i, r.feedback)
        bank.set_local_model(i, r.local_model)
    with pytest.raises(Exception):
        bank.get_contribs(0)                        # nothing computed yet: an error, not stale memory
    bank.predict()
    bank.outvec_contribs()
    torch.cuda.synchronize()
    for i, r in enumerate(rs):
        xw, raw = oracle.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, 1.0, r.feedback, r.local_model, x0[i])
        xt = xw.copy()
        xt[1::2] = xt[1::2] ** 2                     # x_temp(2:n:2) ** 2
        v_p, v_ml = bank.get_contribs(i)
        sc = np.max(np.abs(raw))
        assert np.max(np.abs(v_p - r.wout[:, :132] @ r.local_model)) <= 1e-13 * sc
        assert np.max(np.abs(v_ml - r.wout[:, 132:] @ xt)) <= 1e-13 * sc
        assert np.max(np.abs(v_p + v_ml - raw)) <= 1e-13 * sc, i
        # and the bank's own (un-standardised) outvec is what un-standardising their sum gives
        want = oracle.unstandardize_res(oracle.initializedomain(1152, 954), r.mean, r.std, v_p + v_ml)
        assert np.max(np.abs(bank.get_outvec(i) - want)) <= OUT_TOL * np.max(np.abs(want))


def test_leakage_and_duplicates(oracle):
    r = make_reservoir(n=256, d=16, n_model=4, n_out=6, seed=9, deg=60)
    q = r.k // 4
    r.rows[:q] = r.rows[q:2 * q]
    r.cols[:q] = r.cols[q:2 * q]                     # duplicate (row,col) pairs must accumulate
    bank = ReservoirBank(1, max_d=16, max_n_model=4, max_n_out=6)
    bank.load(0, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None, leakage=0.3)
    x0 = np.random.default_rng(1).standard_normal(r.n)
    bank.set_state(0, x0)
    bank.set_feedback(0, r.feedback)
    bank.set_local_model(0, r.local_model)
    bank.predict(raw=True)
    xw, ow = oracle_predict(oracle, r, x0, leakage=0.3)
    assert np.max(np.abs(bank.get_state(0) - xw)) <= 4 * X_TOL
    assert np.max(np.abs(bank.get_outvec(0) - ow)) <= OUT_TOL * max(1.0, np.max(np.abs(ow)))


def test_dense_win_variant(oracle):
    """The commented-out reference variant fills whole W_in columns (src/mod_reservoir.f90:279): a dense W_in
    must give the same result as the oracle's dense matmul."""
    r = make_reservoir(n=192, d=24, n_model=0, n_out=5, seed=3)
    r.win = np.asfortranarray(np.random.default_rng(4).uniform(-0.5, 0.5, (r.n, r.d)))
    bank = ReservoirBank(1, max_d=24, max_n_model=0, max_n_out=5)
    load(bank, 0, r)
    x0 = np.random.default_rng(2).standard_normal(r.n) * 0.1
    bank.set_state(0, x0)
    bank.set_feedback(0, r.feedback)
    bank.predict(raw=True)
    xw, ow = oracle_predict(oracle, r, x0)
    assert np.max(np.abs(bank.get_state(0) - xw)) <= 10 * X_TOL
    assert np.max(np.abs(bank.get_outvec(0) - ow)) <= OUT_TOL * max(1.0, np.max(np.abs(ow)))


def test_errors_are_loud():
    from speedy_ml_amd._lib import SmlError
    bank = ReservoirBank(2, max_d=8, max_n_model=2, max_n_out=4)
    r = make_reservoir(n=64, d=8, n_model=2, n_out=4, seed=1)
    with pytest.raises(SmlError):
        bank.set_state(0, np.zeros(64))              # nothing loaded
    bad = r.rows.copy()
    bad[0] = 65
    with pytest.raises(SmlError):
        bank.load(0, r.n, r.d, r.n_model, r.n_out, bad, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)
    with pytest.raises(SmlError):
        bank.load(5, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)


def test_slab_ocean_shapes(oracle):
    """the slab-ocean reservoirs reuse the same kernels with m=4000, d=128 -> n=3968, k=23617,
    8 outputs (SST 4 + OHTC 4), no model rows, and one statistics slot for every output
    (predict_slab_ml, src/mod_slab_ocean_reservoir.f90:1318-1363)."""
    n, d, n_out = 3968, 128, 8
    r = make_reservoir(n=n, d=d, n_model=0, n_out=n_out, seed=77, m=4000, sigma=0.6)
    assert r.k == 23617
    stat = np.full(n_out, 3, dtype=np.int32)          # grid%sst_mean_std_idx for every output
    banks = ReservoirBank(4, max_d=d, max_n_model=0, max_n_out=n_out)
    rng = np.random.default_rng(3)
    x0 = rng.standard_normal(n) * 0.2
    for slot in (0, 3):
        banks.load(slot, r.n, r.d, 0, n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, stat)
        banks.set_state(slot, x0)
        banks.set_feedback(slot, r.feedback)
    banks.predict()
    xw, ow = oracle.predict_raw(r.n, r.d, 0, n_out, r.rows, r.cols, r.vals, r.win, r.wout, 1.0, r.feedback, None, x0)
    ow = ow * r.std[3] + r.mean[3]
    for slot in (0, 3):
        assert np.max(np.abs(banks.get_state(slot) - xw)) <= X_TOL
        assert np.max(np.abs(banks.get_outvec(slot) - ow)) <= OUT_TOL * np.max(np.abs(ow))


@pytest.mark.parametrize("variant", ["plain", "persistent", "persistent+drain"])
def test_split_readout_equals_predict(oracle, variant):
    """advance + readout_part(1) + readout_part(2) == predict, for odd/even/zero n_model, empty slots and ragged n_out;
    the persistent work-queue kernels (bounded footprint, and bounded + full-occupancy drain of the same queue) included."""
    shapes = [(600, 60, 12, 16), (640, 64, 0, 8), (330, 30, 7, 17), (1280, 40, 11, 35), (64, 8, 2, 1), (256, 16, 1, 20)]
    rs = [make_reservoir(n=n, d=d, n_model=m, n_out=o, seed=300 + i) for i, (n, d, m, o) in enumerate(shapes)]
    banks = [ReservoirBank(10, max_d=64, max_n_model=12, max_n_out=35) for _ in range(2)]
    rng = np.random.default_rng(3)
    x0 = [rng.standard_normal(r.n) * 0.3 for r in rs]
    for bank in banks:
        for i, r in enumerate(rs):
            stat = (np.arange(r.n_out) % 36).astype(np.int32)
            stat[::4] = -1
            r.stat = stat
            load(bank, i + 2, r, stat)
            bank.set_state(i + 2, x0[i])
            bank.set_feedback(i + 2, r.feedback)
            if r.n_model:
                bank.set_local_model(i + 2, r.local_model)
    a, b = banks
    a.predict()
    b.advance()
    side = torch.cuda.Stream()
    if variant == "plain":
        b.readout_part(1)
    else:
        torch.cuda.synchronize()
        b.readout_part(1, persistent=True, stream=side)
        if variant == "persistent+drain":
            b.readout_part(1, drain=True, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
    b.readout_part(2)
    torch.cuda.synchronize()
    for i, r in enumerate(rs):
        oa, ob = a.get_outvec(i + 2), b.get_outvec(i + 2)
        assert np.max(np.abs(oa - ob)) <= 1e-13 * max(1.0, np.max(np.abs(oa))), (variant, i)
        _, ow = oracle_predict(oracle, r, x0[i], r.stat)
        assert np.max(np.abs(ob - ow)) <= OUT_TOL * max(1.0, np.max(np.abs(ow)))


def test_weights_file_roundtrip_to_device(oracle, tmp_path):
    """The reference's trained-weights file (classic NetCDF, float32 reals: src/mod_reservoir.f90:1727-1736, src/mod_io.f90:2938-2983)
    written, read back and loaded into the bank: the device predicts with exactly the float32-rounded weights the reference would
    have after read_trained_res."""
    from speedy_ml_amd import weights
    r = make_reservoir(n=1152, d=576, n_model=132, n_out=136, seed=77)
    path = str(tmp_path / weights.trained_res_filename(954, "trial"))
    weights.write_trained_res(path, r.win, r.wout, r.rows, r.cols, r.vals, r.mean, r.std)
    bank = ReservoirBank(1)
    stat = (np.arange(r.n_out) % 36).astype(np.int32)
    w = weights.load_trained_res(bank, 0, path, r.n_model, stat)
    assert bank.compact()                          # weights out of a reference file are floats: the bank reads its 4-byte copies
    f32 = lambda a: np.asarray(a).astype(np.float32).astype(np.float64)
    assert np.array_equal(w["win"], f32(r.win)) and np.array_equal(w["wout"], f32(r.wout)) and np.array_equal(w["rows"], r.rows)
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal(r.n) * 0.3
    bank.set_state(0, x0)
    bank.set_feedback(0, r.feedback)
    bank.set_local_model(0, r.local_model)
    bank.predict()
    torch.cuda.synchronize()
    x1, out = oracle.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, f32(r.vals), f32(r.win), f32(r.wout), 1.0, r.feedback,
                                 r.local_model, x0)
    out = out * f32(r.std)[stat] + f32(r.mean)[stat]
    assert np.max(np.abs(bank.get_state(0) - x1)) <= X_TOL
    assert np.max(np.abs(bank.get_outvec(0) - out)) <= OUT_TOL * np.max(np.abs(out))
    with pytest.raises(ValueError):
        weights.load_trained_res(bank, 0, path, r.n_model + 1, stat)          # wout columns must be n + n_model


def test_prediction_start_sequence(oracle):
    """initialize_prediction + start_prediction (src/mod_reservoir.f90:791-961) on the device for two reservoirs of one bank: the
    un-noisy sync from a zero state, the sync of the prediction window continuing from it, then the first feedback / local_model;
    against the oracle's synchronize and predict on the same columns."""
    from speedy_ml_amd import prediction
    rs = [make_reservoir(n=384, d=24, n_model=6, n_out=8, seed=41), make_reservoir(n=200, d=20, n_model=4, n_out=5, seed=42)]
    bank = ReservoirBank(3, max_d=24, max_n_model=6, max_n_out=8)
    for i, r in enumerate(rs):
        load(bank, i + 1, r)                                   # slot 0 stays empty
        bank.set_state(i + 1, np.full(r.n, 0.3))               # must be reset by initialize_prediction
    rng = np.random.default_rng(6)
    timestep, un_noisy, synclength = 6, 360, 84                 # 59 and 13 columns
    pd1 = [None] + [rng.standard_normal((r.d, un_noisy // timestep)) for r in rs]
    pd2 = [None] + [rng.standard_normal((r.d, synclength // timestep + 2)) for r in rs]
    ims = [None] + [rng.standard_normal((r.n_model, synclength // timestep + 2)) for r in rs]
    assert prediction.initialize_prediction(bank, pd1, timestep, un_noisy_sync=un_noisy) == 59
    assert prediction.start_prediction(bank, pd2, ims, synclength, timestep) == 13
    bank.predict(raw=True)
    torch.cuda.synchronize()
    for i, r in enumerate(rs):
        x = oracle.synchronize(r.n, r.d, r.rows, r.cols, r.vals, r.win, 1.0, pd1[i + 1][:, :59], np.zeros(r.n))
        x = oracle.synchronize(r.n, r.d, r.rows, r.cols, r.vals, r.win, 1.0, pd2[i + 1][:, :13], x)
        x1, out = oracle.predict_raw(r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, 1.0,
                                     np.ascontiguousarray(pd2[i + 1][:, 13]), np.ascontiguousarray(ims[i + 1][:, 14]), x)
        assert np.max(np.abs(bank.get_state(i + 1) - x1)) <= 1e-12
        assert np.max(np.abs(bank.get_outvec(i + 1) - out)) <= OUT_TOL * np.max(np.abs(out))



def test_compact_storage_of_float_weights(oracle):
    """Weights that are exactly floats (a reservoir read from the reference's NetCDF weight files: NF90_REAL) are also kept as 4-byte
    copies and the predict kernels read those: same numbers, same fp64 arithmetic, half the bytes.  Config 2 at full size: the compact
    bank against the oracle (on the float-valued weights) and against the same bank told to keep to its 8-byte copies -- state bit for
    bit (the update sums in the same order), outvec to 1e-13 (the readout sums four columns per load instead of two); arbitrary doubles
    keep the bank on the 8-byte copies, and so does a W_out replaced by one (sml_bank_set_wout), until float weights come back."""
    r = make_reservoir(seed=20240954, float32_weights=True)
    _, stat = domain.out_map(1152, 954)
    banks = [ReservoirBank(2), ReservoirBank(2)]
    banks[1].use_compact(False)
    rng = np.random.default_rng(8)
    x0 = rng.standard_normal(r.n) * 0.3
    for b in banks:
        for slot in (0, 1):
            load(b, slot, r, stat)
            b.set_state(slot, x0)
            b.set_feedback(slot, r.feedback)
            b.set_local_model(slot, r.local_model)
    assert banks[0].compact() and not banks[1].compact()
    u0, r0 = banks[0].algorithmic_bytes()
    u1, r1 = banks[1].algorithmic_bytes()
    assert r0 < 0.51 * r1 and u0 < 0.75 * u1                     # W_out in half the bytes, the operator at 6 instead of 10 B per nonzero
    for b in banks:
        b.predict()
    torch.cuda.synchronize()
    xw, ow = oracle_predict(oracle, r, x0, stat)
    for slot in (0, 1):
        assert np.max(np.abs(banks[0].get_state(slot) - xw)) <= X_TOL
        assert np.max(np.abs(banks[0].get_outvec(slot) - ow)) <= OUT_TOL * np.max(np.abs(ow))
        assert np.array_equal(banks[0].get_state(slot), banks[1].get_state(slot))
        assert np.max(np.abs(banks[0].get_outvec(slot) - banks[1].get_outvec(slot))) <= 1e-13 * np.max(np.abs(ow))
    # one reservoir with arbitrary doubles takes the whole bank back to the 8-byte copies ...
    q = make_reservoir(seed=3)
    load(banks[0], 1, q, stat)
    assert not banks[0].compact()
    load(banks[0], 1, r, stat)
    assert banks[0].compact()
    # ... and so does a trained W_out (arbitrary doubles) until it has been through a weights file
    banks[0].set_wout(0, q.wout)
    assert not banks[0].compact()
    banks[0].set_state(0, x0); banks[0].set_feedback(0, r.feedback); banks[0].set_local_model(0, r.local_model)
    banks[0].predict()
    torch.cuda.synchronize()
    r.wout, keep = q.wout, r.wout
    _, ow2 = oracle_predict(oracle, r, x0, stat)
    assert np.max(np.abs(banks[0].get_outvec(0) - ow2)) <= OUT_TOL * np.max(np.abs(ow2))
    r.wout = keep
    banks[0].set_wout(0, r.wout)
    assert banks[0].compact()
