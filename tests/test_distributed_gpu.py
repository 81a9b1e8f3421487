"""GPU, two ranks sharing the one GPU of the test box (gloo rendezvous, slab staged through the host): the sharded hybrid
step must reproduce the single-rank step bit for bit -- every reservoir is independent and the exchange only moves data."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NREG = 1152


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _float_west(r):
    """float-valued weights for the regions of the first of two ranks only (a module-level function: it crosses mp.spawn)"""
    return r < NREG // 2


def _worker(rank, world, port, out_dir, mixed=False):
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import domain, hybrid, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = [int(r) for r in domain.processor_decomposition_manual(rank, world, NREG)]
    m = hybrid.HybridRank(regions, classes, world=world, rank=rank, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2,
                          float32_weights=_float_west if mixed else False)
    compact = m.bank.compact()
    stream = torch.cuda.current_stream()
    for _ in range(2):
        m.step(stream)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), G=m.G.cpu().numpy(), F=m.F.cpu().numpy(),
             fb=m.feedback.cpu().numpy(), lm=m.local_model.cpu().numpy(), regions=np.array(regions), compact=np.array(int(compact)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mixed", [False, True])
def test_two_ranks_equal_one_rank(tmp_path, mixed):
    """mixed: the regions of rank 0 have float-valued weights (its bank alone could read compact copies, which sum the readout in another
    association), those of rank 1 arbitrary doubles: the ranks agree to keep to the 8-byte copies (hybrid.agree_on_storage; the native
    engine does the same over its communicator), so the result is still the single-rank one bit for bit."""
    import torch.multiprocessing as mp
    from speedy_ml_amd import hybrid, synth
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mixed), nprocs=2, join=True)
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    single = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2,
                               float32_weights=_float_west if mixed else False)
    assert not single.bank.compact()
    stream = torch.cuda.current_stream()
    for _ in range(2):
        single.step(stream)
    torch.cuda.synchronize()
    G, F = single.G.cpu().numpy(), single.F.cpu().numpy()
    fb, lm = single.feedback.cpu().numpy(), single.local_model.cpu().numpy()
    for r in range(2):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(d["G"], G) and np.array_equal(d["F"], F), r
        regs = d["regions"]
        assert np.array_equal(d["fb"], fb[regs]) and np.array_equal(d["lm"], lm[regs]), r
        assert int(d["compact"]) == 0, r                       # (rank 0's bank could: the agreement kept it on the 8-byte copies)


def _host_collective_worker(rank, world, port, out_dir):
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import domain, hybrid, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = [int(r) for r in domain.processor_decomposition_manual(rank, world, NREG)]
    m = hybrid.HybridRank(regions, classes, world=world, rank=rank, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2)
    eng = hybrid.NativeEngine(m)                    # no sml_comm: the host's collective between the engine's two half-steps
    assert eng.host_collective
    stream = torch.cuda.current_stream()
    eng.timing(True)
    for _ in range(2):
        eng.step(stream)
    t = eng.timing_collect()
    assert set(t["phases_ms_per_step"]) == {"predict", "allgather", "scatter", "speedy", "gather"} and t["phases_ms_per_step"]["speedy"] > 0
    g, f = eng.state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), G=g, F=f, fb=m.feedback.cpu().numpy(), lm=m.local_model.cpu().numpy(), regions=np.array(regions))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_native_engine_with_the_hosts_collective_equals_one_rank(tmp_path):
    """bench.py's N > 1 path: NativeEngine without an sml_comm -- sml_hybrid_step_predict, the host's all-gather of the outvec slab
    (torch.distributed; gloo here because both ranks share the box's one GPU, RCCL with one GPU per rank), sml_hybrid_step_finish --
    against the single-rank engine, bit for bit."""
    import torch.multiprocessing as mp
    from speedy_ml_amd import hybrid, synth
    mp.spawn(_host_collective_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    single = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2)
    eng = hybrid.NativeEngine(single)
    stream = torch.cuda.current_stream()
    for _ in range(2):
        eng.step(stream)
    G, F = eng.state()
    fb, lm = single.feedback.cpu().numpy(), single.local_model.cpu().numpy()
    for r in range(2):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(d["G"], G) and np.array_equal(d["F"][:152064], F[:152064]), r
        regs = d["regions"]
        assert np.array_equal(d["fb"], fb[regs]) and np.array_equal(d["lm"], lm[regs]), r
    eng.close()


def test_rccl_from_the_cabi_single_rank():
    """sml_comm_*: the C-ABI's own RCCL communicator (for a non-Python multi-rank host).  One rank is all a 1-GPU box can
    rehearse: the all-gather must reproduce the bank's outvec slab; the N > 1 path is the same call with a wider communicator."""
    import ctypes as C
    import numpy as np
    import torch
    from speedy_ml_amd import _lib
    from speedy_ml_amd.reservoir import ReservoirBank
    from speedy_ml_amd.synth import make_reservoir
    L = _lib.lib()
    ident = C.create_string_buffer(128)
    _lib.check(L.sml_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(L.sml_comm_create(1, 0, ident, C.byref(comm)))
    bank = ReservoirBank(3, max_d=12, max_n_model=4, max_n_out=6)
    for i in range(3):
        r = make_reservoir(n=96, d=12, n_model=4, n_out=6, seed=40 + i)
        bank.load(i, r.n, r.d, r.n_model, r.n_out, r.rows, r.cols, r.vals, r.win, r.wout, r.mean, r.std, None)
        bank.set_feedback(i, r.feedback)
        bank.set_local_model(i, r.local_model)
    bank.predict()
    slab = torch.zeros((3, 6), dtype=torch.float64, device="cuda")
    _lib.check(L.sml_comm_allgather_outvec(comm, bank._h, 3, _lib.dp(slab.data_ptr()), None))
    torch.cuda.synchronize()
    want = np.stack([bank.get_outvec(i) for i in range(3)])
    assert np.array_equal(slab.cpu().numpy(), want) and np.abs(want).max() > 0
    _lib.check(L.sml_comm_destroy(comm))


def test_ragged_region_unpack_of_the_cabi():
    """sml_comm_unpack_regions: the reordering sml_comm_allgather_outvec applies after a ragged all-gather (5 ranks x 231 slots for
    1152 regions: ranks 1 and 2 own regions 1150 and 1151 at slot 230), checked on a staged slab built on the host."""
    import ctypes as C
    from speedy_ml_amd import _lib, domain
    nranks, nreg, width = 5, 1152, 136
    slots = nreg // nranks + 1
    stage = np.full((nranks, slots, width), -7.0)
    for p in range(nranks):
        for i, r in enumerate(domain.processor_decomposition_manual(p, nranks, nreg)):
            stage[p, i] = 1000.0 * int(r) + np.arange(width)
    d_stage = torch.from_numpy(stage).cuda()
    out = torch.zeros((nreg, width), dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().sml_comm_unpack_regions(_lib.dp(d_stage.data_ptr()), nranks, slots, nreg, width, _lib.dp(out.data_ptr()), None))
    torch.cuda.synchronize()
    want = 1000.0 * np.arange(nreg)[:, None] + np.arange(width)[None, :]
    assert np.array_equal(out.cpu().numpy(), want)


def _train_worker(rank, world, port, out_dir):
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import training
    from test_distributed_gpu import TRAIN_NREG, build_region, TRAIN_ARGS
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)

    def no_collective(*a, **k):
        raise AssertionError("training must not use a collective")
    for name in ("all_reduce", "all_gather", "all_gather_into_tensor", "broadcast", "reduce_scatter"):
        setattr(dist, name, no_collective)
    res = training.train_sharded(rank, world, TRAIN_NREG, build_region, *TRAIN_ARGS, group=2, out_dir=out_dir, trial="shard")
    np.savez(os.path.join(out_dir, f"train_rank{rank}.npz"), **{str(k): v for k, v in res.items()})
    dist.destroy_process_group()


TRAIN_NREG, TRAIN_ARGS = 5, (246, 6, 6)            # 5 regions over 2 ranks: 2 + 2 and the tail region on rank 1; 246 h, discard 6, timestep 6


def build_region(region):
    from speedy_ml_amd import training
    from speedy_ml_amd.synth import make_reservoir
    n, d, n_model, n_out = 96, 12, 4, 6
    r = make_reservoir(n=n, d=d, n_model=n_model, n_out=n_out, seed=900 + region)
    rng = np.random.default_rng(900 + region)
    win = training.make_win(n, d, 0.5, rng.random((d, n // d)))
    L = TRAIN_ARGS[0]
    truth = np.cumsum(rng.standard_normal((d, L)) * 0.1, axis=1) * 0.2 + np.sin(np.arange(L) / 9.0 + region)[None, :]
    rows_t = np.arange(n_out)
    model = truth[rows_t[:n_model]] * 0.9 + 0.05 * rng.standard_normal((n_model, L))
    noisy = training.add_input_noise(truth, rng.standard_normal(truth.shape), 0.1)
    return dict(n=n, d=d, n_model=n_model, n_out=n_out, rows=r.rows, cols=r.cols, vals=r.vals, win=win, mean=r.mean, std=r.std,
                trainingdata=noisy, clean=truth, imperfect_model=model, target_rows=rows_t)


def test_sharded_training_equals_single_rank(tmp_path):
    """config 4's sharding (src/parallelmain.f90:82-128 over processor_decomposition): two ranks train disjoint region sets with
    no collective; together they produce, bit for bit, the W_out a single rank trains for all regions, and one weight file per
    region."""
    import torch.multiprocessing as mp
    from speedy_ml_amd import training, weights
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    single = training.train_sharded(0, 1, TRAIN_NREG, build_region, *TRAIN_ARGS, group=2)
    got = {}
    for r in range(2):
        d = np.load(tmp_path / f"train_rank{r}.npz")
        for k in d.files:
            assert int(k) not in got
            got[int(k)] = d[k]
    assert sorted(got) == list(range(TRAIN_NREG))
    for region in range(TRAIN_NREG):
        assert np.array_equal(got[region], single[region]), region
        assert os.path.exists(tmp_path / weights.trained_res_filename(region, "shard"))


def _engine_worker(rank, world, name, out_dir, slab):
    """one rank of the NATIVE engine (sml_hybrid_* + sml_comm over the host-staged rehearsal transport: two ranks share the GPU)"""
    import ctypes as C
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["SML_COMM_TRANSPORT"] = "shm"
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import _lib, domain, hybrid, synth
    from test_hybrid_engine_gpu import make_engine
    torch.cuda.set_device(0)
    L, check = _lib.lib(), _lib.check
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    regions = [int(r) for r in domain.processor_decomposition_manual(rank, world, NREG)]
    m = hybrid.HybridRank(regions, classes, world=1, rank=0, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2, slab=slab)
    comm = C.c_void_p()
    if world > 1:
        check(L.sml_comm_bootstrap(world, rank, name.encode(), C.c_uint64(0), C.byref(comm)))
    h = make_engine(m, regions, classes, comm=comm if world > 1 else None)
    stream = torch.cuda.current_stream()
    for _ in range(30 if slab else 3):
        check(L.sml_hybrid_step(h, 2, _lib.vp(stream)))
    torch.cuda.synchronize()
    g, fc = np.zeros(domain.G_SIZE), np.zeros(domain.G_SIZE)
    check(L.sml_hybrid_get_state(h, _lib.dp(g), _lib.dp(fc)))
    np.savez(os.path.join(out_dir, f"eng{world}_{rank}.npz"), G=g, F=fc, fb=m.feedback.cpu().numpy(), lm=m.local_model.cpu().numpy(), regions=np.array(regions))
    check(L.sml_hybrid_destroy(h))
    if world > 1:
        check(L.sml_comm_destroy(comm))


@pytest.mark.parametrize("world,slab", [(2, False), (2, True), (5, True)])
def test_native_engine_ranks_equal_one_rank(tmp_path, world, slab):
    """The C-ABI's own rank exchange inside the engine (sml_hybrid_set_comm -> sml_comm_allgather_outvec, for the atmosphere and
    the slab bank): `world` processes, each with its share of processor_decomposition and its own engine, reproduce the single-rank
    run bit for bit.  Five ranks split 1152 regions raggedly (231, 231, 230, 230, 230): the padded staging buffer and
    sml_comm_unpack_regions are on the path.  All ranks share ONE GPU: the communicator is the host-staged rehearsal transport
    (SML_COMM_TRANSPORT=shm); with one GPU per rank the same calls run over RCCL."""
    import torch.multiprocessing as mp
    # (five ranks + this process = the six GPU processes a box allows at once: SML_TEST_MAX_GPU_PROCS lowers the cap on a stricter pool)
    if world + 1 > int(os.environ.get("SML_TEST_MAX_GPU_PROCS", "6")):
        pytest.skip("more GPU processes than SML_TEST_MAX_GPU_PROCS allows")
    name = f"sml_test_{os.getpid()}_{world}_{int(slab)}"
    mp.spawn(_engine_worker, args=(world, name, str(tmp_path), slab), nprocs=world, join=True)
    _engine_worker(0, 1, name, str(tmp_path), slab)
    one = np.load(tmp_path / "eng1_0.npz")
    seen = 0
    for r in range(world):
        d = np.load(tmp_path / f"eng{world}_{r}.npz")
        assert np.array_equal(d["G"], one["G"]) and np.array_equal(d["F"], one["F"]), r
        regs = d["regions"]
        seen += len(regs)
        assert np.array_equal(d["fb"], one["fb"][regs]) and np.array_equal(d["lm"], one["lm"][regs]), r
    assert seen == NREG
