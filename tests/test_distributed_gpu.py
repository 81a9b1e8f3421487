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


def _worker(rank, world, port, out_dir):
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
    stream = torch.cuda.current_stream()
    for _ in range(2):
        m.step(stream)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), G=m.G.cpu().numpy(), F=m.F.cpu().numpy(),
             fb=m.feedback.cpu().numpy(), lm=m.local_model.cpu().numpy(), regions=np.array(regions))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    import torch.multiprocessing as mp
    from speedy_ml_amd import hybrid, synth
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sea = synth.land_mask()
    classes = hybrid.region_classes(sea)
    single = hybrid.HybridRank(list(range(NREG)), classes, sea_mask=sea, mode="hybrid", n_override=1, leapfrog_steps=2)
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
    _lib.check(L.sml_comm_allgather_outvec(comm, bank._h, _lib.dp(slab.data_ptr()), None))
    torch.cuda.synchronize()
    want = np.stack([bank.get_outvec(i) for i in range(3)])
    assert np.array_equal(slab.cpu().numpy(), want) and np.abs(want).max() > 0
    _lib.check(L.sml_comm_destroy(comm))
