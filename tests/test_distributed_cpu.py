"""CPU, gloo, world_size > 1: the N>1 data path of the hybrid step -- region sharding by processor_decomposition and the
one collective of the step (region-ordered outvec slab on every rank), even and ragged (remainder rule) splits."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NREG, STRIDE = 1152, 136


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import domain, hybrid
    regions = [int(r) for r in domain.processor_decomposition_manual(rank, world, NREG)]
    cap = max(len(domain.processor_decomposition_manual(p, world, NREG)) for p in range(world))
    # each region's outvec is a deterministic function of the region id, so every rank can check the whole slab
    local = torch.zeros((len(regions), STRIDE), dtype=torch.float64)
    for i, r in enumerate(regions):
        local[i] = torch.arange(STRIDE, dtype=torch.float64) + 1000.0 * r
    all_out = torch.full((NREG, STRIDE), -1.0, dtype=torch.float64)
    even = NREG % world == 0
    for _ in range(2):      # twice: the slab is reused every step
        hybrid.gather_outvec_slab(local, regions, all_out, even)
    want = torch.arange(STRIDE, dtype=torch.float64)[None, :] + 1000.0 * torch.arange(NREG, dtype=torch.float64)[:, None]
    ok = bool(torch.equal(all_out, want))
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([ok, len(regions), cap]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_outvec_slab_gloo(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    total = 0
    for r in range(world):
        ok, n, cap = np.load(tmp_path / f"ok_{r}.npy")
        assert ok == 1, f"rank {r} assembled a wrong slab"
        total += int(n)
    assert total == NREG


def _plan_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    load_package()
    from speedy_ml_amd import training
    mine = [r for g in training.shard_plan(rank, world, NREG, group=8) for r in g]
    groups = [len(g) for g in training.shard_plan(rank, world, NREG, group=8)]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)             # only the TEST gathers: training itself has no collective
    flat = sorted(r for lst in everyone for r in lst)
    np.save(os.path.join(out_dir, f"plan_{rank}.npy"), np.array([flat == list(range(NREG)), max(groups) <= 8, len(mine)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_training_shards_are_disjoint_and_complete(world, tmp_path):
    """config 4's 'reservoirs sharded 1 -> 8 GPUs': program main's training loop (src/parallelmain.f90:82-128) gives every rank the
    regions of processor_decomposition; the ranks' sets are disjoint, cover all 1152 regions (remainder rule included) and are cut
    into bank-sized groups."""
    mp.spawn(_plan_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    total = 0
    for r in range(world):
        complete, groups_ok, n = np.load(tmp_path / f"plan_{r}.npy")
        assert complete == 1 and groups_ok == 1
        total += int(n)
    assert total == NREG


def test_bench_gpus_n_launches_its_own_ranks():
    """The driver runs `python bench.py --gpus N ...` with no launcher around it: bench.py must start the N ranks itself (a child
    torch.distributed.run, never an exec).  --dry-run stops after the rendezvous, so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SML_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and sorted(r["rank"] for r in line["ranks"]) == [0, 1]
    assert sorted(r["local_rank"] for r in line["ranks"]) == [0, 1]


def test_bench_gpus_n_relays_a_failing_rank():
    """... and the child's exit code comes back: without a GPU the ranks refuse to run (there is no CPU fallback)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "no HIP device visible" in p.stderr


# ---------------------------------------------------------------- sml_comm_bootstrap: leftovers of an earlier run under the same name
_BOOT = r"""
import ctypes as C, os, sys
L = C.CDLL(sys.argv[1])
L.sml_last_error.restype = C.c_char_p
h = C.c_void_p()
rc = L.sml_comm_bootstrap(int(sys.argv[3]), int(sys.argv[2]), sys.argv[4].encode(), C.c_uint64(1024), C.byref(h))
print("rc", rc, L.sml_last_error().decode() if rc else "", flush=True)
if rc == 0:
    print("segment_name_left", os.path.exists("/dev/shm/" + sys.argv[4]), flush=True)
    L.sml_comm_destroy(h)
sys.exit(0 if rc == 0 else 3)
"""


def _stale_leftovers(name):
    """what a crashed earlier run leaves under /dev/shm: an id file and an initialised segment, both with another launch's token"""
    import struct
    with open(f"/dev/shm/{name}.id", "wb") as f:
        f.write(struct.pack("<Q", 0xdeadbeef) + bytes(128))
    hdr = struct.pack("<4i128sQQ", 0x534d4c43, 0, 0, 0, bytes(128), 1024, 0xdeadbeef)        # ShmHeader of csrc/comm.hip
    with open(f"/dev/shm/{name}", "wb") as f:
        f.write(hdr + bytes(64 + 1024 * 2 * 8 + 64))


def test_bootstrap_ignores_the_leftovers_of_an_earlier_run():
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "speedy-ml_amd", "csrc", "libspeedyml_hip.so")
    name = f"sml_stale_test_{os.getpid()}"
    run = lambda rank, env: subprocess.Popen([sys.executable, "-c", _BOOT, so, str(rank), "2", name], env=dict(os.environ, **env),
                                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        # RCCL transport: a peer that finds only the stale id file must keep waiting for this launch's (and say so), not join a dead id
        _stale_leftovers(name)
        p = run(1, {"SML_COMM_TIMEOUT_S": "2", "SML_COMM_NONCE": "41"})
        out = p.communicate(timeout=60)[0]
        assert p.returncode == 3 and "of this launch" in out, out
        # shm transport: rank 1 starts first and sees the stale segment; rank 0 replaces it; both meet in the new one
        env = {"SML_COMM_TRANSPORT": "shm", "SML_COMM_TIMEOUT_S": "30", "SML_COMM_NONCE": "42"}
        p1 = run(1, env)
        time.sleep(1.0)
        assert p1.poll() is None, p1.communicate()[0]           # still waiting: the leftover did not fool it
        p0 = run(0, env)
        o0, o1 = p0.communicate(timeout=60)[0], p1.communicate(timeout=60)[0]
        assert p0.returncode == 0 and p1.returncode == 0, (o0, o1)
        # rank 0 withdraws the name as soon as everybody is attached (before its bootstrap call returns): nothing for the next run to
        # trip over, even if the host never reaches sml_comm_destroy
        assert "segment_name_left False" in o0, o0
        assert not os.path.exists(f"/dev/shm/{name}")
    finally:
        for f in (f"/dev/shm/{name}", f"/dev/shm/{name}.id", f"/dev/shm/{name}.id.tmp"):
            if os.path.exists(f):
                os.unlink(f)
