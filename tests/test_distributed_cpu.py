"""world_size-2 gloo tests of the N>1 path: sharding + the single all-gather of int8 logits (CPU tensors)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from edison_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(99)
    full = torch.from_numpy(rng.integers(-128, 128, (n_total, 10)).astype(np.int8))   # every rank knows the answer
    lo, hi = parallel.shard_range(n_total, rank, world)
    local = full[lo:hi].clone()
    got = parallel.all_gather_logits(local, n_total=n_total)
    ok1 = bool(torch.equal(got, full))
    got2 = parallel.all_gather_logits(local)            # sizes discovered with a tiny extra all_gather
    ok2 = bool(torch.equal(got2, full))
    ok3 = True
    if n_total % world == 0:
        g = parallel.LogitsGatherer(n_total // world)
        ok3 = bool(torch.equal(g(local), full))
        with pytest.raises(ValueError):
            g(local[:-1])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok1, ok2, ok3))


@pytest.mark.parametrize("n_total", [64, 101, 1])
def test_all_gather_logits_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def test_single_process_gather_is_identity():
    from edison_amd import parallel
    x = torch.arange(40, dtype=torch.int8).reshape(4, 10)
    assert torch.equal(parallel.all_gather_logits(x), x)
    g = parallel.LogitsGatherer(4)
    assert torch.equal(g(x), x)


def test_c_abi_shard_range_equals_the_python_one():
    """edison_dist_shard_range (what a C host uses) and parallel.shard_range must cut the batch identically, and the
    shards must tile [0, n) in rank order (the layout the all-gather assumes)."""
    sys.path.insert(0, ROOT)
    from edison_amd import parallel
    for n in (0, 1, 7, 10, 403, 262144, 2097152, 2097153):
        for w in (1, 2, 3, 4, 8):
            edge = 0
            for r in range(w):
                lo, hi = parallel.shard_range_c(n, r, w)
                assert (lo, hi) == parallel.shard_range(n, r, w)
                assert lo == edge
                edge = hi
            assert edge == n
    with pytest.raises(ValueError):
        parallel.shard_range_c(10, 2, 2)


def _id_worker(rank, world, port, q):
    """The C-ABI communicator bootstrap over gloo: rank 0's RCCL unique id reaches every rank unchanged (the
    ncclCommInitRank that follows needs GPUs and is exercised by tests/test_gpu_dist.py and bench.py --gpus N)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from edison_amd import parallel
    parallel.init_from_env(backend="gloo")
    payload = torch.zeros(128, dtype=torch.uint8)
    mine = None
    if rank == 0:
        mine = parallel.dist_unique_id()
        payload = torch.frombuffer(bytearray(mine), dtype=torch.uint8).clone()
    dist.broadcast(payload, src=0)
    got = bytes(payload.numpy().tobytes())
    gathered = [None] * world
    dist.all_gather_object(gathered, got)
    q.put((rank, len(got) == 128 and all(g == gathered[0] for g in gathered) and (mine is None or mine == got) and any(got)))
    dist.destroy_process_group()


def test_unique_id_reaches_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
