"""Multi-process CPU coverage of the N>1 path (gloo, world_size 2): the static unit partition and the
single weight-blob broadcast that the multi-GPU bench performs over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from render_in_between_amd import distributed as ribdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ribdist.init_process_group("gloo")
    n = 1 << 16
    if rank == 0:
        g = torch.Generator().manual_seed(5)
        blob = torch.randn(n, generator=g)
    else:
        blob = torch.zeros(n)
    ribdist.broadcast_blob(blob, src=0)
    units = ribdist.shard_units(11, rank, world)
    # per-rank timing reduction as bench.py does it (MAX over ranks)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ribdist.blob_checksum(blob), units, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, u0, t0), (r1, c1, u1, t1) = res
    assert c0 == c1 != 0                                   # identical blob on every rank after ONE broadcast
    assert sorted(u0 + u1) == list(range(11)) and not set(u0) & set(u1)
    assert u0 == [0, 2, 4, 6, 8, 10] and u1 == [1, 3, 5, 7, 9]
    assert t0 == t1 == 2.0


def test_shard_units_edge_cases():
    assert ribdist.shard_units(0, 0, 4) == []
    assert ribdist.shard_units(3, 2, 8) == [2] and ribdist.shard_units(3, 3, 8) == [] and ribdist.shard_units(3, 5, 8) == []
    assert sum(len(ribdist.shard_units(32, r, 8)) for r in range(8)) == 32
