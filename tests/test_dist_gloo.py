"""world_size-2 gloo test of the multi-process glue bench.py uses for N > 1 (CPU only)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from segmif_amd import dist
    r, lr, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = dist.shard(11, rank, world)
    dist.fence()
    # rank 1 is the slow one: whole-job time must be its time, whole-job work the sum
    elapsed = 2.0 if rank == 0 else 4.0
    thr = dist.job_throughput(len(mine), elapsed)
    q.put((rank, list(mine), dist.max_over_ranks(elapsed), thr))
    dist.shutdown()


def test_two_process_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] + res[1][1] == list(range(11))  # disjoint, complete, contiguous
    assert abs(len(res[0][1]) - len(res[1][1])) <= 1
    assert res[0][2] == res[1][2] == 4.0
    assert res[0][3] == res[1][3] == 11 / 4.0


def test_shard_partition_properties():
    sys.path.insert(0, ROOT)
    from segmif_amd.dist import shard
    for n in (0, 1, 7, 8, 64, 1000):
        for w in (1, 2, 3, 8):
            parts = [list(shard(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
