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


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from segmif_amd import dist as sdist
    from segmif_amd.parallel import GradAllReducer, allreduce_scalar_mean
    sdist.init(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    dead = torch.nn.Linear(4, 4)  # never used: its grads stay None on every rank (SURVEY F7 analogue)
    params = list(net.parameters()) + list(dead.parameters())
    red = GradAllReducer(params, bucket_mb=0.0002)  # tiny buckets: several collectives per step
    full_x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    full_y = torch.arange(8 * 3, dtype=torch.float32).reshape(8, 3) / 7.0
    mine = slice(rank * 4, rank * 4 + 4)
    out = []
    for step in range(3):  # step 0 = discovery (no overlap), steps 1-2 = hook-driven overlapped buckets
        for p in params:
            p.grad = None
        loss = ((net(full_x[mine]) - full_y[mine]) ** 2).mean()
        loss.backward()
        red.finish()
        out.append([p.grad.tolist() if p.grad is not None else None for p in params])  # plain lists: no shared-memory handles
    mean_loss = allreduce_scalar_mean(float(loss.detach()))
    # (r6) the way segmif_amd.train clears gradients under a reducer: buckets zeroed in place, autograd accumulates straight into
    # them - the hook makes NO per-parameter copy after the first step, and the averages are the same
    from segmif_amd.train import _zero_grads
    opt = torch.optim.SGD(params, lr=0.0)
    copies_before = red.copies
    inplace_ok = copies_before > 0  # (the loop above cleared with p.grad = None: one copy per live parameter per step)
    for step in range(2):
        _zero_grads(opt, red)
        ptrs = [p.grad.data_ptr() for p in params[:4]]
        ((net(full_x[mine]) - full_y[mine]) ** 2).mean().backward()
        red.finish()
        inplace_ok &= ptrs == [p.grad.data_ptr() for p in params[:4]]
        inplace_ok &= all(torch.allclose(p.grad, torch.tensor(g), atol=1e-7) for p, g in zip(params[:4], out[-1][:4]))
    inplace_ok &= red.copies == copies_before and params[4].grad is None
    # the exchange used after a replayed hipGraph (no autograd hooks, p.grad must stay the same tensor): same averages
    red.close()
    red2 = GradAllReducer(params, bucket_mb=0.0002)
    static_ok = True
    for step in range(2):
        for p in params:
            p.grad = None
        ((net(full_x[mine]) - full_y[mine]) ** 2).mean().backward()
        ptrs = [p.grad.data_ptr() if p.grad is not None else 0 for p in params]
        red2.allreduce_static()
        static_ok &= ptrs == [p.grad.data_ptr() if p.grad is not None else 0 for p in params]
        static_ok &= all(torch.allclose(p.grad, torch.tensor(g), atol=1e-7) for p, g in zip(params[:4], out[-1][:4]))
    q.put((rank, out, mean_loss, len(red._buckets), static_ok, inplace_ok))
    sdist.shutdown()


def test_data_parallel_gradient_allreduce_matches_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the full batch of 8
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    y = torch.arange(8 * 3, dtype=torch.float32).reshape(8, 3) / 7.0
    ((net(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in net.parameters()]
    assert res[0][3] > 1  # really bucketed
    for rank in range(world):
        for step in range(3):
            grads = res[rank][1][step]
            for g, r in zip(grads[:4], ref):
                assert torch.allclose(torch.tensor(g), r, atol=1e-6), (rank, step)
            assert grads[4] is None and grads[5] is None  # unused parameters stay grad-less
    assert abs(res[0][2] - res[1][2]) < 1e-12
    assert res[0][4] and res[1][4]  # GradAllReducer.allreduce_static: same averages, gradients updated in place
    assert res[0][5] and res[1][5]  # train._zero_grads under a reducer: no per-parameter copy after step 1, same averages


def test_bench_self_launch_entry_spawns_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (the form the driver
    uses): the same function drives a GPU-less stand-in script here, world size 2 on gloo."""
    import json
    import subprocess
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import argparse, bench; "
            f"bench.self_launch(argparse.Namespace(gpus=2), script={os.path.join(ROOT, 'tests', '_dist_probe.py')!r}, argv=['--steps', '7'])")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert lines == [{"n_gpus": 2, "pairs": 6.0, "elapsed": 2.0, "argv": ["--steps", "7"]}]


def test_bench_main_self_launches_when_no_launcher_set_world_size(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    called = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2"])
    monkeypatch.setattr(bench, "self_launch", lambda args, **k: (called.update(gpus=args.gpus), (_ for _ in ()).throw(SystemExit(0)))[1])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called == {"gpus": 8}


def test_bench_dry_collective_world_2_on_gloo():
    """`bench.py --gpus 2 --dry-collective` under torch.distributed.run with the collectives on gloo:
    rank 0 prints the world size the backend reports, the checked rank-id collectives, one timed all-reduce and the gradient
    bucket plan of both training steps (178 MB / 3.9 MB before the first step drops the grad-less parameters)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SEGMIF_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-collective"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(line) for line in r.stdout.splitlines() if line.startswith('{"dry_collective"')]
    assert len(lines) == 1
    d = lines[0]["dry_collective"]
    assert d["initialized"] and d["world_size"] == 2 and d["backend"] == "gloo" and d["rank_id_collectives_ok"]
    assert d["allreduce_payload_bytes"] == 2 ** 20 and d["allreduce_ms"] > 0
    plans = list(d["gradient_buckets"].values())
    seg, fus = plans[0], plans[1]
    assert seg["bytes"] == 4 * 44604873 and 7 <= seg["buckets"] <= 9 and sum(seg["bucket_bytes"]) == seg["bytes"]
    assert fus["bytes"] == 4 * 1034370 and fus["buckets"] == 1
    assert d["launch"]["WORLD_SIZE"] == 2 and d["launch"]["timeout_s"] == 300.0


def _lonely_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from segmif_amd import dist
    try:
        dist.init(backend="gloo", timeout_s=5)
        q.put("joined")
    except RuntimeError as e:
        q.put(str(e))


def test_a_rank_that_never_joins_fails_with_a_message():
    """WORLD_SIZE = 2 with only rank 0 present: init() gives up after its timeout and says which rank, where, and what to
    check - it must not hang the job."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_lonely_worker, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=120)
    p.join(60)
    assert "rank 0/2" in msg and "could not join the process group within 5 s" in msg and "MASTER_ADDR" in msg, msg
