"""One-process-per-GPU glue (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The forward hot path shards over independent IR/visible pairs: every rank runs the same kernels
on its own pairs and there is NO data-path collective.  The only exchanges are the timing fence
(barrier) and a MAX all-reduce of the elapsed time, used by bench.py to report whole-job
throughput.  (The reference has no distributed code at all: SURVEY F5.)
"""
import datetime
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, timeout_s=None):
    """Initialise the default process group when WORLD_SIZE > 1. Returns (rank, local_rank, world).
    timeout_s (default: SEGMIF_DIST_TIMEOUT or 300): how long the rendezvous and every later collective may take before it
    fails - a rank that never joins must end the job with a message, not hang it."""
    rank, local_rank, world = env_world()
    if world > 1 and not torch.distributed.is_initialized():
        timeout = datetime.timedelta(seconds=float(timeout_s if timeout_s is not None else os.environ.get("SEGMIF_DIST_TIMEOUT", 300)))
        where = (f"rank {rank}/{world} (local {local_rank}) at {os.environ.get('MASTER_ADDR', '127.0.0.1')}:"
                 f"{os.environ.get('MASTER_PORT', '?')}")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SEGMIF_DIST_BACKEND=gloo: run the multi-process path on a box with fewer GPUs than ranks (ranks share devices,
        # collectives go through gloo) - a functional rehearsal of the N > 1 control flow with the real kernels, not a
        # measurement.  RCCL refuses two ranks on one GPU, so the production backend cannot be rehearsed that way.
        backend = backend or os.environ.get("SEGMIF_DIST_BACKEND") or None
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        use_gpu = torch.cuda.is_available() and backend != "gloo"
        try:
            if use_gpu:
                # RCCL must come up on every rank or on none: a per-rank fallback to another backend would leave the
                # ranks in different process groups and deadlock the first collective, so a failure here is fatal.
                torch.distributed.init_process_group(backend or "nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)
                probe = torch.zeros(1, device="cuda")
                torch.distributed.all_reduce(probe)  # surfaces RCCL / xGMI bootstrap problems here, not mid-bench
                torch.cuda.synchronize()
            else:
                torch.distributed.init_process_group(backend or "gloo", timeout=timeout)
        except Exception as exc:
            raise RuntimeError(f"segmif_amd.dist: {where} could not join the process group within {timeout.total_seconds():.0f} s "
                               f"({type(exc).__name__}: {exc}).  Every rank needs the same MASTER_ADDR / MASTER_PORT / WORLD_SIZE, "
                               "its own RANK / LOCAL_RANK, one visible GPU per local rank and HSA_ENABLE_IPC_MODE_LEGACY=0.") from exc
    return rank, local_rank, world


def collective_selftest(param_sizes=None, bucket_mb=25.0, payload_mb=1.0):
    """What the first multi-GPU run should print before anything is timed: the world the backend itself reports, a checked
    sum / max all-reduce and all-gather of rank ids, one timed all-reduce of `payload_mb`, and - given the parameter sizes
    of a training step (elements per tensor) - the gradient bucket plan of segmif_amd.parallel.  Works on every backend
    (RCCL on the GPU box, gloo in the CPU tests); returns a dict, identical on every rank except `rank`."""
    import time
    from .parallel import plan_buckets
    d = torch.distributed
    if not (d.is_available() and d.is_initialized()):
        out = {"initialized": False, "world_size": 1, "rank": 0, "backend": None}
    else:
        w, r, backend = d.get_world_size(), d.get_rank(), d.get_backend()
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([float(r)], dtype=torch.float64, device=dev)
        d.all_reduce(t)
        mx = torch.tensor([float(r)], dtype=torch.float64, device=dev)
        d.all_reduce(mx, op=d.ReduceOp.MAX)
        gathered = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(w)]
        d.all_gather(gathered, torch.tensor([float(r)], dtype=torch.float64, device=dev))
        ok = float(t) == w * (w - 1) / 2 and float(mx) == w - 1 and [float(g) for g in gathered] == [float(i) for i in range(w)]
        buf = torch.ones(int(payload_mb * 2 ** 20) // 4, dtype=torch.float32, device=dev)
        d.all_reduce(buf)  # warm-up (connection set-up)
        fence()
        t0 = time.perf_counter()
        d.all_reduce(buf)
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        ok = ok and float(buf[0]) == float(w * w)
        out = {"initialized": True, "backend": backend, "world_size": w, "rank": r, "rank_id_collectives_ok": bool(ok),
               "allreduce_payload_bytes": buf.numel() * 4, "allreduce_ms": 1e3 * dt,
               "rccl_version": ".".join(map(str, torch.cuda.nccl.version())) if backend == "nccl" else None}
        if not ok:
            raise RuntimeError(f"segmif_amd.dist: collective self-test failed on rank {r}: {out}")
    if param_sizes is not None:
        class _P:  # plan_buckets only asks for numel / element_size
            def __init__(self, n):
                self.n = n

            def numel(self):
                return self.n

            def element_size(self):
                return 4
        plan = {}
        for name, sizes in param_sizes.items():
            buckets = plan_buckets([_P(n) for n in sizes], int(bucket_mb * 2 ** 20))
            plan[name] = {"tensors": len(sizes), "bytes": 4 * sum(sizes), "buckets": len(buckets),
                          "bucket_bytes": [4 * sum(p.numel() for p in b) for b in buckets]}
        out["gradient_buckets"] = plan
        out["bucket_mb"] = bucket_mb
    return out


def shard(n_items, rank, world):
    """Contiguous, balanced partition of `n_items` independent pairs: rank r gets range(lo, hi);
    sizes differ by at most one and the union over ranks is exactly range(n_items)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def fence():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value):
    """MAX all-reduce of a python float (the slowest rank defines the job's elapsed time)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return float(value)
    dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value):
    """-> list of every rank's python float, in rank order (one all_gather: per-rank spread of a timing)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return [float(value)]
    dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size())]
    torch.distributed.all_gather(out, t)
    return [float(o.item()) for o in out]


def sum_over_ranks(value):
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return float(value)
    dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return float(t.item())


def job_throughput(pairs_this_rank, elapsed_this_rank):
    """Whole-job pairs/s = (sum of pairs over ranks) / (max elapsed over ranks)."""
    return sum_over_ranks(pairs_this_rank) / max_over_ranks(elapsed_this_rank)


def shutdown():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
