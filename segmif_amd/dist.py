"""One-process-per-GPU glue (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The forward hot path shards over independent IR/visible pairs: every rank runs the same kernels
on its own pairs and there is NO data-path collective.  The only exchanges are the timing fence
(barrier) and a MAX all-reduce of the elapsed time, used by bench.py to report whole-job
throughput.  (The reference has no distributed code at all: SURVEY F5.)
"""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group when WORLD_SIZE > 1. Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SEGMIF_DIST_BACKEND=gloo: run the multi-process path on a box with fewer GPUs than ranks (ranks share devices,
        # collectives go through gloo) - a functional rehearsal of the N > 1 control flow with the real kernels, not a
        # measurement.  RCCL refuses two ranks on one GPU, so the production backend cannot be rehearsed that way.
        backend = backend or os.environ.get("SEGMIF_DIST_BACKEND") or None
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        use_gpu = torch.cuda.is_available() and backend != "gloo"
        if use_gpu:
            # RCCL must come up on every rank or on none: a per-rank fallback to another backend would leave the
            # ranks in different process groups and deadlock the first collective, so a failure here is fatal.
            torch.distributed.init_process_group(backend or "nccl", device_id=torch.device("cuda", local_rank))
            probe = torch.zeros(1, device="cuda")
            torch.distributed.all_reduce(probe)  # surfaces RCCL / xGMI bootstrap problems here, not mid-bench
            torch.cuda.synchronize()
        else:
            torch.distributed.init_process_group(backend or "gloo")
    return rank, local_rank, world


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
