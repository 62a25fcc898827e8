"""Stand-in workload for the CPU test of bench.py's self-launch entry: the same multi-process glue (segmif_amd.dist) on
the gloo backend, no GPU work.  Rank 0 prints one JSON line like bench.py does."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmif_amd import dist  # noqa: E402

rank, local_rank, world = dist.init(backend="gloo")
dist.fence()
elapsed = dist.max_over_ranks(1.0 + rank)
pairs = dist.sum_over_ranks(3)
if rank == 0:
    print(json.dumps({"n_gpus": world, "pairs": pairs, "elapsed": elapsed, "argv": sys.argv[1:]}), flush=True)
dist.shutdown()
