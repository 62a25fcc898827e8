"""Encoder GEMM shapes at the bench batch (32 images of 480x640) on the candidate tiles, `auto` first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_bench import bench_dense, tile_names
names = tile_names()
for (m, n, k) in ((614400, 64, 64), (614400, 256, 64), (614400, 64, 256), (153600, 128, 128), (153600, 512, 128), (153600, 128, 512),
                  (38400, 320, 320), (38400, 1280, 320), (38400, 320, 1280), (9600, 512, 512), (9600, 2048, 512), (9600, 512, 2048)):
    bench_dense(m, n, k, [-1, 6, 12, 2, 13, 4, 7], names, iters=10)
