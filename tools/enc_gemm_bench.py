"""Encoder GEMM shapes (B=8, 480x640) on tiles 6 (64x64x16), 11 (64x64x32), 2 (128x64x16)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_bench import bench_dense, bench_conv, tile_names
names = tile_names()
for (m, n, k) in ((153600, 64, 64), (153600, 256, 64), (153600, 64, 256), (38400, 128, 128), (38400, 512, 128), (38400, 128, 512),
                  (9600, 320, 320), (9600, 640, 320), (9600, 1280, 320), (9600, 320, 1280), (2400, 512, 512), (2400, 2048, 512), (2400, 512, 2048)):
    bench_dense(m, n, k, [6, 12, 2, 13], names, iters=20)
bench_conv(8, 30, 40, 320, 320, 2, 0, 1, [6, 12], names, stride=2)
bench_conv(8, 120, 160, 64, 64, 8, 0, 1, [6, 12], names, stride=8)
