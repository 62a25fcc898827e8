#!/usr/bin/env python
"""CrossPath kernels (csrc/crosspath.hip) at the bench's shape: B images of 480x640, 64 channels.  Median of interleaved
rounds; algorithmic bytes and fp32-equivalent flops beside the time.  Run through gpurun:  python tools/crosspath_bench.py [B]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops  # noqa: E402


def t(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = 480, 640
N = H * W
g = torch.Generator(device="cuda").manual_seed(0)
x3 = torch.randn(B, N, 64, device="cuda", generator=g)
xi = torch.randn(B, N, 64, device="cuda", generator=g)
w3, wi = (torch.randn(64, 64, device="cuda", generator=g) * 0.1 for _ in range(2))
b3, bi, bend = (torch.randn(64, device="cuda", generator=g) * 0.1 for _ in range(3))
weff = torch.randn(B, 64, 128, device="cuda", generator=g) * 0.1
gamma, beta = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
out = torch.empty(B, N, 64, device="cuda")
planes = ops.Planes(B, H, W, 12, "cuda")
res = {"gram": [], "tail": [], "tail+planes": []}
for _ in range(5):
    res["gram"].append(t(lambda: ops.crosspath_gram(xi, wi, bi)))
    res["tail"].append(t(lambda: ops.crosspath_tail(x3, xi, w3, b3, wi, bi, weff, bend, (gamma, beta, 1e-5), out=out)))
    res["tail+planes"].append(t(lambda: ops.crosspath_tail(x3, xi, w3, b3, wi, bi, weff, bend, (gamma, beta, 1e-5), out=out,
                                                             planes=planes, hw=(H, W))))
px = B * N
for name, byts, flop in (("gram", 256, 2 * 64 * 64 + 3 * 2 * 32 * 32), ("tail", 768, 2 * 2 * 64 * 64 + 2 * 128 * 64),
                         ("tail+planes", 768 + 384, 2 * 2 * 64 * 64 + 2 * 128 * 64)):
    ms = statistics.median(res[name])
    print(f"{name:12s} B {B:3d}: {ms:7.3f} ms   {px * byts / ms / 1e9:6.2f} TB/s algorithmic   {px * flop / ms / 1e9:6.1f} TFLOP/s fp32-equivalent")
