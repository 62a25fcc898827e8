#!/usr/bin/env python
"""Encoder GEMM shapes (B = 32 images of 480x640, mit_b3): bf16x6 split GEMM (csrc/gemm_split.hip) against the fp32 MFMA tiles
igemm picks on its own.  Interleaved rounds, median.  Run through gpurun."""
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
tot = {"fp32": 0.0, "bf16x6": 0.0}
for name, tok, C, reps in (("stage1", B * 19200, 64, 3), ("stage2", B * 4800, 128, 4), ("stage3", B * 1200, 320, 18), ("stage4", B * 300, 512, 3)):
    for lname, N, K in (("q/proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        x = torch.randn(tok, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(tok, N, device="cuda")
        packs = ops.pack_linear(w)
        res = {"fp32": [], "bf16x6": []}
        for _ in range(5):
            res["fp32"].append(t(lambda: ops.linear(x, packs[0], N, bias=b, out=out)))
            res["bf16x6"].append(t(lambda: ops.linear_auto(x, packs, N, bias=b, out=out)))
        m32, m16 = statistics.median(res["fp32"]), statistics.median(res["bf16x6"])
        gf = 2.0 * tok * N * K / 1e9
        mult = reps * (2 if lname == "q/proj" else 1)
        tot["fp32"] += m32 * mult
        tot["bf16x6"] += m16 * mult
        print(f"{name} {lname:7s} M{tok:7d} N{N:5d} K{K:5d}: fp32 {m32:7.3f} ms ({gf / m32:6.1f} TF/s)   bf16x6 {m16:7.3f} ms ({gf / m16:6.1f} TF/s)", flush=True)
print(f"per encoder pass (q, proj, fc1, fc2 of every block): fp32 {tot['fp32']:.2f} ms, bf16x6 {tot['bf16x6']:.2f} ms")
