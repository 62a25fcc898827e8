#!/usr/bin/env python
"""Spatial-reduction attention at the encoder's four stages (mit_b3, B images of 480x640): csrc/attention_split.hip (bf16x6)
against csrc/attention.hip (fp32 MFMA).  Run through gpurun:  python tools/attn_bench.py [B]"""
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
for name, N, Nk, heads, reps in (("stage1", 19200, 300, 1, 3), ("stage2", 4800, 300, 2, 4), ("stage3", 1200, 300, 5, 18), ("stage4", 300, 300, 8, 3)):
    C = heads * 64
    q = torch.randn(B, N, C, device="cuda")
    kv = torch.randn(B, Nk, 2 * C, device="cuda")
    res = {"fp32": [], "bf16x6": []}
    for _ in range(5):
        for mode in res:
            ops.set_attention_mode(mode)
            res[mode].append(t(lambda: ops.sr_attention(q, kv, heads, 0.125)))
    flop = 4.0 * B * heads * N * Nk * 64
    line = f"{name} N {N:6d} Nk {Nk} heads {heads}:"
    for mode in res:
        ms = statistics.median(res[mode])
        tot[mode] += reps * ms
        line += f"  {mode} {ms:7.3f} ms ({flop / ms / 1e9:6.1f} TF/s)"
    print(line)
print("per encoder pass: " + ", ".join(f"{m} {v:.2f} ms" for m, v in tot.items()))
