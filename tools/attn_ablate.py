#!/usr/bin/env python
"""(r6) sr_attention_split (f16x3, pairs output) at the shapes that matter, under whatever library SEGMIF_HIP_LIB names
(tools/attn_ablate.sh builds the -DATTN_ABL variants): ms per launch of the attention kernel alone (the pack launch is timed apart)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops  # noqa: E402


def t(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, B, N, Nk, heads in (("stage3 b64", 64, 1200, 300, 5), ("stage2 b64", 64, 4800, 300, 2), ("b5 stage3 b2", 2, 4096, 1024, 5)):
    C = heads * 64
    q = torch.randn(B, N, C, device="cuda")
    kv = torch.randn(B, Nk, 2 * C, device="cuda")
    g = ops.Planes16Guard("cuda", B)
    g.slot = lambda images=None: (g.amax.data_ptr(), B if images == B else 1)  # (one row re-used: this is a timing loop)
    prev = ops.install_guard(g)
    try:
        ms = t(lambda: ops.sr_attention(q, kv, heads, 0.125, pairs=True))
    finally:
        ops.install_guard(prev)
    flop = 4.0 * B * heads * N * Nk * 64
    print(f"{name}: {1e3 * ms:8.1f} us per call (pack + attention), {flop / ms / 1e9:6.1f} TFLOP/s")
