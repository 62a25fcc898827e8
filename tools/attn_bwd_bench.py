#!/usr/bin/env python
"""(r5) Backward of the spatial-reduction attention at the segmentation step's shapes (8 images of 480x640, mit_b3): the fused
flash-style kernels (csrc/attention_bwd.hip) against round 4's materialising backward.  ms per backward call, interleaved.
    python tools/attn_bwd_bench.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import autograd as ag  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tot = {"fused": 0.0, "materialize": 0.0}
for name, N, heads, blocks in (("stage1", 19200, 1, 3), ("stage2", 4800, 2, 4), ("stage3", 1200, 5, 18), ("stage4", 300, 8, 3)):
    C = heads * 64
    q = torch.randn(B, N, C, device="cuda", requires_grad=True)
    kv = torch.randn(B, 300, 2 * C, device="cuda", requires_grad=True)
    do = torch.randn(B, N, C, device="cuda")
    res = {}
    for mode in ("fused", "materialize", "fused", "materialize"):
        ag.SrAttentionFn.FUSED = mode == "fused"
        out = ag.sr_attention(q, kv, heads, 0.125)
        out.backward(do, retain_graph=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            q.grad = kv.grad = None
            out.backward(do, retain_graph=True)
        e.record()
        torch.cuda.synchronize()
        res[mode] = min(res.get(mode, 1e9), s.elapsed_time(e) / 5)
    ag.SrAttentionFn.FUSED = True
    for m in tot:
        tot[m] += res[m] * blocks
    print(f"{name}: N {N:6d} heads {heads}: fused {res['fused']:.3f} ms, materialising {res['materialize']:.3f} ms per call (x {blocks} blocks)", flush=True)
print(f"per segmentation step (28 attention calls): fused {tot['fused']:.2f} ms, materialising {tot['materialize']:.2f} ms")
