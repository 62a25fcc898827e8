#!/usr/bin/env python
"""(r5) One MiT block of stage 2 / 3 / 4 at the bench's token counts (B images of 480x640, mit_b3), inference inside a guarded scope:
wall time per block with the Linears on gemm_pairs (SEGMIF_GEMM_PAIRS=on) and on round 4's gemm_split.  Under tools/kstats.sh the
kernel table shows where a block's time goes.   python tools/block_bench.py [B] [on|off|both] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from segmif_amd import ops  # noqa: E402
from segmif_amd.core.mix_transformer import Block  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "both"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
for name, C, heads, sr, H, W in (("stage2", 128, 2, 4, 60, 80), ("stage3", 320, 5, 2, 30, 40), ("stage4", 512, 8, 1, 15, 20)):
    blk = Block(dim=C, num_heads=heads, mlp_ratio=4, qkv_bias=True, sr_ratio=sr).cuda().eval()
    x = torch.randn(B, H * W, C, device="cuda")
    for mode in (("on", "off") if which == "both" else (which,)):
        prev = ops.set_pairs_mode(mode)
        guard = ops.Planes16Guard("cuda", B)
        try:
            def run():
                guard.reset()
                ops.install_guard(guard)
                try:
                    with torch.no_grad():
                        return blk.forward_(x.clone(), H, W)
                finally:
                    ops.install_guard(None)
            y = run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                run()
            e.record()
            torch.cuda.synchronize()
            print(f"{name} C{C} tokens {B * H * W}: pairs {mode:3s} {s.elapsed_time(e) / iters:7.3f} ms per block  (guard tripped {int(guard.tripped().sum())}, "
                  f"checksum {float(y.double().abs().mean()):.6f})", flush=True)
        finally:
            ops.set_pairs_mode(prev)
