#!/usr/bin/env python
"""Micro-benchmark of the 3x3 dilation-2 weight gradient (DRDB convs) at the training geometry: fp32 kernel vs bf16x6
kernel, plus the bf16x6 kernel with phases switched off (SEGMIF_WG3_DBG) to see where a tile's time goes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import autograd as ag

B, H, W = 8, 480, 640
torch.manual_seed(0)
for cin in (64, 128, 192):
    x = torch.randn(B, H, W, cin, device="cuda")
    dy = torch.randn(B, H, W, 32, device="cuda")
    flops = 2.0 * B * H * W * 9 * cin * 32

    def run(env):
        for k in ("SEGMIF_WGRAD3X3", "SEGMIF_WG3_DBG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(2):
            ag.conv_wgrad(x, dy, (32, cin, 3, 3), 3, 1, 2, 2, want_bias=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ag.conv_wgrad(x, dy, (32, cin, 3, 3), 3, 1, 2, 2, want_bias=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5

    base = {}
    one = {"SEGMIF_WGRAD3X3": "split1"}
    for name, env in (("fp32", {"SEGMIF_WGRAD3X3": "fp32"}), ("two-team", {}), ("one-team", one), ("no-mfma", {**one, "SEGMIF_WG3_DBG": "1"}),
                      ("no-split", {**one, "SEGMIF_WG3_DBG": "2"}), ("no-loads", {**one, "SEGMIF_WG3_DBG": "4"}),
                      ("mfma-only", {**one, "SEGMIF_WG3_DBG": "6"}), ("loads-only", {**one, "SEGMIF_WG3_DBG": "3"})):
        dt = run(env)
        print(f"Cin {cin:4d} {name:10s} {dt * 1e3:8.3f} ms  {flops / dt / 1e12:7.1f} TF/s", flush=True)
