#!/usr/bin/env python
"""One encoder GEMM shape in a loop (for counter passes): python tools/gemm_one.py M N K [img] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4])
img = len(sys.argv) > 4 and sys.argv[4] == "img"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.05
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
packs = ops.pack_linear(w)
a = ops.rows_image_from_f32(x) if img else x
for _ in range(iters):
    ops.linear_auto(a, packs, N, bias=b, out=out)
torch.cuda.synchronize()
