#!/usr/bin/env python
"""(r5) debugging aid: fp32 -> pairs -> fp32 round trip on a small tensor, positions of the mismatches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops  # noqa: E402

for M, K in ((8, 32), (4096, 128), (777, 320)):
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(M, K, generator=g) * 2 - 1) * (10.0 ** ((torch.rand(M, 1, generator=g) * 2 - 1) * 1.5))
    guard = ops.Planes16Guard("cuda", 1)
    ops.install_guard(guard)
    xp = ops.pairs_from_f32(x.cuda().view(1, M, K))
    back = ops.pairs_to_f32(xp).cpu().view(M, K)
    ops.install_guard(None)
    err = (back.double() - x.double()).abs() / (x.double().abs() + 2.0 ** -14)
    bad = (err > 2.0 ** -21).nonzero()
    print(f"M {M} K {K}: worst {float(err.max()):.3e}, bad {bad.shape[0]} of {M * K}; guard max {guard.maxima().tolist()} (true {float(x.abs().max()):.4f})")
    for i, j in bad[:12].tolist():
        print(f"   [{i},{j}] x {float(x[i, j]):.9g} back {float(back[i, j]):.9g}")
    if bad.shape[0]:
        print("   bad columns mod 16:", sorted(set((bad[:, 1] % 16).tolist())), " bad rows (first 10):", sorted(set(bad[:, 0].tolist()))[:10])
