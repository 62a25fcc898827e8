#!/usr/bin/env python
"""gemm_split epilogue A/B (SEGMIF_GEMM_EPI=direct|lds) over the encoder's Linear shapes at 64 images of 480x640,
with and without the residual / in-place output the blocks use."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops

def t(fn, iters=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

torch.manual_seed(0)
shapes = [("s1 fc1", 1228800, 256, 64, False), ("s1 fc2+res", 1228800, 64, 256, True), ("s2 q", 307200, 128, 128, False),
          ("s2 fc1", 307200, 512, 128, False), ("s2 fc2+res", 307200, 128, 512, True), ("s3 q", 76800, 320, 320, False),
          ("s3 fc1", 76800, 1280, 320, False), ("s3 fc2+res", 76800, 320, 1280, True), ("s3 kv", 19200, 640, 320, False),
          ("s4 fc1", 19200, 2048, 512, False), ("s4 fc2+res", 19200, 512, 2048, True)]
tot = {"direct": 0.0, "lds": 0.0}
for name, M, N, K, res in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    packs = ops.pack_linear(w)
    row = []
    for mode in ("direct", "lds"):
        os.environ["SEGMIF_GEMM_EPI"] = mode
        ms = t(lambda: ops.linear_auto(x, packs, N, bias=b, res=r, out=r))
        row.append(ms); tot[mode] += ms
    print(f"{name:12s} M {M:8d} N {N:5d} K {K:5d}: direct {row[0]:7.3f} ms  lds {row[1]:7.3f} ms  ({2.0*M*N*K/row[1]/1e9:6.1f} TF/s)", flush=True)
print("sum", tot)
