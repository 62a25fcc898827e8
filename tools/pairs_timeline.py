#!/usr/bin/env python
"""Step timeline of csrc/gemm_pairs.hip built with -DPAIRS_DBG=1 (tools/pairs_timeline.sh): s_memtime stamps of thread 0
of the first 1024 workgroups.  Per K step: 0 before the wait + barrier | 1 past the barrier | 2 next stage's DMA issued | 3 MFMAs
issued.  [29]: 0 kernel start, 1 prologue DMA issued; [30]: 0 K loop done, 1 past the closing barrier; [31][0]: stores drained.
    SEGMIF_HIP_LIB=$PWD/segmif_amd/lib/variants/lib_pairs_dbg.so python tools/pairs_timeline.py M N K [tile]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from segmif_amd import ops

M, N, K = (int(a) for a in sys.argv[1:4])
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
x = torch.randn(1, M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.05
b = torch.randn(N, device="cuda")
out = torch.empty(1, M, N, device="cuda")
packs = ops.pack_linear(w, half=True)
guard = ops.Planes16Guard("cuda")
ops.install_guard(guard)
xp = ops.pairs_from_f32(x)
ops.install_guard(None)
for _ in range(3):
    ops.linear_pairs(xp, packs, N, bias=b, out=out, tile_rows=tile)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["SEGMIF_HIP_LIB"])
buf = np.zeros((1024, 32, 4), dtype=np.uint64)
rc = lib.segmif_debug_pairs_timeline(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
nks = min(K // 16, 28)
t = buf.astype(np.int64)
live = t[:, 29, 0] > 0
t = t[live]
nwg = t.shape[0]
print(f"M {M} N {N} K {K} tile {tile or 'auto'}: {nwg} workgroups stamped, {K // 16} K steps (first {nks} stamped); s_memtime ticks (they advance at about the shader clock on this part: a 254 us kernel spans ~5.6e5 of them), mean / p10 / p90")


def row(name, v):
    v = v.ravel()
    print(f"  {name:44s} {v.mean():8.1f} {np.percentile(v, 10):8.1f} {np.percentile(v, 90):8.1f}")


row("prologue: start -> 2 stages issued", t[:, 29, 1] - t[:, 29, 0])
row("first wait + barrier (prologue latency)", t[:, 0, 1] - t[:, 0, 0])
st = t[:, 1:nks, :]
row("wait + barrier (0-1), steps >= 1", st[:, :, 1] - st[:, :, 0])
row("DMA issue (1-2)", st[:, :, 2] - st[:, :, 1])
row("ds_read + MFMA issue (2-3)", st[:, :, 3] - st[:, :, 2])
row("step period", t[:, 2:nks, 0] - t[:, 1:nks - 1, 0])
row("K loop (first barrier -> done)", t[:, 30, 0] - t[:, 0, 1])
row("closing barrier", t[:, 30, 1] - t[:, 30, 0])
row("epilogue (incl. store drain)", t[:, 31, 0] - t[:, 30, 1])
row("workgroup life", t[:, 31, 0] - t[:, 29, 0])
print(f"  kernel span over the stamped workgroups: {t[:, 31, 0].max() - t[:, 29, 0].min()} ticks; start spread {t[:, 29, 0].max() - t[:, 29, 0].min()}")
print("  (96 MFMAs per workgroup step = 768 CU cycles; two workgroups per CU: 1536 cycles if the matrix pipe were the limit)")
