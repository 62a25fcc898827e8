#!/usr/bin/env python
"""Step timeline of csrc/gemm_split.hip built with -DGEMM_DBG=1 (tools/gemm_timeline.sh): s_memtime stamps (100 MHz) of thread 0
of the first 1024 workgroups, per K step:  0 step start | 1 MFMAs issued | 2 past barrier 1 | 3 A split + stored | 4 W DMA issued |
5 DMA landed | 6 past barrier 2;  [15][7] = K loop done, [14][7] = epilogue done.

    SEGMIF_HIP_LIB=$PWD/segmif_amd/lib/variants/lib_gemm_dbg.so python tools/gemm_timeline.py M N K
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from segmif_amd import ops

M, N, K = (int(a) for a in sys.argv[1:4])
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.05
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
packs = ops.pack_linear(w)
for _ in range(3):
    ops.linear_auto(x, packs, N, bias=b, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["SEGMIF_HIP_LIB"])
buf = np.zeros((1024, 16, 8), dtype=np.uint64)
rc = lib.segmif_debug_gemm_timeline(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
nks = K // 32
nwg = min(1024, ((M + 127) // 128) * ((N + 127) // 128))
t = buf[:nwg, :min(nks, 14), :7].astype(np.int64)
names = ["issue loads + MFMAs (0-1)", "barrier 1 wait (1-2)", "A split + store (2-3)", "W DMA issue (3-4)", "DMA wait (4-5)",
         "barrier 2 wait (5-6)"]
seg = np.diff(t[:, :-1], axis=2)  # the last step has no staging
print(f"M {M} N {N} K {K}: per-step segment ticks of 10 ns (thread 0; mean / p10 / p90 over {nwg} workgroups x {t.shape[1] - 1} steps)")
for i, n in enumerate(names):
    v = seg[..., i].ravel()
    print(f"  {n:28s} {v.mean():8.1f} {np.percentile(v, 10):8.1f} {np.percentile(v, 90):8.1f}")
per = (t[:, 1:, 0] - t[:, :-1, 0]).ravel()
print(f"  {'step period':28s} {per.mean():8.1f} {np.percentile(per, 10):8.1f} {np.percentile(per, 90):8.1f}   (48 MFMAs x 32 cycles = 1536 cycles = 64-80 ticks)")
life = buf[:nwg, 14, 7].astype(np.int64) - buf[:nwg, 0, 0].astype(np.int64)
epi = buf[:nwg, 14, 7].astype(np.int64) - buf[:nwg, 15, 7].astype(np.int64)
print(f"  K loop + epilogue per workgroup: {life.mean():.0f} ticks, epilogue alone {epi.mean():.0f}")
start = buf[:nwg, 0, 0].astype(np.int64)
print(f"  first step start spread over workgroups: {start.max() - start.min()} ticks; kernel span {buf[:nwg, 14, 7].max() - start.min()} ticks")
