#!/usr/bin/env python
"""Chunk timeline of csrc/mixffn.hip built with -DMF_DBG=1 (tools/mixffn_timeline.sh): s_memtime stamps (100 MHz ticks of 10 ns)
of one thread of the first 256 workgroups, per 32-channel chunk:
  0 chunk start | 1 image landed (vmcnt) | 2 past the opening barrier | 3 P1 done (fc1 + Hs written) | 4 past barrier |
  5 P2 done (dwconv + GELU [+ Gs written]) | 6 past barrier | next chunk's 0 = P3 done.   [last][7] = all chunks done.

    SEGMIF_HIP_LIB=$PWD/segmif_amd/lib/variants/lib_mixffn_dbg.so python tools/mixffn_timeline.py 64 [B]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from segmif_amd import ops

C = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = (120, 160) if C == 64 else (60, 80)
hid = 4 * C
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.rand(*s, generator=g) * 2 - 1
x = r(B, H * W, C).cuda()
w1, b1, wd, bd, w2, b2 = r(hid, C) * 0.2, r(hid) * 0.3, r(hid, 1, 3, 3) * 0.5, r(hid) * 0.2, r(C, hid) * 0.1, r(C) * 0.3
ln = (torch.ones(C).cuda(), torch.zeros(C).cuda(), 1e-6)
wimg = ops.pack_mixffn(w1.cuda(), b1.cuda(), ops.pack_dw_weight(wd.cuda()), bd.cuda(), w2.cuda())
guard = ops.Planes16Guard("cuda", B)
guard.slot = lambda images=None: (guard.amax.data_ptr(), 1)
ops.install_guard(guard)
for _ in range(3):
    ops.mixffn_fused(x, ln, wimg, b2.cuda(), H, W)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    ops.mixffn_fused(x, ln, wimg, b2.cuda(), H, W)
e.record()
torch.cuda.synchronize()
print(f"C {C} B {B} {H}x{W}: {s.elapsed_time(e) / 5 * 1e3:.0f} us per launch (instrumented build)")
lib = ctypes.CDLL(os.environ["SEGMIF_HIP_LIB"])
buf = np.zeros((256, 16, 8), dtype=np.uint64)
rc = lib.segmif_debug_mixffn_timeline(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
nch = hid // 32
t = buf[:, :min(nch, 16), :7].astype(np.int64)
names = ["wait image (0-1)", "opening barrier (1-2)", "P1 fc1 -> Hs (2-3)", "barrier (3-4)", "P2 dwconv+GELU (4-5)", "barrier (5-6)"]
seg = np.diff(t, axis=2)
print("per-chunk segment ticks of 10 ns (mean / p10 / p90 over 256 workgroups x chunks)")
for i, n in enumerate(names):
    v = seg[..., i].ravel()
    print(f"  {n:28s} {v.mean():8.1f} {np.percentile(v, 10):8.1f} {np.percentile(v, 90):8.1f}")
p3 = (t[:, 1:, 0] - t[:, :-1, 6]).ravel()
print(f"  {'P3 fc2 (6 - next 0)':28s} {p3.mean():8.1f} {np.percentile(p3, 10):8.1f} {np.percentile(p3, 90):8.1f}")
per = (t[:, 1:, 0] - t[:, :-1, 0]).ravel()
print(f"  {'chunk period':28s} {per.mean():8.1f} {np.percentile(per, 10):8.1f} {np.percentile(per, 90):8.1f}")
last = min(nch, 16) - 1
life = buf[:, last, 7].astype(np.int64) - buf[:, 0, 0].astype(np.int64)
print(f"  chunks 0..{last} of a workgroup: {life.mean():.0f} ticks")
