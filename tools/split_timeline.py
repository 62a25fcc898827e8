#!/usr/bin/env python
"""Phase timeline of the bf16x6 3x3 conv (csrc/conv3x3_split.hip built with -DSPLIT_DBG=32 by
tools/split_ablate.sh tl:32): s_memtime stamps of wave 0 of the first 2048 workgroups at
  0 chunk start | 1 LDS stores issued | 2 past barrier 1 | 3 next-chunk loads + first fragments issued |
  4 before the first split step | 5 MFMA steps issued | 6 past barrier 2
Prints, for workgroups in the middle of the launch, the average cycles of each segment, and how the
two workgroups sharing a CU overlap (fraction of a workgroup's MFMA segment during which its partner
is in its own MFMA segment).

    SEGMIF_HIP_LIB=segmif_amd/lib/dbg/libsegmif_hip_tl.so python tools/split_timeline.py [Cin]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from segmif_amd import _lib, ops

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B, H, W = 8, 480, 640
x = torch.randn(B, H, W, 224, device="cuda")
w = torch.randn(32, cin, 3, 3, device="cuda") * 0.05
sw = ops.pack_weight_split(w)
out = torch.empty(B, H, W, 32, device="cuda")
for _ in range(3):
    ops.conv2d(x[..., :cin], sw, 32, 3, pad=2, dil=2, act=1, out=out)
torch.cuda.synchronize()
lib = _lib.load()
NB, NC = 2048, 12
buf = np.zeros((NB, NC + 1, 8), dtype=np.uint64)
rc = lib.segmif_debug_split_timeline(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
nch = cin // 16
t = buf[:, :nch, :7].astype(np.int64)
hw, xcc = buf[:, NC, 0], buf[:, NC, 1]
cu_key = (xcc & 0xf) * 4096 + ((hw >> 8) & 0xf) * 16 + ((hw >> 12) & 0x1) * 256 + ((hw >> 13) & 0x7) * 512  # xcc, cu, sh, se
names = ["LDS stores (0-1)", "barrier 1 wait (1-2)", "issue loads+frags (2-3)", "MFMA steps w/o split (3-4)",
         "MFMA steps with split (4-5)", "barrier 2 wait (5-6)"]
sel = slice(600, 1600)
seg = np.diff(t[sel], axis=2)  # (blocks, chunks, 6)
print(f"Cin={cin}: {nch} chunks; per-chunk segment cycles (wave 0, mean over workgroups {sel.start}..{sel.stop}, chunks 1..)")
for i, n in enumerate(names):
    print(f"  {n:32s} {seg[:, 1:, i].mean():9.0f}")
print(f"  {'chunk total':32s} {(t[sel, 1:, 6] - t[sel, 1:, 0]).mean():9.0f}   (MFMA alone: 108 x 32 = 3456)")
# partner overlap: blocks on the same CU alive at the same time
order = np.argsort(cu_key[sel], kind="stable")
keys = cu_key[sel][order]
ov = []
for k in np.unique(keys):
    idx = np.arange(sel.start, sel.stop)[order[keys == k]]
    for a in idx:
        for b in idx:
            if a == b:
                continue
            a0, a1, b0, b1 = t[a, 0, 0], t[a, nch - 1, 6], t[b, 0, 0], t[b, nch - 1, 6]
            if min(a1, b1) - max(a0, b0) < 0.8 * (a1 - a0):
                continue  # not co-resident for most of a's life
            tot = both = 0
            for c in range(1, nch):
                s0, s1 = t[a, c, 3], t[a, c, 5]
                tot += s1 - s0
                for c2 in range(nch):
                    both += max(0, min(s1, t[b, c2, 5]) - max(s0, t[b, c2, 3]))
            ov.append(both / max(tot, 1))
if ov:
    print(f"co-resident pairs found: {len(ov)}; partner is in ITS MFMA segment during {100 * np.mean(ov):.0f} % of a workgroup's MFMA segment")
else:
    print("no co-resident pairs identified (HW_ID decode)")
