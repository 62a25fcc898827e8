#!/usr/bin/env python
"""Phase timeline of csrc/conv3x3_planes.hip built with -DPLANES_DBG=32 (tools/planes_ablate.sh): s_memtime stamps of wave 0
of each team of every workgroup, per item:  0 LOAD start | 2 DMA issued | 3 own DMA landed | 4 past barrier 1 | 1 epilogue done |
5 MFMA steps done | 6 past barrier 2.   Prints mean / p10 / p90 segment lengths per team.

    SEGMIF_HIP_LIB=segmif_amd/lib/variants/lib_dbg32.so python tools/planes_timeline.py [Cin]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from segmif_amd import ops

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F16 = "f16" in sys.argv[2:]    # (r6) the f16x3 kernels: half-pair planes, SUB = 4 for the plain conv
TAIL = "tail" in sys.argv[2:]  # (r6) the fused tail (conv + 1x1 224 -> 64 + ReLU + residual); Cin is then 192
B, H, W = 8, 480, 640
x = torch.randn(B, H, W, 192, device="cuda")
if F16:
    guard = ops.Planes16Guard("cuda")
    guard.slot = lambda images=None: (guard.amax.data_ptr(), 1)
    pl = ops.Planes(B, H, W, 14, "cuda", guard).load_f32(x)
else:
    pl = ops.Planes(B, H, W, 14, "cuda").load_f32(x)
pack = ops.pack_weight_planes16 if F16 else ops.pack_weight_planes
if TAIL:
    cin = 192
wpl = pack(torch.randn(32, cin, 3, 3, device="cuda") * 0.05)
if TAIL:
    w1 = pack(torch.randn(64, 224, device="cuda") * 0.05)
    b1 = torch.randn(64, device="cuda")
    out = torch.empty(B, H, W, 64, device="cuda")
    x64 = x[..., :64]
    run = lambda: ops.conv3x3_planes(pl, cin, wpl, dil=2, act=1, tail=(w1, b1, None, out, 1, True) if F16 else (w1, b1, x64, out, 1))  # (f16x3: residual from the input planes, as DRDB.forward_planes calls it)
else:
    run = lambda: ops.conv3x3_planes(pl, cin, wpl, dil=2, act=1, out_chunk0=12)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["SEGMIF_HIP_LIB"])
NI = 64
buf = np.zeros((256, 2, NI, 8), dtype=np.uint64)
rc = lib.segmif_debug_planes_timeline(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
t = buf[:, :, 8:56, :7].astype(np.int64)  # steady-state items
t7 = buf[:, :, 8:56, 7].astype(np.int64)
fresh = (np.arange(8, 56) % (cin // 16)) == cin // 16 - 1
d = (t7 - t[..., 4])[:, :, fresh]  # t is re-ordered: index 4 = stamp 1 (MFMA steps done)
print(f"epilogue items only: start -> first sub-tile's activations computed: mean {d.mean():.0f}, whole epilogue {(t[..., 5] - t[..., 4])[:, :, fresh].mean():.0f}")
names = ["DMA issue (0-2)", "own DMA wait (2-3)", "barrier 1 wait (3-4)", "MFMA steps (4-1)", "epilogue (1-5)",
         "barrier 2 wait (5-6)"]
t = t[..., [0, 2, 3, 4, 1, 5, 6]]
seg = np.diff(t, axis=3)
print(f"Cin={cin}: per-item segment ticks (wave 0 of a team; mean / p10 / p90 over 256 workgroups x items 8..55)")
for team in (0, 1):
    print(f" team {team}")
    for i, n in enumerate(names):
        v = seg[:, team, :, i].ravel()
        print(f"  {n:26s} {v.mean():8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
    per = (t[:, team, 1:, 0] - t[:, team, :-1, 0]).ravel()
    print(f"  {'item period (2 phases)':26s} {per.mean():8.0f} {np.percentile(per, 10):8.0f} {np.percentile(per, 90):8.0f}   (MFMA alone per phase: bf16x6 108 x 32 = 3456 ticks; f16x3 SUB=4 108 x 32; f16x3 tail 66 x 32 = 2112)")

if TAIL and hasattr(lib, "segmif_debug_planes_epi_timeline"):
    # (r6) stamps inside the fused tail's epilogue (conv3x3_planes.hip, ETL): where its ticks go, on the items that end a patch
    eb = np.zeros((256, 2, NI, 8), dtype=np.uint64)
    assert lib.segmif_debug_planes_epi_timeline(ctypes.c_void_p(eb.ctypes.data), ctypes.c_size_t(eb.nbytes)) == 0
    e = eb[:, :, 8:56, :7].astype(np.int64)[:, :, fresh, :]
    t1 = buf[:, :, 8:56, 1].astype(np.int64)[:, :, fresh]   # stamp 1: MFMA steps done
    t5 = buf[:, :, 8:56, 5].astype(np.int64)[:, :, fresh]   # stamp 5: epilogue + clear done
    names = ["MFMA steps done -> epilogue entry", "constants in registers (LDS)", "sub-tile 0 activated", "sub-tile 0: split + 1x1 MFMAs",
             "sub-tile 0: out1 values (residual arrived)", "sub-tile 0: stores issued", "sub-tile 1, all of it", "-> accumulators cleared"]
    pts = np.concatenate([t1[..., None], e, t5[..., None]], axis=-1)
    d = np.diff(pts, axis=-1)
    print("fused tail, inside the epilogue (ticks; mean / p10 / p90 over workgroups x patches):")
    for i, n in enumerate(names):
        v = d[..., i].ravel()
        print(f"  {n:44s} {v.mean():8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
