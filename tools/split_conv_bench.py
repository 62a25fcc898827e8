#!/usr/bin/env python
"""The training path's 3x3 convs at 8 x 480 x 640, bf16x6 against f16x3 (csrc/conv3x3_split.hip): the DRDB shapes forward
(64..192 -> 32, dilation 2), the gather-form input gradients (32..160 -> 32 / 64) and conv2 / conv21 (dilation 1)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops
B, H, W = 8, 480, 640
out = {}
for cin, N, dil in ((64, 32, 2), (128, 32, 2), (192, 32, 2), (160, 64, 2), (128, 64, 1), (64, 32, 1)):
    x = torch.rand(B, H, W, cin, device="cuda") - 0.5
    w = (torch.rand(N, cin, 3, 3, device="cuda") - 0.5) * 0.1
    slots = ops.range_slots(6, "cuda")  # as in a DRDB: up to six channel blocks x 8 words handed to one conv
    ops.amax_rows(x, slots[0])
    y = torch.empty(B, H, W, N, device="cuda")
    rec = {}
    fresh = ops.range_slots(16, "cuda")  # a cold (zeroed) report slot per call, as inside a training step
    for name, pk, kw in (("bf16x6", ops.pack_weight_split(w), None), ("f16x3", ops.pack_weight_split16(w), True),
                         ("f16x3, 1-word report", ops.pack_weight_split16(w), 1)):
        for _ in range(3):
            ops.conv2d(x, pk, N, 3, pad=dil, dil=dil, out=y, **({} if kw is None else dict(in_amax=slots.view(-1))))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for it in range(10):
            ops.conv2d(x, pk, N, 3, pad=dil, dil=dil, out=y,
                       **({} if kw is None else dict(in_amax=slots.view(-1), out_amax=fresh[it] if kw is True else fresh[it, :1])))
        b.record(); torch.cuda.synchronize()
        rec[name] = a.elapsed_time(b) / 10
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    fresh.zero_()
    for it in range(10):
        ops.amax_rows(x, fresh[it])  # (cold slot, 8 words)
    t1.record(); torch.cuda.synchronize()
    rec["amax_pass_ms"] = t0.elapsed_time(t1) / 10
    out[f"{cin}->{N} dil{dil}"] = rec
    print(f"{cin:4d} -> {N:3d} dil {dil}:  bf16x6 {rec['bf16x6']:.3f} ms   f16x3 {rec['f16x3']:.3f} ms   (1-word report {rec['f16x3, 1-word report']:.3f} ms; amax pass over the input {rec['amax_pass_ms']:.3f} ms)")
print(json.dumps(out))

# the DRDB backward's gather form as the step issues it: input and output channel slices of wide buffers, residual + ReLU mask in
# the epilogue, range words of several blocks in, a cold report slot out
for cin in (32, 64, 96, 128):
    dzb = torch.rand(B, H, W, 160, device="cuda") * 1e-6
    dbuf = torch.rand(B, H, W, 224, device="cuda") * 1e-6
    buf = (torch.rand(B, H, W, 224, device="cuda") - 0.3).clamp_min(0)
    w = (torch.rand(32, cin, 3, 3, device="cuda") - 0.5) * 0.1
    slots = ops.range_slots(6, "cuda")
    ops.amax_rows(dzb[..., :cin], slots[0])
    fresh = ops.range_slots(16, "cuda")
    rec = {}
    for name, pk in (("bf16x6", ops.pack_weight_split(w)), ("f16x3", ops.pack_weight_split16(w))):
        kw = dict(in_amax=slots.view(-1)) if pk.f16 else {}
        for _ in range(3):
            ops.conv2d(dzb[..., :cin], pk, 32, 3, pad=2, dil=2, res=dbuf[..., 64:96], out=dzb[..., cin:cin + 32], mask=buf[..., 64:96], **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for it in range(10):
            ops.conv2d(dzb[..., :cin], pk, 32, 3, pad=2, dil=2, res=dbuf[..., 64:96], out=dzb[..., cin:cin + 32], mask=buf[..., 64:96],
                       **(dict(kw, out_amax=fresh[it]) if pk.f16 else {}))
        b.record(); torch.cuda.synchronize()
        rec[name] = a.elapsed_time(b) / 10
    print(f"gather {cin:4d} -> 32 dil 2 (+res, mask, slices):  bf16x6 {rec['bf16x6']:.3f} ms   f16x3 {rec['f16x3']:.3f} ms")

# the matching weight gradients (two-team kernel), bf16x6 against f16x3
from segmif_amd import autograd as ag
for cin, N, dil in ((64, 32, 2), (128, 32, 2), (192, 32, 2), (128, 64, 1)):
    x = torch.rand(B, H, W, cin, device="cuda") - 0.5
    dy = (torch.rand(B, H, W, N, device="cuda") - 0.5) * 1e-6
    xs, ys = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.amax_rows(x, xs); ops.amax_rows(dy, ys)
    rec = {}
    for name, am in (("bf16x6", None), ("f16x3", (xs, ys))):
        for _ in range(3):
            ag.conv_wgrad(x, dy, (N, cin, 3, 3), 3, 1, dil, dil, want_bias=True, amax=am)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ag.conv_wgrad(x, dy, (N, cin, 3, 3), 3, 1, dil, dil, want_bias=True, amax=am)
        b.record(); torch.cuda.synchronize()
        rec[name] = a.elapsed_time(b) / 10
    print(f"wgrad {cin:4d} -> {N:3d} dil {dil}:  bf16x6 {rec['bf16x6']:.3f} ms   f16x3 {rec['f16x3']:.3f} ms")
