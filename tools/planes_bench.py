#!/usr/bin/env python
"""DRDB conv micro-benchmark: conv3x3_planes.hip (pre-split activations, LDS-DMA staging) against the round-1
split kernel (tile 14) on the five DRDB shapes, plus the fused 1x1 tail against conv + separate 1x1 GEMM and the
fp32 -> planes converter.  One process, interleaved rounds (median / min).  Run through gpurun."""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops  # noqa: E402


def time_all(fns, rounds=7, iters=5):
    """fns: {name: callable}; interleaved rounds -> {name: (median ms, min ms)}"""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / iters)
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", type=int, default=0, help="one Cin only (PMC passes)")
    ap.add_argument("--kernel", default="both", choices=["both", "planes", "split", "planes16", "planes+16"])
    ap.add_argument("--fill", default="randn", choices=["randn", "zeros", "relu", "ones"], help="activation data (DVFS probe)")
    args = ap.parse_args()
    B, H, W = args.batch, 480, 640
    dev = "cuda"
    buf = torch.randn(B, H, W, 224, device=dev)
    if args.fill == "zeros":
        buf.zero_()
    elif args.fill == "relu":
        buf.relu_()
    elif args.fill == "ones":
        buf.fill_(1.0)
    pl = ops.Planes(B, H, W, 14, dev).load_f32(buf[..., :192])
    guard = ops.Planes16Guard(dev)
    guard.slot = lambda images=None: (guard.amax.data_ptr(), 1)  # a benchmark re-launches forever: one shared slot
    pl16 = ops.Planes(B, H, W, 14, dev, guard).load_f32(buf[..., :192])
    bias = torch.randn(32, device=dev)
    peak = 2500.0 / 6
    for cin in ((args.only,) if args.only else (64, 96, 128, 160, 192)):
        w = torch.randn(32, cin, 3, 3, device=dev) * 0.05
        wsplit, wpl, wpl16 = ops.pack_weight_split(w), ops.pack_weight_planes(w), ops.pack_weight_planes16(w)
        fns = {}
        if args.kernel in ("both", "split"):
            fns["split(r1)"] = lambda: ops.conv2d(buf[..., :cin], wsplit, 32, 3, pad=2, dil=2, bias=bias, act=1, out=buf[..., 192:224])
        if args.kernel in ("both", "planes", "planes+16"):
            fns["planes"] = lambda: ops.conv3x3_planes(pl, cin, wpl, dil=2, bias=bias, act=1, out_chunk0=12)
        if args.kernel in ("planes16", "planes+16"):
            fns["planes16"] = lambda: ops.conv3x3_planes(pl16, cin, wpl16, dil=2, bias=bias, act=1, out_chunk0=12)
        flops = 2.0 * B * H * W * 32 * 9 * cin
        for k, (med, mn) in time_all(fns).items():
            tf = flops / med / 1e9
            print(f"dcov {cin:3d}->32 B{B} {k:10s} median {med:7.3f} ms (min {mn:7.3f})  {tf:6.1f} TF/s = {100 * tf / peak:5.1f}% of 416.7", flush=True)
    if args.only:
        return
    # DRDB tail: Dcov5 + 1x1 (224 -> 64) + ReLU + residual
    cin = 192
    w = torch.randn(32, cin, 3, 3, device=dev) * 0.05
    w1 = torch.randn(64, 224, device=dev) * 0.05
    b1 = torch.randn(64, device=dev)
    wsplit, wpl, w1pl, w1pk = ops.pack_weight_split(w), ops.pack_weight_planes(w), ops.pack_weight_planes(w1), ops.pack_weight(w1)
    x64 = buf[..., :64]
    out = torch.empty(B, H, W, 64, device=dev)

    def r1():
        ops.conv2d(buf[..., :cin], wsplit, 32, 3, pad=2, dil=2, bias=bias, act=1, out=buf[..., 192:224])
        ops.linear(buf, w1pk, 64, bias=b1, act=1, res=x64, out=out)

    wpl16, w1pl16 = ops.pack_weight_planes16(w), ops.pack_weight_planes16(w1)
    fns = {"split + 1x1 gemm (r1)": r1,
           "planes fused tail": lambda: ops.conv3x3_planes(pl, cin, wpl, dil=2, bias=bias, act=1, tail=(w1pl, b1, x64, out, 1)),
           "planes16 fused tail": lambda: ops.conv3x3_planes(pl16, cin, wpl16, dil=2, bias=bias, act=1, tail=(w1pl16, b1, x64, out, 1)),
           "planes16 fused tail, residual from planes": lambda: ops.conv3x3_planes(pl16, cin, wpl16, dil=2, bias=bias, act=1, tail=(w1pl16, b1, None, out, 1, True)),
           "planes_from_f32 64ch": lambda: pl.load_f32(x64, 0),
           "planes16_from_f32 64ch": lambda: pl16.load_f32(x64, 0),
           "1x1 gemm 224->64 alone": lambda: ops.linear(buf, w1pk, 64, bias=b1, act=1, res=x64, out=out)}
    if args.kernel in ("planes16", "planes+16"):
        fns.pop("split + 1x1 gemm (r1)"), fns.pop("1x1 gemm 224->64 alone")
    for k, (med, mn) in time_all(fns).items():
        print(f"tail B{B} {k:26s} median {med:7.3f} ms (min {mn:7.3f})", flush=True)
    # whole DRDB, both paths, through the module
    from segmif_amd.core.model_fusion import DRDB
    blk = DRDB().to(dev).eval()
    x = torch.randn(B, H, W, 64, device=dev)
    big = blk.new_buffer(B, H, W, dev)

    def drdb_r1():
        big[..., :64].copy_(x)
        prev = ops.set_conv3x3_mode("bf16x6")
        blk.forward_buffer(big, out=out)
        ops.set_conv3x3_mode(prev)

    with torch.no_grad():
        fns = {"DRDB bf16x6 (r1)": drdb_r1, "DRDB planes": lambda: blk.forward_planes(x, pl, out=out),
               "DRDB planes16": lambda: blk.forward_planes(x, pl16, out=out)}
        if args.kernel in ("planes16", "planes+16"):
            fns.pop("DRDB bf16x6 (r1)")
        for k, (med, mn) in time_all(fns, rounds=5, iters=2).items():
            print(f"drdb B{B} {k:20s} median {med:7.3f} ms (min {mn:7.3f})", flush=True)


if __name__ == "__main__":
    main()
