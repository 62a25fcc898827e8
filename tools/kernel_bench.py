#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the MI355X: every igemm tile on the hot-path shapes, plus the
bandwidth kernels.  Prints one line per (shape, variant): time, TFLOP/s (fraction of the 157.3
TFLOP/s fp32-MFMA peak) or GB/s (fraction of 8 TB/s).  Run through gpurun."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import _lib, ops  # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def tile_names():
    lib = _lib.load()
    return [lib.segmif_igemm_tile_name(i).decode() for i in range(lib.segmif_igemm_num_tiles())]


def bench_conv(B, H, W, Cin, N, k, pad, dil, tiles, names, cbuf=None, stride=1, iters=10):
    cbuf = cbuf or Cin
    x = torch.randn(B, H, W, cbuf, device="cuda")
    wraw = torch.randn(N, Cin, k, k, device="cuda") * 0.05
    w = ops.pack_weight(wraw)
    wsplit = ops.pack_weight_split(wraw) if 14 in tiles else None
    b = torch.randn(N, device="cuda")
    OH = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    out = torch.empty(B, OH, OW, N, device="cuda")
    flops = 2.0 * B * OH * OW * N * k * k * Cin
    for t in tiles:
        if t >= 0 and names[t].endswith("x32") and Cin % 32:
            continue
        try:
            ms = timeit(lambda: ops.conv2d(x[..., :Cin], wsplit if t == 14 else w, N, k, stride=stride, pad=pad, dil=dil,
                                           bias=b, act=1, out=out, tile=t), iters)
        except RuntimeError as ex:
            print(f"  conv {Cin}->{N} k{k} tile {t}: {ex}")
            continue
        tf = flops / ms / 1e9
        print(f"  conv B{B} {H}x{W} {Cin:4d}->{N:4d} k{k} d{dil} s{stride} tile {names[t] if t >= 0 else 'auto':12s} "
              f"{ms:8.3f} ms {tf:7.1f} TF/s ({100 * tf / PEAK_TF:5.1f}%)", flush=True)


def bench_dense(M, N, K, tiles, names, iters=10, act=0):
    x = torch.randn(M, K, device="cuda")
    w = ops.pack_weight(torch.randn(N, K, device="cuda") * 0.05)
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    flops = 2.0 * M * N * K
    for t in tiles:
        if t >= 0 and names[t].endswith("x32") and K % 32:
            continue
        ms = timeit(lambda: ops.linear(x, w, N, bias=b, act=act, out=out, tile=t), iters)
        tf = flops / ms / 1e9
        print(f"  dense M{M:8d} N{N:5d} K{K:5d} tile {names[t] if t >= 0 else 'auto':12s} {ms:8.3f} ms "
              f"{tf:7.1f} TF/s ({100 * tf / PEAK_TF:5.1f}%)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--convs-only", action="store_true")
    ap.add_argument("--pointwise-only", action="store_true", help="full-resolution 1x1 convs / CrossPath linears")
    ap.add_argument("--drdb-only", type=int, default=0, metavar="CIN", help="one DRDB conv shape, tiles 10 and 14 (PMC passes)")
    args = ap.parse_args()
    lib = _lib.load()
    import ctypes
    buf = ctypes.create_string_buffer(256)
    lib.segmif_device_name(buf, 256)
    print("device:", buf.value.decode())
    names = tile_names()
    B, H, W = args.batch, 480, 640
    if args.drdb_only:
        bench_conv(B, H, W, args.drdb_only, 32, 3, 2, 2, [10, 14], names, cbuf=224)
        return
    if args.pointwise_only:
        M = B * H * W
        bench_dense(M, 64, 224, [-1, 2, 7, 13], names)
        bench_dense(M, 128, 64, [-1, 2, 4, 7, 13], names, act=1)
        bench_dense(M, 64, 128, [-1, 2, 7, 13], names)
        bench_dense(M, 64, 64, [-1, 2, 7, 13], names)
        return
    print("== DRDB dilated 3x3 (reads a 224-pitch concat buffer)")
    for cin in ((64, 192) if args.quick else (64, 96, 128, 160, 192)):
        bench_conv(B, H, W, cin, 32, 3, 2, 2, [10, 14] if args.convs_only else [0, 9, 10, 14], names, cbuf=224)
    print("== fusion-net plain convs")
    bench_conv(B, H, W, 128, 64, 3, 1, 1, [7, 9, 10, 14], names)
    bench_conv(B, H, W, 64, 32, 3, 1, 1, [0, 9, 10, 14], names)
    bench_conv(B, H, W, 32, 1, 3, 1, 1, [0, 9, 10], names)
    bench_conv(B, H, W, 1, 64, 3, 1, 1, [-1], names)
    if args.convs_only:
        return
    print("== fusion-net 1x1 / CrossPath linears (M = B*H*W)")
    M = B * H * W
    bench_dense(M, 64, 224, [2, 3, 7, 8], names)
    bench_dense(M, 128, 64, [2, 3, 4, 5, 7, 8], names, act=1)
    bench_dense(M, 64, 128, [2, 3, 7, 8], names)
    print("== MiT encoder GEMMs (B = 8 images of 480x640)")
    for (m, n, k) in ((8 * 19200, 64, 64), (8 * 19200, 256, 64), (8 * 19200, 64, 256), (8 * 4800, 128, 128),
                      (8 * 4800, 512, 128), (8 * 4800, 128, 512), (8 * 1200, 320, 320), (8 * 1200, 1280, 320),
                      (8 * 1200, 320, 1280), (8 * 300, 512, 512), (8 * 300, 2048, 512), (8 * 300, 512, 2048),
                      (8 * 300, 640, 320), (8 * 19200, 256, 1024)):
        bench_dense(m, n, k, [-1, 2, 3, 4, 5, 6, 7, 8] if not args.quick else [-1], names)
    print("== patch-embed / sr convs (B = 8)")
    bench_conv(8, 480, 640, 3, 64, 7, 3, 1, [-1], names, stride=4)
    bench_conv(8, 120, 160, 64, 128, 3, 1, 1, [-1, 2, 3, 4, 5, 6], names, stride=2)
    bench_conv(8, 120, 160, 64, 64, 8, 0, 1, [-1, 2, 3, 6], names, stride=8)
    bench_conv(8, 30, 40, 320, 320, 2, 0, 1, [-1, 2, 3, 4, 6], names, stride=2)
    print("== fused sr-attention")
    for (b, h, n, nk, hd) in ((8, 1, 19200, 300, 64), (8, 2, 4800, 300, 64), (8, 5, 1200, 300, 64),
                              (8, 8, 300, 300, 64), (2, 1, 65536, 1024, 64)):
        C = h * hd
        q = torch.randn(b, n, C, device="cuda")
        kv = torch.randn(b, nk, 2 * C, device="cuda")
        ms = timeit(lambda: ops.sr_attention(q, kv, h, hd ** -0.5))
        tf = 4.0 * b * h * n * nk * hd / ms / 1e9
        print(f"  attn B{b} h{h} N{n} Nk{nk} hd{hd}: {ms:8.3f} ms {tf:7.1f} TF/s ({100 * tf / PEAK_TF:5.1f}%)", flush=True)
    print("== bandwidth kernels (B = 8)")
    for (rows, C) in ((8 * 19200, 64), (8 * 4800, 128), (8 * 1200, 320), (4 * 307200, 64)):
        x = torch.randn(rows, C, device="cuda")
        g = torch.ones(C, device="cuda")
        y = torch.empty_like(x)
        ms = timeit(lambda: ops.layernorm(x, g, g, 1e-6, out=y))
        gbs = 2 * rows * C * 4 / ms / 1e6
        print(f"  layernorm rows {rows} C {C}: {ms:8.3f} ms {gbs:7.0f} GB/s ({100 * gbs / PEAK_GBS:5.1f}%)")
    for (b, h, w, C) in ((8, 120, 160, 256), (8, 60, 80, 512), (8, 30, 40, 1280), (8, 15, 20, 2048)):
        x = torch.randn(b, h * w, C, device="cuda")
        w9 = torch.randn(9, C, device="cuda")
        bias = torch.randn(C, device="cuda")
        ms = timeit(lambda: ops.dwconv3x3_gelu(x, w9, bias, h, w))
        gbs = 2 * x.numel() * 4 / ms / 1e6
        print(f"  dwconv+gelu B{b} {h}x{w} C{C}: {ms:8.3f} ms {gbs:7.0f} GB/s ({100 * gbs / PEAK_GBS:5.1f}%)")
    for (b, ih, iw, oh, ow, C) in ((4, 120, 160, 480, 640, 64), (4, 60, 80, 480, 640, 128), (8, 15, 20, 120, 160, 256)):
        x = torch.randn(b, ih, iw, C, device="cuda")
        y = torch.empty(b, oh, ow, C, device="cuda")
        ms = timeit(lambda: ops.bilinear(x, oh, ow, out=y))
        gbs = (x.numel() + y.numel()) * 4 / ms / 1e6
        print(f"  bilinear B{b} {ih}x{iw}->{oh}x{ow} C{C}: {ms:8.3f} ms {gbs:7.0f} GB/s ({100 * gbs / PEAK_GBS:5.1f}%)")
    kv = torch.randn(B, H * W, 128, device="cuda")
    ms = timeit(lambda: ops.linattn_partial(kv))
    gbs = kv.numel() * 4 / ms / 1e6
    print(f"  linattn_partial B{B} N{H * W}: {ms:8.3f} ms {gbs:7.0f} GB/s ({100 * gbs / PEAK_GBS:5.1f}%)")


if __name__ == "__main__":
    main()
