#!/usr/bin/env python
"""Encoder GEMM shapes (B = 32 images of 480x640, mit_b3): the split GEMM (csrc/gemm_split.hip) in its bf16x6 and f16x3 forms
against the fp32 MFMA tiles igemm picks on its own.  Interleaved rounds, median.  Run through gpurun.
    python tools/gemm_bench.py [B] [--no-fp32]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmif_amd import ops  # noqa: E402


def t(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


NO32 = "--no-fp32" in sys.argv
if NO32:
    sys.argv.remove("--no-fp32")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tot = {"fp32": 0.0, "bf16x6": 0.0, "f16x3": 0.0, "pairs": 0.0}
guard = ops.Planes16Guard("cuda")
guard.slot = lambda images=None: (guard.amax.data_ptr(), 1)  # a benchmark re-launches forever: one shared slot


def in_scope(fn):
    ops.install_guard(guard)
    try:
        fn()
    finally:
        ops.install_guard(None)


for name, tok, C, reps in (("stage1", B * 19200, 64, 3), ("stage2", B * 4800, 128, 4), ("stage3", B * 1200, 320, 18), ("stage4", B * 300, 512, 3)):
    for lname, N, K in (("q/proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        x = torch.randn(tok, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(tok, N, device="cuda")
        packs = ops.pack_linear(w, half=True)
        res = {"fp32": [], "bf16x6": [], "f16x3": [], "pairs": []}
        xp = None
        if packs[1] is not None and packs[1].pairs is not None:  # (r5) gemm_pairs: A pre-split once by its producer
            ops.install_guard(guard)
            xp = ops.pairs_from_f32(x.view(1, tok, K))
            ops.install_guard(None)
        outp = out.view(1, tok, N)
        for _ in range(3 if NO32 else 5):
            res["fp32"].append(0.0 if NO32 else t(lambda: ops.linear(x, packs[0], N, bias=b, out=out)))
            res["bf16x6"].append(t(lambda: ops.linear_auto(x, packs, N, bias=b, out=out)))
            res["f16x3"].append(t(lambda: in_scope(lambda: ops.linear_auto(x, packs, N, bias=b, out=out))))
            res["pairs"].append(t(lambda: ops.linear_pairs(xp, packs, N, bias=b, out=outp)) if xp is not None else float("nan"))
        m32, m16, mh = (statistics.median(res[k]) for k in ("fp32", "bf16x6", "f16x3"))
        mp = statistics.median(res["pairs"])
        gf = 2.0 * tok * N * K / 1e9
        mult = reps * (2 if lname == "q/proj" else 1)
        tot["fp32"] += m32 * mult
        tot["bf16x6"] += m16 * mult
        tot["f16x3"] += mh * mult
        tot["pairs"] += (mp if mp == mp else mh) * mult
        print(f"{name} {lname:7s} M{tok:7d} N{N:5d} K{K:5d}: fp32 {m32:7.3f} ms ({gf / max(m32, 1e-9):6.1f} TF/s)   bf16x6 {m16:7.3f} ms "
              f"({gf / m16:6.1f} TF/s)   f16x3 {mh:7.3f} ms ({gf / mh:6.1f} TF/s)   pairs {mp:7.3f} ms ({gf / mp:6.1f} TF/s)", flush=True)
print(f"per encoder pass (q, proj, fc1, fc2 of every block): fp32 {tot['fp32']:.2f} ms, bf16x6 {tot['bf16x6']:.2f} ms, "
      f"f16x3 {tot['f16x3']:.2f} ms, pairs (gemm_split where N < 128) {tot['pairs']:.2f} ms")
