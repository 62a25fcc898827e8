#!/usr/bin/env python
"""(r6, VERDICT r5 item 6 ii) False-negative search for the f16x3 guard's conditioning half at FULL size (480 x 640).

Question: is there a pair that the guard lets THROUGH (no range trip, conditioning estimate <= COND_BOUND) whose f16x3 result is
worse than max(1e-3, 1.5 x the error of the repo's exact-fp32 MFMA path) against the float64 truth?  The float64 oracle costs ~20 s of
CPU per pair at this size, so the search runs in two steps:
  1. every pair: the default guarded arithmetic (f16x3, nothing repeated) against the SAME pair on exact-fp32 MFMA kernels throughout
     (SEGMIF_CONV3X3 = SEGMIF_LINEAR = SEGMIF_ATTENTION = fp32, SEGMIF_CROSSPATH = gemm) -> d = max |f16x3 - fp32| / range of the fused
     image.  With e16 / e32 the two paths' errors against the truth, e16 <= e32 + d; so d <= 1e-4 implies e16 <= max(1e-3, 1.5 e32)
     whatever e32 is (e32 <= 9e-4: e16 <= 1e-3; e32 >= 2e-4: e32 + 1e-4 <= 1.5 e32).  Such a pair cannot be a false negative.
  2. pairs with d > 1e-4 that the guard would NOT repeat: the float64 oracle decides (e16, e32 measured).
Families (>= 200 pairs in total): the bench's generator U[0,1) at exposures x1 x2 x4 x8; image-like inputs (uint8 grid, > 50 % black,
saturated highlights: tests/test_gpu_round4.py) at the same exposures; hash weights and weights whose per-layer scale is drawn
log-uniformly over 1e-3 .. 1e1 (with and without the layer's bias scaled along); mit_b1 (the fusion net - where the context
softmaxes live - is the same for every backbone) and one mit_b3 group.
    python tools/cond_search.py [pairs_per_group=8] > profiles/r06_cond_search.txt"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import detweights as dw  # noqa: E402
import segmif_oracle as so  # noqa: E402
from segmif_amd import ops  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.pipeline import PairForward  # noqa: E402

# (r6 record: run with the round-5 bound to SEE what it let through: SEGMIF_GUARD_COND_BOUND=2e-3; the default bound is the one this search set)
PB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 480, 640
BOUND = ops.Planes16Guard.COND_BOUND


def image_like(B, seed):
    g = torch.Generator().manual_seed(seed)

    def one(c):
        x = torch.rand(B, c, H, W, generator=g)
        x = (torch.nn.functional.avg_pool2d(x, 5, 1, 2) - 0.5) * 8 + 0.5
        x = (x.clamp(0, 1) * 255).floor() / 255
        m = torch.ones(B, 1, H, W)
        m[:, :, : H // 2 + 3, :] = 0
        m[:, :, :, : W // 5] = 0
        return x * m

    return one(1), one(3), one(1).repeat(1, 3, 1, 1)


def uniform(B, seed):
    return (dw.det_input(f"cs_ir_{seed}", (B, 1, H, W)), dw.det_input(f"cs_vis_{seed}", (B, 3, H, W)),
            dw.det_input(f"cs_mask_{seed}", (B, 1, H, W)).repeat(1, 3, 1, 1))


def build(backbone, wseed):
    seg, fus = Network3(backbone, 9, pretrained=None), Fusion_Network3_ac()
    sd_seg, sd_fus = dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
    if wseed:
        g = torch.Generator().manual_seed(100 + wseed)
        for sd, net in ((sd_seg, seg), (sd_fus, fus)):
            layers = sorted({k.rsplit(".", 1)[0] for k in sd if k.endswith(".weight") and sd[k].dim() >= 2})
            for layer in layers:
                s = 10.0 ** float(torch.rand(1, generator=g) * 4 - 3)
                sd[layer + ".weight"] = sd[layer + ".weight"] * s
                if wseed % 2 == 0 and layer + ".bias" in sd:
                    sd[layer + ".bias"] = sd[layer + ".bias"] * s
            net.load_state_dict(sd)
    return seg.cuda().eval(), fus.cuda().eval(), sd_seg, sd_fus


def all_fp32(fn):
    prev = (ops.set_conv3x3_mode("fp32"), ops.set_linear_mode("fp32"), ops.set_attention_mode("fp32"), ops.set_crosspath_mode("gemm"))
    try:
        return ops.run_unguarded(fn, images=0, repeated=0)
    finally:
        ops.set_conv3x3_mode(prev[0]), ops.set_linear_mode(prev[1]), ops.set_attention_mode(prev[2]), ops.set_crosspath_mode(prev[3])


print(f"# f16x3 guard: false-negative search at {H} x {W}, {PB} pairs per group, COND_BOUND = {BOUND:g}, COND_EPS = {ops.Planes16Guard.COND_EPS:g}")
print("# group | pair | range trip | kappa_1 kappa_2 -> estimate | guard | d = max |f16x3 - fp32 MFMA| / range (fused) | labels that differ | e16 / e32 vs float64 (pairs with d > 1e-4 the guard lets through)")
groups = [("mit_b1", 0, "uniform"), ("mit_b1", 0, "image"), ("mit_b3", 0, "image")] + [("mit_b1", ws, "image" if ws % 3 else "uniform") for ws in (1, 2, 3, 4)]
total = passed = flagged = tripped = suspects = false_neg = 0
worst_passed = 0.0
t0 = time.time()
ops.Planes16Guard.COND_BOUND = math.inf  # (measure, do not repeat)
for backbone, wseed, family in groups:
    seg, fus, sd_seg, sd_fus = build(backbone, wseed)
    pipe = PairForward(seg, fus)
    for expo in (1.0, 2.0, 4.0, 8.0):
        ir, vis, mask = (image_like if family == "image" else uniform)(PB, 1000 * wseed + int(expo))
        ir, vis, mask = ir * expo, vis * expo, mask * expo
        with torch.no_grad():
            g = ops.Planes16Guard("cuda", PB)
            prev = ops.install_guard(g)
            try:
                f16, l16 = pipe._eager_body(ir.cuda(), vis.cuda(), mask.cuda())
            finally:
                ops.install_guard(prev)
            f32, l32 = all_fp32(lambda: pipe._eager_body(ir.cuda(), vis.cuda(), mask.cuda()))
        bad = g.tripped()
        kk = g.kappa()
        est = g.cond_estimate(kk)
        finite = torch.isfinite(f32).flatten(1).all(1).cpu()
        rng = f32.abs().flatten(1).max(1).values.clamp_min(1e-30)
        d = ((f16 - f32).abs().flatten(1).max(1).values / rng).cpu()
        ld = (l16 != l32).float().flatten(1).mean(1).cpu()
        for b in range(PB):
            name = f"{backbone}/w{wseed}/{family}/x{expo:g}"
            if not bool(finite[b]):
                print(f"{name} | {b} | - | - | fp32 reference not finite: skipped")
                continue
            total += 1
            sat = not (float(est[b]) <= BOUND)
            verdict = "range-repeat" if bool(bad[b]) else ("cond-repeat" if sat else "pass")
            tripped += bool(bad[b])
            flagged += (not bool(bad[b])) and sat
            extra = ""
            if verdict == "pass":
                passed += 1
                worst_passed = max(worst_passed, float(d[b]))
                if float(d[b]) > 1e-4:  # step 2: the float64 oracle decides
                    suspects += 1
                    with torch.no_grad():
                        sd64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()} for sd in (sd_seg, sd_fus)]
                        truth = so.pair_forward(sd64[0], sd64[1], ir[b:b + 1].double(), vis[b:b + 1].double(), mask[b:b + 1].double(), backbone,
                                                return_all=True)["fused"]
                    sc = float(truth.abs().max())
                    e16 = float((f16[b:b + 1].double().cpu() - truth).abs().max()) / sc
                    e32 = float((f32[b:b + 1].double().cpu() - truth).abs().max()) / sc
                    fn = e16 > max(1e-3, 1.5 * e32)
                    false_neg += fn
                    extra = f" | e16 {e16:.2e} e32 {e32:.2e} {'FALSE NEGATIVE' if fn else 'ok'}"
            print(f"{name} | {b} | {int(bool(bad[b]))} | {float(kk[0, b]):9.3g} {float(kk[1, b]):9.3g} -> {float(est[b]):8.2e} | {verdict} | {float(d[b]):.2e} | "
                  f"{float(ld[b]):.2e}{extra}", flush=True)
    del seg, fus, pipe
    torch.cuda.empty_cache()
print(f"# summary: {total} pairs; guard: {passed} pass, {flagged} repeated for conditioning, {tripped} repeated for range; among the pairs it lets "
      f"through: largest d = {worst_passed:.2e}, {suspects} with d > 1e-4 (sent to the float64 oracle), FALSE NEGATIVES: {false_neg}; {time.time() - t0:.0f} s")
