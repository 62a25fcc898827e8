#!/usr/bin/env python
"""(r5) Calibration of the f16x3 guard's conditioning bound (ops.Planes16Guard.COND_BOUND).

For several input families (mit_b1, 64x96 - the size at which the CPU oracle in float64 takes seconds) prints, PER PAIR: the
conditioning figure kappa (how far a context softmax moves per unit relative perturbation of its Gram matrix under a fixed +-1
pattern: csrc/crosspath.hip) its CrossPath context softmaxes reported (csrc/crosspath.hip,
crosspath_fold_kernel), and the error of the fused image against the oracle evaluated in float64 (the truth) for
  f16x3 (default, no repeat) | bf16x6 | f16x3 but 3x3 convs in exact fp32 (what a saturated pair is repeated with) |
  everything on exact-fp32 MFMA | the oracle in float32 (= the reference's own arithmetic).
Errors are max |a - truth| over the pair's image / max |truth| over the batch (the tests' norm).  Also prints kappa of the
bench workload's inputs (mit_b3, 480x640, U[0,1)), for which there is no float64 truth - only to show where the default
workload sits relative to the bound.  Run through gpurun; the output is committed as profiles/r05_cond_calibration.txt."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import detweights as dw  # noqa: E402
import segmif_oracle as so  # noqa: E402
from segmif_amd import ops  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.pipeline import PairForward  # noqa: E402
from test_gpu_round4 import _image_like, _inputs  # noqa: E402


def per_pair_err(t, truth):
    d = (t.double().cpu() - truth).abs().flatten(1).max(1).values
    return (d / (truth.abs().max() + 1e-30)).tolist()


def main():
    ops.Planes16Guard.COND_BOUND = math.inf  # observe, never repeat
    seg, fus = Network3("mit_b1", 9, pretrained=None), Fusion_Network3_ac()
    sd_seg, sd_fus = dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda().eval()
    pipe = PairForward(seg, fus)
    sd64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()} for sd in (sd_seg, sd_fus)]
    img = _image_like(3, 64, 96, 11)
    fams = [("U[0,1) det", _inputs(3, 64, 96, 1), 1.0)]
    fams += [(f"image-like x{s:g}", img, s) for s in (1.0, 1.5, 2.0, 3.0, 4.0, 6.0)]
    fams += [("U[0,1) det x3", _inputs(3, 64, 96, 2), 3.0)]
    fams += [(f"image-like(seed 12) x{s:g}", _image_like(3, 64, 96, 12), s) for s in (2.0, 4.0)]
    print("# family | pair | kappa_1 kappa_2 -> estimate | err f16x3 | err bf16x6 | err f16x3+fp32 convs | err all-fp32-MFMA | err oracle fp32 (reference arithmetic)")
    for name, (ir, vis, mask), s in fams:
        ir, vis, mask = ir * s, vis * s, mask * s
        with torch.no_grad():
            ref = so.pair_forward(sd_seg, sd_fus, ir, vis, mask, "mit_b1", return_all=True)
            truth = so.pair_forward(sd64[0], sd64[1], ir.double(), vis.double(), mask.double(), "mit_b1", return_all=True)["fused"]
            g = ops.Planes16Guard("cuda", ir.shape[0])
            prev = ops.install_guard(g)
            try:
                f16 = pipe._eager_body(ir.cuda(), vis.cuda(), mask.cuda())[0]
            finally:
                ops.install_guard(prev)
            kap = g.kappa()
            est = g.cond_estimate(kap).tolist()
            kappa = [f"{kap[0, b]:8.3g} {kap[1, b]:8.3g} -> {est[b]:8.2e}" for b in range(ir.shape[0])]
            tripped = g.tripped().tolist()
            b6 = ops.run_unguarded(lambda: pipe._eager_body(ir.cuda(), vis.cuda(), mask.cuda()), images=0, repeated=0)[0]
            prev = ops.set_conv3x3_mode("fp32")
            try:
                c32 = ops.run_unguarded(lambda: pipe._eager_body(ir.cuda(), vis.cuda(), mask.cuda()), images=0, repeated=0)[0]
            finally:
                ops.set_conv3x3_mode(prev)
            prev = (ops.set_conv3x3_mode("fp32"), ops.set_linear_mode("fp32"), ops.set_attention_mode("fp32"))
            try:
                a32 = pipe._eager_body(ir.cuda(), vis.cuda(), mask.cuda())[0]
            finally:
                ops.set_conv3x3_mode(prev[0]), ops.set_linear_mode(prev[1]), ops.set_attention_mode(prev[2])
        cols = [per_pair_err(t, truth) for t in (f16, b6, c32, a32, ref["fused"])]
        for b in range(ir.shape[0]):
            print(f"{name:24s} | {b} | {kappa[b]} | " + " | ".join(f"{c[b]:.2e}" for c in cols) + (" | RANGE-TRIPPED" if tripped[b] else ""),
                  flush=True)
    # the bench workload: mit_b3, 480x640, U[0,1) inputs (bench.py's generator) - kappa only
    del seg, fus, pipe
    seg, fus = Network3("mit_b3", 9, pretrained=None), Fusion_Network3_ac()
    dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda().eval()
    pipe = PairForward(seg, fus)
    B = 4
    for name, s in (("bench inputs (mit_b3 480x640 U[0,1))", 1.0), ("the same x2", 2.0), ("the same x4", 4.0)):
        gen = torch.Generator(device="cuda").manual_seed(0)
        ir = torch.rand((B, 1, 480, 640), device="cuda", generator=gen) * s
        vis = torch.rand((B, 3, 480, 640), device="cuda", generator=gen) * s
        mask = torch.rand((B, 1, 480, 640), device="cuda", generator=gen).repeat(1, 3, 1, 1) * s
        with torch.no_grad():
            g = ops.Planes16Guard("cuda", B)
            prev = ops.install_guard(g)
            try:
                pipe._eager_body(ir, vis, mask)
            finally:
                ops.install_guard(prev)
        print(f"# {name}: estimate per pair {[float('%.3g' % k) for k in g.cond_estimate().tolist()]}  kappa_1 {[float('%.3g' % k) for k in g.kappa()[0].tolist()]} "
              f"kappa_2 {[float('%.3g' % k) for k in g.kappa()[1].tolist()]}  range-tripped {g.tripped().tolist()}", flush=True)


if __name__ == "__main__":
    main()
