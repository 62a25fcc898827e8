#!/usr/bin/env python
"""(r5) WHERE does a pair forward amplify the f16x3 convs' operand rounding?  Runs the mit_b1 64x96 image-like x4 batch of
tests/test_gpu_round4.py (pair 0 is the ill-conditioned one: 4.6e-3 on f16x3 against 1.6e-3 with exact-fp32 convs) twice - default
f16x3, and with the 3x3 convs in exact fp32 - and compares, per CrossPath call and per pair, the tensors on the way: the tokens
entering the FeatureFusionModule (what the DRDBs produced), the context-folded end_proj weights (after the 8 x 8 softmax), the
closing LayerNorm's output.  Relative differences (max |a - b| / max |b| per pair).  Run through gpurun."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import detweights as dw  # noqa: E402
from segmif_amd import ops  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.core import model_fusion as mfu  # noqa: E402
from segmif_amd.pipeline import PairForward  # noqa: E402
from test_gpu_round4 import _image_like  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    ops.Planes16Guard.COND_BOUND = math.inf
    seg, fus = Network3("mit_b1", 9, pretrained=None), Fusion_Network3_ac()
    dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda().eval()
    pipe = PairForward(seg, fus)
    ir, vis, mask = (t.cuda() * scale for t in _image_like(3, 64, 96, 11))
    log = []
    fold0, gram0, tail0 = ops.crosspath_fold, ops.crosspath_gram, ops.crosspath_tail

    def fold(part, wkv, wend, weff, wofs, kofs, scale):
        out = fold0(part, wkv, wend, weff, wofs, kofs, scale)
        log.append(("gram partial sum", part.sum(1).clone()))
        log.append((f"weff[kofs={kofs}]", weff[:, :, kofs:kofs + 64].clone()))
        return out

    def tail(x3, xi, *a, **k):
        log.append(("tail in x3", x3.clone()))
        log.append(("tail in xi", xi.clone()))
        k2 = dict(k)
        k2["planes_only"] = False
        out = tail0(x3, xi, *a, **k2)
        log.append(("tail out (LN)", out.clone()))
        return out if not k.get("planes_only") else None

    ops.crosspath_fold, ops.crosspath_tail = fold, tail
    mfu.ops.crosspath_fold, mfu.ops.crosspath_tail = fold, tail
    runs = {}
    with torch.no_grad():
        g = ops.Planes16Guard("cuda", 3)
        prev = ops.install_guard(g)
        try:
            f16 = pipe._eager_body(ir, vis, mask)[0]
        finally:
            ops.install_guard(prev)
        runs["f16x3"] = (log[:], f16)
        del log[:]
        prev = ops.set_conv3x3_mode("fp32")
        try:
            c32 = ops.run_unguarded(lambda: pipe._eager_body(ir, vis, mask), images=0, repeated=0)[0]
        finally:
            ops.set_conv3x3_mode(prev)
        runs["fp32conv"] = (log[:], c32)
    print(f"# image-like x{scale:g}; kappa (interaction 1, 2) {g.kappa().tolist()} -> estimate {g.cond_estimate().tolist()}")
    a, b = runs["f16x3"], runs["fp32conv"]
    print("# per pair: max |f16x3 - fp32conv| / max |fp32conv|")
    assert len(a[0]) == len(b[0]), (len(a[0]), len(b[0]))
    for (na, ta), (nb, tb) in zip(a[0], b[0]):
        d = (ta.double() - tb.double()).abs().flatten(1).max(1).values / tb.double().abs().flatten(1).max(1).values.clamp_min(1e-30)
        print(f"{na:22s} " + "  ".join(f"{v:.2e}" for v in d.tolist()))
    d = (a[1].double() - b[1].double()).abs().flatten(1).max(1).values / b[1].double().abs().max()
    print(f"{'fused image':22s} " + "  ".join(f"{v:.2e}" for v in d.tolist()))


if __name__ == "__main__":
    main()
