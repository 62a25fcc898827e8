"""(r6) Which arithmetic carries the false-negative pair (U[0,1) x 8, cs_*_8 pair 3, 480 x 640, mit_b1)?  Distance of the fused image
from the all-exact-fp32 result (convs / linear / attention fp32, CrossPath in GEMM form) with one family at a time switched."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import detweights as dw  # noqa: E402
from segmif_amd import ops  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.pipeline import PairForward  # noqa: E402

seg, fus = Network3("mit_b1", 9, pretrained=None), Fusion_Network3_ac()
dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
seg, fus = seg.cuda().eval(), fus.cuda().eval()
pipe = PairForward(seg, fus)
H, W = 480, 640
ir = (dw.det_input("cs_ir_8", (8, 1, H, W))[3:4] * 8.0).cuda()
vis = (dw.det_input("cs_vis_8", (8, 3, H, W))[3:4] * 8.0).cuda()
mask = (dw.det_input("cs_mask_8", (8, 1, H, W))[3:4].repeat(1, 3, 1, 1) * 8.0).cuda()


def run(conv, lin, att, cp, guarded=False):
    prev = (ops.set_conv3x3_mode(conv), ops.set_linear_mode(lin), ops.set_attention_mode(att), ops.set_crosspath_mode(cp))
    try:
        with torch.no_grad():
            if guarded:
                g = ops.Planes16Guard("cuda", 1)
                p0 = ops.install_guard(g)
                try:
                    return pipe._eager_body(ir, vis, mask)[0]
                finally:
                    ops.install_guard(p0)
            return ops.run_unguarded(lambda: pipe._eager_body(ir, vis, mask), images=0, repeated=0)[0]
    finally:
        ops.set_conv3x3_mode(prev[0]), ops.set_linear_mode(prev[1]), ops.set_attention_mode(prev[2]), ops.set_crosspath_mode(prev[3])


ref = run("fp32", "fp32", "fp32", "gemm")
rng = float(ref.abs().max())
cases = [("default f16x3 (guarded scope)", ("planes16", "f16x3", "f16x3", "gram"), True),
         ("all bf16x6 (what a range repeat runs)", ("planes", "bf16x6", "bf16x6", "gram"), False),
         ("fp32 convs, rest bf16x6 (the conditioning repeat)", ("fp32", "bf16x6", "bf16x6", "gram"), False),
         ("fp32 convs + fp32 linear", ("fp32", "fp32", "bf16x6", "gram"), False),
         ("fp32 convs + linear + attention, CrossPath gram", ("fp32", "fp32", "fp32", "gram"), False),
         ("bf16x6 convs, fp32 linear + attention, CrossPath gemm", ("planes", "fp32", "fp32", "gemm"), False),
         ("bf16x6 everything, CrossPath gemm", ("planes", "bf16x6", "bf16x6", "gemm"), False),
         ("fp32 convs, bf16x6 linear + attention, CrossPath gemm", ("fp32", "bf16x6", "bf16x6", "gemm"), False)]
for name, modes, guarded in cases:
    out = run(*modes, guarded=guarded)
    print(f"{name:60s} {float((out - ref).abs().max()) / rng:.3e}", flush=True)
