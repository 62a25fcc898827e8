#!/usr/bin/env python
"""Where does a pair forward's error against the CPU oracle come from on image-like inputs?  Runs the mit_b1 64x96 'x4'
case of tests/test_gpu_round4.py stage by stage (forward_fusion features, fusion output, fused image, logits) under several
arithmetic modes and prints the relative error of each stage.  Run through gpurun."""
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
from segmif_amd.core import Fusion_Network3_ac, Network3, fuse_to_rgb  # noqa: E402
from test_gpu_round4 import _image_like  # noqa: E402


def rel(a, b):
    b = b.double()
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    seg, fus = Network3("mit_b1", 9, pretrained=None), Fusion_Network3_ac()
    sd_seg, sd_fus = dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda().eval()
    ir, vis, mask = (t * scale for t in _image_like(3, 64, 96, 11))
    with torch.no_grad():
        ref = so.pair_forward(sd_seg, sd_fus, ir, vis, mask, "mit_b1", return_all=True)
    print("ref ranges: out0 %.3g out1 %.3g y_f [%.3g, %.3g] fused [%.3g, %.3g] logits %.3g" % (
        ref["out0"].abs().max(), ref["out1"].abs().max(), ref["y_fused"].min(), ref["y_fused"].max(), ref["fused"].min(),
        ref["fused"].max(), ref["logits"].abs().max()))

    def run():
        with torch.no_grad():
            out0, out1 = seg.denoise_net.encoder.forward_fusion(mask.cuda())
            y_f = fus(ir.cuda(), vis.cuda(), out0, out1)
            y_f_ref_in = fus(ir.cuda(), vis.cuda(), ref["out0"].cuda(), ref["out1"].cuda())
            fused = fuse_to_rgb(vis.cuda(), y_f)
            _, _, logits = seg(fused)
            _, _, logits_ref_in = seg(ref["fused"].cuda())
        return {"out0": rel(out0, ref["out0"]), "out1": rel(out1, ref["out1"]), "y_f": rel(y_f, ref["y_fused"]),
                "y_f|ref feats": rel(y_f_ref_in, ref["y_fused"]), "fused": rel(fused, ref["fused"]), "seg": rel(logits, ref["seg"]),
                "seg|ref fused": rel(logits_ref_in, ref["seg"])}

    def show(name, d):
        print(f"{name:34s} " + "  ".join(f"{k} {v:.2e}" for k, v in d.items()), flush=True)

    show("default guarded (f16x3)", ops.run_guarded(run, "cuda"))
    show("unguarded (bf16x6 everywhere)", ops.run_unguarded(run, images=0, repeated=0))
    for what in ("conv3x3", "linear", "attention", "crosspath"):
        setter = getattr(ops, f"set_{what}_mode")
        prev = setter("gemm" if what == "crosspath" else "fp32")
        try:
            show(f"bf16x6 but {what} fp32", ops.run_unguarded(run, images=0, repeated=0))
            show(f"f16x3  but {what} fp32", ops.run_guarded(run, "cuda"))
        finally:
            setter(prev)
    prev = (ops.set_conv3x3_mode("fp32"), ops.set_linear_mode("fp32"), ops.set_attention_mode("fp32"), ops.set_crosspath_mode("gemm"))
    show("all fp32 MFMA", run())


if __name__ == "__main__":
    main()
