"""Round-6 fixture: the FLOAT64 evaluation of the mit_b1 64 x 96 pair by the REAL upstream reference.

  pair_b1_64x96_fp64.npz   y_fused, fused, seg, logits of tests/golden/pair_b1_64x96.npz's pair, computed by the reference's own
                           modules cast to double (same key-hash weights, same inputs), plus the ELEMENT-WISE distance of the
                           reference's float32 record from it (ref32_<name>_{ew_max, ew_p999, rms}: |a - b| / |b| over the elements
                           above 1 % of the tensor's range; RMS-relative over all).

Why: every gate so far was a global max-norm against the float32 record.  Read element by element, float32 arithmetic itself is
5e-3 away from the truth on the small values of these tensors (the reference's own record: y_fused 5.3e-3 above the 1 % floor,
p99.9 3.0e-3) - so the element-wise gate of the GPU tests is "no further from the float64 truth than 1.5 x the reference's own
float32 result is", which needs the truth.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r6_truth.py
Container-only (needs /root/reference)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detweights as dw  # noqa: E402
import make_golden as mg  # noqa: E402
import refload  # noqa: E402
import segmif_oracle as so  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
NAMES = ("y_fused", "fused", "seg", "logits")


def elementwise(a, b, floor=1e-2):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    big = b.abs() > floor * b.abs().max()
    ew = ((a - b).abs() / b.abs())[big]
    return float(ew.max()), float(torch.quantile(ew, 0.999)), float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def main():
    torch.manual_seed(0)
    _, _, mf = refload.load_reference()
    g = np.load(os.path.join(OUT, "pair_b1_64x96.npz"))
    ir, vis, mask = (torch.from_numpy(g[k]) for k in ("ir", "vis", "mask"))
    net = mg.quiet(mf.Network3, "mit_b1", mg.NUM_CLASSES, pretrained=None).eval()
    fus = mg.quiet(mf.Fusion_Network3_ac).eval()
    dw.load_det_weights(net, seed=0)
    dw.load_det_weights(fus, seed=0)
    with torch.no_grad():
        r32 = mg.ref_pair_forward(net, fus, ir, vis, mask)
        for k in NAMES:  # the float32 record is reproduced first: the truth below belongs to the same pair
            assert np.array_equal(mg.npy(r32[k]), g[k]), k
        net, fus = net.double(), fus.double()
        # (the colour helper builds float32 constants: evaluate it through the oracle's dtype-generic restatement instead)
        out0, out1 = net.denoise_net.encoder.forward_fusion(mask.double())
        y_f = fus(ir.double(), vis.double(), out0, out1)
        ycc = so.rgb2ycrcb(vis.double())
        fused = so.ycrcb2rgb(torch.cat((y_f, ycc[:, 1:2], ycc[:, 2:3]), dim=1)).clamp(0.0, 1.0).contiguous()
        _, _, seg1 = net.forward(fused)
        logits = torch.nn.functional.interpolate(seg1, size=vis.shape[2:], mode="bilinear", align_corners=False)
        r64 = dict(y_fused=y_f, fused=fused, seg=seg1, logits=logits)
        # the oracle in float64 is the same function (it is what the larger tests use as their truth)
        sd_seg = {k: v.double() if v.is_floating_point() else v for k, v in dw.det_state_dict(so.network3_shapes("mit_b1", 9), seed=0).items()}
        sd_fus = {k: v.double() if v.is_floating_point() else v for k, v in dw.det_state_dict(so.fusion_shapes(), seed=0).items()}
        o64 = so.pair_forward(sd_seg, sd_fus, ir.double(), vis.double(), mask.double(), "mit_b1", return_all=True)
    rec = {}
    for k in NAMES:
        assert r64[k].dtype == torch.float64
        d = float((o64[k] - r64[k]).abs().max() / r64[k].abs().max())
        assert d < 1e-12, (k, d)
        rec[k] = mg.npy(r64[k])
        mx, p999, rms = elementwise(torch.from_numpy(g[k]), r64[k])
        rec[f"ref32_{k}_ew_max"], rec[f"ref32_{k}_ew_p999"], rec[f"ref32_{k}_rms"] = np.float64(mx), np.float64(p999), np.float64(rms)
        print(f"{k}: reference float32 vs its own float64: element-wise max {mx:.3e} p99.9 {p999:.3e} rms-rel {rms:.3e}; oracle fp64 vs reference fp64 {d:.1e}")
    np.savez_compressed(os.path.join(OUT, "pair_b1_64x96_fp64.npz"), **rec)


if __name__ == "__main__":
    main()
