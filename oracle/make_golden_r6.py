"""Round-6 fixtures (SURVEY 8(f) N4 remainder): the ablation / variant classes of core/model_fusion.py,
recorded from the REAL upstream reference.

  variants.npz   for each of the reference's variant networks and interaction modules (model_fusion.py:158-1025) one
                 forward at 2 x 24 x 40 with key-hash weights (detweights) and named deterministic inputs:
                   <Class>|out      the network's (first) result
                   <Class>|extra<i> further returned tensors (Fusion_Network3_obtainattention, Fusion_Network_rmseg_att)
                 and the state_dict key / shape table of every class (variants_keys.json).

Inputs are re-derived by name on the test side (detweights.det_input); only outputs are stored.  The classes'
`print(in_ch_, in_ch)` (model_fusion.py:131) is silenced; `.cuda()` in the colour functions is an identity here.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r6.py
Container-only (needs /root/reference)."""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detweights as dw  # noqa: E402
import make_golden_train as mgt  # noqa: E402
import refload  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
B, H, W = 2, 24, 40

# networks with the (ir, vis, out1, out2) signature
NETS4 = ["Fusion_Network3", "Fusion_Network3_Con", "Fusion_Network3_Add", "Fusion_Network3_Average", "Fusion_Network3_S",
         "Fusion_Network3_M", "Fusion_Network3_obtainattention"]
# networks with the (ir, vis) signature
NETS2 = ["Fusion_Network", "Fusion_Network_rmseg", "Fusion_Network_rmseg_att"]
# interaction modules on (B, C, H, W) maps / (B, N, C) tokens, dim 32
FFMS = ["FeatureFusionModule_SoAM", "FeatureFusionModule_MoAM", "FeatureFusionModule_ShowAttention"]
PATHS = ["CrossPath_M", "CrossPath_S", "CrossPath_showAttention"]


def inputs():
    return {"ir": dw.det_input("r6v_ir", (B, 1, H, W)), "vis": dw.det_input("r6v_vis", (B, 3, H, W)),
            "out1": dw.det_input("r6v_out1", (B, 64, H, W), lo=-1.0, hi=1.0),
            "out2": dw.det_input("r6v_out2", (B, 128, H, W), lo=-1.0, hi=1.0),
            "x1": dw.det_input("r6v_x1", (B, 32, H, W), lo=-1.0, hi=1.0), "x2": dw.det_input("r6v_x2", (B, 32, H, W), lo=-1.0, hi=1.0),
            "x3": dw.det_input("r6v_x3", (B, 32, H, W), lo=-1.0, hi=1.0)}


def flat(res):
    if torch.is_tensor(res):
        return [res]
    outs = []
    for r in res:
        outs += flat(r)
    return outs


def main():
    torch.manual_seed(0)
    mt, sh, mf = refload.load_reference()
    inp = inputs()
    rec, keys, raises = {}, {}, {}
    tok = lambda t: t.flatten(2).transpose(1, 2).contiguous()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in NETS4 + NETS2 + FFMS + PATHS + ["AttentionModule", "Network_fused"]:
            cls = getattr(mf, name)
            if name in FFMS:
                net = cls(32)
            elif name in PATHS:
                net = cls(32)
            elif name == "Network_fused":
                net = mgt.quiet(cls, torch.nn.CrossEntropyLoss(ignore_index=255), "mit_b0", 9, pretrained=None)
            else:
                net = mgt.quiet(cls)
            net = net.eval()
            dw.load_det_weights(net, seed=0)
            keys[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
            if name in NETS4:
                res = net(inp["ir"], inp["vis"], inp["out1"], inp["out2"])
            elif name in NETS2:
                try:
                    res = net(inp["ir"], inp["vis"])
                except RuntimeError as e:  # Fusion_Network: conv1 makes 64 channels, its DRDBs take 32 (model_fusion.py:161-163)
                    raises[name] = str(e).splitlines()[0]
                    print(name, "raises:", raises[name])
                    continue
            elif name in FFMS:
                res = net(inp["x1"], inp["x2"], inp["x3"])
            elif name in PATHS:
                res = net(tok(inp["x1"]), tok(inp["x2"]), tok(inp["x3"]))
            elif name == "AttentionModule":
                res = net(inp["x1"])
            else:  # Network_fused: forward(fused) = WeTr(fused) on a 64 x 64 three-channel image; _loss with the CE criterion
                img = dw.det_input("r6v_img", (1, 3, 64, 64))
                lab = dw.det_labels("r6v_lab", (1, 64, 64), 9)
                res = [net(img), net._loss(img, lab)]
            outs = flat(res)
            rec[f"{name}|out"] = mgt.npy(outs[0])
            for i, t in enumerate(outs[1:]):
                rec[f"{name}|extra{i}"] = mgt.npy(t)
            print(name, [tuple(t.shape) for t in outs], float(outs[0].abs().max()))
    np.savez_compressed(os.path.join(OUT, "variants.npz"), **rec)
    with open(os.path.join(OUT, "variants_keys.json"), "w") as f:
        json.dump({"keys": keys, "forward_raises": raises}, f, indent=0, sort_keys=True)
    print("wrote", os.path.join(OUT, "variants.npz"), os.path.getsize(os.path.join(OUT, "variants.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
