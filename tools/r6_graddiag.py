"""Diagnosis (r6): is the full-size fusion-step gradient record's error deterministic run to run, and does the frozen encoder
pass trip the guard?  Runs tests/test_train_golden.py's full-size step twice and prints the worst tensors."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import detweights as dw  # noqa: E402
from segmif_amd import ops  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.train import FusionTrainer  # noqa: E402
from segmif_amd.utils.optimizer import PolyWarmupAdamW  # noqa: E402
from test_train_golden import fus_kw  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "grads_fusion_step_b3_480x640.npz"))
B, H, W, iter_ = 1, 480, 640, 2
res = []
for rep in range(2):
    net = Network3("mit_b3", 9, pretrained=None)
    dw.load_det_weights(net, seed=0)
    fus = Fusion_Network3_ac()
    dw.load_det_weights(fus, seed=0)
    net, fus = net.cuda().eval(), fus.cuda().eval()
    ir3 = dw.det_input("r5g_ir", (B, 1, H, W)).repeat(1, 3, 1, 1)
    vis3 = dw.det_input("r5g_vis", (B, 3, H, W))
    mask3 = dw.det_input("r5g_mask", (B, 1, H, W)).repeat(1, 3, 1, 1)
    labels = dw.det_labels("r5g_lab", (B, H, W), 9)
    labels[0, 11:40, 100:300] = 255
    opt = PolyWarmupAdamW([{"params": fus.parameters(), "lr": 8e-5 / iter_, "weight_decay": 0.01}], **fus_kw(iter_))
    tr = FusionTrainer(net, fus, opt, torch.nn.CrossEntropyLoss(ignore_index=255), iter_=iter_)
    s0 = ops.range_stats()
    total = tr.step(ir3.cuda(), vis3.cuda(), mask3.cuda(), labels.cuda())
    s1 = ops.range_stats()
    errs = {}
    for n, p in fus.named_parameters():
        if p.grad is None:
            continue
        got = p.grad.detach().double().cpu().reshape(-1)
        head = torch.from_numpy(g[n + "|head"]).double()
        rms = float(g[n + "|norm"]) / max(got.numel(), 1) ** 0.5 + 1e-30
        scale = max(rms, float(head.abs().max()))
        errs[n] = (float((got[:head.numel()] - head).abs().max()) / scale, got.clone())
    res.append(errs)
    top = sorted(((e, n) for n, (e, _) in errs.items()), reverse=True)[:5]
    print("run", rep, "losses", tr.history[0], "trips", {k: s1[k] - s0[k] for k in ("images", "images_repeated", "images_repeated_fp32conv")}, flush=True)
    print("  worst:", [(n, round(e, 6)) for e, n in top], flush=True)
same = all(torch.equal(res[0][n][1], res[1][n][1]) for n in res[0])
print("bitwise equal across the two runs:", same)
