#!/usr/bin/env python
"""Timing of the segmentation training step (train.py:217-227 in its eval-mode regime):
forward + CE + backward + PolyWarmupAdamW_seg on synthetic data, one MI355X."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import detweights as dw
from segmif_amd.core import Network3
from segmif_amd.utils.optimizer import PolyWarmupAdamW_seg

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="mit_b3"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--height", type=int, default=480); ap.add_argument("--width", type=int, default=640)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
net = Network3(a.backbone, 9, pretrained=None); dw.load_det_weights(net, seed=0); net = net.cuda().eval()
g = net.denoise_net.get_param_groups()
opt = PolyWarmupAdamW_seg([{"params": g[0], "lr": 8e-5, "weight_decay": 0.01}, {"params": g[1], "lr": 8e-5, "weight_decay": 0.0},
                           {"params": g[2], "lr": 8e-4, "weight_decay": 0.01}], lr=8e-5, weight_decay=0.01, betas=(0.9, 0.999),
                          iter_curr=0, warmup_iter=3000, max_iter=80000, warmup_ratio=1e-6, power=1.0)
x = dw.det_input("trb_x", (a.batch, 3, a.height, a.width)).cuda()
y = dw.det_labels("trb_y", (a.batch, a.height, a.width), 9).cuda()
crit = torch.nn.CrossEntropyLoss(ignore_index=255)
def step():
    opt.zero_grad(set_to_none=True)
    loss = net._loss(x, y, crit); loss.backward(); opt.step(); return loss
for _ in range(a.warmup): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
gf = {"mit_b3": 300.4, "mit_b1": 110.0}.get(a.backbone, 0) * a.batch  # ~3 x (encoder + head) GFLOP per image
print(json.dumps({"what": "seg-train step (fwd+CE+bwd+AdamW), eval-mode regime", "backbone": a.backbone, "batch": a.batch,
                  "ms_per_step": 1e3 * dt, "images_per_s": a.batch / dt, "approx_tflops": gf / dt / 1e3,
                  "loss": float(l.detach()), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
