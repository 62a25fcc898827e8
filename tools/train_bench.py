#!/usr/bin/env python
"""Timing of the two training steps (configs[2] / [3] of BASELINE.json) on synthetic data.
  python tools/train_bench.py --step seg|fusion [--batch 8] [--backbone mit_b3]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...  (data parallel)
Eval-mode regime for the segmentation net; loss terms of the fusion step run on torch-ROCm ops
(segmif_amd/losses.py, §8(f) N1)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import detweights as dw
from segmif_amd import dist
from segmif_amd.core import Fusion_Network3_ac, Network3
from segmif_amd.parallel import GradAllReducer
from segmif_amd.train import FusionTrainer, GraphedSegTrainStep, seg_train_step
from segmif_amd.utils.optimizer import PolyWarmupAdamW, PolyWarmupAdamW_seg

ap = argparse.ArgumentParser()
ap.add_argument("--step", default="seg", choices=["seg", "fusion"])
ap.add_argument("--backbone", default="mit_b3"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--height", type=int, default=480); ap.add_argument("--width", type=int, default=640)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--graph", action="store_true", help="seg step: forward + backward replayed from a hipGraph")
ap.add_argument("--train-mode", action="store_true", help="module.train(): DropPath, Dropout2d, BatchNorm batch statistics")
a = ap.parse_args()
rank, local_rank, world = dist.env_world()
torch.cuda.set_device(local_rank)
dist.init()
B, H, W = a.batch, a.height, a.width
seg = Network3(a.backbone, 9, pretrained=None); dw.load_det_weights(seg, seed=0); seg = seg.cuda().eval()
crit = torch.nn.CrossEntropyLoss(ignore_index=255)
labels = dw.det_labels(f"trb_y{rank}", (B, H, W), 9).cuda()
if a.step == "seg":
    g = seg.denoise_net.get_param_groups()
    opt = PolyWarmupAdamW_seg([{"params": g[0], "lr": 8e-5, "weight_decay": 0.01}, {"params": g[1], "lr": 8e-5, "weight_decay": 0.0},
                               {"params": g[2], "lr": 8e-4, "weight_decay": 0.01}], lr=8e-5, weight_decay=0.01, betas=(0.9, 0.999),
                              iter_curr=0, warmup_iter=3000, max_iter=80000, warmup_ratio=1e-6, power=1.0)
    red = GradAllReducer([p for grp in g for p in grp]) if world > 1 else None
    x = dw.det_input(f"trb_x{rank}", (B, 3, H, W)).cuda()
    if a.train_mode:
        seg.train()
    if a.graph:
        gstep = GraphedSegTrainStep(seg, opt, crit, x, labels, reducer=red)
        step = lambda: gstep()
    else:
        step = lambda: seg_train_step(seg, opt, x, labels, crit, red)
    gflop = {"mit_b3": 300.4, "mit_b1": 110.0}.get(a.backbone, 0)
else:
    fus = Fusion_Network3_ac(); dw.load_det_weights(fus, seed=0); fus = fus.cuda()
    opt = PolyWarmupAdamW([{"params": fus.parameters(), "lr": 1e-4 / 2, "weight_decay": 0.01}], lr=3e-4 / 2, weight_decay=0.01,
                          betas=(0.9, 0.999), warmup_iter=3e-5 / 2, max_iter=80000, warmup_ratio=1e-6, power=1.0)
    red = GradAllReducer(list(fus.parameters())) if world > 1 else None
    tr = FusionTrainer(seg, fus, opt, crit, iter_=2, reducer=red)
    ir = dw.det_input(f"trb_ir{rank}", (B, 3, H, W)).cuda(); vis = dw.det_input(f"trb_vis{rank}", (B, 3, H, W)).cuda()
    mask = dw.det_input(f"trb_m{rank}", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    step = lambda: tr.step(ir, vis, mask, labels)
    gflop = {"mit_b3": 2304.0, "mit_b1": 2000.0}.get(a.backbone, 0)
for _ in range(a.warmup): l = step()
dist.fence(); t0 = time.perf_counter()
for _ in range(a.steps): l = step()
dist.fence(); dt = dist.max_over_ranks((time.perf_counter() - t0) / a.steps)
if rank == 0:
    print(json.dumps({"what": f"{a.step}-train step (fwd+loss+bwd+AdamW), " + ("train mode" if a.train_mode else "seg net in eval-mode regime") + (", hipGraph replay" if a.graph else ""), "backbone": a.backbone,
                      "n_gpus": world, "batch_per_gpu": B, "ms_per_step": 1e3 * dt, "samples_per_s": world * B / dt,
                      "approx_tflops_per_gpu": gflop * B / dt / 1e3, "loss": float(l),
                      "grad_bytes_exchanged": red.gradient_bytes() if red else 0,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
dist.shutdown()
