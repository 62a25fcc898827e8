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
ap.add_argument("--native-sites", default="", help="write a table of the torch (aten) ops one step still issues, by call site, to this file")
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
if a.native_sites and rank == 0:
    # which Python lines still hand device work to torch's own kernels (at::native): every aten op of ONE step, grouped by
    # (op, innermost segmif_amd frame), ranked by the bytes its tensors span.  Views and metadata ops move nothing and are skipped.
    import collections, traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils._pytree import tree_flatten
    SKIP = ("view", "reshape", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
            "as_strided", "t.default", "size", "stride", "is_", "_unsafe_view", "unbind", "split", "narrow", "chunk", "sym_",
            "empty", "_local_scalar_dense", "unfold", "lift_fresh", "record_stream", "numel", "dim", "storage_offset")
    agg = collections.defaultdict(lambda: [0, 0])

    class Sites(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            if not any(k in name for k in SKIP):
                nbytes = sum(t.numel() * t.element_size() for t in tree_flatten((args, kwargs, out))[0]
                             if isinstance(t, torch.Tensor) and t.is_cuda)
                site = "autograd engine (no Python frame)"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if "segmif_amd" in fr.filename and "train_bench" not in fr.filename:
                        site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                        break
                e = agg[(name, site)]
                e[0] += 1
                e[1] += nbytes
            return out

    with Sites():
        l = step()
    torch.cuda.synchronize()
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(a.native_sites, "w") as f:
        f.write(f"# aten ops of one {a.step} step ({'train' if a.train_mode else 'eval-regime'} mode, batch {B}) by call site; MB = bytes spanned by the op's device tensors\n")
        f.write(f"# total {sum(v[0] for _, v in rows)} ops, {sum(v[1] for _, v in rows) / 1e6:.0f} MB\n")
        for (name, site), (n, nb) in rows:
            f.write(f"{n:6d} {nb / 1e6:10.1f} MB  {name:44s} {site}\n")
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
