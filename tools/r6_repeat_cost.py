"""(r6) What a conditioning repeat costs: the 64-pair step on the bench inputs of a rank whose batch holds flagged pairs, under two
bounds (per-step wall ms, pairs repeated per step)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import detweights as dw  # noqa: E402
from segmif_amd import ops  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.pipeline import PairForward  # noqa: E402

B, H, W = 64, 480, 640
seg, fus = Network3("mit_b3", 9, pretrained=None), Fusion_Network3_ac()
dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
seg, fus = seg.cuda().eval(), fus.cuda().eval()
pipe = PairForward(seg, fus)
for rank in (0, 6, 2):
    ir = dw.det_input(f"bench_ir_{rank}", (B, 1, H, W)).cuda()
    vis = dw.det_input(f"bench_vis_{rank}", (B, 3, H, W)).cuda()
    mask = dw.det_input(f"bench_mask_{rank}", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    for bound in (2e-3, 2e-4):
        ops.Planes16Guard.COND_BOUND = bound
        with torch.no_grad():
            for _ in range(2):
                pipe(ir, vis, mask)
            torch.cuda.synchronize()
            s0 = ops.range_stats()
            t0 = time.perf_counter()
            for _ in range(5):
                pipe(ir, vis, mask)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            s1 = ops.range_stats()
        print(f"rank {rank} bound {bound:g}: {1e3 * dt:.1f} ms per step, pairs repeated per step {(s1['images_repeated_fp32conv'] - s0['images_repeated_fp32conv']) / 5:.1f}", flush=True)
