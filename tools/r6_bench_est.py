"""(r6) The guard's conditioning estimate over the bench's own inputs (ranks 0..7, 64 pairs each, mit_b3 480 x 640): how far below a
candidate COND_BOUND does the headline workload sit?"""
import math
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

ops.Planes16Guard.COND_BOUND = math.inf
B, H, W = 64, 480, 640
seg, fus = Network3("mit_b3", 9, pretrained=None), Fusion_Network3_ac()
dw.load_det_weights(seg, seed=0), dw.load_det_weights(fus, seed=0)
seg, fus = seg.cuda().eval(), fus.cuda().eval()
pipe = PairForward(seg, fus)
allest = []
for rank in range(8):
    ir = dw.det_input(f"bench_ir_{rank}", (B, 1, H, W)).cuda()
    vis = dw.det_input(f"bench_vis_{rank}", (B, 3, H, W)).cuda()
    mask = dw.det_input(f"bench_mask_{rank}", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    with torch.no_grad():
        g = ops.Planes16Guard("cuda", B)
        prev = ops.install_guard(g)
        try:
            pipe._eager_body(ir, vis, mask)
        finally:
            ops.install_guard(prev)
    est = g.cond_estimate().tolist()
    allest += est
    top = sorted(est, reverse=True)[:4]
    print(f"rank {rank}: tripped {int(g.tripped().sum())}, largest estimates {['%.2e' % e for e in top]}", flush=True)
s = sorted(allest, reverse=True)
print("all 512 pairs: top 8", ["%.2e" % e for e in s[:8]], "count > 1e-4:", sum(e > 1e-4 for e in s), "> 2e-4:", sum(e > 2e-4 for e in s), "> 5e-5:", sum(e > 5e-5 for e in s))
