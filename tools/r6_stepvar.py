"""Diagnosis (r6): per-step wall / device time of the 64-pair forward, eager and under hipGraph replay, with the device's clocks and
power sampled beside it - is a slow run uniformly slow or made of spikes?"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import detweights as dw  # noqa: E402
from segmif_amd.core import Fusion_Network3_ac, Network3  # noqa: E402
from segmif_amd.pipeline import PairForward  # noqa: E402

B, H, W = 64, 480, 640
seg, fus = Network3("mit_b3", 9, pretrained=None), Fusion_Network3_ac()
dw.load_det_weights(seg, seed=0)
dw.load_det_weights(fus, seed=0)
seg, fus = seg.cuda().eval(), fus.cuda().eval()
ir = dw.det_input("bench_ir_0", (B, 1, H, W)).cuda()
vis = dw.det_input("bench_vis_0", (B, 3, H, W)).cuda()
mask = dw.det_input("bench_mask_0", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
pipe = PairForward(seg, fus)

samples, stop = [], False


def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            samples.append({k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "power", "junction", "edge"))})
        except Exception as e:  # noqa: BLE001
            samples.append({"err": str(e)[:80]})
        time.sleep(0.5)


th = threading.Thread(target=smi, daemon=True)
th.start()


def run(n, label):
    walls, devs = [], []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.record()
        pipe(ir, vis, mask)
        e.record()
        torch.cuda.synchronize()
        walls.append(1e3 * (time.perf_counter() - t0))
        devs.append(s.elapsed_time(e))
    print(label, "wall", [round(w, 1) for w in walls], flush=True)
    print(label, "dev ", [round(w, 1) for w in devs], flush=True)


with torch.no_grad():
    for _ in range(3):
        pipe(ir, vis, mask)
    run(16, "eager")
    pipe.capture(ir, vis, mask)
    for _ in range(2):
        pipe(ir, vis, mask)
    run(16, "graph")
stop = True
time.sleep(0.6)
print("smi samples:", len(samples))
for s_ in samples[:: max(1, len(samples) // 12)]:
    print(s_)
