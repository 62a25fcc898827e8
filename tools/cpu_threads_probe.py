"""How many host threads should the CPU baseline (oracle) use on the GPU box?  Times one mit_b3 pair forward per thread count
(`--full`: at the bench's 480x640; default: quarter size 240x320); bench.py's CPU_BASELINE_THREADS is the fastest setting.
    python tools/cpu_threads_probe.py [--full] 4 8 12 16 24 32"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import detweights as dw
import segmif_oracle as so
FULL = "--full" in sys.argv
if FULL:
    sys.argv.remove("--full")
H, W = (480, 640) if FULL else (240, 320)
sd_seg = dw.det_state_dict(so.network3_shapes("mit_b3", 9), seed=0)
sd_fus = dw.det_state_dict(so.fusion_shapes(), seed=0)
ir = dw.det_input("cpu_ir", (1, 1, H, W)); vis = dw.det_input("cpu_vis", (1, 3, H, W))
mask = dw.det_input("cpu_mask", (1, 1, H, W)).repeat(1, 3, 1, 1)
print("cpu_count", os.cpu_count(), flush=True)
for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    torch.set_num_threads(nt)
    with torch.no_grad():
        so.pair_forward(sd_seg, sd_fus, ir, vis, mask, "mit_b3")  # warm-up
        t0 = time.perf_counter(); so.pair_forward(sd_seg, sd_fus, ir, vis, mask, "mit_b3"); dt = time.perf_counter() - t0
    print(f"threads {nt:4d}: {dt:7.2f} s per {H}x{W} pair", flush=True)
