#!/usr/bin/env python
"""bench.py — IR+visible image-pairs/s, forward (BASELINE.json metric) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic pairs already resident in HBM:
  out0,out1 = seg.denoise_net.encoder.forward_fusion(mask3)   (MiT encoder + 2 bilinear resizes)
  Yf        = fusion(ir, vis, out0, out1)                      (Fusion_Network3_ac)
  fused     = clamp01(YCrCb2RGB([Yf, Cr(vis), Cb(vis)]))
  logits    = seg(fused) -> bilinear x4 -> argmax              (Network3 = MiT encoder + SegFormer head)
exactly the in-memory chain of test_fusion.py:100-111 + test_segmentation.py:169-174.

Launch: `python bench.py --gpus N` (for N > 1 without a launcher it re-executes itself under
torch.distributed.run on 127.0.0.1) or, explicitly,
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
One process per GPU; the path shards over independent pairs, so there is no data-path collective
(weak scaling: per-GPU batch fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

# algorithmic work, GFLOP per pair forward at 480x640 (torch FlopCounterMode, 2*MAC)
GFLOP_PER_PAIR = {("mit_b1", 480, 640): 700.2, ("mit_b3", 480, 640): 827.1,
                  ("mit_b5", 1024, 1024): 2 * 798.47 + 38.8 + 2178.0}  # b5: BASELINE.md table, head / fusion scaled by pixels


def gflop_removed_by_n4(H, W):
    """FLOPs of the textbook order that the measured pipeline does not execute (exact in real
    arithmetic): conv3 / conv4 run at H/4 x W/4 and H/8 x W/8 instead of H x W, and the head's 1024 -> 256 fuse conv is
    composed with the four per-scale Linears (once per weight version) instead of running on the concatenation."""
    px = H * W
    conv3 = 2.0 * px * 64 * 64 * (1 - 1 / 16)
    conv4 = 2.0 * px * 128 * 64 * (1 - 1 / 64)
    fuse = 2.0 * (px / 16) * 1024 * 256
    return (conv3 + conv4 + fuse) / 1e9


CPU_BASELINE_THREADS = 16  # fastest of {16,32,64,128,256} on the GPU box host (profiles/r01_cpu_threads.txt)
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
# bf16x6 (csrc/conv3x3_split.hip): six bf16 MFMA products per fp32-equivalent multiply-add, so the
# ceiling in algorithmic (fp32) flops is the dense BF16 MFMA peak / 6
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step (288 GB of HBM: ~100 GB at 64; +3-5 %% over 32, flat beyond: profiles/r02_bench_batch_sweep.txt)")
    ap.add_argument("--backbone", default="mit_b3")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the step into a hipGraph and replay it")
    ap.add_argument("--no-extras", action="store_true", help="skip the second timed region (forward_fusion without its discarded stages)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (BASELINE configs [2] / [3])")
    ap.add_argument("--train-batch", type=int, default=8, help="samples per GPU per training step (config[2]: 8)")
    ap.add_argument("--train-steps", type=int, default=0, help="timed steps per training variant (0: min(--steps, 8))")
    ap.add_argument("--no-configs", action="store_true", help="skip the short timed legs of BASELINE configs [1] (mit_b1, batch 4) and [4] (mit_b5, batch 2, 1024x1024)")
    ap.add_argument("--dry-collective", action="store_true", help="multi-GPU readiness check only: bring the process group up, run the collective "
                    "self-test (world size the backend reports, checked rank-id collectives, one timed all-reduce) and print the gradient bucket "
                    "plan of both training steps; nothing is benchmarked")
    return ap.parse_args()


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(backbone, H, W):
    """The oracle (CPU port of the reference path) timed on this host's cores on a bounded sample: one warm-up pair,
    then the best of three timed pairs (batch 1: SURVEY §8(d) — larger batches are slower per pair on CPU)."""
    import detweights as dw
    import segmif_oracle as so
    # one thread per core up to 16: beyond that torch-CPU / oneDNN goes backwards on this path (measured in round 1
    # on the GPU box's host: 16 threads fastest of {16..256}; 256 threads: 284 s per pair)
    host_cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(CPU_BASELINE_THREADS, host_cores)))
    sd_seg = dw.det_state_dict(so.network3_shapes(backbone, 9), seed=0)
    sd_fus = dw.det_state_dict(so.fusion_shapes(), seed=0)
    ir = dw.det_input("cpu_ir", (1, 1, H, W))
    vis = dw.det_input("cpu_vis", (1, 3, H, W))
    mask = dw.det_input("cpu_mask", (1, 1, H, W)).repeat(1, 3, 1, 1)
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        so.pair_forward(sd_seg, sd_fus, ir, vis, mask, backbone)  # warm-up (allocator, oneDNN primitive cache)
        for _ in range(3):
            t0 = time.perf_counter()
            so.pair_forward(sd_seg, sd_fus, ir, vis, mask, backbone)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > 40.0:
                break
    best = min(times)
    return {"value": 1.0 / best, "unit": "img-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cores": host_cores, "cpu_model": cpu_model_name(),
            "sample": f"{backbone} {H}x{W} pairs at batch 1 through oracle/segmif_oracle.py (torch-CPU fp32): 1 warm-up "
                      f"pair, best of {len(times)} timed pairs ({', '.join(f'{t:.2f}' for t in times)} s)"}


# algorithmic work of the two training steps, GFLOP per sample at 480x640: the reference's autograd
# forms every gradient, 3 x forward of the differentiated parts ...
GFLOP_TRAIN = {("mit_b3", "seg"): 300.0, ("mit_b3", "fusion"): 2304.0}
# ... and what the fusion step EXECUTES with FusionTrainer(seg_weight_grads=False) (default: train_fusion's optimizer holds the
# fusion net only, train.py:316-327; nothing reads the segmentation net's .grad): its backward runs THROUGH the segmentation
# net (input gradients, 1 x forward) without the weight gradients (another 1 x): 88.8 + 3 x 638.2 + 2 x 100.1
GFLOP_TRAIN_EXECUTED = {("mit_b3", "fusion", False): 88.8 + 3 * 638.2 + 2 * 100.1}


def train_leg(args, rank, world, seg, fus):
    """BASELINE configs [2] (one GPU) and [3] (data parallel): the two steps of the reference's train.py on synthetic data,
    B = --train-batch per GPU, AFTER the forward metric's timed region (never part of `value`):
      seg     train.py:217-227  forward, x4 bilinear + CE(ignore 255), backward, PolyWarmupAdamW_seg
      fusion  train.py:351-385  no-grad forward_fusion, fusion net, MSE + 1.1 (1 - SSIM) (+ LapLoss2 evaluated as a
                                reported extra term), CE through the segmentation net, backward, PolyWarmupAdamW
    each in TRAIN mode (DropPath, Dropout2d(0.1), BatchNorm batch statistics: the mode BASELINE.md section 3 states) and,
    as a second figure, in the eval-mode regime the reference's train_seg drifts into.  With WORLD_SIZE > 1
    gradients go through segmif_amd.parallel.GradAllReducer (bucketed RCCL all-reduce launched from autograd hooks) and the
    non-overlapped part of the exchange is reported.  Same fencing as the forward leg: barrier + device sync on both sides,
    max over ranks."""
    import detweights as dw
    from segmif_amd import dist
    from segmif_amd.parallel import GradAllReducer
    from segmif_amd.train import FusionTrainer, seg_train_step
    from segmif_amd.utils.optimizer import PolyWarmupAdamW, PolyWarmupAdamW_seg

    B, H, W = args.train_batch, args.height, args.width
    steps = args.train_steps or max(1, min(args.steps, 8))
    warm = 2
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    labels = dw.det_labels(f"trb_y{rank}", (B, H, W), 9).cuda()
    x = dw.det_input(f"trb_x{rank}", (B, 3, H, W)).cuda()
    ir3 = dw.det_input(f"trb_ir{rank}", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    vis3 = dw.det_input(f"trb_vis{rank}", (B, 3, H, W)).cuda()
    mask3 = dw.det_input(f"trb_m{rank}", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    g = seg.denoise_net.get_param_groups()
    # constructor arguments of train.py:171-199 / :316-331 (configs/voc.yaml; iter_ = 2)
    opt_seg = PolyWarmupAdamW_seg([{"params": g[0], "lr": 8e-5, "weight_decay": 0.01}, {"params": g[1], "lr": 8e-5, "weight_decay": 0.0},
                                   {"params": g[2], "lr": 8e-4, "weight_decay": 0.01}], lr=8e-5, weight_decay=0.01,
                                  betas=(0.9, 0.999), iter_curr=0, warmup_iter=3000, max_iter=160000, warmup_ratio=1e-6, power=1.0)
    opt_fus = PolyWarmupAdamW([{"params": fus.parameters(), "lr": 8e-5 / 2, "weight_decay": 0.01}], lr=3e-4 / 2, weight_decay=0.01,
                              betas=(0.9, 0.999), warmup_iter=3e-5 / 2, max_iter=160000, warmup_ratio=1e-6, power=1.0)
    red_seg = GradAllReducer([p for grp in g for p in grp]) if world > 1 else None
    red_fus = GradAllReducer(list(fus.parameters())) if world > 1 else None
    trainer = FusionTrainer(seg, fus, opt_fus, crit, iter_=2, reducer=red_fus, report_lap=True)
    trainer_full = FusionTrainer(seg, fus, opt_fus, crit, iter_=2, reducer=red_fus, report_lap=False, seg_weight_grads=True)

    def timed(step, reducer):
        for _ in range(warm):
            loss = step()
        if reducer is not None:
            reducer.time_exposed_wait = True
        exposed, per_bucket = [], None
        dist.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
            if reducer is not None:
                exposed.append(reducer.exposed_wait_s)
                pb = reducer.exposed_wait_per_bucket_s
                per_bucket = pb if per_bucket is None or len(pb) != len(per_bucket) else [a + b for a, b in zip(per_bucket, pb)]
        dist.fence()
        mine = (time.perf_counter() - t0) / steps
        dt = dist.max_over_ranks(mine)
        spread = dist.gather_over_ranks(1e3 * mine)
        if reducer is not None:
            reducer.time_exposed_wait = False
        timed.last = {"ms_per_step_per_rank": {"min": min(spread), "max": max(spread), "all": spread},
                      "allreduce_exposed_ms_per_bucket": [1e3 * v / steps for v in per_bucket] if per_bucket else None}
        return dt, float(loss), (1e3 * sum(exposed) / len(exposed) if exposed else None)

    out = {"batch_per_gpu": B, "global_batch": B * world, "steps": steps, "warmup": warm, "world_size": world,
           "backbone": args.backbone, "height": H, "width": W}
    # One GPU: the segmentation step's forward + loss + backward is replayed from a hipGraph (segmif_amd.train.GraphedSegTrainStep:
    # the same kernels on the same data, bitwise the eager step - tests/test_gpu_round3.py; ~3 000 launches of ~15 us leave the host
    # out of the loop), the optimizer update stays outside it; the eager time is reported beside it.  Data parallel runs stay eager:
    # their gradient buckets are launched from autograd hooks DURING backward, which a replay has none of.
    graph_seg = world == 1 and os.environ.get("SEGMIF_BENCH_SEG_GRAPH", "1") != "0"
    for mode in ("train", "eval_regime"):
        seg.train(mode == "train")
        fus.train(mode == "train")
        for name, step, red in (("seg", lambda: seg_train_step(seg, opt_seg, x, labels, crit, red_seg), red_seg),
                                ("fusion", lambda: trainer.step(ir3, vis3, mask3, labels), red_fus)):
            dt, loss, exposed_ms = timed(step, red)
            eager_ms = graph_note = None
            if name == "seg" and graph_seg:
                try:
                    from segmif_amd.train import GraphedSegTrainStep
                    gstep = GraphedSegTrainStep(seg, opt_seg, crit, x, labels)
                    dt_g, loss_g, _ = timed(lambda: gstep(), None)
                    eager_ms, dt, loss = 1e3 * dt, dt_g, loss_g
                    del gstep
                except Exception as e:  # (capture refused: keep the eager figure and say so)
                    graph_note = f"hipGraph capture failed, eager step reported: {type(e).__name__}: {e}"[:300]
                opt_seg.zero_grad(set_to_none=True)
            gf = GFLOP_TRAIN.get((args.backbone, name))
            gfx = GFLOP_TRAIN_EXECUTED.get((args.backbone, name, False), gf)  # (the default fusion step skips the seg net's weight gradients)
            rec = {"ms_per_step": 1e3 * dt, "samples_per_s": world * B / dt, "loss": loss,
                   "tflops_per_gpu": (gfx * B / dt / 1e3) if gfx and (H, W) == (480, 640) else None,
                   "gflop_per_sample": gf, "gflop_executed_per_sample": gfx, "allreduce_exposed_ms": exposed_ms,
                   "grad_bytes": red.gradient_bytes() if red is not None else
                   4 * sum(p.numel() for p in (fus.parameters() if name == "fusion" else [q for grp in g for q in grp])
                           if p.grad is not None)}
            if rec["tflops_per_gpu"] is not None:
                # (r6) the step's rate against the two matrix pipes it runs on: the f16x3 ceiling the forward is
                # priced against (dense F16 2500 / 3 products) and the exact-fp32 MFMA pipe most of the segmentation step still uses
                rec["peak_tflops"] = 2500.0 / 3
                rec["frac"] = rec["tflops_per_gpu"] / (2500.0 / 3)
                rec["frac_of_fp32_mfma_pipe"] = rec["tflops_per_gpu"] / 157.3
                rec["dominant_kernel"] = TRAIN_DOMINANT[name]
            if world > 1:  # (r5) so that the first real multi-GPU run is diagnosable from this one line
                rec.update(timed.last)
            if name == "fusion" and trainer.last_lap is not None:
                rec["lap_loss2_reported"] = float(trainer.last_lap)
            if eager_ms is not None:
                rec["hipgraph_replay"], rec["eager_ms_per_step"] = True, eager_ms
            if graph_note is not None:
                rec["hipgraph_replay"], rec["note"] = False, graph_note
            out[f"{name}_{mode}"] = rec
    # the reference's side effect reproduced: the fusion step also forming the segmentation net's weight gradients (3 x forward
    # of everything differentiated: the full 2 304 GFLOP per sample), train mode
    seg.train(True)
    fus.train(True)
    if red_seg is not None:
        # this leg's backward also reaches the segmentation net's parameters, whose gradients nothing reads and no rank
        # exchanges: the segmentation step's reducer (its hooks sit on those parameters) has done its work - detach it, or its
        # hooks would see a second backward without a finish() in between
        red_seg.close()
    dt, loss, _ = timed(lambda: trainer_full.step(ir3, vis3, mask3, labels), red_fus)
    gf = GFLOP_TRAIN.get((args.backbone, "fusion"))
    out["fusion_train_with_seg_weight_grads"] = {
        "ms_per_step": 1e3 * dt, "samples_per_s": world * B / dt, "loss": loss, "gflop_executed_per_sample": gf,
        "tflops_per_gpu": (gf * B / dt / 1e3) if gf and (H, W) == (480, 640) else None,
        "note": "FusionTrainer(seg_weight_grads=True): the .grad the reference accumulates on the segmentation net and never reads"}
    seg.eval()
    fus.eval()
    out["modes"] = {"train": "DropPath, Dropout2d(0.1), BatchNorm batch statistics active (BASELINE.md section 3)",
                    "eval_regime": "module.eval() with gradients: the regime train_seg drifts into after its first validation (SURVEY F11)"}
    out["fusion_step_seg_weight_grads"] = False
    out["peak_bf16x6_tflops"] = PEAK_BF16X6_TFLOPS
    if world > 1:
        out["collective"] = {"backend": torch.distributed.get_backend(), "rccl_version": ".".join(map(str, torch.cuda.nccl.version())),
                             "world_size": torch.distributed.get_world_size(), "bucket_mb": 25.0,
                             "overlap": "buckets launched from post-accumulate-grad hooks during backward"}
    out["peak_mem_GB"] = torch.cuda.max_memory_allocated() / 2 ** 30
    return out


# the kernel with the largest share of each training step's kernel time (static record: the rocprofv3 tables named here, taken with
# tools/kstats.sh over tools/train_bench.py on this code; not re-measured by a bench run)
TRAIN_DOMINANT = {
    "seg": {"kernel": "igemm_kernel<64,64,32,32,16,0,2> (the ~290 small Linears of the MiT blocks at 8 images: exact-fp32 MFMA tiles)",
            "share_of_kernel_time": 0.242, "source": "profiles/r06_segtrain_kernel_stats.txt"},
    "fusion": {"kernel": "conv3x3_split_kernel<32,2,8,f16x3> (DRDB dilated convs, forward and input gradients, range made on the device)",
               "share_of_kernel_time": 0.134, "source": "profiles/r06_fusiontrain_kernel_stats.txt"},
}


def config_leg(backbone, B, H, W, steps, graph=False):
    """A short timed forward leg of another BASELINE config (the same pair forward in the default arithmetic, inputs resident
    in HBM, one warm-up pass that also packs the weights): pairs/s and the whole-path rate."""
    import detweights as dw
    from segmif_amd import dist, ops
    from segmif_amd.core import Fusion_Network3_ac, Network3
    from segmif_amd.pipeline import PairForward
    seg, fus = Network3(backbone, 9, pretrained=None), Fusion_Network3_ac()
    dw.load_det_weights(seg, seed=0)
    dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda().eval()
    ir = dw.det_input("cfg_ir", (B, 1, H, W)).cuda()
    vis = dw.det_input("cfg_vis", (B, 3, H, W)).cuda()
    mask = dw.det_input("cfg_mask", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    pipe = PairForward(seg, fus)
    s0 = ops.range_stats()
    with torch.no_grad():
        for _ in range(2):
            pipe(ir, vis, mask)
        dist.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            labels = pipe(ir, vis, mask)[1]
        dist.fence()
        dt = (time.perf_counter() - t0) / steps
    assert labels.shape == (B, H, W)
    s1 = ops.range_stats()
    gf = GFLOP_PER_PAIR.get((backbone, H, W))
    rec = {"backbone": backbone, "height": H, "width": W, "pairs_per_step": B, "steps": steps, "warmup": 2,
           "value": B / dt, "unit": "img-pairs/s", "ms_per_step": 1e3 * dt, "launch": "eager",
           "f16x3_pairs_repeated": s1["images_repeated"] - s0["images_repeated"],
           "f16x3_pairs_repeated_fp32conv": s1["images_repeated_fp32conv"] - s0["images_repeated_fp32conv"],
           "f16x3_pairs_seen": s1["images"] - s0["images"]}
    if graph:
        # (r6) small steps are launch-bound (~900 launches for a 15 ms step at 4 pairs): the same step captured
        # once into a hipGraph and replayed - same kernels, same guard (the graph carries its own range slots, read back after every
        # replay; a tripped pair is repeated eagerly) - reported beside the eager figure
        try:
            with torch.no_grad():
                pipe.capture(ir, vis, mask)
                for _ in range(2):
                    pipe(ir, vis, mask)
                dist.fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    lab2 = pipe(ir, vis, mask)[1]
                dist.fence()
                dtg = (time.perf_counter() - t0) / steps
            rec["hipgraph_replay"] = {"value": B / dtg, "ms_per_step": 1e3 * dtg, "labels_equal_eager": bool(torch.equal(lab2, labels))}
            del lab2
        except Exception as exc:  # (capture refused: the eager figure stands)
            rec["hipgraph_replay"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if gf is not None:
        rec["gflop_per_pair"] = {"executed": gf - gflop_removed_by_n4(H, W), "textbook_order": gf}
        rec["whole_path_tflops"] = B / dt * (gf - gflop_removed_by_n4(H, W)) / 1e3
    del pipe, seg, fus, ir, vis, mask, labels
    torch.cuda.empty_cache()
    return rec


def dry_collective(args, rank, world):
    """--dry-collective: what the first real multi-GPU run should print before anything is timed."""
    from segmif_amd import dist
    from segmif_amd.core import Fusion_Network3_ac, Network3
    seg = Network3(args.backbone, 9, pretrained=None)
    g = seg.denoise_net.get_param_groups()
    sizes = {"seg_step (upper bound: parameters that receive no gradient are dropped at the first step)":
             [p.numel() for grp in g for p in grp if p.requires_grad],
             "fusion_step (upper bound, same rule: ffm2.* receives none)": [p.numel() for p in Fusion_Network3_ac().parameters()]}
    rec = dist.collective_selftest(sizes)
    rec["launch"] = {"WORLD_SIZE": world, "RANK": rank, "MASTER_ADDR": os.environ.get("MASTER_ADDR"),
                     "MASTER_PORT": os.environ.get("MASTER_PORT"), "timeout_s": float(os.environ.get("SEGMIF_DIST_TIMEOUT", 300))}
    if rank == 0:
        print(json.dumps({"dry_collective": rec}), flush=True)
    dist.shutdown()


def self_launch(args, script=None, argv=None):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU
    (script / argv: what to run, for the CPU test of this entry; default: this file with the same arguments)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: the launcher binds its own free rendezvous port (no bind-then-close race with other jobs on the host)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", script or os.path.abspath(__file__)]
    cmd += sys.argv[1:] if argv is None else list(argv)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    from segmif_amd import dist
    rank, local_rank, world = dist.env_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.dry_collective:  # (works on any backend: SEGMIF_DIST_BACKEND=gloo exercises it without GPUs)
        dist.init()
        return dry_collective(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist.init()  # RCCL process group when WORLD_SIZE > 1; replicas only, no data-path collective
    torch.manual_seed(1234 + rank)  # per-rank RNG stream (nothing on the eval path draws from it; train-mode masks would)

    import detweights as dw
    from segmif_amd import ops
    from segmif_amd.core import Fusion_Network3_ac, Network3, fuse_to_rgb

    B, H, W = args.batch, args.height, args.width
    seg = Network3(args.backbone, 9, pretrained=None)
    fus = Fusion_Network3_ac()
    dw.load_det_weights(seg, seed=0)
    dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda().eval()
    ir = dw.det_input(f"bench_ir_{rank}", (B, 1, H, W)).cuda()
    vis = dw.det_input(f"bench_vis_{rank}", (B, 3, H, W)).cuda()
    mask = dw.det_input(f"bench_mask_{rank}", (B, 1, H, W)).repeat(1, 3, 1, 1).cuda()

    from segmif_amd.pipeline import PairForward
    pipe = PairForward(seg, fus)
    if args.graph:
        args.no_kernel_timer = True  # HIP events cannot be recorded inside a captured graph
        pipe.capture(ir, vis, mask)  # (f16x3: the graph carries its own range guard, read back after every replay)

    def step():
        return pipe(ir, vis, mask)[1]

    fence = dist.fence

    with torch.no_grad():
        for _ in range(args.warmup):
            labels = step()
        timer, side = None, {}
        if not args.no_kernel_timer:
            timer = ops.LaunchTimer("drdb_dcov")
            side = {t: ops.LaunchTimer(t) for t in ("dwconv", "cp_gram", "cp_tail", "bilinear", "mixffn")}
            ops.set_launch_timer(timer, side)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            labels = step()
        fence()
        elapsed = time.perf_counter() - t0
        ops.set_launch_timer(None)
    assert labels.shape == (B, H, W)

    rank_ms = dist.gather_over_ranks(1000.0 * elapsed / args.steps)  # per-rank spread: a slow GPU / link shows here
    elapsed = dist.max_over_ranks(elapsed)
    trip = ops.range_stats()  # (warm-up + timed steps of the headline configuration)
    trip["granularity"] = ("one pair: range slots and a CrossPath-softmax conditioning word per image; pairs that left the half's range are "
                           "repeated on bf16x6 (trip_rate), pairs whose context softmaxes are ill-conditioned (Planes16Guard.cond_estimate > COND_BOUND) with "
                           "exact-fp32 3x3 convs (cond_repeat_rate)")
    trip["cond_bound"], trip["cond_eps"] = ops.Planes16Guard.COND_BOUND, ops.Planes16Guard.COND_EPS

    # for continuity with rounds 1-2: the same step on bf16 triples throughout (6 products per MAC, no range guard)
    elapsed_bf16 = None
    if ops.conv3x3_mode() == "planes16" and not args.graph and not args.no_extras:
        prev_mode, prev_lin = ops.set_conv3x3_mode("planes"), ops.set_linear_mode("bf16x6" if ops.linear_mode() == "f16x3" else ops.linear_mode())
        with torch.no_grad():
            step()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            elapsed_bf16 = dist.max_over_ranks(time.perf_counter() - t0)
        ops.set_conv3x3_mode(prev_mode)
        ops.set_linear_mode(prev_lin)

    # beside the headline, never in it: the same step with the two encoder stages whose outputs forward_fusion() discards
    # (the reference computes and drops them, core/mix_transformer.py:358-375) not computed - identical results
    enc = seg.denoise_net.encoder
    elapsed_dse = None
    if not args.graph and not args.no_extras:
        enc.skip_unused_fusion_stages = True
        with torch.no_grad():
            step()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            elapsed_dse = dist.max_over_ranks(time.perf_counter() - t0)
        enc.skip_unused_fusion_stages = False

    # the other single-GPU forward configs of BASELINE.json, short legs in the same arithmetic (never part of `value`)
    configs = None
    if world == 1 and not args.no_configs and not args.graph and (args.backbone, H, W) == ("mit_b3", 480, 640):
        configs = {}
        for key, cfg in (("config1_mit_b1_b4_480x640", ("mit_b1", 4, 480, 640)), ("config4_mit_b5_b2_1024x1024", ("mit_b5", 2, 1024, 1024)),
                         ("mit_b1_b64_480x640", ("mit_b1", 64, 480, 640))):
            try:
                configs[key] = config_leg(*cfg, steps=3, graph=cfg[1] <= 8)
            except Exception as exc:
                configs[key] = {"error": f"{type(exc).__name__}: {exc}"}

    train = None
    if not args.no_train and not args.graph:
        del pipe, labels
        torch.cuda.empty_cache()
        try:
            train = train_leg(args, rank, world, seg, fus)  # every rank: the data-parallel steps hold collectives
        except Exception as exc:  # the forward metric above is already measured: report the failure instead of losing the line
            train = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        all_fp32 = (ops.conv3x3_mode(), ops.linear_mode(), ops.crosspath_mode(), ops.attention_mode()) == ("fp32", "fp32", "gemm", "fp32")
        out = {
            "metric": "IR+visible image-pairs/sec fwd", "value": value, "unit": "img-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if all_fp32 else (
                "f32 (large contractions on split operands, fp32-class: the fusion net's 3x3 convs, the encoder's tall GEMMs, fused Mix-FFN "
                "and attention and the CrossPath tail as half pairs x 3 f16 MFMA products under a per-pair range guard, the rest as bf16 "
                "triples x 6 products; see arithmetic_modes)"
                if ops.conv3x3_mode() == "planes16" else
                "f32 (large contractions: fp32-equivalent 3-way bf16 split, 6 MFMA products; see arithmetic_modes)"),
            "f16x3_range_fallbacks": ops.range_fallbacks(),
            "f16x3_trip_rate": trip["trip_rate"], "f16x3_cond_repeat_rate": trip["cond_repeat_rate"], "f16x3_guard": trip,
            "ms_per_step_per_rank": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "conv3x3_mode": ops.conv3x3_mode(),
            "arithmetic_modes": {"conv3x3": ops.conv3x3_mode(), "linear": ops.linear_mode(), "crosspath": ops.crosspath_mode(),
                                 "attention": ops.attention_mode(), "mixffn": ops.mixffn_mode(),
                                 "encoder_gemms": "gemm_pairs (A pre-split by its producer)" if ops.pairs_mode() == "on" else "gemm_split",
                                 "crosspath_seg_feature": "read at its own resolution by the Gram / tail kernels (ops.LazySeg)"
                                 if ops.lazy_seg_mode() else "resized to H x W first"},
            "data": "synthetic",
            "config": {"workload": f"{args.backbone} pair forward (forward_fusion + Fusion_Network3_ac + Network3 "
                                   f"+ x4 bilinear + argmax), {H}x{W}, {B} pairs per GPU per step, eval mode, "
                                   "seeded deterministic weights; conv3/conv4 (1x1) applied before the "
                                   "bilinear resize of forward_fusion, and that resize done by CrossPath's kernels as they read "
                                   "(same function, SURVEY 8(f) N4)",
                       "backbone": args.backbone, "height": H, "width": W, "pairs_per_gpu": B,
                       "parallelism": f"replicas x{world} (independent pairs, no collective)",
                       "launch": "hipGraph replay" if args.graph else "eager"},
        }
        gf = GFLOP_PER_PAIR.get((args.backbone, H, W))
        if gf is not None:
            # per GPU: executed FLOPs (N4 removes part of the textbook count) and, beside it, the textbook figure
            out["whole_path_tflops"] = value * (gf - gflop_removed_by_n4(H, W)) / 1000.0 / world
            out["whole_path_tflops_textbook_order"] = value * gf / 1000.0 / world
            # the whole path against the split-operand ceiling of its arithmetic: 2500 / 3 for the default (the fusion net's convs
            # and the encoder's GEMMs on f16x3; attention / CrossPath still issue six bf16 products, fp32-MFMA patch embeds and
            # bandwidth-bound row kernels only lower the figure), 2500 / 6 in the bf16x6 modes
            ceiling = PEAK_BF16X6_TFLOPS * (2.0 if ops.conv3x3_mode() == "planes16" else 1.0)
            out["whole_path_frac"] = None if all_fp32 else out["whole_path_tflops"] / ceiling
            out["whole_path_ceiling_tflops"] = None if all_fp32 else ceiling
            out["gflop_per_pair"] = {"executed": gf - gflop_removed_by_n4(H, W), "textbook_order": gf}
        if timer is not None:
            n, ms, flops = timer.summary()
            achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic, traffic_src = None, None
            mode = ops.conv3x3_mode()
            peak = PEAK_FP32_MFMA_TFLOPS if mode == "fp32" else (PEAK_BF16X6_TFLOPS * 2.0 if mode == "planes16" else PEAK_BF16X6_TFLOPS)
            p16_name = next((n for n in (f"r06_pmc_dominant_b{B}_planes16.json", f"r05_pmc_dominant_b{B}_planes16.json", f"r04_pmc_dominant_b{B}_planes16.json")
                             if os.path.exists(os.path.join(ROOT, "profiles", n))), f"r03_pmc_dominant_b{B}_planes16.json")
            pmc_name = {"planes16": p16_name, "planes": f"r03_pmc_dominant_b{B}_planes.json" if os.path.exists(os.path.join(ROOT, "profiles", f"r03_pmc_dominant_b{B}_planes.json")) else f"r02_pmc_dominant_b{B}_planes.json", "bf16x6": f"r01_pmc_dominant_b{B}_bf16x6.json",
                        "fp32": "r01_pmc_dominant.json" if B == 8 else f"r01_pmc_dominant_b{B}.json"}[mode]
            pmc = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc) and (H, W) == (480, 640):
                # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
                rec = json.load(open(pmc))
                traffic, traffic_src = rec["hbm_bytes_per_launch"], ("static record, not measured in this run: profiles/" + os.path.basename(pmc) +
                                                                     " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.sh)")
            kernel = {"planes": "conv3x3_planes_kernel<2,false> (DRDB dilated 3x3 convs 1-4 on pre-split activations, bf16 MFMA x 6 "
                                "split products, fp32-class; the fifth conv carries the DRDB's 1x1 tail and is a separate kernel)",
                      "planes16": "conv3x3_planes_kernel<2,false,f16x3,SUB=4> (DRDB dilated 3x3 convs 1-4 on half-pair activations, 16 x 32 patches, f16 MFMA x 3 "
                                  "split products, fp32-class inside the half's exponent range, guarded)",
                      "bf16x6": "conv3x3_split_kernel<32,2,8> (DRDB dilated 3x3 conv, bf16 MFMA x 6 split products, fp32-class)",
                      "fp32": "conv3x3_halo_kernel<32,8,2> (DRDB dilated 3x3 conv, fp32 MFMA)"}[mode]
            # algorithmic bytes per pixel of an average timed launch: planes mode reads Cin x 6 B (three bf16 planes) and
            # writes 32 x 6 B, Cin averaging 112 over Dcov1-4; the fp32 layouts read Cin x 4 B (average 128) + write 128 B
            alg_bytes = (112 * 6 + 32 * 6 if mode == "planes" else 112 * 4 + 32 * 4 if mode == "planes16" else 128 * 4 + 32 * 4) * float(B) * H * W
            out["roofline"] = {
                "kernel": kernel, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "peak_basis": ("dense fp32 MFMA" if mode == "fp32" else "dense F16 MFMA 2500 TFLOP/s / 3 products per fp32-equivalent "
                               "MAC; achieved counts algorithmic fp32 flops" if mode == "planes16" else "dense BF16 MFMA 2500 "
                               "TFLOP/s / 6 products per fp32-equivalent MAC; achieved counts algorithmic fp32 flops"),
                "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                "launches_timed": n, "avg_launch_ms": ms, "avg_launch_gflop": flops / 1e9,
                # the same launch against the memory side (planes16 sits near the ridge: 112 flop per algorithmic byte
                # against 833 TFLOP/s / 8 TB/s = 104)
                "hbm_side": {"algorithmic_GBps": alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "peak_GBps": 8000.0,
                             "frac": alg_bytes / (ms * 1e-3) / 8e12 if ms > 0 else 0.0},
            }
        if elapsed_bf16 is not None:
            out["with_bf16x6_planes"] = {
                "value": pairs / elapsed_bf16, "ms_per_step": 1000.0 * elapsed_bf16 / args.steps,
                "note": "the same step with SEGMIF_CONV3X3=planes SEGMIF_LINEAR=bf16x6 (rounds 1-2 arithmetic: bf16 triples, six "
                        "products per MAC, no range guard needed) - what a pair falls back to when the f16x3 guard trips"}
        if elapsed_dse is not None:
            out["without_discarded_encoder_stages"] = {
                "value": pairs / elapsed_dse, "ms_per_step": 1000.0 * elapsed_dse / args.steps,
                "note": "NOT the headline: same pair forward, same outputs bit for bit, with stages 3-4 of the FIRST encoder pass "
                        "(forward_fusion on the mask; the reference computes and discards them, mix_transformer.py:358-375) not "
                        "computed (encoder.skip_unused_fusion_stages = True); 67.7 GFLOP per pair fewer executed"}
        if side:
            # bandwidth-bound kernels: algorithmic HBM bytes per launch / HIP-event time around the launch (peak 8 TB/s,
            # ~6.3 achievable: MI355X_MICROARCH.md)
            names = {"dwconv": "dwconv3x3_gelu_kernel (Mix-FFN middle)", "cp_gram": "crosspath_gram_kernel (the two modalities' features; the segmentation feature's Gram reads a low-resolution map and is not timed here)",
                     "cp_tail": "crosspath_tail_kernel (x_i in + planes out: the segmentation feature is an L2-resident low-resolution map)" if ops.lazy_seg_mode() else "crosspath_tail_kernel", "bilinear": "bilinear_kernel (forward_fusion / logits resize)",
                     "mixffn": "mixffn_kernel (norm2 + fc1 + dwconv + GELU + fc2 + residual of a stage-1/2 block in one launch; bytes = x in + out)"}
            hb = {}
            for tag, t in side.items():
                n, ms, nbytes = t.summary()
                if n and tag == "mixffn":
                    # NOT an HBM-bound kernel: its hidden tensor never reaches HBM; it is bound by its vector-ALU /
                    # matrix phases.  Work per token element of x: 2 GEMMs of 2 * 4C flops each + 9 * 2 * 4 depthwise flops.
                    # The timer carries x in + out bytes (8 per element); C is 64 | 128 per launch: both bounds are given.
                    elems = nbytes / 8.0
                    lo, hi = elems * (16 * 64 + 72) / (ms * 1e-3) / 1e12, elems * (16 * 128 + 72) / (ms * 1e-3) / 1e12
                    out["compute_kernels"] = {"mixffn": {"kernel": names[tag], "launches_timed": n, "avg_launch_ms": ms, "bound": "vector ALU / mfma",
                                                         "tflops_if_all_C64": lo, "tflops_if_all_C128": hi, "peak_tflops": PEAK_BF16X6_TFLOPS * 2.0,
                                                         "frac_of_peak_range": [lo / (PEAK_BF16X6_TFLOPS * 2.0), hi / (PEAK_BF16X6_TFLOPS * 2.0)],
                                                         "hbm_GBps": nbytes / (ms * 1e-3) / 1e9}}
                elif n:
                    hb[tag] = {"kernel": names[tag], "launches_timed": n, "avg_launch_ms": ms,
                               "algorithmic_GB_per_launch": nbytes / 1e9, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
                               "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12}
            out["hbm_bound_kernels"] = hb
        if configs is not None:
            out["configs"] = configs
        if train is not None:
            out["train"] = train
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.backbone, H, W)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    dist.shutdown()


if __name__ == "__main__":
    main()
