"""Generate the training-path fixtures of tests/golden/ by running the REAL upstream reference on CPU:

  optim_steps.npz        utils/optimizer.py (PolyWarmupAdamW, PolyWarmupAdamW_seg): three steps on seeded
                         parameters / gradients with the exact constructor arguments of train.py:171-199 and
                         :316-331 (incl. the warmup_iter = 3e-5 / iter_ quirk), one parameter without a gradient
  losses.npz             pytorch_ssim.ssim, core/loss.py Fusionloss_grad3 / Fusionloss3 / Sobelxy: values and
                         gradients w.r.t. the fused image
  train_seg_b1.npz       three iterations of train.py:217-227 (forward, x4 bilinear, CE ignore 255, backward,
                         PolyWarmupAdamW_seg over get_param_groups()) on Network3('mit_b1'), 2 x 64 x 96
  train_fusion_b1.npz    three iterations of train.py:351-385 with iter_ = 2 (Fusionloss_grad3 + CE through
                         the segmentation net, PolyWarmupAdamW on the fusion net), 2 x 32 x 48

Modules run in eval() mode: the reference trains in train() mode, whose DropPath / Dropout2d draws cannot be
reproduced across frameworks; BatchNorm then uses its running statistics.  Everything else —
losses, schedules, update arithmetic, which parameters receive gradients — is the reference's own code.
core/loss.py calls .cuda() in constructors (Sobelxy :645-646, LapLoss2 default device): during generation
torch.Tensor.cuda / nn.Module.cuda are identity functions, nothing else is changed.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_train.py
Container-only (needs /root/reference); the fixtures are data (inputs, expected outputs), never source.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detweights as dw  # noqa: E402
import refload  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = refload.REF_ROOT
NUM_CLASSES = 9


def npy(t):
    return t.detach().cpu().numpy().copy()  # a copy: the numpy view of a CPU tensor follows later in-place updates


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def load_by_path(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def cuda_is_identity():
    """core/loss.py and lap_loss.py move constants to a CUDA device in constructors; on this CPU-only
    container those calls become no-ops (device placement is not arithmetic)."""
    t_cuda, m_cuda = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = t_cuda, m_cuda


def load_reference_losses():
    sys.dont_write_bytecode = True
    src = open(os.path.join(REF, "lap_loss.py")).read()
    # LapLoss2's default argument is torch.device('cuda'), evaluated at import: exec the module with the default
    # device string rewritten to 'cpu' in a private namespace (no file is written, nothing is shipped)
    lap = types.ModuleType("lap_loss")
    exec(compile(src.replace("torch.device('cuda')", "torch.device('cpu')"), os.path.join(REF, "lap_loss.py"), "exec"), lap.__dict__)
    sys.modules["lap_loss"] = lap
    ssim_mod = load_by_path("pytorch_ssim", os.path.join(REF, "pytorch_ssim", "__init__.py"))
    pkg = sys.modules["_segmif_ref_core"]
    ent = load_by_path("_segmif_ref_core.Entropy", os.path.join(REF, "core", "Entropy.py"))
    pkg.Entropy = ent
    loss = load_by_path("_segmif_ref_core.loss", os.path.join(REF, "core", "loss.py"))
    return ssim_mod, loss


def param_record(module, prefix, rec, head=64):
    for name, p in module.named_parameters():
        t = p.detach()
        rec[prefix + name + "|norm"] = np.float64(t.double().norm())
        rec[prefix + name + "|head"] = npy(t.reshape(-1)[:head])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    mt, sh, mf = refload.load_reference()
    opt_mod = load_by_path("_segmif_ref_optimizer", os.path.join(REF, "utils", "optimizer.py"))
    with cuda_is_identity():
        ssim_mod, loss_mod = load_reference_losses()

    # ---- 1. optimizer -----------------------------------------------------------------------------
    rec = {}
    shapes = [(7, 5), (33,), (4, 3, 3, 3), (1,), (6,)]

    def fresh_params(tag):
        return [torch.nn.Parameter(dw.det_input(f"opt_{tag}_p{i}", s, lo=-1.0, hi=1.0)) for i, s in enumerate(shapes)]

    def run(opt, params, tag, steps=3):
        for st in range(steps):
            for i, p in enumerate(params):
                p.grad = None if i == 4 else dw.det_input(f"opt_{tag}_g{i}_s{st}", p.shape, lo=-1.0, hi=1.0)  # param 4: no grad
            opt.step()
            rec[f"{tag}|lr|{st}"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
            for i, p in enumerate(params):
                rec[f"{tag}|p{i}|{st}"] = npy(p)

    # segmentation: train.py:171-199 with configs/voc.yaml (lr 8e-5, wd 0.01, warmup 3000 / ratio 1e-6, power 1, max 160000)
    for it_start in (0, 10000, 159999):
        ps = fresh_params(f"seg{it_start}")
        opt = opt_mod.PolyWarmupAdamW_seg(
            params=[{"params": ps[:2], "lr": 8e-5, "weight_decay": 0.01}, {"params": ps[2:3], "lr": 8e-5, "weight_decay": 0.0},
                    {"params": ps[3:], "lr": 8e-4, "weight_decay": 0.01}],
            lr=8e-5, weight_decay=0.01, betas=[0.9, 0.999], iter_curr=it_start, warmup_iter=3000, max_iter=160000,
            warmup_ratio=1e-6, power=1.0)
        run(opt, ps, f"seg{it_start}")
    # fusion: train.py:316-331 with iter_ = 2 (lr passed per group = 8e-5 / iter_, warmup_iter = 3e-5 / iter_)
    for iter_ in (1, 2):
        ps = fresh_params(f"fus{iter_}")
        opt = opt_mod.PolyWarmupAdamW(
            params=[{"params": ps, "lr": 8e-5 / iter_, "weight_decay": 0.01}], lr=3e-4 / iter_, weight_decay=0.01,
            betas=[0.9, 0.999], warmup_iter=3e-5 / iter_, max_iter=160000, warmup_ratio=1e-6, power=1.0)
        run(opt, ps, f"fus{iter_}")
    rec["shapes"] = np.array([len(s) for s in shapes])
    np.savez_compressed(os.path.join(OUT, "optim_steps.npz"), **rec)

    # ---- 2. losses --------------------------------------------------------------------------------
    rec = {}
    B, H, W = 2, 40, 56
    gen = dw.det_input("loss_gen", (B, 1, H, W), lo=-0.1, hi=1.1).requires_grad_(True)
    mask = dw.det_input("loss_mask", (B, 3, H, W))
    ir = dw.det_input("loss_ir", (B, 1, H, W))
    vis = dw.det_input("loss_vis", (B, 3, H, W))
    rec.update(gen=npy(gen), mask=npy(mask), ir=npy(ir), vis=npy(vis))
    s = ssim_mod.ssim(gen, mask[:, :1])
    (g,) = torch.autograd.grad(s, gen)
    rec["ssim"], rec["ssim_grad"] = np.float64(s.detach()), npy(g)
    with cuda_is_identity():
        lg3, l3, sob = loss_mod.Fusionloss_grad3(), loss_mod.Fusionloss3(), loss_mod.Sobelxy()
    for name, fn in (("grad3", lg3), ("loss3", l3)):
        v = fn(ir, vis, gen, mask)
        (g,) = torch.autograd.grad(v, gen)
        rec[name], rec[name + "_grad"] = np.float64(v.detach()), npy(g)
    rec["sobel"] = npy(sob(gen.detach()))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **rec)

    # ---- 3. segmentation training step x 3 (train.py:217-227) --------------------------------------
    rec = {}
    net = quiet(mf.Network3, "mit_b1", NUM_CLASSES, pretrained=None).eval()
    dw.load_det_weights(net, seed=0)
    groups = net.denoise_net.get_param_groups()
    opt = opt_mod.PolyWarmupAdamW_seg(
        params=[{"params": groups[0], "lr": 8e-5, "weight_decay": 0.01}, {"params": groups[1], "lr": 8e-5, "weight_decay": 0.0},
                {"params": groups[2], "lr": 8e-4, "weight_decay": 0.01}],
        lr=8e-5, weight_decay=0.01, betas=[0.9, 0.999], iter_curr=10000, warmup_iter=3000, max_iter=160000,
        warmup_ratio=1e-6, power=1.0)
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    B, H, W = 2, 64, 96
    losses = []
    for st in range(3):
        x = dw.det_input(f"trs_x{st}", (B, 3, H, W))
        y = dw.det_labels(f"trs_y{st}", (B, H, W), NUM_CLASSES)
        y[0, 3:9, 5:40] = 255
        _, _, segmap = net(x)
        out = F.interpolate(segmap, size=y.shape[1:], mode="bilinear", align_corners=False)
        loss = crit(out, y.type(torch.long))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    rec["losses"] = np.array(losses, dtype=np.float64)
    rec["no_grad_params"] = np.array([n for n, p in net.named_parameters() if p.grad is None])
    param_record(net, "", rec)
    np.savez_compressed(os.path.join(OUT, "train_seg_b1.npz"), **rec)

    # ---- 4. fusion training step x 3 (train.py:351-385, iter_ = 2, n_iter <= 10) -------------------
    rec = {}
    iter_ = 2
    net = quiet(mf.Network3, "mit_b1", NUM_CLASSES, pretrained=None).eval()
    dw.load_det_weights(net, seed=0)
    fus = quiet(mf.Fusion_Network3_ac).eval()
    dw.load_det_weights(fus, seed=0)
    opt = opt_mod.PolyWarmupAdamW(
        params=[{"params": fus.parameters(), "lr": 8e-5 / iter_, "weight_decay": 0.01}], lr=3e-4 / iter_, weight_decay=0.01,
        betas=[0.9, 0.999], warmup_iter=3e-5 / iter_, max_iter=160000, warmup_ratio=1e-6, power=1.0)
    with cuda_is_identity():
        floss = loss_mod.Fusionloss_grad3()
    B, H, W = 2, 32, 48
    l1s, l2s, tot = [], [], []
    for st in range(3):
        ir3 = dw.det_input(f"trf_ir{st}", (B, 1, H, W)).repeat(1, 3, 1, 1)
        vis3 = dw.det_input(f"trf_vis{st}", (B, 3, H, W))
        mask3 = dw.det_input(f"trf_mask{st}", (B, 1, H, W)).repeat(1, 3, 1, 1)
        labels = dw.det_labels(f"trf_y{st}", (B, H, W), NUM_CLASSES)
        ir = ir3[:, 0:1]
        # RGB2YCrCb / YCrCb2RGB of core/model_fusion.py hard-wire .cuda(): identity here
        with cuda_is_identity():
            vis = mf.RGB2YCrCb(vis3)
            with torch.no_grad():
                out0, out1 = net.denoise_net.encoder.forward_fusion(mask3)
            fusion = fus(ir, vis, out0, out1)
            opt.zero_grad()
            ycc = vis.clone()
            ycc[:, 0:1] = fusion
            rgb = mf.YCrCb2RGB(ycc)
            loss1 = floss(ir, vis, fusion, mask3)
            loss2 = net._loss(rgb, labels, crit)
        seg_loss = (0.4 / iter_) * loss1 + 0.8 * loss2
        seg_loss.backward()
        opt.step()
        l1s.append(float(loss1.detach()))
        l2s.append(float(loss2.detach()))
        tot.append(float(seg_loss.detach()))
    rec["loss1"], rec["loss2"], rec["total"] = (np.array(v, dtype=np.float64) for v in (l1s, l2s, tot))
    rec["no_grad_params"] = np.array([n for n, p in fus.named_parameters() if p.grad is None])
    param_record(fus, "", rec)
    np.savez_compressed(os.path.join(OUT, "train_fusion_b1.npz"), **rec)

    for fn in ("optim_steps.npz", "losses.npz", "train_seg_b1.npz", "train_fusion_b1.npz"):
        print(f"  {fn:28s} {os.path.getsize(os.path.join(OUT, fn)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
