"""Fusion losses used by the reference's train_fusion (core/loss.py:459-476 Fusionloss3, :506-517
Fusionloss_grad3, :634-650 Sobelxy; pytorch_ssim/__init__.py:8-43) and LapLoss2 (lap_loss.py:100-118).

On the GPU both objectives are fused HIP kernels, forward and backward (csrc/losses.hip + the separable blur of
csrc/rowops.hip; autograd.FusionLossGrad3Fn / FusionLoss3Fn): SURVEY §8(f) N1.  Device tensors the kernels do not cover
(non-fp32, more than one channel, mismatched shapes) are REFUSED like everywhere else in the package - there is no torch /
MIOpen fallback on the GPU.  The torch formulations below run on CPU tensors only: they are what the CPU tests pin against
the reference (tests/golden/losses.npz).
"""
import math

import torch
import torch.nn.functional as F


def _gaussian_window(size=11, sigma=1.5, device=None, dtype=torch.float32):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)], dtype=dtype)
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).to(device)[None, None]


def ssim(img1, img2, window_size=11):
    """Gaussian-window SSIM averaged over the image (single channel per group).  On the GPU the five
    window convolutions (and their backward) run in the separable HIP blur kernel."""
    C = img1.shape[1]
    if img1.is_cuda:
        if window_size != 11 or img1.dtype != torch.float32 or img2.dtype != torch.float32 or not img2.is_cuda:
            raise RuntimeError("segmif_amd.losses.ssim: the HIP blur kernel is the 11-tap float32 one (no torch fallback on the GPU)")
        from . import autograd as ag
        blur = ag.gauss_blur11
    else:
        w = _gaussian_window(window_size, 1.5, img1.device, img1.dtype).expand(C, 1, window_size, window_size).contiguous()
        pad = window_size // 2
        blur = lambda t: F.conv2d(t, w, padding=pad, groups=C)
    mu1, mu2 = blur(img1), blur(img2)
    s11 = blur(img1 * img1) - mu1 * mu1
    s22 = blur(img2 * img2) - mu2 * mu2
    s12 = blur(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean()


def _hip_ok(generate_img, *others):
    """True: the fused HIP kernels take it; False: CPU tensors (the torch formulation, test infrastructure); a device
    tensor the kernels do not cover raises - no library-convolution fallback on the GPU."""
    ts = (generate_img,) + others
    if not any(t.is_cuda for t in ts):
        return False
    if all(t.is_cuda and t.dtype == torch.float32 and t.shape == generate_img.shape for t in ts) and generate_img.shape[1] == 1:
        return True
    raise RuntimeError("segmif_amd.losses: the fused loss kernels take float32 single-channel device tensors of one shape, got "
                       + ", ".join(f"{tuple(t.shape)} {t.dtype} {t.device.type}" for t in ts) + " (no torch fallback on the GPU)")


def fusion_loss_grad3(generate_img, mask):
    """MSE(mask_0, fused) + 1.1 * (1 - SSIM(fused, mask_0))  — the round >= 2 intensity term."""
    m = mask[:, :1]
    if _hip_ok(generate_img, m):
        from . import autograd as ag
        return ag.FusionLossGrad3Fn.apply(generate_img, m.detach())
    return F.mse_loss(m, generate_img) + 1.1 * (1 - ssim(generate_img, m))


def sobel_xy(x):
    """|Sobel_x| + |Sobel_y| with zero padding (core/loss.py:634-650), written as shifted differences so
    that neither the forward nor the backward goes through a library convolution."""
    p = F.pad(x, (1, 1, 1, 1))
    top, mid, bot = p[:, :, :-2], p[:, :, 1:-1], p[:, :, 2:]
    gx = (top[..., 2:] + 2 * mid[..., 2:] + bot[..., 2:]) - (top[..., :-2] + 2 * mid[..., :-2] + bot[..., :-2])
    gy = (top[..., :-2] + 2 * top[..., 1:-1] + top[..., 2:]) - (bot[..., :-2] + 2 * bot[..., 1:-1] + bot[..., 2:])
    return gx.abs() + gy.abs()


def fusion_loss3(generate_img, mask):
    """L1(mask_0, fused) + L1(Sobel(mask_0), Sobel(fused))  — the round-1 objective."""
    m = mask[:, :1]
    if _hip_ok(generate_img, m):
        from . import autograd as ag
        return ag.FusionLoss3Fn.apply(generate_img, m.detach())
    return F.l1_loss(m, generate_img) + F.l1_loss(sobel_xy(m), sobel_xy(generate_img))


def _lap_window(size, sigma=2.0, device=None, dtype=torch.float32):
    """lap_loss.py:39-80 `smoothing`: the normalised size x size Gaussian (zero padding size // 2 at the call site)."""
    x = torch.arange(size, dtype=torch.float64) - (size - 1) / 2.0
    g = torch.exp(-(x[:, None] ** 2 + x[None, :] ** 2) / (2.0 * sigma ** 2))
    return (g / g.sum()).to(dtype).to(device)[None, None]


def lap_loss2(generate_img, ir, vis):
    """LapLoss2.forward (lap_loss.py:100-118): d_k(img) = img - G_k * img for the 3 / 5 / 7-tap sigma-2 Gaussians;
    10 (L1_3 + L1_5) + L1_7 with L1_k = mean |d_k(gen) - max(d_k(ir), d_k(vis))|.  Fusionloss_grad3 builds one and never
    evaluates it (core/loss.py:509); FusionTrainer(report_lap=True) reports it beside the loss (BASELINE config[2])."""
    if _hip_ok(generate_img, ir, vis):
        from . import autograd as ag
        return ag.LapLoss2Fn.apply(generate_img, ir.detach(), vis.detach())
    C = generate_img.shape[1]
    total = 0.0
    for size, coef in ((3, 10.0), (5, 10.0), (7, 1.0)):
        w = _lap_window(size, 2.0, generate_img.device, generate_img.dtype).expand(C, 1, size, size).contiguous()
        d = [t - F.conv2d(t, w, padding=size // 2, groups=C) for t in (generate_img, ir, vis)]
        total = total + coef * F.l1_loss(d[0], torch.maximum(d[1], d[2]))
    return total
