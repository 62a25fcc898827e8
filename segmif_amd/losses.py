"""Fusion losses used by the reference's train_fusion (core/loss.py:459-476 Fusionloss3, :506-517
Fusionloss_grad3, :634-650 Sobelxy; pytorch_ssim/__init__.py:8-43).

On the GPU both objectives are fused HIP kernels, forward and backward (csrc/losses.hip + the separable blur of
csrc/rowops.hip; autograd.FusionLossGrad3Fn / FusionLoss3Fn): SURVEY §8(f) N1.  The torch formulations below are what
the CPU tests pin against the reference (tests/golden/losses.npz) and what non-fp32 / multi-channel inputs fall back to.
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
    if img1.is_cuda and window_size == 11 and img1.dtype == torch.float32:
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


def _hip_ok(generate_img, m):
    return generate_img.is_cuda and generate_img.dtype == torch.float32 and m.dtype == torch.float32 \
        and generate_img.shape == m.shape and generate_img.shape[1] == 1


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
