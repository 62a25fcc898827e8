"""Names the reference's train.py pulls from `core` (train.py:111-112: a star-import of core/loss.py).

On the executed path are Fusionloss3 (round 1, core/loss.py:459-476), Fusionloss_grad3 (rounds >= 2,
:506-517), Sobelxy (:634-650) and RGB2YCrCb; they wrap segmif_amd.losses.  The other names train.py
imports are variants it never instantiates (dead code); they exist so that the import line
works unchanged and raise if somebody does call them.
"""
import torch.nn as nn

from .. import losses
from .model_fusion import RGB2YCrCb  # noqa: F401

__all__ = ["Sobelxy", "Fusionloss3", "Fusionloss_grad3", "LapLoss2", "RGB2YCrCb", "Total_fusion_loss", "Total_fusion_loss2",
           "Fusionloss", "Fusionloss_add", "Fusionloss2", "Fusionloss4"]


class Sobelxy(nn.Module):
    def forward(self, x):
        return losses.sobel_xy(x)


class Fusionloss3(nn.Module):
    """L1(mask_0, fused) + L1(Sobel(mask_0), Sobel(fused)); image_ir / image_vis are accepted and unused,
    as in the reference."""

    def __init__(self):
        super().__init__()
        self.sobelconv = Sobelxy()

    def forward(self, image_ir, image_vis, generate_img, mask):
        return losses.fusion_loss3(generate_img, mask)


class LapLoss2(nn.Module):
    """lap_loss.py:100-118: three Gaussian-difference levels (3 / 5 / 7 taps, sigma 2) of the fused image against the
    pixel-wise maximum of the same levels of the two sources; the `device` argument is accepted for signature
    compatibility (the windows are constants of the HIP kernel)."""

    def __init__(self, max_levels=3, channels=1, device=None):
        super().__init__()
        if max_levels != 3 or channels != 1:
            raise NotImplementedError("LapLoss2 is built for the reference's only use: 3 levels, single-channel images")
        self.max_levels = max_levels

    def forward(self, input, ir, vis):
        return losses.lap_loss2(input, ir, vis)


class Fusionloss_grad3(nn.Module):
    """MSE(mask_0, fused) + 1.1 * (1 - SSIM(fused, mask_0)).  Like the reference (core/loss.py:509) it owns a LapLoss2
    that its forward never evaluates; segmif_amd.train.FusionTrainer(report_lap=True) reports that term beside the loss."""

    def __init__(self):
        super().__init__()
        self.lap = LapLoss2()

    def forward(self, image_ir, image_vis, generate_img, mask):
        return losses.fusion_loss_grad3(generate_img, mask)


def _unused(name):
    class _Unused(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            raise NotImplementedError(f"core.{name} is not on any executed path of the reference's scripts "
                                      "(SURVEY.md §2) and is not implemented in segmif_amd")

    _Unused.__name__ = _Unused.__qualname__ = name
    return _Unused


Total_fusion_loss = _unused("Total_fusion_loss")
Total_fusion_loss2 = _unused("Total_fusion_loss2")
Fusionloss = _unused("Fusionloss")
Fusionloss_add = _unused("Fusionloss_add")
Fusionloss2 = _unused("Fusionloss2")
Fusionloss4 = _unused("Fusionloss4")
