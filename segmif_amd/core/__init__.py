"""Drop-in counterpart of the reference's `core` package for the hot path.

The reference's own core/__init__.py imports a class that does not exist (`Network`)
and star-imports core/loss.py; here the hot-path names are exported, `Network` is an alias of
`Network3` so that `import core` succeeds, and the loss names train.py imports come from .loss.
"""
from . import loss, mix_transformer, model_fusion, segformer_head  # noqa: F401
from .loss import *  # noqa: F401,F403
from .mix_transformer import *  # noqa: F401,F403
from .model_fusion import (DRDB, CrossAttention, CrossAttention2, CrossPath, FeatureFusionModule,  # noqa: F401
                           Fusion_Network3_ac, Mean, Network3, RGB2YCrCb, WeTr, YCrCb2RGB, fuse_to_rgb)
from .segformer_head import SegFormerHead  # noqa: F401

Network = Network3
