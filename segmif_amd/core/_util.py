"""Host-side helpers shared by the module mirror (no arithmetic on activations except the
train-mode stochastic-depth mask)."""
import math

import torch
import torch.nn as nn


class PackedCache:
    """Caches kernel-layout copies of parameters (tap-major conv weights, folded BN, ...).

    An entry is rebuilt whenever any of the source tensors changed storage or was modified in
    place (optimizer step, load_state_dict, .to(device)) — tracked through data_ptr/_version."""

    def __init__(self):
        self._entries = {}

    @staticmethod
    def _tag(params):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in params)

    def get(self, key, param, fn):
        return self.get_multi(key, (param,), lambda: fn(param))

    def get_multi(self, key, params, fn):
        tag = self._tag(params)
        hit = self._entries.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, fn())
            self._entries[key] = hit
        return hit[1]

    def clear(self):
        self._entries.clear()

    # caches hold device tensors derived from parameters: never copy / pickle them with the module
    def __deepcopy__(self, memo):
        return PackedCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._entries = {}


def init_reference_style(module):
    """Initialisation rule the reference applies inside MiT and the FeatureFusionModule
    (core/mix_transformer.py:31-44, core/model_fusion.py:438-451): truncated-normal(0.02) Linear
    weights, unit LayerNorm, fan-out-scaled normal Conv2d weights, zero biases."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                nn.init.zeros_(m.bias)


def drop_path_scale(y, drop_prob):
    """timm DropPath in train mode: per-sample Bernoulli(keep) / keep."""
    keep = 1.0 - drop_prob
    mask = y.new_empty((y.shape[0],) + (1,) * (y.dim() - 1)).bernoulli_(keep)
    return y * (mask / keep)


def require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"segmif_amd.core: {what} lives on {t.device}; this package runs on the MI355X HIP kernels only "
            "(no CPU fallback). Move the module and its inputs to the GPU with .cuda().")


def wants_grad(module, *tensors):
    """True when this call must record an autograd graph: grad mode is on and either an input or one
    of the module's parameters requires grad.  The HIP inference path (in-place buffers, cached
    packed weights) is used otherwise."""
    if not torch.is_grad_enabled():
        return False
    if any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in module.parameters())
