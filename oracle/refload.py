"""Container-only loader for the upstream reference (read-only at /root/reference).

TEST INFRASTRUCTURE — used only by oracle/make_golden.py to *generate* golden vectors.
Nothing here is imported by the product path, by `-m gpu` tests, by smoke() or by
bench.py: /root/reference does not exist on the GPU box.

SURVEY.md F1/F10: the published package is not importable as-is
(core/__init__.py:4 imports a class that does not exist; timm / mmcv are absent
here).  We register a synthetic `core` package and exec the three hot-path files by
path, with the four third-party symbols they need provided as minimal stand-ins:

  timm.models.layers.DropPath      per-sample Bernoulli(keep)/keep in train, identity in eval
  timm.models.layers.to_2tuple     x -> (x, x)
  timm.models.layers.trunc_normal_ init only
  mmcv.cnn.ConvModule              conv(bias=False when a norm follows) -> bn -> ReLU,
                                   sub-module names conv/bn/activate (mmcv 1.x semantics)
"""
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference"


def _install_third_party_stubs():
    import torch
    import torch.nn as nn

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")

        class DropPath(nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                if self.drop_prob == 0.0 or not self.training:
                    return x
                keep = 1.0 - self.drop_prob
                shape = (x.shape[0],) + (1,) * (x.ndim - 1)
                mask = x.new_empty(shape).bernoulli_(keep)
                return x * mask / keep

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
            return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

        timm_layers.DropPath = DropPath
        timm_layers.to_2tuple = to_2tuple
        timm_layers.trunc_normal_ = trunc_normal_
        timm.models = timm_models
        timm_models.layers = timm_layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = timm_models
        sys.modules["timm.models.layers"] = timm_layers

    if "mmcv" not in sys.modules:
        mmcv = types.ModuleType("mmcv")
        mmcv_cnn = types.ModuleType("mmcv.cnn")

        class ConvModule(nn.Module):
            def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, **kw):
                super().__init__()
                self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=norm_cfg is None)
                self.with_norm = norm_cfg is not None
                if self.with_norm:
                    self.bn = nn.BatchNorm2d(out_channels)
                self.activate = nn.ReLU(inplace=True)

            def forward(self, x):
                x = self.conv(x)
                if self.with_norm:
                    x = self.bn(x)
                return self.activate(x)

        mmcv_cnn.ConvModule = ConvModule
        mmcv_cnn.DepthwiseSeparableConvModule = ConvModule  # imported by the reference, never used
        mmcv.cnn = mmcv_cnn
        sys.modules["mmcv"] = mmcv
        sys.modules["mmcv.cnn"] = mmcv_cnn


def load_reference():
    """Returns (mix_transformer, segformer_head, model_fusion) reference modules."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (only available in the build container)")
    sys.dont_write_bytecode = True
    _install_third_party_stubs()
    pkg_name = "_segmif_ref_core"
    if pkg_name + ".model_fusion" in sys.modules:
        m = sys.modules
        return m[pkg_name + ".mix_transformer"], m[pkg_name + ".segformer_head"], m[pkg_name + ".model_fusion"]
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [os.path.join(REF_ROOT, "core")]
    sys.modules[pkg_name] = pkg
    mods = []
    for name in ("mix_transformer", "segformer_head", "model_fusion"):
        spec = importlib.util.spec_from_file_location(
            pkg_name + "." + name, os.path.join(REF_ROOT, "core", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[pkg_name + "." + name] = mod
        setattr(pkg, name, mod)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)
