"""SegFormer all-MLP decode head on the MI355X HIP kernels.

Mirror of the reference's core/segformer_head.py (:13-82): same names, signatures and state_dict
keys (`linear_c{1-4}.proj`, `linear_fuse.{conv,bn}`, `linear_pred`).  The reference builds
`linear_fuse` with mmcv's ConvModule (conv without bias -> BatchNorm2d -> ReLU); `ConvModule` below
is a parameter container with the same child names.

Data flow (NHWC): each scale's Linear writes (c1) or is bilinearly resized (c2..c4) straight into
its channel slice of one (B, H/4, W/4, 4E) buffer — the torch.cat of ref :77 never happens — then
the 1x1 fuse conv runs as a GEMM with eval-mode BatchNorm folded into its weight/bias and ReLU in
the epilogue, then the 1x1 prediction conv.
"""
import torch
import torch.nn as nn

from .. import autograd as ag
from .. import ops
from ._util import PackedCache, require_device, wants_grad

__all__ = ["MLP", "ConvModule", "SegFormerHead"]


class MLP(nn.Module):
    """Linear embedding of one feature scale (ref :13-24)."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)
        self._pk = PackedCache()

    def forward_nhwc(self, x_nhwc, out=None):
        return ops.linear(x_nhwc, self._pk.get("proj", self.proj.weight, ops.pack_weight),
                          self.proj.out_features, bias=self.proj.bias, out=out)

    def forward(self, x):
        """(B, C, H, W) -> (B, H*W, E) like the reference."""
        y = self.forward_nhwc(ops.to_nhwc(x))
        return y.view(y.shape[0], -1, y.shape[3])


class ConvModule(nn.Module):
    """conv (bias only without a norm) -> bn -> ReLU: child names follow mmcv 1.x."""

    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, **kwargs):
        super().__init__()
        if kernel_size != 1:
            raise NotImplementedError("the SegFormer head only uses a 1x1 fuse conv")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=norm_cfg is None)
        self.with_norm = norm_cfg is not None
        if self.with_norm:
            self.bn = nn.BatchNorm2d(out_channels)
        self.activate = nn.ReLU(inplace=True)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out", nonlinearity="relu")
        self._pk = PackedCache()

    def _folded(self):
        """(packed weight, bias) with eval-mode BatchNorm folded in: y = conv(x) * s + t."""
        if not self.with_norm:
            return self._pk.get("w", self.conv.weight, ops.pack_weight), self.conv.bias
        bn = self.bn
        srcs = (self.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)

        def fold():
            s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            w = (self.conv.weight.double().flatten(1) * s[:, None]).float()
            t = (bn.bias.double() - bn.running_mean.double() * s).float().contiguous()
            return ops.pack_weight(w.contiguous()), t

        return self._pk.get_multi("folded", srcs, fold)

    def forward_nhwc(self, x_nhwc):
        if self.training and self.with_norm:
            raise RuntimeError("ConvModule.forward_nhwc is the eval-mode (folded BatchNorm) path; train mode is "
                               "handled by SegFormerHead.forward_train_nhwc")
        w, b = self._folded()
        return ops.linear(x_nhwc, w, self.conv.out_channels, bias=b, act=ops.ACT_RELU)

    def forward(self, x):
        return ops.as_nchw(self.forward_nhwc(ops.to_nhwc(x)))


class SegFormerHead(nn.Module):
    def __init__(self, feature_strides=None, in_channels=128, embedding_dim=256, num_classes=20, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        assert len(feature_strides) == len(self.in_channels)
        assert min(feature_strides) == feature_strides[0]
        self.feature_strides = feature_strides
        c1, c2, c3, c4 = self.in_channels
        self.linear_c4 = MLP(input_dim=c4, embed_dim=embedding_dim)
        self.linear_c3 = MLP(input_dim=c3, embed_dim=embedding_dim)
        self.linear_c2 = MLP(input_dim=c2, embed_dim=embedding_dim)
        self.linear_c1 = MLP(input_dim=c1, embed_dim=embedding_dim)
        self.dropout = nn.Dropout2d(0.1)
        self.linear_fuse = ConvModule(in_channels=embedding_dim * 4, out_channels=embedding_dim, kernel_size=1,
                                      norm_cfg=dict(type='BN', requires_grad=True))
        self.linear_pred = nn.Conv2d(embedding_dim, self.num_classes, kernel_size=1)
        self.commute_resize = True  # eval: apply linear_fuse per scale before the bilinear resize (same function)
        self._pk = PackedCache()

    def _per_scale_fused(self):
        """Packed (E, C_s) matrices fuse_slot_s @ proj_s for s = c4, c3, c2, c1 (eval-mode BatchNorm folded into
        fuse) and the total shift  bn_shift + sum_s fuse_slot_s @ proj_s.bias ; composed in float64."""
        fuse, bn = self.linear_fuse, self.linear_fuse.bn
        mlps = (self.linear_c4, self.linear_c3, self.linear_c2, self.linear_c1)
        srcs = (fuse.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var) + tuple(
            p for m in mlps for p in (m.proj.weight, m.proj.bias))

        def build():
            # (r6) weight preparation, once per parameter version, on the HOST in float64 - like the optimizer's bias corrections:
            # ~1.5 MB down, 1 MB up; the device-side float64 `@` it replaces went through rocBLAS (Cijk_* / gemvt kernels in the
            # forward table)
            dev = fuse.conv.weight.device
            E = fuse.conv.out_channels
            h = lambda t: t.detach().to("cpu", torch.float64)
            s = h(bn.weight) / torch.sqrt(h(bn.running_var) + bn.eps)
            wf = h(fuse.conv.weight).flatten(1) * s[:, None]  # (E, 4E), columns in cat order [c4 c3 c2 c1]
            shift = h(bn.bias) - h(bn.running_mean) * s
            ws = []
            for slot, m in enumerate(mlps):
                blk = wf[:, slot * E:(slot + 1) * E]
                ws.append(ops.pack_weight((blk @ h(m.proj.weight)).float().contiguous().to(dev)))
                shift = shift + blk @ h(m.proj.bias)
            return ws, shift.float().contiguous().to(dev)

        return self._pk.get_multi("per_scale_fused", srcs, build)

    def forward_train_nhwc(self, feats):
        """autograd path: BatchNorm on batch statistics in train() mode (bn_colstats / bn_apply kernels, running statistics
        updated like nn.BatchNorm2d), on running statistics in eval() mode."""
        c1, c2, c3, c4 = feats
        B, H1, W1, _ = c1.shape
        fuse = self.linear_fuse
        # every scale writes its slice of the concatenated tensor in place (ag.Out placements + ag.join): no torch.cat, and the
        # backward reads the slices of the concatenated gradient in place as well
        E = self.linear_c1.proj.out_features
        whole = torch.empty((B, H1, W1, 4 * E), device=c1.device, dtype=torch.float32)
        parts = []
        for slot, (mlp, c) in enumerate(((self.linear_c4, c4), (self.linear_c3, c3), (self.linear_c2, c2))):
            parts.append(ag.bilinear(ag.linear(c.contiguous(), mlp.proj.weight, mlp.proj.bias), H1, W1,
                                     out=ag.Out(whole[..., slot * E:(slot + 1) * E])))
        parts.append(ag.linear(c1.contiguous(), self.linear_c1.proj.weight, self.linear_c1.proj.bias, out=ag.Out(whole[..., 3 * E:])))
        cat = ag.join(ag.Out(whole), *parts)
        w = fuse.conv.weight.flatten(1)
        b = fuse.conv.bias
        if fuse.with_norm and self.training:
            # train mode: batch statistics (biased variance for the normalisation, unbiased for the running
            # estimate, momentum 0.1 — nn.BatchNorm2d defaults, as mmcv's ConvModule builds it)
            bn = fuse.bn
            pre = ag.linear(cat, w, None)
            y, mean, var = ag.batchnorm_relu_train(pre, bn.weight, bn.bias, bn.eps)
            if bn.track_running_stats:
                with torch.no_grad():
                    n = pre.numel() // pre.shape[-1]
                    m = bn.momentum if bn.momentum is not None else 0.1
                    bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
                    bn.num_batches_tracked += 1
        else:
            if fuse.with_norm:  # fold eval-mode BN differentiably: grads reach conv.weight, bn.weight, bn.bias
                s = fuse.bn.weight / torch.sqrt(fuse.bn.running_var + fuse.bn.eps)
                w = w * s[:, None]
                b = fuse.bn.bias - fuse.bn.running_mean * s
            y = ag.linear(cat, w, b, act=ops.ACT_RELU)
        if self.training and self.dropout.p > 0:
            # Dropout2d(0.1) zeroes whole channels per sample and scales the rest by 1 / keep (ref :1090-1097 via mmseg's head):
            # (y * m_b) @ W^T == y @ (W * m_b)^T, so the mask goes onto per-image copies of linear_pred's (classes x E) weight - a
            # few KB - instead of two full passes over the (B, H/4, W/4, E) tensor (forward multiply + its backward)
            E = y.shape[-1]
            keep = 1.0 - self.dropout.p
            if keep <= 0.0:  # Dropout2d(p = 1): every channel dropped - zeros, not 0 / 0
                mask = torch.zeros((B, 1, E), device=y.device, dtype=torch.float32)
            else:
                mask = torch.empty((B, 1, E), device=y.device, dtype=torch.float32).bernoulli_(keep) / keep
            wb = self.linear_pred.weight.flatten(1).unsqueeze(0) * mask
            return ag.batched_linear(y.contiguous().reshape(B, H1 * W1, E), wb, self.linear_pred.bias).view(B, H1, W1, -1)
        return ag.linear(y, self.linear_pred.weight, self.linear_pred.bias)

    def forward_nhwc(self, feats):
        """feats: [c1..c4] NHWC -> logits NHWC (B, H/4, W/4, num_classes)."""
        if wants_grad(self, *feats) or (self.training and self.linear_fuse.with_norm):
            return self.forward_train_nhwc(feats)  # also used under no_grad in train mode (batch-stat BN)
        c1, c2, c3, c4 = feats
        B, H1, W1, _ = c1.shape
        E = self.linear_c1.proj.out_features
        if self.commute_resize and not self.training and self.linear_fuse.with_norm:
            # linear_fuse is a 1x1 conv over the concatenation, i.e. a sum over scales of 1x1 convs of resized maps,
            # and a 1x1 conv commutes with a bilinear resize: fuse at each scale's own resolution (composed
            # with the scale's Linear into one C_s -> E matrix), then resize, sum, shift, ReLU in one pass.
            # No (B, H/4, W/4, 4E) buffer, 3x fewer FLOPs.
            ws, b_total = self._per_scale_fused()
            low = [ops.linear(c.contiguous(), w, E) for w, c in zip(ws[:3], (c4, c3, c2))]
            p1 = ops.linear(c1, ws[3], E)
            y = ops.upsum_act(p1, low, H1, W1, bias=b_total, act=ops.ACT_RELU)
            return ops.linear(y, self._pk.get("pred", self.linear_pred.weight, ops.pack_weight), self.num_classes,
                              bias=self.linear_pred.bias)
        cat = torch.empty((B, H1, W1, 4 * E), device=c1.device, dtype=torch.float32)
        for slot, (mlp, c) in enumerate(((self.linear_c4, c4), (self.linear_c3, c3), (self.linear_c2, c2))):
            ops.bilinear(mlp.forward_nhwc(c), H1, W1, out=cat[..., slot * E:(slot + 1) * E])
        self.linear_c1.forward_nhwc(c1, out=cat[..., 3 * E:])
        y = self.linear_fuse.forward_nhwc(cat)
        if self.training:
            y = ops.to_nhwc(self.dropout(ops.as_nchw(y)))  # Dropout2d: whole channels, train mode only
        return ops.linear(y, self._pk.get("pred", self.linear_pred.weight, ops.pack_weight), self.num_classes,
                          bias=self.linear_pred.bias)

    def forward(self, x):
        c1 = x[0]
        require_device(c1, "SegFormerHead input")
        return ops.as_nchw(self.forward_nhwc([ops.to_nhwc(c) for c in x]))
