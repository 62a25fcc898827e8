"""Fusion + segmentation networks of SegMiF on the MI355X HIP kernels.

Mirror of the classes the reference's scripts use from core/model_fusion.py: WeTr (:9-68),
RGB2YCrCb / YCrCb2RGB (:69-111), DRDB (:117-157), CrossAttention / CrossAttention2 (:250-328),
CrossPath (:329-361), FeatureFusionModule (:430-463), Fusion_Network3_ac (:1026-1067),
Network3 (:1068-1104) — same constructor/forward signatures and state_dict keys.  (r6) The file's
ablation / variant classes (Fusion_Network3 and the rest) are in .variants and re-exported here.

Layout: every feature map is NHWC.  A DRDB owns one (B, H, W, 224) buffer; each dilated conv reads
the first Cin channels and writes its 32 output channels in place (the five torch.cat copies of
ref :137-153 do not exist).  The cross-modal interaction is linear attention: its K^T V reduction,
softmax and Q@ctx are folded into a per-image end_proj weight (csrc/linattn.hip).
"""
import torch
import torch.nn as nn

from .. import autograd as ag
from .. import ops
from . import mix_transformer
from ._util import PackedCache, init_reference_style, require_device, wants_grad
from .segformer_head import SegFormerHead

import os

_DRDB_RES_PLANES = os.environ.get("SEGMIF_DRDB_RES", "planes") != "fp32"
_CONV1_STENCIL = os.environ.get("SEGMIF_CONV1", "stencil") != "igemm"  # A/B switch (r6): conv1_ir / conv1_vis as a stencil kernel

__all__ = ["WeTr", "RGB2YCrCb", "YCrCb2RGB", "DRDB", "CrossAttention", "CrossAttention2", "CrossPath",
           "FeatureFusionModule", "Fusion_Network3_ac", "Network3", "Mean", "fuse_to_rgb"]


class WeTr(nn.Module):
    def __init__(self, backbone, num_classes=20, embedding_dim=256, pretrained=None):
        super().__init__()
        self.num_classes = num_classes
        self.embedding_dim = embedding_dim
        self.backbone = backbone
        self.feature_strides = [4, 8, 16, 32]
        self.encoder = getattr(mix_transformer, backbone)()
        self.in_channels = self.encoder.embed_dims
        if pretrained:
            self.initialize()
        self.decoder = SegFormerHead(feature_strides=self.feature_strides, in_channels=self.in_channels,
                                     embedding_dim=self.embedding_dim, num_classes=self.num_classes)
        self.classifier = nn.Conv2d(in_channels=self.in_channels[-1], out_channels=self.num_classes,
                                    kernel_size=1, bias=False)

    def initialize(self):
        state_dict = torch.load('pretrained/' + self.backbone + '.pth')
        state_dict.pop('head.weight')
        state_dict.pop('head.bias')
        self.encoder.load_state_dict(state_dict)

    def get_param_groups(self):
        """[encoder non-norm, encoder norm, decoder + classifier]  (ref :44-60)."""
        groups = [[], [], []]
        for name, param in self.encoder.named_parameters():
            groups[1 if "norm" in name else 0].append(param)
        groups[2].extend(self.decoder.parameters())
        groups[2].append(self.classifier.weight)
        return groups

    def forward_nhwc(self, x):
        # the reference also evaluates classifier(_x4) and discards it (ref :66); it does not
        # influence the result and is not computed here.
        return self.decoder.forward_nhwc(self.encoder.forward_features_nhwc(x))

    def forward(self, x):
        require_device(x, "WeTr input")
        return ops.as_nchw(self.forward_nhwc(x))


def RGB2YCrCb(input_im):
    """(B,3,H,W) RGB -> YCrCb (ref :69-91): one HIP kernel, with its backward (autograd.Rgb2YCrCbFn); CPU tensors take the
    torch formulation below (test infrastructure: the CPU tests pin it against the reference)."""
    if input_im.is_cuda:
        return ag.rgb2ycrcb(input_im)
    R, G, B = input_im[:, 0:1], input_im[:, 1:2], input_im[:, 2:3]
    Y = 0.299 * R + 0.587 * G + 0.114 * B
    return torch.cat((Y, (R - Y) * 0.713 + 0.5, (B - Y) * 0.564 + 0.5), dim=1)


def YCrCb2RGB(input_im, y=None):
    """(B,3,H,W) YCrCb -> RGB (ref :93-111): one HIP kernel, with its backward (autograd.YCrCb2RgbFn).  y (B,1,H,W), an
    extension of the reference signature: takes the place of channel 0 (train.py:362-365's clone + slice assignment).  CPU
    tensors take the torch formulation below (test infrastructure)."""
    if input_im.is_cuda:
        return ag.ycrcb2rgb(input_im, y)
    if y is not None:
        input_im = torch.cat((y, input_im[:, 1:]), dim=1)
    mat = input_im.new_tensor([[1.0, 1.0, 1.0], [1.403, -0.714, 0.0], [0.0, -0.344, 1.773]])
    bias = input_im.new_tensor([0.0, -0.5, -0.5])
    flat = input_im.permute(0, 2, 3, 1).reshape(-1, 3)
    out = (flat + bias).mm(mat)
    return out.reshape(input_im.shape[0], input_im.shape[2], input_im.shape[3], 3).permute(0, 3, 1, 2)


def fuse_to_rgb(vis, y_fused):
    """test_fusion.py:102-111 in one kernel: clamp01(YCrCb2RGB([y_fused, Cr(vis), Cb(vis)]))."""
    return ops.fuse_ycrcb(vis, y_fused)


class DRDB(nn.Module):
    def __init__(self, in_ch=64, growth_rate=32):
        super().__init__()
        self.in_ch, self.growth = in_ch, growth_rate
        ch = in_ch
        for i in range(1, 6):
            setattr(self, f"Dcov{i}", nn.Conv2d(ch, growth_rate, 3, padding=2, dilation=2))
            ch += growth_rate
        self.conv = nn.Conv2d(ch, in_ch, 1, padding=0)
        self.total_ch = ch
        self._pk = PackedCache()

    def new_buffer(self, B, H, W, device):
        return torch.empty((B, H, W, self.total_ch), device=device, dtype=torch.float32)

    def forward_buffer(self, buf, out=None):
        """buf: (B,H,W,224) whose first in_ch channels hold x. Returns x + relu(conv1x1(concat))."""
        ch = self.in_ch
        for i in range(1, 6):
            conv = getattr(self, f"Dcov{i}")
            ops.conv2d(buf[..., :ch], self._pk.get(f"d{i}:{ops.conv3x3_mode()}", conv.weight, ops.pack_conv3x3), self.growth, 3,
                       pad=2, dil=2, bias=conv.bias, act=ops.ACT_RELU, out=buf[..., ch:ch + self.growth], tag="drdb_dcov")
            ch += self.growth
        return ops.linear(buf, self._pk.get("conv", self.conv.weight, ops.pack_weight), self.in_ch,
                          bias=self.conv.bias, act=ops.ACT_RELU, res=buf[..., :self.in_ch], out=out)

    PLANES_CHUNKS = 12  # 64 input + 4 x 32 grown channels; the fifth conv's result never leaves the kernel

    def forward_planes(self, x, planes, out=None, preloaded=False):
        """Inference on pre-split activations (csrc/conv3x3_planes.hip): x (B,H,W,64) fp32 rows view, planes a
        12-chunk ops.Planes scratch buffer.  Dcov1-4 read / append chunk images, Dcov5 also carries the closing
        1x1 conv + ReLU + residual (ref :153-157), so the 224-channel concat is never materialised in fp32.
        x = None (r4; f16x3 planes already holding the input as chunks 0..3): the residual is read back from those chunks
        (hi + 2^-11 lo), so the producer - conv1, the CrossPath tail - need not write an fp32 copy at all."""
        B, H, W = planes.B, planes.H, planes.W
        if x is None and not (preloaded and planes.f16):
            raise RuntimeError("DRDB.forward_planes: x = None needs f16x3 planes preloaded with the input")
        if out is None:
            out = torch.empty((B, H, W, self.in_ch), device=planes.data.device, dtype=torch.float32)
        if not preloaded:  # (a producer may already have written x as chunks 0..3, e.g. the CrossPath tail)
            planes.load_f32(x, 0)
        ch = self.in_ch
        sfx, pack = ("h", ops.pack_weight_planes16) if planes.f16 else ("", ops.pack_weight_planes)
        for i in range(1, 6):
            conv = getattr(self, f"Dcov{i}")
            wt = self._pk.get(f"p{i}{sfx}", conv.weight, pack)
            if i < 5:
                ops.conv3x3_planes(planes, ch, wt, dil=2, bias=conv.bias, act=ops.ACT_RELU, out_chunk0=ch // 16,
                                   tag="drdb_dcov")
            else:
                w1 = self._pk.get(f"p1x1{sfx}", self.conv.weight, pack)
                ops.conv3x3_planes(planes, ch, wt, dil=2, bias=conv.bias, act=ops.ACT_RELU,
                                   tail=(w1, self.conv.bias, x, out, ops.ACT_RELU, x is None), tag="drdb_tail")
            ch += self.growth
        return out

    def planes_ok(self):
        """The planes kernels take 16-byte bias loads: parameters living at odd offsets of a flattened buffer fall back to
        the fp32-buffer DRDB (forward_buffer) instead of failing."""
        return ops.conv3x3_mode() in ("planes", "planes16") and self.in_ch == 64 and self.growth == 32 \
            and ops.aligned16(self.conv.bias, *(getattr(self, f"Dcov{i}").bias for i in range(1, 6)))

    def _params(self):
        ps = []
        for i in range(1, 6):
            conv = getattr(self, f"Dcov{i}")
            ps += [conv.weight, conv.bias]
        return ps + [self.conv.weight, self.conv.bias]

    def forward_train_nhwc(self, x, home=None):
        """home: ag.Out of a new_buffer() whose first in_ch channels x already is (its producer wrote there): no copy in."""
        return ag.drdb(x if home is not None else x.contiguous(), self._params(), home)

    def forward(self, x):
        require_device(x, "DRDB input")
        if wants_grad(self, x):
            return self.forward_train_nhwc(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        B, _, H, W = x.shape
        if self.planes_ok():
            xh = ops.to_nhwc(x)
            # 'planes16': half pairs inside a guarded scope; repeated on bf16 triples if a tensor left the half's range
            return ops.as_nchw(ops.run_guarded(
                lambda: self.forward_planes(xh, ops.Planes(B, H, W, self.PLANES_CHUNKS, x.device, ops.active_guard())),
                x.device, enabled=ops.conv3x3_mode() == "planes16"))
        buf = self.new_buffer(B, H, W, x.device)
        buf[..., :self.in_ch].copy_(x.permute(0, 2, 3, 1))
        return ops.as_nchw(self.forward_buffer(buf))


class _LinearCrossAttention(nn.Module):
    """Shared machinery of CrossAttention / CrossAttention2 (ref :250-328): softmax_{dim=-2}((K^T V) * scale) per head is an
    8 x 8 "context" ctx[b][h][i][j]; applying it to a query, out[b, n, 8 h + j] = sum_i x[b, n, 8 h + i] ctx[b, h, i, j]
    with the heads re-interleaved (ref :285-286, :325-326), is one GEMM per image against the block-diagonal 64 x 64
    matrix of that image's eight contexts.  The kernels (csrc/linattn.hip) are built for the one configuration SegMiF
    instantiates: dim 64, 8 heads of 8."""

    def _check(self):
        if self.dim != 64 or self.num_heads != 8:
            raise NotImplementedError("the linear-attention kernels are built for dim 64, 8 heads of 8 (the only "
                                      "configuration SegMiF instantiates)")

    def _partial(self, lin, name, x):
        if lin.bias is None:  # the configuration SegMiF uses: fused projection + reduction
            return ops.linattn_kvpartial(x, lin.weight, self.num_heads)
        kv = ops.linear(x, self._pk.get(name, lin.weight, ops.pack_weight), 2 * self.dim, bias=lin.bias)
        return ops.linattn_partial(kv, self.num_heads)

    def _context_weight(self, lin, name, x):
        """Inference: (B, 64, 64) block-diagonal W with q @ ctx == q @ W^T, straight from the fp64 K^T V partial sums
        (segmif_linattn_fold_f32 against an identity end_proj)."""
        part = self._partial(lin, name, x)
        eye = self._pk.get_multi(f"eye:{x.device}", (), lambda: torch.eye(self.dim, device=x.device, dtype=torch.float32))
        w = torch.empty((x.shape[0], self.dim, self.dim), device=x.device, dtype=torch.float32)
        return ops.linattn_fold(part, eye, w, wofs=0, kofs=0, scale=self.scale, heads=self.num_heads)

    def _context_weight_train(self, lin, x):
        if lin.bias is not None:
            raise NotImplementedError("training through a linear cross attention with qkv_bias=True is not implemented "
                                      "(SegMiF builds these modules without bias)")
        ctx = torch.softmax(ag.kv_context(x, lin.weight) * self.scale, dim=-2)  # (B, 8, 8, 8) fp64, [h][i][j]
        return _block_diag_batched(ctx).float()

    @staticmethod
    def _tokens(*ts):
        for t in ts:
            require_device(t, "CrossAttention input")
            if t.dim() != 3 or t.shape[-1] != 64:
                raise RuntimeError(f"CrossAttention expects (B, N, 64) tokens, got {tuple(t.shape)}")
        return [t.contiguous() for t in ts]


def _block_diag_batched(ctx):
    """(B, h, d, d) contexts [i][j] -> (B, h d, h d) with W[b][h d + j][h d + i] = ctx[b][h][i][j] (q @ ctx == q @ W^T)."""
    B, h, d, _ = ctx.shape
    eye = torch.eye(h, device=ctx.device, dtype=ctx.dtype).view(1, h, 1, h, 1)
    return (ctx.transpose(2, 3).reshape(B, h, d, 1, d) * eye).reshape(B, h * d, h * d)  # one broadcast product, not h slice writes


class CrossAttention(_LinearCrossAttention):
    """Context from the segmentation feature, queries = the two modality features (ref :250-288)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim, self.num_heads = dim, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.kv3 = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self._pk = PackedCache()

    def context_partial(self, seg):
        return self._partial(self.kv3, "kv3", seg)

    def forward(self, x1, x2, segfeature):
        """(B, N, C) x 3 -> (q1 @ ctx3, q2 @ ctx3), ctx3 = softmax_{dim=-2}(K3^T V3 * scale) per head (ref :263-288)."""
        self._check()
        x1, x2, seg = self._tokens(x1, x2, segfeature)
        if wants_grad(self, x1, x2, seg):
            w = self._context_weight_train(self.kv3, seg)
            return ag.batched_linear(x1, w), ag.batched_linear(x2, w)
        w = self._context_weight(self.kv3, "kv3", seg)
        return ops.linear(x1, w, self.dim, batched_weight=True), ops.linear(x2, w, self.dim, batched_weight=True)


class CrossAttention2(_LinearCrossAttention):
    """Contexts from each modality, query = the segmentation feature (ref :290-328)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim, self.num_heads = dim, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.kv1 = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.kv2 = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self._pk = PackedCache()

    def context_partial(self, which, x):
        return self._partial(self.kv1 if which == 1 else self.kv2, f"kv{which}", x)

    def forward(self, x1, x2, segfeature):
        """(B, N, C) x 3 -> (q3 @ ctx1, q3 @ ctx2), ctx_i = softmax_{dim=-2}(K_i^T V_i * scale) per head (ref :303-328)."""
        self._check()
        x1, x2, seg = self._tokens(x1, x2, segfeature)
        if wants_grad(self, x1, x2, seg):
            return (ag.batched_linear(seg, self._context_weight_train(self.kv1, x1)),
                    ag.batched_linear(seg, self._context_weight_train(self.kv2, x2)))
        w1 = self._context_weight(self.kv1, "kv1", x1)
        w2 = self._context_weight(self.kv2, "kv2", x2)
        return ops.linear(seg, w1, self.dim, batched_weight=True), ops.linear(seg, w2, self.dim, batched_weight=True)


class CrossPath(nn.Module):
    def __init__(self, dim, reduction=1, num_heads=8, norm_layer=nn.LayerNorm):
        super().__init__()
        if reduction != 1 or dim // num_heads != 8 or num_heads != 8:
            raise NotImplementedError("the linear-attention kernels are built for dim 64, 8 heads of 8 (the "
                                      "only configuration SegMiF instantiates)")
        self.dim = dim
        self.channel_proj1 = nn.Linear(dim, dim * 2)
        self.channel_proj2 = nn.Linear(dim, dim * 2)
        self.channel_proj3 = nn.Linear(dim, dim * 2)
        self.act1 = nn.ReLU(inplace=True)
        self.act2 = nn.ReLU(inplace=True)
        self.act3 = nn.ReLU(inplace=True)
        self.cross_attn = CrossAttention(dim, num_heads=num_heads)
        self.cross_attn2 = CrossAttention2(dim, num_heads=num_heads)
        self.end_proj1 = nn.Linear(dim * 2, dim)
        self.end_proj2 = nn.Linear(dim * 2, dim)
        self.norm1 = norm_layer(dim)
        self.norm2 = norm_layer(dim)
        self._pk = PackedCache()

    def forward_tokens(self, x1, x2, seg, out1=None, out2=None):
        """x1, x2, seg: (B, N, 64) rows views.  Returns LN(x_i + end_proj_i(cat(z_i, v_i))) written to out_i."""
        C = self.dim
        pk = self._pk
        proj = []
        for i, x in ((1, x1), (2, x2), (3, seg)):
            lin = getattr(self, f"channel_proj{i}")
            proj.append(ops.linear(x, pk.get(f"cp{i}", lin.weight, ops.pack_weight), 2 * C, bias=lin.bias,
                                   act=ops.ACT_RELU))
        p1, p2, p3 = proj  # each (B, N, 128) = [y_i | u_i]   (ref :351-353)
        part3 = self.cross_attn.context_partial(p3[..., C:])  # ctx3 from u3
        part1 = self.cross_attn2.context_partial(1, p1[..., :C])  # ctx1 from y1
        part2 = self.cross_attn2.context_partial(2, p2[..., :C])  # ctx2 from y2
        B = x1.shape[0]
        outs = []
        for i, (x, p, part, o) in enumerate(((x1, p1, part1, out1), (x2, p2, part2, out2)), start=1):
            end = getattr(self, f"end_proj{i}")
            weff = torch.empty((B, C, 2 * C), device=x.device, dtype=torch.float32)
            # cat(z_i, v_i) @ Wend^T  ==  [y3 | u_i] @ Weff^T with the contexts folded in (ref :357-360)
            ops.linattn_fold(part, end.weight, weff, wofs=0, kofs=0, scale=self.cross_attn2.scale)
            ops.linattn_fold(part3, end.weight, weff, wofs=C, kofs=C, scale=self.cross_attn.scale)
            norm = getattr(self, f"norm{i}")
            # LN(x + [y3 | u_i] @ Weff^T + b): GEMM, residual and LayerNorm in one kernel
            outs.append(ops.linear(p3[..., :C], weff, C, bias=end.bias, res=x, x2=p[..., C:], batched_weight=True,
                                   ln=(norm.weight, norm.bias, norm.eps), out=o))
        return outs[0], outs[1]

    def forward_tokens_gram(self, x1, x2, seg, out1=None, out2=None, planes1=None, planes2=None, hw=None, planes_only=False):
        """forward_tokens without the 128-wide intermediates (csrc/crosspath.hip): K^T V = Wk (Y^T Y) Wv^T needs only the
        Gram matrix of each projected half, and each consumer recomputes the 64-wide half of channel_proj it needs.
        Three Gram passes + two fused tails instead of three GEMMs, three kv reductions and two two-source GEMMs.
        planes_i: optional ops.Planes receiving out_i as chunks 0..3 (the next DRDB's input, already split)."""
        C = self.dim
        cp = [getattr(self, f"channel_proj{i}") for i in (1, 2, 3)]
        halves = self._pk.get_multi("cp_halves", [m.weight for m in cp],
                                    lambda: [(m.weight[:C].contiguous(), m.weight[C:].contiguous()) for m in cp])
        bias = [(m.bias[:C], m.bias[C:]) if m.bias is not None else (None, None) for m in cp]
        g1 = ops.crosspath_gram(x1, halves[0][0], bias[0][0])   # y1 -> ctx1 (cross_attn2.kv1)
        g2 = ops.crosspath_gram(x2, halves[1][0], bias[1][0])   # y2 -> ctx2 (cross_attn2.kv2)
        lazy = isinstance(seg, ops.LazySeg)
        if lazy:
            # (r5) the segmentation feature stays at its own resolution: channel_proj3 (both halves, no ReLU yet) runs on the low-
            # resolution map - a resize is a convex combination per channel, so Linear(resize(x)) = resize(Linear(x)) - and the
            # Gram / tail kernels resize the projected rows as they read them (csrc/crosspath.hip, LAZY)
            proj3 = ops.linear(seg.low, self._pk.get("cp3", cp[2].weight, ops.pack_weight), 2 * C, bias=cp[2].bias)
            seg = proj3[..., :C]  # y half: the tails' x3
            g3 = ops.crosspath_gram_lazy(proj3[..., C:], hw[0], hw[1])  # u3 -> ctx3 (cross_attn.kv3)
        else:
            g3 = ops.crosspath_gram(seg, halves[2][1], bias[2][1])  # u3 -> ctx3 (cross_attn.kv3)
        B = x1.shape[0]
        outs = []
        for i, (x, g, kv, o, pl) in enumerate(((x1, g1, self.cross_attn2.kv1, out1, planes1),
                                               (x2, g2, self.cross_attn2.kv2, out2, planes2)), start=1):
            end = getattr(self, f"end_proj{i}")
            norm = getattr(self, f"norm{i}")
            weff = torch.empty((B, C, 2 * C), device=x.device, dtype=torch.float32)
            ops.crosspath_fold(g, kv.weight, end.weight, weff, wofs=0, kofs=0, scale=self.cross_attn2.scale)
            ops.crosspath_fold(g3, self.cross_attn.kv3.weight, end.weight, weff, wofs=C, kofs=C, scale=self.cross_attn.scale)
            outs.append(ops.crosspath_tail(seg, x, halves[2][0], bias[2][0], halves[i - 1][1], bias[i - 1][1], weff, end.bias,
                                           (norm.weight, norm.bias, norm.eps), out=o, planes=pl, hw=hw,
                                           planes_only=planes_only, lazy=lazy))
        return outs[0], outs[1]

    def gram_ok(self):
        return ops.crosspath_mode() == "gram" and self.cross_attn.kv3.bias is None and self.cross_attn2.kv1.bias is None \
            and self.cross_attn2.kv2.bias is None \
            and ops.aligned16(*(p for p in self.parameters() if p.dim() == 1))  # 16-byte bias / LayerNorm loads in the tail

    def forward_tokens_train(self, x1, x2, seg, out1=None, out2=None):
        """autograd path: every node a HIP Function - the heavy contractions and, (r6), the 8x8 context softmaxes with their fold
        into end_proj (ag.context_fold; tensors of a few KB).  out_i: optional ag.Out placements of the two results.
        (r4) Three nodes carry the full-resolution tensors - ag.cross_proj (the three channel_proj with every 64-channel half
        its own output), ag.kv_context x 3, ag.tail_pair (both closing projections) - arranged so that each big tensor has one
        consumer: no autograd accumulation passes, no zero-padded slice gradients."""
        C = self.dim
        cp = [getattr(self, f"channel_proj{i}") for i in (1, 2, 3)]
        # (the sink lets the consumers' backward GEMMs write each half's gradient through its ReLU mask into cross_proj's buffer)
        sink = ag.ProjSink()
        y1, u1, y2, u2, y3, u3, x1r, x2r = ag.cross_proj(x1, x2, seg, cp[0].weight, cp[0].bias, cp[1].weight, cp[1].bias,
                                                         cp[2].weight, cp[2].bias, sink)
        # (r6) the 8 x 8 context softmaxes and their fold into end_proj are one autograd node with HIP kernels on both sides
        # (ag.context_fold): no torch softmax / einsum / cat on the way
        k3 = ag.kv_context(u3, self.cross_attn.kv3.weight, sink, (2, 1))
        k1 = ag.kv_context(y1, self.cross_attn2.kv1.weight, sink, (0, 0))
        k2 = ag.kv_context(y2, self.cross_attn2.kv2.weight, sink, (1, 0))
        weffs = [ag.context_fold(k_i, k3, end.weight, self.cross_attn2.scale, self.cross_attn.scale)
                 for end, k_i in ((self.end_proj1, k1), (self.end_proj2, k2))]
        # x_i + [y3 | u_i] @ Weff_i^T + b_i for both modalities as one node (two-source GEMMs, residual in the epilogue)
        t1, t2 = ag.tail_pair(y3, u1, u2, weffs[0], weffs[1], self.end_proj1.bias, self.end_proj2.bias, x1r, x2r, sink)
        return (ag.layernorm(t1, self.norm1.weight, self.norm1.bias, self.norm1.eps, out=out1),
                ag.layernorm(t2, self.norm2.weight, self.norm2.bias, self.norm2.eps, out=out2))

    def forward(self, x1, x2, segfeature):
        require_device(x1, "CrossPath input")
        if wants_grad(self, x1, x2, segfeature):
            return self.forward_tokens_train(x1.contiguous(), x2.contiguous(), segfeature.contiguous())
        fn = self.forward_tokens_gram if self.gram_ok() else self.forward_tokens
        return fn(x1.contiguous(), x2.contiguous(), segfeature.contiguous())


class FeatureFusionModule(nn.Module):
    def __init__(self, dim, reduction=1, num_heads=8, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.cross = CrossPath(dim=dim, reduction=reduction, num_heads=num_heads)
        init_reference_style(self)

    def forward_nhwc(self, x1, x2, seg, out1=None, out2=None, planes1=None, planes2=None, planes_only=False):
        """NHWC in / out; out_i may be channel slices of wider buffers (e.g. a DRDB concat buffer); planes_i: optional
        ops.Planes that also receive out_i pre-split (inference on the Gram path only); planes_only: the planes are the only
        output (returns None, None)."""
        B, H, W, C = x1.shape
        guard = ops.active_guard()
        if guard is not None:  # (r5) the conditioning words of this interaction's context softmaxes go to their own row
            guard.next_interaction()
        lazy = isinstance(seg, ops.LazySeg)
        if lazy and (seg.H, seg.W) != (H, W):
            raise RuntimeError("FeatureFusionModule: the lazy segmentation feature targets another image size")
        if lazy and not (self.cross.gram_ok() and seg.fits() and not wants_grad(self, x1, x2)):
            seg, lazy = seg.materialise(), False  # (only the Gram-form inference kernels resize as they read)
        if lazy:
            r1, r2 = self.cross.forward_tokens_gram(x1.view(B, H * W, C), x2.view(B, H * W, C), seg,
                                                    None if out1 is None else out1.view(B, H * W, out1.shape[-1]),
                                                    None if out2 is None else out2.view(B, H * W, out2.shape[-1]),
                                                    planes1, planes2, (H, W), planes_only=planes_only)
            return (None, None) if planes_only else (r1.view(B, H, W, C), r2.view(B, H, W, C))
        if wants_grad(self, x1, x2, seg):
            # (out_i: ag.Out placements here - the next DRDB's buffer or the halves of conv2's input)
            r1, r2 = self.cross.forward_tokens_train(x1.reshape(B, H * W, C), x2.reshape(B, H * W, C),
                                                     seg.reshape(B, H * W, C), out1, out2)
            return r1.view(B, H, W, C), r2.view(B, H, W, C)
        tok = lambda t: None if t is None else t.view(B, H * W, t.shape[-1])
        if self.cross.gram_ok():
            r1, r2 = self.cross.forward_tokens_gram(tok(x1), tok(x2), tok(seg), tok(out1), tok(out2), planes1, planes2, (H, W),
                                                    planes_only=planes_only)
            if planes_only:
                return None, None
        else:
            if planes1 is not None or planes2 is not None:
                raise RuntimeError("planes outputs need the Gram CrossPath path")
            r1, r2 = self.cross.forward_tokens(tok(x1), tok(x2), tok(seg), tok(out1), tok(out2))
        return r1.view(B, H, W, C), r2.view(B, H, W, C)

    def forward(self, x1, x2, segfeature):
        require_device(x1, "FeatureFusionModule input")
        r1, r2 = self.forward_nhwc(ops.to_nhwc(x1), ops.to_nhwc(x2), ops.to_nhwc(segfeature))
        return ops.as_nchw(r1), ops.as_nchw(r2)


class Fusion_Network3_ac(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1_ir = nn.Conv2d(1, 64, 3, padding=1)
        self.conv1_vis = nn.Conv2d(1, 64, 3, padding=1)
        self.DRDB1 = DRDB(in_ch=64)
        self.DRDB2 = DRDB(in_ch=64)
        self.DRDB3 = DRDB(in_ch=64)
        self.DRDB4 = DRDB(in_ch=64)
        self.conv2 = nn.Conv2d(128, 64, 3, padding=1)
        self.relu = nn.PReLU()
        self.ffm = FeatureFusionModule(64)
        self.ffm2 = FeatureFusionModule(64)  # present in checkpoints, never used by forward
        self.conv3 = nn.Conv2d(64, 64, 1, padding=0)
        self.conv4 = nn.Conv2d(128, 64, 1, padding=0)
        self.conv21 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv22 = nn.Conv2d(32, 1, 3, padding=1)
        self._pk = PackedCache()

    def _w(self, name):
        return self._pk.get(name, getattr(self, name).weight, ops.pack_weight)

    def _w3(self, name):  # stride-1 "same" 3x3 convs: split-bf16 image when ops.conv3x3_mode() allows
        return self._pk.get(f"{name}:{ops.conv3x3_mode()}", getattr(self, name).weight, ops.pack_conv3x3)

    def _conv22(self, f, slope):
        """conv22 (32 -> 1 channel) + the shared PReLU: a bandwidth-bound stencil kernel instead of a 32-wide matrix tile."""
        if ops.aligned16(f) and f.shape[-1] == 32:
            return ops.conv3x3_c32to1(f, self._w("conv22"), bias=self.conv22.bias, act=ops.ACT_PRELU, prelu=slope)
        return ops.conv2d(f, self._w("conv22"), 1, 3, pad=1, bias=self.conv22.bias, act=ops.ACT_PRELU, prelu=slope)

    @staticmethod
    def _first_channel_nhwc(x):
        """x[:, 0:1] of an NCHW image as an NHWC (B,H,W,1) tensor (identical memory for C == 1)."""
        B, _, H, W = x.shape
        return x[:, 0:1].contiguous().view(B, H, W, 1)

    def forward_train(self, ir, vis, out1, out2):
        """autograd path (train.py:360): every conv / DRDB / interaction block is an autograd node whose
        forward and backward are HIP kernels."""
        B, _, H, W = ir.shape
        slope = self.relu.weight

        def nhwc1(x):  # x[:, 0:1] as NHWC, autograd-aware
            return x[:, 0:1].permute(0, 2, 3, 1).contiguous()

        def conv_prelu(x, conv, out=None):  # (ag.conv2d keeps the shared PReLU a node of its own: exact backward for any slope)
            return ag.conv2d(x, conv.weight, conv.bias, k=3, pad=1, act=ops.ACT_PRELU, slope=slope, out=out)

        def home(drdb):  # a DRDB's concat buffer; its producer writes the first 64 channels in place (no copy in)
            buf = drdb.new_buffer(B, H, W, ir.device)
            return ag.Out(buf), ag.Out(buf[..., :drdb.in_ch])

        def tokens_out(o):  # the same placement as a (B, H * W, C) token view (what CrossPath's LayerNorm writes)
            return ag.Out(o.t.view(B, H * W, o.t.shape[-1])) if o.t.is_contiguous() else \
                ag.Out(o.t.as_strided((B, H * W, o.t.shape[-1]), (H * W * o.t.stride(2), o.t.stride(2), 1), o.t.storage_offset()))

        h1, o1 = home(self.DRDB1)
        h2, o2 = home(self.DRDB2)
        x1 = self.DRDB1.forward_train_nhwc(conv_prelu(nhwc1(ir), self.conv1_ir, o1), h1)
        x2 = self.DRDB2.forward_train_nhwc(conv_prelu(nhwc1(vis), self.conv1_vis, o2), h2)
        seg = ag.linear(out1.permute(0, 2, 3, 1).contiguous(), self.conv3.weight, self.conv3.bias)
        h1, o1 = home(self.DRDB3)
        h2, o2 = home(self.DRDB4)
        x1, x2 = self.ffm.forward_nhwc(x1, x2, seg, out1=tokens_out(o1), out2=tokens_out(o2))
        x1 = self.DRDB3.forward_train_nhwc(x1, h1)
        x2 = self.DRDB4.forward_train_nhwc(x2, h2)
        seg = ag.linear(out2.permute(0, 2, 3, 1).contiguous(), self.conv4.weight, self.conv4.bias)
        # the second interaction writes the two halves of conv2's input in place: no torch.cat, no slice-gradient copies
        cat = torch.empty((B, H, W, 128), device=ir.device, dtype=torch.float32)
        x1, x2 = self.ffm.forward_nhwc(x1, x2, seg, out1=tokens_out(ag.Out(cat[..., :64])), out2=tokens_out(ag.Out(cat[..., 64:])))
        f = conv_prelu(ag.join(ag.Out(cat), x1, x2), self.conv2)
        f = conv_prelu(f, self.conv21)
        f = conv_prelu(f, self.conv22)
        return f.permute(0, 3, 1, 2)

    def forward(self, ir, vis, out1, out2):
        require_device(ir, "Fusion_Network3_ac input")
        if out1.shape[1] == 64 and out2.shape[1] == 128 and wants_grad(self, ir, vis, out1, out2):
            return self.forward_train(ir, vis, out1, out2)
        if out1.shape[1] != 64 or out2.shape[1] != 128:
            # same failure the reference hits inside conv3/conv4 (mit_b0 features do not fit)
            raise RuntimeError(f"Fusion_Network3_ac expects 64/128-channel segmentation features, got "
                               f"{out1.shape[1]}/{out2.shape[1]} channels")
        return self._forward_eval(
            ir, vis,
            lambda: ops.linear(ops.to_nhwc(out1), self._w("conv3"), 64, bias=self.conv3.bias),
            lambda: ops.linear(ops.to_nhwc(out2), self._w("conv4"), 64, bias=self.conv4.bias))

    def forward_from_features(self, ir, vis, f1, f2):
        """Same result as forward(ir, vis, up(f1), up(f2)) for the NHWC feature maps of
        MixVisionTransformer.forward_fusion_features(): conv3 / conv4 are 1x1 convs with bias, and a
        bilinear resize is a per-channel convex combination (weights sum to 1), so the two commute
        exactly in real arithmetic (to rounding in fp32; tests/test_gpu_modules.py).  Running the convs
        at H/4 x W/4 and H/8 x W/8 removes the 64- and 128-channel full-resolution tensors (236 MB per
        image) and 16x / 64x of the two GEMMs.  Inference only."""
        require_device(ir, "Fusion_Network3_ac input")
        if f1.shape[-1] != 64 or f2.shape[-1] != 128:
            raise RuntimeError(f"Fusion_Network3_ac expects 64/128-channel segmentation features, got "
                               f"{f1.shape[-1]}/{f2.shape[-1]} channels")
        H, W = ir.shape[2], ir.shape[3]
        # (r5, SURVEY §8(f) N4 second half) the resized features are not formed either: CrossPath's kernels read the low-resolution
        # maps through ops.LazySeg (SEGMIF_LAZY_SEG=0: resize first, as before)
        up = (lambda low: ops.LazySeg(low, H, W)) if ops.lazy_seg_mode() else (lambda low: ops.bilinear(low, H, W))
        return self._forward_eval(
            ir, vis,
            lambda: up(ops.linear(f1, self._w("conv3"), 64, bias=self.conv3.bias)),
            lambda: up(ops.linear(f2, self._w("conv4"), 64, bias=self.conv4.bias)))

    def _forward_eval(self, ir, vis, seg1_fn, seg2_fn):
        if all(d.planes_ok() for d in (self.DRDB1, self.DRDB2, self.DRDB3, self.DRDB4)):
            return self._forward_eval_planes(ir, vis, seg1_fn, seg2_fn)
        return self._eval_buffers_body(ir, vis, seg1_fn, seg2_fn)

    def _eval_dispatch(self, ir, vis, seg1_fn, seg2_fn):
        """The body a guarded scope runs and, on a trip, runs AGAIN: re-reads the conv mode each time, so that the repeat
        finish_guarded makes under set_conv3x3_mode('fp32') (an ill-conditioned CrossPath softmax) really is the exact-fp32
        buffer path - (r6, ADVICE r5: the planes body ignores the mode, and a standalone call's 'exact' repeat had silently
        been a bf16x6 planes repeat)."""
        if all(d.planes_ok() for d in (self.DRDB1, self.DRDB2, self.DRDB3, self.DRDB4)):
            return self._eval_planes_body(ir, vis, seg1_fn, seg2_fn)
        return self._eval_buffers_body(ir, vis, seg1_fn, seg2_fn)

    def _eval_buffers_body(self, ir, vis, seg1_fn, seg2_fn):
        B, _, H, W = ir.shape
        dev, slope = ir.device, self.relu.weight
        PRELU = ops.ACT_PRELU
        bufs = []
        for x, conv, drdb in ((ir, self.conv1_ir, self.DRDB1), (vis, self.conv1_vis, self.DRDB2)):
            buf = drdb.new_buffer(B, H, W, dev)
            name = "conv1_ir" if conv is self.conv1_ir else "conv1_vis"
            ops.conv2d(self._first_channel_nhwc(x), self._w(name), 64, 3, pad=1, bias=conv.bias, act=PRELU,
                       prelu=slope, out=buf[..., :64])
            bufs.append(buf)
        x1 = self.DRDB1.forward_buffer(bufs[0])
        x2 = self.DRDB2.forward_buffer(bufs[1])
        seg = seg1_fn()
        # first interaction writes straight into the DRDB3 / DRDB4 concat buffers (reused storage)
        x1, x2 = self.ffm.forward_nhwc(x1, x2, seg, out1=bufs[0][..., :64], out2=bufs[1][..., :64])
        x1 = self.DRDB3.forward_buffer(bufs[0])
        x2 = self.DRDB4.forward_buffer(bufs[1])
        del bufs
        seg = seg2_fn()
        cat = torch.empty((B, H, W, 128), device=dev, dtype=torch.float32)
        self.ffm.forward_nhwc(x1, x2, seg, out1=cat[..., :64], out2=cat[..., 64:])
        f = ops.conv2d(cat, self._w3("conv2"), 64, 3, pad=1, bias=self.conv2.bias, act=PRELU, prelu=slope)
        f = ops.conv2d(f, self._w3("conv21"), 32, 3, pad=1, bias=self.conv21.bias, act=PRELU, prelu=slope)
        f = self._conv22(f, slope)
        return f.view(B, 1, H, W)


    def _forward_eval_planes(self, ir, vis, seg1_fn, seg2_fn):
        """_forward_eval with the four DRDBs and the closing convs on pre-split activations.  In 'planes16' mode the body
        runs in a guarded scope (its own, or the caller's - segmif_amd.pipeline.PairForward opens one around the whole pair
        forward): half pairs while a guard is active, and the scope repeats the forward on bf16 triples if a planes tensor
        left the half's exponent range."""
        before = ops.range_fallbacks()
        # (images = the batch: the scope's guard then has per-image rows and the conditioning half of the guard sees this call's
        # CrossPath softmaxes; no per-image redo here - a tripped standalone call repeats as a whole)
        out = ops.run_guarded(lambda: self._eval_dispatch(ir, vis, seg1_fn, seg2_fn), ir.device,
                              enabled=ops.conv3x3_mode() == "planes16", images=ir.shape[0])
        self.planes16_fallbacks += ops.range_fallbacks() - before
        return out

    def _eval_planes_body(self, ir, vis, seg1_fn, seg2_fn):
        """Two planes scratch buffers (one per modality, reused by DRDB1 -> DRDB3 and DRDB2 -> DRDB4), 64-channel fp32
        tensors between the blocks; a third buffer for the concatenated tensor and conv2's output."""
        B, _, H, W = ir.shape
        dev, slope = ir.device, self.relu.weight
        PRELU = ops.ACT_PRELU
        guard = ops.active_guard() if ops.conv3x3_mode() == "planes16" else None  # half pairs iff a guarded scope is running
        xs, pls = [], []
        # (r4) on f16x3 planes a DRDB takes its residual from its own input chunks (hi + 2^-11 lo), so neither conv1 nor the
        # first interaction's CrossPath tails write an fp32 copy of the DRDB inputs: 20 GB of stores and as many of reads per
        # 64-pair step.  SEGMIF_DRDB_RES=fp32 keeps round 3's fp32 residual tensors (A/B switch; bf16 planes always do).
        lean = guard is not None and _DRDB_RES_PLANES
        for x, conv, name in ((ir, self.conv1_ir, "conv1_ir"), (vis, self.conv1_vis, "conv1_vis")):
            pls.append(ops.Planes(B, H, W, DRDB.PLANES_CHUNKS, dev, guard))
            # conv1 writes its 64 channels split, as the DRDB's first four chunks (and as fp32 - the DRDB's residual input - unless lean)
            if pls[-1].f16 and _CONV1_STENCIL and conv.weight.is_contiguous():
                # (r6) Cin = 1: a store-bound stencil, not a K = 9 scalar-gather GEMM (2.4 -> ~1 ms per 64-image launch)
                xs.append(ops.conv3x3_c1(self._first_channel_nhwc(x), conv.weight, bias=conv.bias, act=PRELU, prelu=slope,
                                         planes=pls[-1], planes_only=lean))
                continue
            xs.append(ops.conv2d(self._first_channel_nhwc(x), self._w(name), 64, 3, pad=1, bias=conv.bias, act=PRELU,
                                 prelu=slope, planes=pls[-1], planes_only=lean))
        y1 = self.DRDB1.forward_planes(xs[0], pls[0], preloaded=True)
        y2 = self.DRDB2.forward_planes(xs[1], pls[1], preloaded=True)
        seg = seg1_fn()
        pre = self.ffm.cross.gram_ok()  # the CrossPath tail then writes the DRDB inputs pre-split as well
        if pre and lean:
            self.ffm.forward_nhwc(y1, y2, seg, planes1=pls[0], planes2=pls[1], planes_only=True)
            x1 = x2 = None
        else:
            if lean:  # (the GEMM-form CrossPath has no planes epilogue: it needs fp32 outputs)
                xs = [torch.empty((B, H, W, 64), device=dev, dtype=torch.float32) for _ in range(2)]
            x1, x2 = self.ffm.forward_nhwc(y1, y2, seg, out1=xs[0], out2=xs[1], planes1=pls[0] if pre else None,
                                           planes2=pls[1] if pre else None)
        y1 = self.DRDB3.forward_planes(x1, pls[0], out=y1, preloaded=pre)
        y2 = self.DRDB4.forward_planes(x2, pls[1], out=y2, preloaded=pre)
        del pls, xs, x1, x2
        seg = seg2_fn()
        if pre and ops.aligned16(self.conv2.bias, self.conv21.bias):
            # conv2 (128 -> 64) and conv21 (64 -> 32) on the planes kernel as well: the second interaction's tails write the
            # concatenated tensor pre-split (chunks 0-7), conv2 = two 32-channel launches appending chunks 8-11, conv21
            # reads those and hands fp32 rows to the conv22 stencil
            pc = ops.Planes(B, H, W, 12, dev, guard)
            # (r4: no fp32 copy of the concatenated tensor - nothing reads it: 10 GB of stores per 64-pair step)
            self.ffm.forward_nhwc(y1, y2, seg, planes1=pc, planes2=pc.at(4), planes_only=True)
            del y1, y2
            sfx, pack = ("h", ops.pack_weight_planes16) if pc.f16 else ("", ops.pack_weight_planes)
            for half in (0, 1):
                rows = slice(32 * half, 32 * half + 32)
                wt = self._pk.get(f"conv2:p{half}{sfx}", self.conv2.weight, lambda w, rows=rows: pack(w[rows]))
                ops.conv3x3_planes(pc, 128, wt, dil=1, bias=self.conv2.bias[rows], act=PRELU, prelu=slope, out_chunk0=8 + 2 * half)
            f = torch.empty((B, H, W, 32), device=dev, dtype=torch.float32)
            ops.conv3x3_planes(pc.at(8), 64, self._pk.get(f"conv21:p{sfx}", self.conv21.weight, pack), dil=1,
                               bias=self.conv21.bias, act=PRELU, prelu=slope, out=f)
            del pc
        else:
            cat = torch.empty((B, H, W, 128), device=dev, dtype=torch.float32)
            self.ffm.forward_nhwc(y1, y2, seg, out1=cat[..., :64], out2=cat[..., 64:])
            del y1, y2
            f = ops.conv2d(cat, self._w3("conv2"), 64, 3, pad=1, bias=self.conv2.bias, act=PRELU, prelu=slope)
            f = ops.conv2d(f, self._w3("conv21"), 32, 3, pad=1, bias=self.conv21.bias, act=PRELU, prelu=slope)
        f = self._conv22(f, slope)
        return f.view(B, 1, H, W)

    planes16_fallbacks = 0  # forwards repeated on the bf16x6 kernels because the f16x3 range guard tripped


def seg_criterion_loss(seg, label, criterion):
    """criterion(bilinear-up(seg -> label size), label) for NHWC logits `seg` (ref :1095-1096, :241-244).  With the criterion
    train.py builds - nn.CrossEntropyLoss(ignore_index=...), mean reduction, no class weights - the whole chain (x4 bilinear,
    softmax-CE with ignore_index, their backward) runs in two HIP kernels.  Any other criterion is the CALLER's code: it receives
    the up-sampled logits as an NCHW view of the HIP bilinear kernel's output (with its HIP backward) - (r6) no F.interpolate, no
    aten op of this package's own on the way."""
    H, W = label.shape[1:]
    up = ag.bilinear(seg, H, W) if seg.requires_grad else ops.bilinear(seg, H, W)
    # the HIP CE kernel holds a row's logits in 32 registers; labels outside [0, C) other than ignore_index are
    # treated as ignored there (torch raises), so the data pipeline must already guarantee the range (train.py's does)
    if isinstance(criterion, nn.CrossEntropyLoss) and criterion.weight is None and criterion.reduction == "mean" \
            and getattr(criterion, "label_smoothing", 0.0) == 0.0 and seg.shape[-1] <= 32:
        return ag.softmax_ce(up, label.type(torch.long), criterion.ignore_index)
    return criterion(ops.as_nchw(up), label.type(torch.long))


class Network3(nn.Module):
    def __init__(self, backbone, num_classes=20, embedding_dim=256, pretrained=True):
        super().__init__()
        self.fusion_nums = 2
        self.seg_nums = 2
        self.fusion_channel = 48
        self.seg_channel = 64
        self.denoise_net = WeTr(backbone, num_classes, embedding_dim, pretrained)
        self.mean = [123.675, 116.28, 103.53]
        self.std = [58.395, 57.12, 57.375]

    def _segment_nhwc(self, fused):
        require_device(fused, "Network3 input")
        if torch.is_grad_enabled() and fused.requires_grad:
            # gradient flows back to the fused image (train.py:368): keep the normalisation in autograd
            mean = fused.new_tensor(self.mean).view(1, 3, 1, 1)
            std = fused.new_tensor(self.std).view(1, 3, 1, 1)
            return self.denoise_net.forward_nhwc((fused * 255 - mean) / std)
        # (x*255 - mean)/std fused with the NCHW -> NHWC transpose (ref :1083-1085)
        def body(x):
            return self.denoise_net.forward_nhwc(ops.as_nchw(ops.seg_normalize(x)))

        if torch.is_grad_enabled():
            return body(fused)  # (training path: its kernels make their own ranges; no guarded scope)
        # (r5) plain inference, also when the net is called on its own (test_segmentation.py:169): a guarded scope of its own -
        # f16x3 GEMMs / attention / Mix-FFN on half pairs, images that leave the half's range repeated on bf16x6 - unless a
        # caller (pipeline.PairForward) already opened one, which this call then joins
        def redo(out, idx):
            out.index_copy_(0, idx, body(fused.index_select(0, idx)))
            return out

        return ops.run_guarded(lambda: body(fused), fused.device, images=fused.shape[0], redo=redo)

    def forward(self, fused_seg1):
        return fused_seg1, fused_seg1, ops.as_nchw(self._segment_nhwc(fused_seg1))

    def predict_labels(self, fused, size=None):
        """test_segmentation.py:169-174 on device: logits -> bilinear to `size` -> argmax (int32, B,H,W)."""
        seg = self._segment_nhwc(fused)
        H, W = size if size is not None else fused.shape[2:]
        return ops.bilinear_argmax(seg, H, W)  # (r6: one pass - the resized logits, 708 MB per 64 images, are never written)

    def _loss(self, fused_seg1, label, criterion):
        """CE(bilinear-up(seg_map), label) (ref :1090-1097): seg_criterion_loss below."""
        return seg_criterion_loss(self._segment_nhwc(fused_seg1), label, criterion)

    def denoise_net_parameters(self):
        return self.denoise_net.parameters()


class Mean(nn.Module):
    """Y-channel substitution baseline (ref :184-214): put `mask` in the Y channel of `vis`,
    convert back to RGB, clamp to [0,1] and min-max normalise over the whole batch."""

    def __init__(self):
        super().__init__()
        self.fusion_nums = 2
        self.seg_nums = 2
        self.fusion_channel = 48
        self.seg_channel = 64
        self.mean = [123.675, 116.28, 103.53]
        self.std = [58.395, 57.12, 57.375]

    def forward(self, mask, vis):
        require_device(vis, "Mean input")
        rgb = ops.fuse_ycrcb(vis, mask[:, 0:1])
        lo, hi = rgb.min(), rgb.max()
        return (rgb - lo) / (hi - lo)


# (r6) the reference's ablation / variant classes (core/model_fusion.py:158-1025) live in .variants and are part of this module's
# namespace like upstream: `from core.model_fusion import Fusion_Network3` (val_performance.py:565) resolves here.
from .variants import *  # noqa: E402,F401,F403
from . import variants as _variants  # noqa: E402

__all__ += _variants.__all__
