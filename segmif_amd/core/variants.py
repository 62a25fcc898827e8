"""The ablation / variant classes of the reference's core/model_fusion.py on the HIP kernels.

`val_performance.py:565` instantiates `Fusion_Network3`; the other classes are the paper's ablations (interaction replaced by a
concatenation, a sum, gated averages, one of the two attentions only, no segmentation feature at all).  None of them is on the
measured path, so they are compositions of the package's generic primitives - NHWC end to end, the implicit-GEMM convs
(`ops.conv2d`), the buffer form of the DRDB (any `in_ch`), `ops.linear`, `ops.layernorm`, the linear-attention partial sums and fold at
any head geometry with dim <= 64 (csrc/linattn.hip, r6: these modules run at dim 32 = 8 heads of 4) and one pointwise kernel
(`ops.pointwise2`) - with the reference's constructor / forward signatures and state_dict keys (tests/golden/variants_keys.json).
Inference only: called with gradients wanted they raise (the training path exists for the classes train.py builds).

Reference lines: Fusion_Network :158-183 (its forward cannot run upstream either: conv1 makes 64 channels, its DRDBs take 32 - the
same RuntimeError is raised here), Network_fused :218-246, CrossPath_M :363-395, CrossPath_S :397-429, FeatureFusionModule_SoAM /
_MoAM :467-536, CrossPath_showAttention :538-572, FeatureFusionModule_ShowAttention :573-624, Fusion_Network3 :626-660, _Con :662-711,
_Add :714-757, AttentionModule :759-770, _Average :772-819, _S :821-854, _M :856-889, _obtainattention :891-932,
Fusion_Network_rmseg :934-979, Fusion_Network_rmseg_att :981-1025."""
import torch
import torch.nn as nn

from .. import autograd as ag
from .. import ops
from ._util import PackedCache, init_reference_style, require_device, wants_grad
from .model_fusion import DRDB, CrossAttention, CrossAttention2, FeatureFusionModule, WeTr

__all__ = ["Fusion_Network", "Network_fused", "CrossPath_M", "CrossPath_S", "FeatureFusionModule_SoAM", "FeatureFusionModule_MoAM",
           "CrossPath_showAttention", "FeatureFusionModule_ShowAttention", "Fusion_Network3", "Fusion_Network3_Con",
           "Fusion_Network3_Add", "AttentionModule", "Fusion_Network3_Average", "Fusion_Network3_S", "Fusion_Network3_M",
           "Fusion_Network3_obtainattention", "Fusion_Network_rmseg", "Fusion_Network_rmseg_att"]

PRELU, RELU, NONE = ops.ACT_PRELU, ops.ACT_RELU, ops.ACT_NONE


def _inference_only(module, *tensors):
    if wants_grad(module, *tensors):
        raise NotImplementedError(
            f"{type(module).__name__}: the ablation variants run inference only on the HIP path - call under torch.no_grad() "
            "(the training path covers the classes train.py builds: Fusion_Network3_ac, Network3)")


def _first_channel_nhwc(x):
    B, _, H, W = x.shape
    return x[:, 0:1].contiguous().view(B, H, W, 1)


class _Convs(nn.Module):
    """Shared conv plumbing: packed weights cached per parameter version, NHWC rows views in and out."""

    def _pk_(self):
        pk = self.__dict__.get("_pk")
        if pk is None:
            pk = self.__dict__["_pk"] = PackedCache()
        return pk

    def _conv(self, name, x, act=NONE, out=None, mod=None):
        conv = mod if mod is not None else getattr(self, name)
        k = conv.kernel_size[0]
        wt = self._pk_().get(name, conv.weight, ops.pack_weight)
        prelu = self.relu.weight if act == PRELU else None
        if k == 1:
            return ops.linear(x, wt, conv.out_channels, bias=conv.bias, act=act, prelu=prelu, out=out)
        return ops.conv2d(x, wt, conv.out_channels, k, pad=conv.padding[0], dil=conv.dilation[0], bias=conv.bias, act=act,
                          prelu=prelu, out=out)


def _drdb_run(drdb, buf, out=None):
    """buf: the block's concat buffer whose first in_ch channels its producer has already written (no copy in)."""
    return drdb.forward_buffer(buf, out=out)


# ----------------------------------------------------------------------------------------------------------------------------------
# interaction modules at any dim <= 64 (heads * d), tokens (B, N, C)
# ----------------------------------------------------------------------------------------------------------------------------------
class _CrossPathGeneric(nn.Module):
    """channel_proj x 3 -> ReLU -> chunk (y | u) -> linear cross attention(s) -> end_proj -> residual -> LayerNorm, with
    the attention(s) named by `USE` ('v': CrossAttention on u, 'z': CrossAttention2 on y; ref :351-361, :385-395, :419-429)."""
    USE = "zv"

    def __init__(self, dim, reduction=1, num_heads=8, norm_layer=nn.LayerNorm):
        super().__init__()
        if reduction != 1 or dim % num_heads or dim > 64 or dim // num_heads > 8 or dim % 16:
            raise NotImplementedError("the generic linear-attention kernels take dim <= 64 (a multiple of 16), head size <= 8, reduction 1")
        self.dim, self.num_heads = dim, num_heads
        self.channel_proj1 = nn.Linear(dim, dim * 2)
        self.channel_proj2 = nn.Linear(dim, dim * 2)
        self.channel_proj3 = nn.Linear(dim, dim * 2)
        self.act1 = nn.ReLU(inplace=True)
        self.act2 = nn.ReLU(inplace=True)
        self.act3 = nn.ReLU(inplace=True)
        if "v" in self.USE:
            self.cross_attn = CrossAttention(dim, num_heads=num_heads)
        if "z" in self.USE:
            self.cross_attn2 = CrossAttention2(dim, num_heads=num_heads)
        self.end_proj1 = nn.Linear(dim * len(self.USE), dim)
        self.end_proj2 = nn.Linear(dim * len(self.USE), dim)
        self.norm1 = norm_layer(dim)
        self.norm2 = norm_layer(dim)
        self._pk = PackedCache()

    def _kv_partial(self, lin, name, x):
        kv = ops.linear(x, self._pk.get(name, lin.weight, ops.pack_weight), 2 * self.dim, bias=lin.bias)
        return ops.linattn_partial(kv, self.num_heads)

    def _apply_ctx(self, q, part, scale):
        """q @ softmax-context, materialised (the attention maps the *_showAttention classes hand back)."""
        C = self.dim
        eye = self._pk.get_multi(f"eye:{q.device}", (), lambda: torch.eye(C, device=q.device, dtype=torch.float32))
        w = torch.empty((q.shape[0], C, C), device=q.device, dtype=torch.float32)
        ops.linattn_fold(part, eye, w, wofs=0, kofs=0, scale=scale, heads=self.num_heads)
        return ops.linear(q, w, C, batched_weight=True)

    def forward_tokens(self, x1, x2, seg, want_maps=False, outs=(None, None)):
        """outs: optional (B, N, C) rows views (channel slices of wider buffers) receiving the two results."""
        C, pk = self.dim, self._pk
        p = []
        for i, x in ((1, x1), (2, x2), (3, seg)):
            lin = getattr(self, f"channel_proj{i}")
            p.append(ops.linear(x, pk.get(f"cp{i}", lin.weight, ops.pack_weight), 2 * C, bias=lin.bias, act=RELU))
        p1, p2, p3 = p  # [y_i | u_i]
        B = x1.shape[0]
        part3 = self._kv_partial(self.cross_attn.kv3, "kv3", p3[..., C:]) if "v" in self.USE else None
        part = [self._kv_partial(getattr(self.cross_attn2, f"kv{i}"), f"kv{i}", pp[..., :C]) if "z" in self.USE else None
                for i, pp in ((1, p1), (2, p2))]
        res = []
        for i, (x, pp) in enumerate(((x1, p1), (x2, p2))):
            end, norm = getattr(self, f"end_proj{i + 1}"), getattr(self, f"norm{i + 1}")
            weff = torch.empty((B, C, C * len(self.USE)), device=x.device, dtype=torch.float32)
            # cat(z_i, v_i) @ Wend^T == [y3 | u_i] @ Weff^T with the contexts folded in; one attention only: the matching half
            k = 0
            if "z" in self.USE:
                ops.linattn_fold(part[i], end.weight, weff, wofs=k, kofs=k, scale=self.cross_attn2.scale, heads=self.num_heads)
                k += C
            if "v" in self.USE:
                ops.linattn_fold(part3, end.weight, weff, wofs=k, kofs=k, scale=self.cross_attn.scale, heads=self.num_heads)
            if self.USE == "zv":
                t = ops.linear(p3[..., :C], weff, C, bias=end.bias, res=x, x2=pp[..., C:], batched_weight=True)
            elif self.USE == "z":
                t = ops.linear(p3[..., :C], weff, C, bias=end.bias, res=x, batched_weight=True)
            else:
                t = ops.linear(pp[..., C:], weff, C, bias=end.bias, res=x, batched_weight=True)
            res.append(ops.layernorm(t, norm.weight, norm.bias, norm.eps, out=t if outs[i] is None else outs[i]))
        if not want_maps:
            return res[0], res[1]
        v1 = self._apply_ctx(p1[..., C:], part3, self.cross_attn.scale)
        v2 = self._apply_ctx(p2[..., C:], part3, self.cross_attn.scale)
        z1 = self._apply_ctx(p3[..., :C], part[0], self.cross_attn2.scale)
        z2 = self._apply_ctx(p3[..., :C], part[1], self.cross_attn2.scale)
        return res[0], res[1], [v1, z1, z2, v2]

    def forward(self, x1, x2, segfeature):
        require_device(x1, f"{type(self).__name__} input")
        _inference_only(self, x1, x2, segfeature)
        return self.forward_tokens(x1.contiguous(), x2.contiguous(), segfeature.contiguous())


class CrossPath_M(_CrossPathGeneric):
    """CrossAttention only: out_i = LN(x_i + end_proj_i(u_i @ ctx3))  (ref :363-395)."""
    USE = "v"


class CrossPath_S(_CrossPathGeneric):
    """CrossAttention2 only: out_i = LN(x_i + end_proj_i(y3 @ ctx_i))  (ref :397-429)."""
    USE = "z"


class CrossPath_showAttention(_CrossPathGeneric):
    """CrossPath that also returns its four attention results [v1, z1, z2, v2]  (ref :538-572)."""
    USE = "zv"

    def forward(self, x1, x2, segfeature):
        require_device(x1, "CrossPath_showAttention input")
        _inference_only(self, x1, x2, segfeature)
        return self.forward_tokens(x1.contiguous(), x2.contiguous(), segfeature.contiguous(), want_maps=True)


class _CrossPathAny(_CrossPathGeneric):
    """The reference's CrossPath at a dim the tuned kernels are not built for (dim 32 inside Fusion_Network3, ref :639)."""
    USE = "zv"


class _FfmGeneric(nn.Module):
    PATH = _CrossPathAny

    def __init__(self, dim, reduction=1, num_heads=8, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.cross = self.PATH(dim=dim, reduction=reduction, num_heads=num_heads)
        init_reference_style(self)

    def forward_nhwc(self, x1, x2, seg, out1=None, out2=None):
        """NHWC in / out; out_i may be channel slices of wider buffers (a DRDB's concat buffer, the halves of conv2's input)."""
        B, H, W, C = x1.shape
        tok = lambda t: None if t is None else t.view(B, H * W, t.shape[-1]) if t.is_contiguous() else \
            t.as_strided((B, H * W, t.shape[-1]), (H * W * t.stride(2), t.stride(2), 1), t.storage_offset())
        r = self.cross.forward_tokens(tok(x1), tok(x2), tok(seg), outs=(tok(out1), tok(out2)))
        return (out1 if out1 is not None else r[0].view(B, H, W, C)), (out2 if out2 is not None else r[1].view(B, H, W, C))

    def forward(self, x1, x2, segfeature):
        require_device(x1, f"{type(self).__name__} input")
        _inference_only(self, x1, x2, segfeature)
        r1, r2 = self.forward_nhwc(ops.to_nhwc(x1), ops.to_nhwc(x2), ops.to_nhwc(segfeature))
        return ops.as_nchw(r1), ops.as_nchw(r2)


class FeatureFusionModule_SoAM(_FfmGeneric):
    PATH = CrossPath_S


class FeatureFusionModule_MoAM(_FfmGeneric):
    PATH = CrossPath_M


class FeatureFusionModule_ShowAttention(_FfmGeneric):
    """Returns (x1, x2, [x1_in, x2_in]) - the reference builds the attention maps and hands back copies of its INPUTS (ref :612-624)."""
    PATH = CrossPath_showAttention

    def forward(self, x1, x2, segfeature):
        require_device(x1, "FeatureFusionModule_ShowAttention input")
        _inference_only(self, x1, x2, segfeature)
        a, b = ops.to_nhwc(x1), ops.to_nhwc(x2)
        r1, r2 = self.forward_nhwc(a, b, ops.to_nhwc(segfeature))
        return ops.as_nchw(r1), ops.as_nchw(r2), [ops.as_nchw(a).clone(), ops.as_nchw(b).clone()]


def _ffm(dim):
    """FeatureFusionModule at `dim`: the tuned Gram-form module at 64, the generic composition elsewhere (same state_dict keys)."""
    return FeatureFusionModule(dim) if dim == 64 else _FfmGeneric(dim)


# ----------------------------------------------------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------------------------------------------------
class _FusionBase(_Convs):
    """conv1_ir / conv1_vis -> DRDB1 / DRDB2 -> [interaction 1] -> DRDB3 / DRDB4 -> [interaction 2] -> closing convs, all sharing
    one scalar PReLU (the skeleton of every Fusion_Network3* class, ref :643-660)."""
    CH = 32

    def _build(self, ch):
        self.conv1_ir = nn.Conv2d(1, ch, 3, padding=1)
        self.conv1_vis = nn.Conv2d(1, ch, 3, padding=1)
        self.DRDB1 = DRDB(in_ch=ch)
        self.DRDB2 = DRDB(in_ch=ch)
        self.DRDB3 = DRDB(in_ch=ch)
        self.DRDB4 = DRDB(in_ch=ch)

    def _stem(self, ir, vis, o1=None, o2=None):
        """x1 = DRDB1(PReLU(conv1_ir(ir[:, :1]))), x2 likewise (ref :646-653); conv1 writes straight into the DRDB's concat buffer,
        the DRDBs into o1 / o2 when given."""
        B, _, H, W = ir.shape
        r = []
        for name, img, drdb, o in (("conv1_ir", ir, self.DRDB1, o1), ("conv1_vis", vis, self.DRDB2, o2)):
            buf = drdb.new_buffer(B, H, W, ir.device)
            self._conv(name, _first_channel_nhwc(img), PRELU, out=buf[..., :drdb.in_ch])
            r.append(_drdb_run(drdb, buf, o))
        return r

    def _bufs(self, ir):
        B, _, H, W = ir.shape
        b3, b4 = self.DRDB3.new_buffer(B, H, W, ir.device), self.DRDB4.new_buffer(B, H, W, ir.device)
        ch = self.DRDB3.in_ch
        return b3, b4, b3[..., :ch], b4[..., :ch]

    def _catbuf(self, ir, ch):
        B, _, H, W = ir.shape
        cat = torch.empty((B, H, W, 2 * ch), device=ir.device, dtype=torch.float32)
        return cat, cat[..., :ch], cat[..., ch:]

    @staticmethod
    def _cat(x1, x2):  # (the _Con ablation's concatenations: two strided copies, as the reference's torch.concat)
        B, H, W, C = x1.shape
        cat = torch.empty((B, H, W, C + x2.shape[-1]), device=x1.device, dtype=torch.float32)
        cat[..., :C].copy_(x1)
        cat[..., C:].copy_(x2)
        return cat

    def _check(self, ir, vis, out1=None, out2=None):
        require_device(ir, f"{type(self).__name__} input")
        _inference_only(self, ir, vis, out1, out2)
        if out1 is not None and (out1.shape[1] != 64 or out2.shape[1] != 128):
            raise RuntimeError(f"{type(self).__name__} expects 64/128-channel segmentation features, got "
                               f"{out1.shape[1]}/{out2.shape[1]} channels")


class Fusion_Network3(_FusionBase):
    """The 32-channel interaction network val_performance.py:565 builds (ref :626-660): as Fusion_Network3_ac with 32-channel
    blocks and without conv22."""
    FFM = staticmethod(_ffm)

    def __init__(self):
        super().__init__()
        self._build(32)
        self.conv2 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv21 = nn.Conv2d(32, 1, 3, padding=1)
        self.relu = nn.PReLU()
        self.ffm = self.FFM(32)
        if self.HAS_FFM2:
            self.ffm2 = _ffm(32)  # present in checkpoints, never used by forward (as in Fusion_Network3_ac)
        self.conv3 = nn.Conv2d(64, 32, 1, padding=0)
        self.conv4 = nn.Conv2d(128, 32, 1, padding=0)

    HAS_FFM2 = True

    def _tail(self, cat):
        return self._conv("conv21", self._conv("conv2", cat, PRELU), PRELU)

    def _body(self, ir, vis, out1, out2):
        """Everything up to the concatenated input of conv2; also returns the first interaction's inputs."""
        a, b = self._stem(ir, vis)
        b3, b4, o3, o4 = self._bufs(ir)
        self.ffm.forward_nhwc(a, b, self._conv("conv3", ops.to_nhwc(out1)), out1=o3, out2=o4)
        y1, y2 = _drdb_run(self.DRDB3, b3), _drdb_run(self.DRDB4, b4)
        cat, c1, c2 = self._catbuf(ir, 32)
        self.ffm.forward_nhwc(y1, y2, self._conv("conv4", ops.to_nhwc(out2)), out1=c1, out2=c2)
        return cat, a, b

    def forward(self, ir, vis, out1, out2):
        self._check(ir, vis, out1, out2)
        return ops.as_nchw(self._tail(self._body(ir, vis, out1, out2)[0]))


class Fusion_Network3_S(Fusion_Network3):
    """Fusion_Network3 with CrossAttention2 only (ref :821-854)."""
    FFM = FeatureFusionModule_SoAM
    HAS_FFM2 = False


class Fusion_Network3_M(Fusion_Network3):
    """Fusion_Network3 with CrossAttention only (ref :856-889)."""
    FFM = FeatureFusionModule_MoAM
    HAS_FFM2 = False


class Fusion_Network3_obtainattention(Fusion_Network3):
    """Fusion_Network3 that also returns [x1, x2 entering the first interaction, conv2's pre-activation]  (ref :891-932)."""
    FFM = FeatureFusionModule_ShowAttention

    def forward(self, ir, vis, out1, out2):
        self._check(ir, vis, out1, out2)
        cat, a, b = self._body(ir, vis, out1, out2)
        f2 = self._conv("conv2", cat)
        f = self._conv("conv21", ag.prelu(f2, self.relu.weight), PRELU)
        return ops.as_nchw(f), [ops.as_nchw(a), ops.as_nchw(b), ops.as_nchw(f2)]


class _FusionNoFfm(_FusionBase):
    def __init__(self):
        super().__init__()
        self._build(32)
        self.conv2 = nn.Conv2d(64, 32, 3, padding=1)
        self._extra()
        self.conv21 = nn.Conv2d(32, 1, 3, padding=1)
        self.relu = nn.PReLU()
        self.conv3 = nn.Conv2d(64, 32, 1, padding=0)
        self.conv4 = nn.Conv2d(128, 32, 1, padding=0)

    def forward(self, ir, vis, out1, out2):
        self._check(ir, vis, out1, out2)
        x1, x2 = self._stem(ir, vis)
        b3, b4, o3, o4 = self._bufs(ir)
        self._mix(x1, x2, self._conv("conv3", ops.to_nhwc(out1)), 0, o3, o4)
        y1, y2 = _drdb_run(self.DRDB3, b3), _drdb_run(self.DRDB4, b4)
        cat, c1, c2 = self._catbuf(ir, 32)
        self._mix(y1, y2, self._conv("conv4", ops.to_nhwc(out2)), 1, c1, c2)
        return ops.as_nchw(self._conv("conv21", self._conv("conv2", cat, PRELU), PRELU))


class Fusion_Network3_Con(_FusionNoFfm):
    """Interaction replaced by concatenation + 3x3 conv (ref :662-711)."""

    def _extra(self):
        self.conv211 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv221 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv411 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv421 = nn.Conv2d(64, 32, 3, padding=1)

    def _mix(self, x1, x2, s, stage, o1, o2):
        n1, n2 = (("conv211", "conv221"), ("conv411", "conv421"))[stage]
        self._conv(n1, self._cat(x1, s), out=o1)
        self._conv(n2, self._cat(x2, s), out=o2)


class Fusion_Network3_Add(_FusionNoFfm):
    """Interaction replaced by a sum + 3x3 conv (ref :714-757)."""

    def _extra(self):
        self.conv211 = nn.Conv2d(32, 32, 3, padding=1)
        self.conv221 = nn.Conv2d(32, 32, 3, padding=1)
        self.conv411 = nn.Conv2d(32, 32, 3, padding=1)
        self.conv421 = nn.Conv2d(32, 32, 3, padding=1)

    def _mix(self, x1, x2, s, stage, o1, o2):
        n1, n2 = (("conv211", "conv221"), ("conv411", "conv421"))[stage]
        self._conv(n1, ops.pointwise2(x1, s, 0), out=o1)
        self._conv(n2, ops.pointwise2(x2, s, 0), out=o2)


class AttentionModule(_Convs):
    """conv3x3 -> ReLU -> conv3x3 -> z * sigmoid(z)  (ref :759-770)."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(32, 32, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(32, 32, 3, padding=1))

    def pre_nhwc(self, x):
        """The two convs; the closing z * sigmoid(z) is the caller's (fused with the sum in Fusion_Network3_Average)."""
        h = self._conv("c0", x, RELU, mod=self.conv[0])
        return self._conv("c2", h, mod=self.conv[2])

    def forward(self, x1):
        require_device(x1, "AttentionModule input")
        _inference_only(self, x1)
        return ops.as_nchw(ops.pointwise2(self.pre_nhwc(ops.to_nhwc(x1)), None, 2))


class Fusion_Network3_Average(_FusionNoFfm):
    """Interaction replaced by gated sums att(x) + att(seg)  (ref :772-819)."""

    def _extra(self):
        for i in range(1, 9):
            setattr(self, f"att{i}", AttentionModule())

    def _mix(self, x1, x2, s, stage, o1, o2):
        a = (self.att1, self.att2, self.att3, self.att4) if stage == 0 else (self.att5, self.att6, self.att7, self.att8)
        ops.pointwise2(a[0].pre_nhwc(x1), a[1].pre_nhwc(s), 1, out=o1)
        ops.pointwise2(a[2].pre_nhwc(x2), a[3].pre_nhwc(s), 1, out=o2)


class Fusion_Network_rmseg(_FusionBase):
    """Fusion_Network3_ac without the segmentation features and interactions: forward(ir, vis)  (ref :934-979)."""

    def __init__(self):
        super().__init__()
        self._build(64)
        self.conv2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv21 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv22 = nn.Conv2d(32, 1, 3, padding=1)
        self.relu = nn.PReLU()

    def _run(self, ir, vis):
        self._check(ir, vis)
        b3, b4, o3, o4 = self._bufs(ir)
        self._stem(ir, vis, o3, o4)  # DRDB1 / DRDB2 write straight into DRDB3's / DRDB4's concat buffers
        cat, c1, c2 = self._catbuf(ir, 64)
        _drdb_run(self.DRDB3, b3, c1)
        _drdb_run(self.DRDB4, b4, c2)
        f = self._conv("conv2", cat, PRELU)
        f = self._conv("conv21", f, PRELU)
        return self._conv("conv22", f, PRELU), c1, c2

    def forward(self, ir, vis):
        return ops.as_nchw(self._run(ir, vis)[0])


class Fusion_Network_rmseg_att(Fusion_Network_rmseg):
    """...that also returns the two branch features [x1, x2]  (ref :981-1025)."""

    def forward(self, ir, vis):
        f, x1, x2 = self._run(ir, vis)
        return ops.as_nchw(f), [ops.as_nchw(x1), ops.as_nchw(x2)]


class Fusion_Network(_FusionBase):
    """ref :158-183.  Upstream this class cannot run: conv1 produces 64 channels and DRDB1 takes 32, so forward raises inside
    the first dilated conv (recorded in tests/golden/variants_keys.json: forward_raises).  The constructor and the state_dict are
    reproduced; forward raises the same RuntimeError instead of guessing an intent."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(2, 64, 3, padding=1)
        self.DRDB1 = DRDB(in_ch=32)
        self.DRDB2 = DRDB(in_ch=32)
        self.conv2 = nn.Conv2d(64, 1, 3, padding=1)
        self.relu = nn.PReLU()

    def forward(self, ir, vis):
        require_device(ir, "Fusion_Network input")
        B, _, H, W = ir.shape
        raise RuntimeError(f"Given groups=1, weight of size [32, 32, 3, 3], expected input[{B}, 64, {H}, {W}] to have 32 channels, "
                           "but got 64 channels instead")


class Network_fused(nn.Module):
    """WeTr + a stored segmentation criterion  (ref :218-246): forward(fused) -> logits (no input normalisation, unlike Network3);
    _loss(fused, labels) = criterion(bilinear-up(logits), labels)."""

    def __init__(self, segloss, backbone, num_classes=20, embedding_dim=256, pretrained=None):
        super().__init__()
        self.fusion_nums = 2
        self.seg_nums = 2
        self.fusion_channel = 48
        self.seg_channel = 64
        self.seg_loss = segloss
        self.denoise_net = WeTr(backbone, num_classes, embedding_dim, pretrained)
        self.mean = [123.675, 116.28, 103.53]
        self.std = [58.395, 57.12, 57.375]

    def forward(self, fused):
        return self.denoise_net(fused)

    def _loss(self, fused, labels):
        from .model_fusion import seg_criterion_loss
        seg = self.denoise_net.forward_nhwc(fused)
        return seg_criterion_loss(seg, labels, self.seg_loss)

    def denoise_net_parameters(self):
        return self.denoise_net.parameters()
