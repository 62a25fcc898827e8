"""CPU oracle: a functional, state_dict-driven restatement of SegMiF's hot path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file.  The product path (segmif_amd/) never does:
it runs hand-written HIP kernels only and raises when they are unavailable.

Every function restates what the upstream reference computes (file:line cited, paths
relative to /root/reference) as plain fp32 torch-CPU arithmetic over a flat
``state_dict`` — there are no nn.Modules here, so nothing can be confused with (or
copied from) the reference classes.  The restatement is pinned against golden vectors
produced by importing the real reference in the build container
(oracle/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Parity status: PINNED by generated fixtures (the reference ships no tests of its own,
SURVEY.md §4).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --- MiT variant table: core/mix_transformer.py:389-434 -----------------------------------
MIT_VARIANTS = {
    "mit_b0": dict(dims=[32, 64, 160, 256], depths=[2, 2, 2, 2]),
    "mit_b1": dict(dims=[64, 128, 320, 512], depths=[2, 2, 2, 2]),
    "mit_b2": dict(dims=[64, 128, 320, 512], depths=[3, 4, 6, 3]),
    "mit_b3": dict(dims=[64, 128, 320, 512], depths=[3, 4, 18, 3]),
    "mit_b4": dict(dims=[64, 128, 320, 512], depths=[3, 8, 27, 3]),
    "mit_b5": dict(dims=[64, 128, 320, 512], depths=[3, 6, 40, 3]),
}
MIT_HEADS = [1, 2, 5, 8]
MIT_SR = [8, 4, 2, 1]
MIT_PATCH = [(7, 4), (3, 2), (3, 2), (3, 2)]  # (kernel, stride); padding = kernel // 2
BLOCK_LN_EPS = 1e-6  # partial(nn.LayerNorm, eps=1e-6): mix_transformer.py:393
DEFAULT_LN_EPS = 1e-5  # OverlapPatchEmbed.norm :173, Attention.norm :75, CrossPath.norm1/2

SEG_MEAN = (123.675, 116.28, 103.53)  # model_fusion.py:1079
SEG_STD = (58.395, 57.12, 57.375)  # model_fusion.py:1080


def _ln(x, sd, pfx, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + ".weight"], sd[pfx + ".bias"], eps)


def _lin(x, sd, pfx):
    return F.linear(x, sd[pfx + ".weight"], sd.get(pfx + ".bias"))


# --- MiT encoder ----------------------------------------------------------------------------
def overlap_patch_embed(sd, pfx, x, k, s):
    """core/mix_transformer.py:192-198: conv(k, s, pad k//2) -> tokens -> LayerNorm(1e-5)."""
    y = F.conv2d(x, sd[pfx + ".proj.weight"], sd[pfx + ".proj.bias"], stride=s, padding=k // 2)
    H, W = y.shape[2], y.shape[3]
    t = y.flatten(2).transpose(1, 2)
    return _ln(t, sd, pfx + ".norm", DEFAULT_LN_EPS), H, W


def sr_attention(sd, pfx, x, H, W, heads, sr):
    """core/mix_transformer.py:94-115: spatial-reduction attention."""
    B, N, C = x.shape
    hd = C // heads
    q = _lin(x, sd, pfx + ".q").reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        img = x.transpose(1, 2).reshape(B, C, H, W)
        red = F.conv2d(img, sd[pfx + ".sr.weight"], sd[pfx + ".sr.bias"], stride=sr)
        red = _ln(red.flatten(2).transpose(1, 2), sd, pfx + ".norm", DEFAULT_LN_EPS)
    else:
        red = x
    kv = _lin(red, sd, pfx + ".kv").reshape(B, -1, 2, heads, hd)
    k = kv[:, :, 0].permute(0, 2, 1, 3)  # first C outputs are K, head-major (:102-105)
    v = kv[:, :, 1].permute(0, 2, 1, 3)
    att = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(o, sd, pfx + ".proj")


def mix_ffn(sd, pfx, x, H, W):
    """core/mix_transformer.py:46-53 and :381-387: fc1 -> depthwise 3x3 -> exact GELU -> fc2."""
    B, N, _ = x.shape
    h = _lin(x, sd, pfx + ".fc1")
    Ch = h.shape[-1]
    img = h.transpose(1, 2).reshape(B, Ch, H, W)
    img = F.conv2d(img, sd[pfx + ".dwconv.dwconv.weight"], sd[pfx + ".dwconv.dwconv.bias"],
                   padding=1, groups=Ch)
    h = F.gelu(img.flatten(2).transpose(1, 2))
    return _lin(h, sd, pfx + ".fc2")


def mit_block(sd, pfx, x, H, W, heads, sr):
    """core/mix_transformer.py:151-155 (eval mode: DropPath is identity)."""
    x = x + sr_attention(sd, pfx + ".attn", _ln(x, sd, pfx + ".norm1", BLOCK_LN_EPS), H, W, heads, sr)
    x = x + mix_ffn(sd, pfx + ".mlp", _ln(x, sd, pfx + ".norm2", BLOCK_LN_EPS), H, W)
    return x


def mit_forward_features(sd, pfx, x, variant, taps=None):
    """core/mix_transformer.py:312-348 -> list of 4 NCHW feature maps."""
    cfg = MIT_VARIANTS[variant]
    B = x.shape[0]
    outs = []
    for s in range(4):
        k, st = MIT_PATCH[s]
        t, H, W = overlap_patch_embed(sd, f"{pfx}patch_embed{s + 1}", x, k, st)
        if taps is not None:
            taps[f"stage{s + 1}.embed"] = t
        for i in range(cfg["depths"][s]):
            t = mit_block(sd, f"{pfx}block{s + 1}.{i}", t, H, W, MIT_HEADS[s], MIT_SR[s])
        t = _ln(t, sd, f"{pfx}norm{s + 1}", BLOCK_LN_EPS)
        x = t.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def mit_forward_fusion(sd, pfx, x, variant):
    """core/mix_transformer.py:358-375: stage-1/2 features bilinearly resized to the input size."""
    H, W = x.shape[2], x.shape[3]
    outs = mit_forward_features(sd, pfx, x, variant)
    up = lambda t: F.interpolate(t, size=[H, W], mode="bilinear", align_corners=False)
    return up(outs[0]), up(outs[1])


# --- SegFormer head -------------------------------------------------------------------------
def segformer_head(sd, pfx, feats, bn_training=False):
    """core/segformer_head.py:59-82.  Default: eval mode (BatchNorm running stats, Dropout2d off);
    bn_training=True evaluates linear_fuse.bn with batch statistics as nn.BatchNorm2d does in train
    mode (running statistics in `sd` are updated in place, momentum 0.1); Dropout2d stays off.

    linear_fuse is mmcv ConvModule = conv(no bias) -> BatchNorm2d(eps 1e-5) -> ReLU
    (semantics from mmcv 1.x, not verifiable offline).
    """
    c1 = feats[0]
    n = c1.shape[0]
    size = c1.shape[2:]
    embedded = []
    for idx in (4, 3, 2, 1):
        c = feats[idx - 1]
        e = _lin(c.flatten(2).transpose(1, 2), sd, f"{pfx}linear_c{idx}.proj")
        e = e.permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
        if idx != 1:
            e = F.interpolate(e, size=size, mode="bilinear", align_corners=False)
        embedded.append(e)
    cat = torch.cat(embedded, dim=1)
    y = F.conv2d(cat, sd[pfx + "linear_fuse.conv.weight"])
    y = F.batch_norm(y, sd[pfx + "linear_fuse.bn.running_mean"], sd[pfx + "linear_fuse.bn.running_var"],
                     sd[pfx + "linear_fuse.bn.weight"], sd[pfx + "linear_fuse.bn.bias"], bn_training, 0.1, 1e-5)
    y = F.relu(y)
    return F.conv2d(y, sd[pfx + "linear_pred.weight"], sd[pfx + "linear_pred.bias"])


def network3_forward(sd, x, backbone, bn_training=False):
    """core/model_fusion.py:1081-1088 + WeTr.forward :62-68 -> seg logits (B, K, H/4, W/4).

    The reference also evaluates `classifier(_x4)` and drops it (:66); it has no effect on
    the returned value and is omitted here.
    """
    mean = torch.tensor(SEG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(SEG_STD, dtype=x.dtype).view(1, 3, 1, 1)
    xn = (x * 255 - mean) / std
    feats = mit_forward_features(sd, "denoise_net.encoder.", xn, backbone)
    return segformer_head(sd, "denoise_net.decoder.", feats, bn_training=bn_training)


# --- Fusion network -------------------------------------------------------------------------
def drdb(sd, pfx, x):
    """core/model_fusion.py:134-157: five dilated(2) 3x3 convs over a growing concat, 1x1, residual."""
    cat = x
    for i in range(1, 6):
        y = F.conv2d(cat, sd[f"{pfx}.Dcov{i}.weight"], sd[f"{pfx}.Dcov{i}.bias"], padding=2, dilation=2)
        cat = torch.cat([cat, F.relu(y)], dim=1)
    y = F.conv2d(cat, sd[pfx + ".conv.weight"], sd[pfx + ".conv.bias"])
    return x + F.relu(y)


def _linear_attention_context(kv, heads):
    """softmax_{dim=-2}((K^T V) * d^-1/2) per head: core/model_fusion.py:281-282, 316-319."""
    B, N, C2 = kv.shape
    C = C2 // 2
    d = C // heads
    kv = kv.reshape(B, N, 2, heads, d)
    k = kv[:, :, 0].permute(0, 2, 1, 3)  # (B, h, N, d)
    v = kv[:, :, 1].permute(0, 2, 1, 3)
    ctx = (k.transpose(-2, -1) @ v) * (d ** -0.5)
    return torch.softmax(ctx, dim=-2)


def _apply_context(q, ctx, heads):
    B, N, C = q.shape
    qh = q.reshape(B, N, heads, C // heads).permute(0, 2, 1, 3)
    return (qh @ ctx).permute(0, 2, 1, 3).reshape(B, N, C)


def cross_path(sd, pfx, x1, x2, seg, heads=8):
    """core/model_fusion.py:350-361 with CrossAttention :263-288 and CrossAttention2 :303-328."""
    p1 = F.relu(_lin(x1, sd, pfx + ".channel_proj1"))
    p2 = F.relu(_lin(x2, sd, pfx + ".channel_proj2"))
    p3 = F.relu(_lin(seg, sd, pfx + ".channel_proj3"))
    y1, u1 = p1.chunk(2, dim=-1)
    y2, u2 = p2.chunk(2, dim=-1)
    y3, u3 = p3.chunk(2, dim=-1)
    # CrossAttention(u1, u2, u3): context from the seg feature, queries = raw u1/u2
    ctx3 = _linear_attention_context(_lin(u3, sd, pfx + ".cross_attn.kv3"), heads)
    v1 = _apply_context(u1, ctx3, heads)
    v2 = _apply_context(u2, ctx3, heads)
    # CrossAttention2(y1, y2, y3): contexts from each modality, query = seg feature
    ctx1 = _linear_attention_context(_lin(y1, sd, pfx + ".cross_attn2.kv1"), heads)
    ctx2 = _linear_attention_context(_lin(y2, sd, pfx + ".cross_attn2.kv2"), heads)
    z1 = _apply_context(y3, ctx1, heads)
    z2 = _apply_context(y3, ctx2, heads)
    o1 = _ln(x1 + _lin(torch.cat((z1, v1), dim=-1), sd, pfx + ".end_proj1"), sd, pfx + ".norm1", DEFAULT_LN_EPS)
    o2 = _ln(x2 + _lin(torch.cat((z2, v2), dim=-1), sd, pfx + ".end_proj2"), sd, pfx + ".norm2", DEFAULT_LN_EPS)
    return o1, o2


def cross_attention(sd, pfx, x1, x2, seg, heads=8):
    """CrossAttention.forward on its own (core/model_fusion.py:263-288): one context from the segmentation feature (kv3,
    no bias), applied to the two modality features used as queries."""
    ctx3 = _linear_attention_context(_lin(seg, sd, pfx + "kv3"), heads)
    return _apply_context(x1, ctx3, heads), _apply_context(x2, ctx3, heads)


def cross_attention2(sd, pfx, x1, x2, seg, heads=8):
    """CrossAttention2.forward on its own (core/model_fusion.py:303-328): one context per modality (kv1 / kv2, no bias),
    each applied to the segmentation feature used as the query."""
    ctx1 = _linear_attention_context(_lin(x1, sd, pfx + "kv1"), heads)
    ctx2 = _linear_attention_context(_lin(x2, sd, pfx + "kv2"), heads)
    return _apply_context(seg, ctx1, heads), _apply_context(seg, ctx2, heads)


def dwconv_tokens(sd, pfx, x, H, W):
    """DWConv.forward (core/mix_transformer.py:381-387): tokens -> image, depthwise 3x3 (pad 1) + bias, back to tokens."""
    B, N, C = x.shape
    img = x.transpose(1, 2).reshape(B, C, H, W)
    img = F.conv2d(img, sd[pfx + "dwconv.weight"], sd[pfx + "dwconv.bias"], padding=1, groups=C)
    return img.flatten(2).transpose(1, 2)


def _lap_window(size, sigma=2.0):
    """lap_loss.py:39-80 `smoothing`: exp(-((x - m)^2 + (y - m)^2) / (2 sigma^2)) / (2 pi sigma^2), m = (size - 1) / 2,
    normalised to sum 1, fp32."""
    c = torch.arange(size, dtype=torch.float32)
    xg = c.repeat(size).view(size, size)
    d2 = (xg - (size - 1) / 2.0) ** 2 + (xg.t() - (size - 1) / 2.0) ** 2
    g = (1.0 / (2.0 * math.pi * sigma ** 2)) * torch.exp(-d2 / (2 * sigma ** 2))
    return (g / g.sum()).view(1, 1, size, size)


def lap_loss2(gen, ir, vis):
    """lap_loss.LapLoss2.forward (lap_loss.py:100-118, laplacian_pyramid :71-79): "levels" img - G_k * img for the 3 / 5 / 7
    windows (zero padding k // 2), 10 * (L1 of levels 0 and 1) + L1 of level 2 against max(level(ir), level(vis))."""
    C = gen.shape[1]
    levels = []
    for size in (3, 5, 7):
        w = _lap_window(size).repeat(C, 1, 1, 1)
        levels.append([t - F.conv2d(t, w, padding=size // 2, groups=C) for t in (gen, ir, vis)])
    loss = 10.0 * sum(F.l1_loss(a, torch.maximum(b, c)) for a, b, c in levels[:-1])
    a, b, c = levels[-1]
    return loss + F.l1_loss(a, torch.maximum(b, c))


def feature_fusion_module(sd, pfx, x1, x2, seg):
    """core/model_fusion.py:453-463: NCHW -> tokens -> CrossPath -> NCHW."""
    B, C, H, W = x1.shape
    tok = lambda t: t.flatten(2).transpose(1, 2)
    o1, o2 = cross_path(sd, pfx + ".cross", tok(x1), tok(x2), tok(seg))
    img = lambda t: t.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
    return img(o1), img(o2)


def fusion_network3_ac(sd, ir, vis, out1, out2, taps=None):
    """core/model_fusion.py:1047-1067.  One scalar PReLU shared by all five uses; `ffm` is
    applied twice and `ffm2` never."""
    a = sd["relu.weight"]
    prelu = lambda t: F.prelu(t, a)
    conv = lambda t, name, pad: F.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], padding=pad)
    x1 = drdb(sd, "DRDB1", prelu(conv(ir[:, 0:1], "conv1_ir", 1)))
    x2 = drdb(sd, "DRDB2", prelu(conv(vis[:, 0:1], "conv1_vis", 1)))
    if taps is not None:
        taps["drdb1"], taps["drdb2"] = x1, x2
    x1, x2 = feature_fusion_module(sd, "ffm", x1, x2, conv(out1, "conv3", 0))
    if taps is not None:
        taps["ffm_a1"], taps["ffm_a2"] = x1, x2
    x1 = drdb(sd, "DRDB3", x1)
    x2 = drdb(sd, "DRDB4", x2)
    x1, x2 = feature_fusion_module(sd, "ffm", x1, x2, conv(out2, "conv4", 0))
    if taps is not None:
        taps["ffm_b1"], taps["ffm_b2"] = x1, x2
    f = prelu(conv(torch.cat([x1, x2], dim=1), "conv2", 1))
    f = prelu(conv(f, "conv21", 1))
    return prelu(conv(f, "conv22", 1))


# --- (r6) ablation / variant networks (core/model_fusion.py:363-1025) ---------------------------------------------------
def cross_path_variant(sd, pfx, x1, x2, seg, use, heads=8, want_maps=False):
    """CrossPath_M (use = 'v', ref :385-395), CrossPath_S ('z', :419-429), CrossPath / CrossPath_showAttention ('zv', :350-361,
    :560-572): the attention(s) kept, end_proj over what they return."""
    p1 = F.relu(_lin(x1, sd, pfx + ".channel_proj1"))
    p2 = F.relu(_lin(x2, sd, pfx + ".channel_proj2"))
    p3 = F.relu(_lin(seg, sd, pfx + ".channel_proj3"))
    (y1, u1), (y2, u2), (y3, u3) = p1.chunk(2, dim=-1), p2.chunk(2, dim=-1), p3.chunk(2, dim=-1)
    parts1, parts2, maps = [], [], {}
    if "z" in use:
        ctx1 = _linear_attention_context(_lin(y1, sd, pfx + ".cross_attn2.kv1"), heads)
        ctx2 = _linear_attention_context(_lin(y2, sd, pfx + ".cross_attn2.kv2"), heads)
        maps["z1"], maps["z2"] = _apply_context(y3, ctx1, heads), _apply_context(y3, ctx2, heads)
        parts1.append(maps["z1"])
        parts2.append(maps["z2"])
    if "v" in use:
        ctx3 = _linear_attention_context(_lin(u3, sd, pfx + ".cross_attn.kv3"), heads)
        maps["v1"], maps["v2"] = _apply_context(u1, ctx3, heads), _apply_context(u2, ctx3, heads)
        parts1.append(maps["v1"])
        parts2.append(maps["v2"])
    o1 = _ln(x1 + _lin(torch.cat(parts1, dim=-1), sd, pfx + ".end_proj1"), sd, pfx + ".norm1", DEFAULT_LN_EPS)
    o2 = _ln(x2 + _lin(torch.cat(parts2, dim=-1), sd, pfx + ".end_proj2"), sd, pfx + ".norm2", DEFAULT_LN_EPS)
    if want_maps:
        return o1, o2, [maps["v1"], maps["z1"], maps["z2"], maps["v2"]]
    return o1, o2


def ffm_variant(sd, pfx, x1, x2, seg, use):
    """FeatureFusionModule / _SoAM / _MoAM / _ShowAttention (ref :453-463, :490-501, :526-536, :596-605): tokens, CrossPath*, back."""
    B, C, H, W = x1.shape
    tok = lambda t: t.flatten(2).transpose(1, 2)
    o1, o2 = cross_path_variant(sd, pfx + ".cross", tok(x1), tok(x2), tok(seg), use)
    img = lambda t: t.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
    return img(o1), img(o2)


def attention_module(sd, pfx, x):
    """AttentionModule (ref :759-770): z = conv(relu(conv(x))); sigmoid(z) * z."""
    z = F.conv2d(F.relu(F.conv2d(x, sd[pfx + ".conv.0.weight"], sd[pfx + ".conv.0.bias"], padding=1)),
                 sd[pfx + ".conv.2.weight"], sd[pfx + ".conv.2.bias"], padding=1)
    return torch.sigmoid(z) * z


def fusion_variant(sd, name, ir, vis, out1=None, out2=None):
    """The reference's ablation networks by class name: Fusion_Network3 (:643-660), _S (:838-854), _M (:873-889),
    _obtainattention (:913-932), _Con (:680-711), _Add (:732-757), _Average (:796-819), Fusion_Network_rmseg (:949-979),
    Fusion_Network_rmseg_att (:996-1025).  Returns what the class's forward returns."""
    a = sd["relu.weight"]
    prelu = lambda t: F.prelu(t, a)
    conv = lambda t, n, pad: F.conv2d(t, sd[n + ".weight"], sd[n + ".bias"], padding=pad)
    x1 = drdb(sd, "DRDB1", prelu(conv(ir[:, 0:1], "conv1_ir", 1)))
    x2 = drdb(sd, "DRDB2", prelu(conv(vis[:, 0:1], "conv1_vis", 1)))
    if name in ("Fusion_Network_rmseg", "Fusion_Network_rmseg_att"):
        x1, x2 = drdb(sd, "DRDB3", x1), drdb(sd, "DRDB4", x2)
        f = prelu(conv(prelu(conv(prelu(conv(torch.cat([x1, x2], dim=1), "conv2", 1)), "conv21", 1)), "conv22", 1))
        return f if name == "Fusion_Network_rmseg" else (f, [x1, x2])
    s1, s2 = conv(out1, "conv3", 0), conv(out2, "conv4", 0)
    a_in = [x1, x2]
    if name in ("Fusion_Network3", "Fusion_Network3_S", "Fusion_Network3_M", "Fusion_Network3_obtainattention"):
        use = {"Fusion_Network3_S": "z", "Fusion_Network3_M": "v"}.get(name, "zv")
        mix = lambda p, q, s, stage: ffm_variant(sd, "ffm", p, q, s, use)
    elif name == "Fusion_Network3_Con":
        mix = lambda p, q, s, stage: (conv(torch.cat([p, s], dim=1), ("conv211", "conv411")[stage], 1),
                                      conv(torch.cat([q, s], dim=1), ("conv221", "conv421")[stage], 1))
    elif name == "Fusion_Network3_Add":
        mix = lambda p, q, s, stage: (conv(p + s, ("conv211", "conv411")[stage], 1), conv(q + s, ("conv221", "conv421")[stage], 1))
    elif name == "Fusion_Network3_Average":
        mix = lambda p, q, s, stage: (
            attention_module(sd, f"att{1 + 4 * stage}", p) + attention_module(sd, f"att{2 + 4 * stage}", s),
            attention_module(sd, f"att{3 + 4 * stage}", q) + attention_module(sd, f"att{4 + 4 * stage}", s))
    else:
        raise KeyError(name)
    x1, x2 = mix(x1, x2, s1, 0)
    x1, x2 = drdb(sd, "DRDB3", x1), drdb(sd, "DRDB4", x2)
    x1, x2 = mix(x1, x2, s2, 1)
    f2 = conv(torch.cat([x1, x2], dim=1), "conv2", 1)
    f = prelu(conv(prelu(f2), "conv21", 1))
    return (f, a_in + [f2]) if name == "Fusion_Network3_obtainattention" else f


# --- colour transforms ----------------------------------------------------------------------
def rgb2ycrcb(x):
    """core/model_fusion.py:69-91 (coefficients :75-77)."""
    R, G, B = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    Y = 0.299 * R + 0.587 * G + 0.114 * B
    Cr = (R - Y) * 0.713 + 0.5
    Cb = (B - Y) * 0.564 + 0.5
    return torch.cat((Y, Cr, Cb), dim=1)


def ycrcb2rgb(x):
    """core/model_fusion.py:93-111: ([Y,Cr,Cb] + [0,-.5,-.5]) @ M."""
    mat = torch.tensor([[1.0, 1.0, 1.0], [1.403, -0.714, 0.0], [0.0, -0.344, 1.773]], dtype=x.dtype)
    bias = torch.tensor([0.0, -0.5, -0.5], dtype=x.dtype)
    flat = x.permute(0, 2, 3, 1).reshape(-1, 3)
    out = (flat + bias).mm(mat)
    return out.reshape(x.shape[0], x.shape[2], x.shape[3], 3).permute(0, 3, 1, 2)


# --- the measured unit of work: one IR+visible pair forward ---------------
def pair_forward(sd_seg, sd_fus, ir, vis, mask3, backbone, return_all=False):
    """test_fusion.py:100-111 followed by test_segmentation.py:169-174, in memory.

    ir (B,1,H,W), vis (B,3,H,W) RGB in [0,1], mask3 (B,3,H,W).
    Returns (fused_rgb, logits_fullres, labels).
    """
    H, W = vis.shape[2], vis.shape[3]
    out0, out1 = mit_forward_fusion(sd_seg, "denoise_net.encoder.", mask3, backbone)
    y_f = fusion_network3_ac(sd_fus, ir, vis, out0, out1)
    ycc = rgb2ycrcb(vis)
    fused = ycrcb2rgb(torch.cat((y_f, ycc[:, 1:2], ycc[:, 2:3]), dim=1)).clamp(0.0, 1.0)
    seg = network3_forward(sd_seg, fused, backbone)
    logits = F.interpolate(seg, size=[H, W], mode="bilinear", align_corners=False)
    labels = logits.argmax(1)
    if return_all:
        return dict(out0=out0, out1=out1, y_fused=y_f, fused=fused, seg=seg, logits=logits, labels=labels)
    return fused, logits, labels


# --- metric ---------------------------------------------------------------------------------
def confusion(labels_true, labels_pred, n_class=9):
    t = np.asarray(labels_true).reshape(-1).astype(np.int64)
    p = np.asarray(labels_pred).reshape(-1).astype(np.int64)
    keep = (t >= 0) & (t < n_class)
    return np.bincount(t[keep] * n_class + p[keep], minlength=n_class * n_class).reshape(n_class, n_class)


def miou(conf):
    """util/util.py:31-55: per-class IoU = TP / (TP+FP+FN), NaN for empty classes; mean over classes
    as test_segmentation.py:192 prints it (nan-to-num variant)."""
    conf = np.asarray(conf, dtype=np.float64)
    tp = np.diag(conf)
    denom = conf.sum(0) + conf.sum(1) - tp
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = np.where(denom > 0, tp / denom, np.nan)
    return float(np.mean(np.nan_to_num(iou))), iou


def compute_results(conf_total):
    """util/util.py:31-55 with consider_unlabeled=True: per class cid, precision = conf[cid,cid] / column
    sum, recall = conf[cid,cid] / row sum, IoU = conf[cid,cid] / (row + column - diagonal); NaN when the
    denominator is 0.  Loop form, as the reference writes it."""
    conf = np.asarray(conf_total)
    n = conf.shape[0]
    prec, rec, iou = np.zeros(n), np.zeros(n), np.zeros(n)
    for cid in range(n):
        col, row, tp = conf[:, cid].sum(), conf[cid, :].sum(), conf[cid, cid]
        prec[cid] = np.nan if col == 0 else float(tp) / float(col)
        rec[cid] = np.nan if row == 0 else float(tp) / float(row)
        iou[cid] = np.nan if (row + col - tp) == 0 else float(tp) / float(row + col - tp)
    return prec, rec, iou


def quantize_fused_u8(fused):
    """test_fusion.py:112-120 on the clamped fused image (B,3,H,W) float32: np.uint8(255.0 * x), NHWC
    transpose, (a - min) / (max - min) with the GLOBAL min / max of the batch (uint8 - uint8, then a true
    division -> float64), np.uint8(255.0 * .).  max == min divides 0 by 0; the write-out is then 0."""
    x = np.asarray(fused, dtype=np.float32)
    a = np.uint8(255.0 * x).transpose((0, 2, 3, 1))
    lo, hi = np.min(a), np.max(a)
    if hi == lo:
        return np.zeros_like(a)
    return np.uint8(255.0 * ((a - lo) / (hi - lo)))


def top2_margin(logits):
    """Gap between the best and second-best class logit per pixel (argmax stability gate)."""
    top = torch.topk(logits, 2, dim=1).values
    return top[:, 0] - top[:, 1]


# --- shape tables (state_dict keys) ---------------------------------------------------------
def mit_shapes(variant, pfx=""):
    cfg = MIT_VARIANTS[variant]
    dims, depths = cfg["dims"], cfg["depths"]
    sh = {}
    cin = 3
    for s in range(4):
        C = dims[s]
        k, _ = MIT_PATCH[s]
        pe = f"{pfx}patch_embed{s + 1}"
        sh[pe + ".proj.weight"] = (C, cin, k, k)
        sh[pe + ".proj.bias"] = (C,)
        sh[pe + ".norm.weight"] = (C,)
        sh[pe + ".norm.bias"] = (C,)
        for i in range(depths[s]):
            b = f"{pfx}block{s + 1}.{i}"
            for n in ("norm1", "norm2"):
                sh[f"{b}.{n}.weight"] = (C,)
                sh[f"{b}.{n}.bias"] = (C,)
            sh[b + ".attn.q.weight"] = (C, C)
            sh[b + ".attn.q.bias"] = (C,)
            sh[b + ".attn.kv.weight"] = (2 * C, C)
            sh[b + ".attn.kv.bias"] = (2 * C,)
            sh[b + ".attn.proj.weight"] = (C, C)
            sh[b + ".attn.proj.bias"] = (C,)
            if MIT_SR[s] > 1:
                sh[b + ".attn.sr.weight"] = (C, C, MIT_SR[s], MIT_SR[s])
                sh[b + ".attn.sr.bias"] = (C,)
                sh[b + ".attn.norm.weight"] = (C,)
                sh[b + ".attn.norm.bias"] = (C,)
            sh[b + ".mlp.fc1.weight"] = (4 * C, C)
            sh[b + ".mlp.fc1.bias"] = (4 * C,)
            sh[b + ".mlp.dwconv.dwconv.weight"] = (4 * C, 1, 3, 3)
            sh[b + ".mlp.dwconv.dwconv.bias"] = (4 * C,)
            sh[b + ".mlp.fc2.weight"] = (C, 4 * C)
            sh[b + ".mlp.fc2.bias"] = (C,)
        sh[f"{pfx}norm{s + 1}.weight"] = (C,)
        sh[f"{pfx}norm{s + 1}.bias"] = (C,)
        cin = C
    return sh


def network3_shapes(backbone, num_classes=9, embed=256):
    dims = MIT_VARIANTS[backbone]["dims"]
    sh = mit_shapes(backbone, "denoise_net.encoder.")
    d = "denoise_net.decoder."
    for i in range(4):
        sh[f"{d}linear_c{i + 1}.proj.weight"] = (embed, dims[i])
        sh[f"{d}linear_c{i + 1}.proj.bias"] = (embed,)
    sh[d + "linear_fuse.conv.weight"] = (embed, 4 * embed, 1, 1)
    for n in ("weight", "bias", "running_mean", "running_var"):
        sh[d + "linear_fuse.bn." + n] = (embed,)
    sh[d + "linear_fuse.bn.num_batches_tracked"] = ()
    sh[d + "linear_pred.weight"] = (num_classes, embed, 1, 1)
    sh[d + "linear_pred.bias"] = (num_classes,)
    sh["denoise_net.classifier.weight"] = (num_classes, dims[3], 1, 1)
    return sh


def fusion_shapes():
    sh = {}

    def conv(name, co, ci, k):
        sh[name + ".weight"] = (co, ci, k, k)
        sh[name + ".bias"] = (co,)

    conv("conv1_ir", 64, 1, 3)
    conv("conv1_vis", 64, 1, 3)
    for d in range(1, 5):
        for i in range(5):
            conv(f"DRDB{d}.Dcov{i + 1}", 32, 64 + 32 * i, 3)
        conv(f"DRDB{d}.conv", 64, 224, 1)
    conv("conv2", 64, 128, 3)
    sh["relu.weight"] = (1,)
    for f in ("ffm", "ffm2"):
        c = f + ".cross."
        for i in (1, 2, 3):
            sh[f"{c}channel_proj{i}.weight"] = (128, 64)
            sh[f"{c}channel_proj{i}.bias"] = (128,)
        sh[c + "cross_attn.kv3.weight"] = (128, 64)
        sh[c + "cross_attn2.kv1.weight"] = (128, 64)
        sh[c + "cross_attn2.kv2.weight"] = (128, 64)
        for i in (1, 2):
            sh[f"{c}end_proj{i}.weight"] = (64, 128)
            sh[f"{c}end_proj{i}.bias"] = (64,)
            sh[f"{c}norm{i}.weight"] = (64,)
            sh[f"{c}norm{i}.bias"] = (64,)
    conv("conv3", 64, 64, 1)
    conv("conv4", 64, 128, 1)
    conv("conv21", 32, 64, 3)
    conv("conv22", 1, 32, 3)
    return sh
