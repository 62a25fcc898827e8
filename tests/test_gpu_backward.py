"""GPU parity tests for the training path: every backward kernel against torch-CPU fp64 autograd
of the aten op the reference differentiates, the seg-training step's parameter gradients against
the CPU oracle's autograd, and the fused AdamW against torch.optim.AdamW."""
import pytest
import torch
import torch.nn.functional as F

import detweights as dw
import segmif_oracle as so

pytestmark = pytest.mark.gpu

TOL = 5e-5


@pytest.fixture(scope="module")
def ag():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from segmif_amd import autograd as _ag
    return _ag


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def leaf(t, dev=None, double=False):
    t = t.double() if double else t.clone()
    if dev:
        t = t.to(dev)
    return t.requires_grad_(True)


@pytest.mark.parametrize("M,N,K,act", [(1000, 64, 64, 0), (777, 128, 320, 1), (300, 9, 256, 0), (4100, 32, 224, 1),
                                       (2500, 512, 128, 2)])
def test_linear_backward(ag, M, N, K, act):
    x, w, b, g = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    slope = torch.tensor([0.2])
    xr, wr, br, sr = leaf(x, double=True), leaf(w, double=True), leaf(b, double=True), leaf(slope, double=True)
    y = F.linear(xr, wr, br)
    y = F.relu(y) if act == 1 else (F.prelu(y, sr) if act == 2 else y)
    y.backward(g.double())
    xg, wg, bg, sg = leaf(x, "cuda"), leaf(w, "cuda"), leaf(b, "cuda"), leaf(slope, "cuda")
    yg = ag.linear(xg, wg, bg, act=act, slope=sg if act == 2 else None)
    assert err(yg, y) < TOL
    yg.backward(g.cuda())
    assert err(xg.grad, xr.grad) < TOL and err(wg.grad, wr.grad) < TOL and err(bg.grad, br.grad) < TOL
    if act == 2:
        assert err(sg.grad, sr.grad) < 1e-4


CONV_CASES = [
    # B, H, W, Cin, N, k, stride, pad, dil, act
    (2, 20, 28, 64, 32, 3, 1, 2, 2, 1),  # DRDB dilated conv + ReLU
    (1, 17, 23, 64, 128, 3, 2, 1, 1, 0),  # overlap patch embed (strided, overlapping)
    (1, 16, 24, 128, 128, 4, 4, 0, 1, 0),  # sr conv
    (1, 9, 13, 32, 64, 2, 2, 0, 1, 0),  # sr conv dropping the remainder
    (2, 21, 19, 3, 32, 7, 4, 3, 1, 0),  # stage-1 patch embed
    (2, 12, 20, 32, 1, 3, 1, 1, 1, 2),  # conv22 + PReLU
    (1, 14, 18, 128, 64, 3, 1, 1, 1, 2),  # conv2 + PReLU
    (2, 10, 12, 1, 64, 3, 1, 1, 1, 2),  # conv1
    (1, 19, 45, 96, 64, 3, 1, 2, 2, 0),  # dilated, odd width, two 32-channel output tiles, three input chunks (bf16x6 wgrad)
    (3, 40, 70, 160, 32, 3, 1, 2, 2, 1),  # DRDB Dcov4 geometry: several 8x32 tiles per strip, partial tiles on both edges
    (2, 23, 37, 64, 32, 3, 1, 1, 1, 2),  # conv21 geometry, ragged: bf16x6 weight gradient at dilation 1 (funnel-shifted middle tap)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_backward(ag, case):
    B, H, W, Cin, N, k, s, p, d, act = case
    x, w, b = rnd(B, Cin, H, W, seed=5), rnd(N, Cin, k, k, seed=6), rnd(N, seed=7)
    slope = torch.tensor([0.25])
    xr, wr, br, sr = leaf(x, double=True), leaf(w, double=True), leaf(b, double=True), leaf(slope, double=True)
    y = F.conv2d(xr, wr, br, stride=s, padding=p, dilation=d)
    y = F.relu(y) if act == 1 else (F.prelu(y, sr) if act == 2 else y)
    g = rnd(*y.shape, seed=8)
    y.backward(g.double())
    xg = leaf(x.permute(0, 2, 3, 1).contiguous(), "cuda")
    wg, bg, sg = leaf(w, "cuda"), leaf(b, "cuda"), leaf(slope, "cuda")
    yg = ag.conv2d(xg, wg, bg, k=k, stride=s, pad=p, dil=d, act=act, slope=sg if act == 2 else None)
    assert err(yg, y.permute(0, 2, 3, 1)) < TOL
    yg.backward(g.permute(0, 2, 3, 1).contiguous().cuda())
    assert err(xg.grad, xr.grad.permute(0, 2, 3, 1)) < TOL
    assert err(wg.grad, wr.grad) < TOL and err(bg.grad, br.grad) < TOL
    if act == 2:
        assert err(sg.grad, sr.grad) < 1e-4


@pytest.mark.parametrize("rows,C", [(1003, 64), (517, 320), (40, 512), (9000, 128), (333, 32)])
def test_layernorm_backward(ag, rows, C):
    x, gm, bt, g = rnd(rows, C, seed=9, lo=-3, hi=5), rnd(C, seed=10), rnd(C, seed=11), rnd(rows, C, seed=12)
    xr, gr, br = leaf(x, double=True), leaf(gm, double=True), leaf(bt, double=True)
    F.layer_norm(xr, (C,), gr, br, 1e-6).backward(g.double())
    xg, gg, bg = leaf(x, "cuda"), leaf(gm, "cuda"), leaf(bt, "cuda")
    ag.layernorm(xg, gg, bg, 1e-6).backward(g.cuda())
    assert err(xg.grad, xr.grad) < TOL and err(gg.grad, gr.grad) < TOL and err(bg.grad, br.grad) < TOL


@pytest.mark.parametrize("B,H,W,C", [(2, 9, 13, 128), (1, 30, 40, 1280), (1, 5, 3, 256)])
def test_dwconv_gelu_backward(ag, B, H, W, C):
    h, w, b, g = rnd(B, H * W, C, seed=13, lo=-2, hi=2), rnd(C, 1, 3, 3, seed=14), rnd(C, seed=15), rnd(B, H * W, C, seed=16)
    hr, wr, br = leaf(h, double=True), leaf(w, double=True), leaf(b, double=True)
    img = hr.transpose(1, 2).reshape(B, C, H, W)
    F.gelu(F.conv2d(img, wr, br, padding=1, groups=C)).flatten(2).transpose(1, 2).backward(g.double())
    hg, wg, bg = leaf(h, "cuda"), leaf(w, "cuda"), leaf(b, "cuda")
    ag.dwconv_gelu(hg, wg, bg, H, W).backward(g.cuda())
    assert err(hg.grad, hr.grad) < TOL and err(wg.grad, wr.grad) < TOL and err(bg.grad, br.grad) < TOL


@pytest.mark.parametrize("B,IH,IW,OH,OW,C", [(2, 16, 24, 64, 96, 64), (1, 15, 20, 120, 160, 256), (1, 5, 7, 18, 26, 32),
                                             (2, 18, 26, 72, 104, 9), (1, 30, 40, 30, 40, 8), (1, 64, 48, 16, 12, 4)])
def test_bilinear_backward(ag, B, IH, IW, OH, OW, C):
    x, g = rnd(B, IH, IW, C, seed=17), rnd(B, OH, OW, C, seed=18)
    xr = leaf(x.permute(0, 3, 1, 2), double=True)
    F.interpolate(xr, size=[OH, OW], mode="bilinear", align_corners=False).backward(g.permute(0, 3, 1, 2).double())
    xg = leaf(x, "cuda")
    ag.bilinear(xg, OH, OW).backward(g.cuda())
    assert err(xg.grad, xr.grad.permute(0, 2, 3, 1)) < TOL


@pytest.mark.parametrize("B,heads,N,Nk,hd", [(2, 2, 96, 6, 64), (1, 1, 2000, 300, 64), (2, 5, 130, 35, 64),
                                             (1, 8, 300, 300, 64), (1, 2, 256, 64, 32)])
def test_sr_attention_backward(ag, B, heads, N, Nk, hd):
    C = heads * hd
    q, kv, g = rnd(B, N, C, seed=19, lo=-2, hi=2), rnd(B, Nk, 2 * C, seed=20, lo=-2, hi=2), rnd(B, N, C, seed=21)
    qr, kvr = leaf(q, double=True), leaf(kv, double=True)
    scale = hd ** -0.5
    qh = qr.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kvh = kvr.reshape(B, Nk, 2, heads, hd)
    k, v = kvh[:, :, 0].permute(0, 2, 1, 3), kvh[:, :, 1].permute(0, 2, 1, 3)
    (torch.softmax(qh @ k.transpose(-2, -1) * scale, -1) @ v).transpose(1, 2).reshape(B, N, C).backward(g.double())
    qg, kvg = leaf(q, "cuda"), leaf(kv, "cuda")
    ag.sr_attention(qg, kvg, heads, scale).backward(g.cuda())
    assert err(qg.grad, qr.grad) < TOL and err(kvg.grad, kvr.grad) < TOL


def test_softmax_ce_with_ignore(ag):
    B, H, W, C = 2, 17, 23, 9
    x = rnd(B, H, W, C, seed=22, lo=-3, hi=3)
    labels = dw.det_labels("ce", (B, H, W), 9)
    labels[0, :3] = 255
    xr = leaf(x.permute(0, 3, 1, 2), double=True)
    lr = F.cross_entropy(xr, labels, ignore_index=255)
    lr.backward()
    xg = leaf(x, "cuda")
    lg = ag.softmax_ce(xg, labels.cuda(), 255)
    lg.backward()
    assert abs(float(lg.detach()) - float(lr.detach())) / abs(float(lr.detach())) < 1e-5
    assert err(xg.grad, xr.grad.permute(0, 2, 3, 1)) < TOL


def test_seg_training_step_gradients_match_oracle_autograd(ag):
    """Network3('mit_b1') in eval mode with grad (the deterministic regime of train_seg):
    loss = CE(bilinear-up(seg), labels).  Every parameter gradient vs the oracle's autograd."""
    from segmif_amd.core import Network3
    B, H, W = 2, 64, 96
    x = dw.det_input("tr_x", (B, 3, H, W))
    labels = dw.det_labels("tr_y", (B, H, W), 9)
    labels[0, 5:9, 7:30] = 255
    sd = {}
    for k, v in dw.det_state_dict(so.network3_shapes("mit_b1", 9), seed=0).items():
        is_param = v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))
        sd[k] = v.double().requires_grad_(True) if is_param else (v.double() if v.dtype.is_floating_point else v)
    seg = so.network3_forward(sd, x.double(), "mit_b1")
    ref_loss = F.cross_entropy(F.interpolate(seg, size=[H, W], mode="bilinear", align_corners=False), labels,
                               ignore_index=255)
    ref_loss.backward()
    net = Network3("mit_b1", 9, pretrained=None)
    dw.load_det_weights(net, seed=0)
    net = net.cuda().eval()
    loss = net._loss(x.cuda(), labels.cuda(), torch.nn.CrossEntropyLoss(ignore_index=255))
    assert abs(float(loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach())) < 1e-4
    loss.backward()
    worst = ("", 0.0)
    checked = 0
    for name, p in net.named_parameters():
        ref = sd[name].grad if sd[name].requires_grad else None
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name  # classifier.weight: no grad
            continue
        assert p.grad is not None, name
        e = err(p.grad, ref)
        checked += 1
        if e > worst[1]:
            worst = (name, e)
    assert checked > 150
    assert worst[1] < 1e-3, worst  # north-star tolerance; typically ~1e-5


def test_fused_adamw_matches_torch(ag):
    from segmif_amd.utils.optimizer import FusedAdamW, PolyWarmupAdamW_seg
    torch.manual_seed(0)
    shapes = [(64, 64), (128,), (32, 16, 3, 3), (1,), (100000,)]
    ps = [torch.randn(s) for s in shapes]
    a = [p.clone().cuda().requires_grad_(True) for p in ps]
    b = [p.clone().cuda().requires_grad_(True) for p in ps]
    oa = FusedAdamW([{"params": a[:2], "lr": 1e-3, "weight_decay": 0.01}, {"params": a[2:], "lr": 1e-2, "weight_decay": 0.0}],
                    lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ob = torch.optim.AdamW([{"params": b[:2], "lr": 1e-3, "weight_decay": 0.01}, {"params": b[2:], "lr": 1e-2, "weight_decay": 0.0}],
                           lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for it in range(4):
        for i, (pa, pb) in enumerate(zip(a, b)):
            g = torch.randn(pa.shape, device="cuda")
            if i == 3 and it < 2:
                pa.grad = pb.grad = None  # a parameter without a gradient is skipped, not zero-filled
            else:
                pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        assert err(pa, pb.detach().cpu()) < 1e-6
    # schedule mirror: linear warm-up for warmup_iter steps, then polynomial decay, written into param_groups
    p = torch.zeros(4, device="cuda", requires_grad=True)
    opt = PolyWarmupAdamW_seg([{"params": [p], "lr": 1e-2, "weight_decay": 0.0}], lr=1e-2, weight_decay=0.0,
                              betas=(0.9, 0.999), iter_curr=0, warmup_iter=10, max_iter=100, warmup_ratio=0.1, power=1.0)
    seen = []
    for _ in range(12):
        p.grad = torch.ones_like(p)
        opt.step()
        seen.append(opt.param_groups[0]["lr"])
    assert abs(seen[0] - 1e-2 * 0.1) < 1e-12 and abs(seen[5] - 1e-2 * (1 - 0.5 * 0.9)) < 1e-12
    assert abs(seen[11] - 1e-2 * (1 - 11 / 100)) < 1e-12


def _oracle_params(shapes, seed=0):
    sd = {}
    for k, v in dw.det_state_dict(shapes, seed=seed).items():
        is_param = v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))
        sd[k] = v.double().requires_grad_(True) if is_param else (v.double() if v.dtype.is_floating_point else v)
    return sd


def _compare_param_grads(module, sd, tol=1e-3, sd32=None):
    """Every parameter gradient vs the fp64 oracle.  A gradient may exceed `tol` only if the oracle
    evaluated in the reference's own precision (fp32, `sd32`) deviates from fp64 comparably — i.e.
    the quantity is ill-conditioned (gradients through the saturated 8x8 context softmax), not wrong."""
    bad, checked = [], 0
    gmax = max(float(v.grad.abs().max()) for v in sd.values() if isinstance(v, torch.Tensor) and v.requires_grad
               and v.grad is not None)
    for name, p in module.named_parameters():
        ref = sd[name].grad if sd[name].requires_grad else None
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        # gradients that are mathematically zero (e.g. biases in front of a batch-statistics BatchNorm) are
        # compared against a floor tied to the largest gradient in the model, not to their own ~1e-17 magnitude
        floor = 1e-4 * gmax  # fp32 cancellation residue of a ~4e4-term sum is ~1e-8 of the terms
        e = float((p.grad.detach().double().cpu() - ref.double()).abs().max()) / max(float(ref.abs().max()), floor)
        checked += 1
        if e >= tol:
            # ill-conditioned tensors: when the reference's own fp32 evaluation is already outside `tol`
            # of fp64 (roundoff amplified ~1e5x by a softmax with 1-p ~ 1e-5), only the order of magnitude
            # of the deviation is comparable between two fp32 implementations
            e32 = err(sd32[name].grad, ref) if sd32 is not None else 0.0
            if not (e32 >= tol and e <= 10 * e32):
                bad.append((name, e, e32, float(ref.abs().max())))
    assert not bad, bad
    return checked


def _oracle_params32(shapes, seed=0):
    sd = {}
    for k, v in dw.det_state_dict(shapes, seed=seed).items():
        is_param = v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))
        sd[k] = v.clone().requires_grad_(True) if is_param else v
    return sd


def test_drdb_backward(ag):
    from segmif_amd.core import Fusion_Network3_ac
    fus = Fusion_Network3_ac()
    dw.load_det_weights(fus, seed=0)
    fus = fus.cuda()
    sd = _oracle_params(so.fusion_shapes())
    x = rnd(2, 64, 20, 28, seed=30)
    g = rnd(2, 64, 20, 28, seed=31)
    xr = leaf(x, double=True)
    so.drdb(sd, "DRDB1", xr).backward(g.double())
    xg = leaf(x, "cuda")
    fus.DRDB1(xg).backward(g.cuda())
    assert err(xg.grad, xr.grad) < TOL
    for i in range(1, 6):
        assert err(getattr(fus.DRDB1, f"Dcov{i}").weight.grad, sd[f"DRDB1.Dcov{i}.weight"].grad) < TOL, i
        assert err(getattr(fus.DRDB1, f"Dcov{i}").bias.grad, sd[f"DRDB1.Dcov{i}.bias"].grad) < TOL, i
    assert err(fus.DRDB1.conv.weight.grad, sd["DRDB1.conv.weight"].grad) < TOL
    assert err(fus.DRDB1.conv.bias.grad, sd["DRDB1.conv.bias"].grad) < TOL


def test_feature_fusion_module_backward(ag):
    from segmif_amd.core import Fusion_Network3_ac
    fus = Fusion_Network3_ac()
    dw.load_det_weights(fus, seed=0)
    fus = fus.cuda()
    sd = _oracle_params(so.fusion_shapes())
    xs = [rnd(2, 64, 12, 16, seed=32 + i) for i in range(3)]
    g1, g2 = rnd(2, 64, 12, 16, seed=36), rnd(2, 64, 12, 16, seed=37)
    xr = [leaf(t, double=True) for t in xs]
    o1, o2 = so.feature_fusion_module(sd, "ffm", *xr)
    (o1 * g1.double()).sum().add((o2 * g2.double()).sum()).backward()
    xg = [leaf(t, "cuda") for t in xs]
    p1, p2 = fus.ffm(*xg)
    assert err(p1, o1) < TOL and err(p2, o2) < TOL
    (p1 * g1.cuda()).sum().add((p2 * g2.cuda()).sum()).backward()
    for a, b in zip(xg, xr):
        assert err(a.grad, b.grad) < 2e-4
    n = 0
    for name, p in fus.ffm.named_parameters():
        assert err(p.grad, sd["ffm." + name].grad) < 2e-4, name
        n += 1
    assert n == 17


def test_fusion_network_gradients_match_oracle_autograd(ag):
    """Fusion_Network3_ac parameter gradients (the tensors train_fusion's optimizer updates) vs the
    oracle's autograd; ffm2.* must stay without gradient."""
    from segmif_amd.core import Fusion_Network3_ac
    B, H, W = 2, 24, 40
    ir, vis = dw.det_input("g_ir", (B, 1, H, W)), dw.det_input("g_vis", (B, 3, H, W))
    o1, o2 = rnd(B, 64, H, W, seed=40), rnd(B, 128, H, W, seed=41)
    g = rnd(B, 1, H, W, seed=42)
    sd = _oracle_params(so.fusion_shapes())
    ref = so.fusion_network3_ac(sd, ir.double(), vis.double(), o1.double(), o2.double())
    (ref * g.double()).sum().backward()
    fus = Fusion_Network3_ac()
    dw.load_det_weights(fus, seed=0)
    fus = fus.cuda()
    out = fus(ir.cuda(), vis.cuda(), o1.cuda(), o2.cuda())
    assert err(out, ref) < 1e-4
    (out * g.cuda()).sum().backward()
    checked = _compare_param_grads(fus, sd, tol=1e-3)
    assert checked == 97 - 17  # everything except the unused ffm2.* copy (17 tensors)
    assert all(p.grad is None for n, p in fus.named_parameters() if n.startswith("ffm2."))


def test_fusion_training_loss_through_seg_net(ag):
    """The round>=2 objective of train_fusion (train.py:363-374) without its host-side weighting:
    CE(seg(YCrCb2RGB([fusion, Cr, Cb])), labels) + MSE(fusion, mask) back-propagated into the fusion
    net through the (frozen-in-effect) segmentation net, vs the oracle's autograd."""
    from segmif_amd.core import Fusion_Network3_ac, Network3, YCrCb2RGB
    B, H, W = 1, 64, 96
    ir, vis = dw.det_input("l_ir", (B, 1, H, W)), dw.det_input("l_vis", (B, 3, H, W))
    mask = dw.det_input("l_mask", (B, 1, H, W))
    labels = dw.det_labels("l_y", (B, H, W), 9)
    sd_f = _oracle_params(so.fusion_shapes())
    sd_s = {k: (v.double() if v.dtype.is_floating_point else v)
            for k, v in dw.det_state_dict(so.network3_shapes("mit_b1", 9), seed=0).items()}
    with torch.no_grad():
        o0, o1 = so.mit_forward_fusion(sd_s, "denoise_net.encoder.", mask.repeat(1, 3, 1, 1).double(), "mit_b1")
    ycc = so.rgb2ycrcb(vis.double())

    def objective(fusion, seg_fn, rgb_fn):
        rgb = rgb_fn(torch.cat((fusion, ycc_dev[:, 1:2], ycc_dev[:, 2:3]), dim=1))
        return seg_fn(rgb) + ((fusion - mask_dev) ** 2).mean()

    ycc_dev, mask_dev = ycc, mask.double()
    f_ref = so.fusion_network3_ac(sd_f, ir.double(), ycc, o0, o1)
    seg_ref = lambda rgb: F.cross_entropy(
        F.interpolate(so.network3_forward(sd_s, rgb, "mit_b1"), size=[H, W], mode="bilinear", align_corners=False), labels)
    l_ref = objective(f_ref, seg_ref, so.ycrcb2rgb)
    l_ref.backward()
    # the same objective in the reference's own precision (fp32) — the conditioning yardstick
    sd_f32 = _oracle_params32(so.fusion_shapes())
    sd_s32 = dw.det_state_dict(so.network3_shapes("mit_b1", 9), seed=0)
    ycc_dev, mask_dev = ycc.float(), mask
    f32 = so.fusion_network3_ac(sd_f32, ir, ycc.float(), o0.float(), o1.float())
    seg32 = lambda rgb: F.cross_entropy(
        F.interpolate(so.network3_forward(sd_s32, rgb, "mit_b1"), size=[H, W], mode="bilinear", align_corners=False), labels)
    objective(f32, seg32, so.ycrcb2rgb).backward()

    seg = Network3("mit_b1", 9, pretrained=None)
    fus = Fusion_Network3_ac()
    dw.load_det_weights(seg, seed=0)
    dw.load_det_weights(fus, seed=0)
    seg, fus = seg.cuda().eval(), fus.cuda()
    ycc_dev, mask_dev = ycc.float().cuda(), mask.cuda()
    with torch.no_grad():
        g0, g1 = seg.denoise_net.encoder.forward_fusion(mask.repeat(1, 3, 1, 1).cuda())
    f_hip = fus(ir.cuda(), ycc_dev, g0, g1)
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    l_hip = objective(f_hip, lambda rgb: seg._loss(rgb, labels.cuda(), crit), YCrCb2RGB)
    assert abs(float(l_hip.detach()) - float(l_ref.detach())) / abs(float(l_ref.detach())) < 1e-4
    l_hip.backward()
    assert _compare_param_grads(fus, sd_f, tol=2e-3, sd32=sd_f32) == 80


def test_ssim_loss_matches_reference_formula(ag):
    """segmif_amd.losses.ssim (HIP separable blur, fwd + bwd) vs the reference's conv2d formulation
    (pytorch_ssim/__init__.py:19-43) in fp64 on the CPU."""
    from segmif_amd import losses
    a, b = rnd(2, 1, 37, 53, seed=50, lo=0, hi=1), rnd(2, 1, 37, 53, seed=51, lo=0, hi=1)
    ar = leaf(a, double=True)
    ref = losses.ssim(ar, b.double())
    ref.backward()
    ag_ = leaf(a, "cuda")
    got = losses.ssim(ag_, b.cuda())
    got.backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 1e-5
    assert err(ag_.grad, ar.grad) < 1e-4


def _golden_grad_check(module, g, tol, sd64=None):
    """HIP gradients vs the reference-autograd fixture.  The fixture is torch-CPU fp32; where it is itself
    further than tol/4 from the fp64 oracle (`sd64`, ill-conditioned tensors) the HIP result only has to be
    as close to fp64 as 10x the fixture's own deviation."""
    names = sorted(k[:-5] for k in g if k.endswith("|norm"))
    produced = {n for n, p in module.named_parameters() if p.grad is not None}
    assert produced == set(names)
    bad = []
    for n, p in module.named_parameters():
        if p.grad is None:
            continue
        got = p.grad.detach().double().cpu().reshape(-1)
        ref_head = torch.from_numpy(g[n + "|head"]).double()
        k = ref_head.numel()
        rms = float(g[n + "|norm"]) / max(got.numel(), 1) ** 0.5 + 1e-30
        scale = max(rms, float(ref_head.abs().max()))
        e = float((got[:k] - ref_head).abs().max()) / scale
        if e < tol:
            continue
        if sd64 is not None:
            truth = sd64[n].grad.double().reshape(-1)[:k]
            e_fix = float((ref_head - truth).abs().max()) / scale
            e_hip = float((got[:k] - truth).abs().max()) / scale
            if e_fix >= tol / 4 and e_hip <= 10 * e_fix:
                continue
            bad.append((n, e, e_hip, e_fix))
        else:
            bad.append((n, e))
    assert not bad, bad


def test_hip_gradients_match_reference_autograd_fixtures(ag, golden_dir):
    """HIP training path vs gradients recorded from the REAL reference's autograd (tests/golden/grads_*)."""
    import os
    import numpy as np
    from segmif_amd.core import Fusion_Network3_ac, Network3
    g = dict(np.load(os.path.join(golden_dir, "grads_seg_b1_64x96.npz")))
    net = Network3("mit_b1", 9, pretrained=None)
    dw.load_det_weights(net, seed=0)
    net = net.cuda().eval()
    x = dw.det_input("tr_x", (2, 3, 64, 96)).cuda()
    y = dw.det_labels("tr_y", (2, 64, 96), 9)
    y[0, 5:9, 7:30] = 255
    loss = net._loss(x, y.cuda(), torch.nn.CrossEntropyLoss(ignore_index=255))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    loss.backward()
    _golden_grad_check(net, g, tol=2e-3)

    g = dict(np.load(os.path.join(golden_dir, "grads_fusion_24x40.npz")))
    fus = Fusion_Network3_ac()
    dw.load_det_weights(fus, seed=0)
    fus = fus.cuda()
    ir, vis = dw.det_input("g_ir", (2, 1, 24, 40)).cuda(), dw.det_input("g_vis", (2, 3, 24, 40)).cuda()
    out = fus(ir, vis, torch.from_numpy(g["o1"]).cuda(), torch.from_numpy(g["o2"]).cuda())
    assert err(out, torch.from_numpy(g["out"])) < 1e-4
    (out * torch.from_numpy(g["cot"]).cuda()).sum().backward()
    sd64 = _oracle_params(so.fusion_shapes())
    ref64 = so.fusion_network3_ac(sd64, ir.cpu().double(), vis.cpu().double(), torch.from_numpy(g["o1"]).double(),
                                  torch.from_numpy(g["o2"]).double())
    (ref64 * torch.from_numpy(g["cot"]).double()).sum().backward()
    _golden_grad_check(fus, g, tol=2e-3, sd64=sd64)


def test_batchnorm_relu_train_mode(ag):
    rows, C = 5000, 256
    x, gm, bt, g = rnd(rows, C, seed=60, lo=-2, hi=3), rnd(C, seed=61, lo=0.5, hi=1.5), rnd(C, seed=62), rnd(rows, C, seed=63)
    xr, gr, br = leaf(x, double=True), leaf(gm, double=True), leaf(bt, double=True)
    yr = F.relu(F.batch_norm(xr.t()[None], None, None, gr, br, True, 0.1, 1e-5))[0].t()
    yr.backward(g.double())
    xg, gg, bg = leaf(x, "cuda"), leaf(gm, "cuda"), leaf(bt, "cuda")
    y, mean, var = ag.batchnorm_relu_train(xg, gg, bg, 1e-5)
    assert err(y, yr) < TOL
    assert err(mean, x.double().mean(0)) < 1e-6 and err(var, x.double().var(0, unbiased=False)) < 1e-5
    y.backward(g.cuda())
    assert err(xg.grad, xr.grad) < TOL and err(gg.grad, gr.grad) < TOL and err(bg.grad, br.grad) < TOL


def test_seg_net_train_mode_gradients_and_running_stats(ag):
    """Network3 in train() mode (BatchNorm on batch statistics — the regime of train_fusion and of the
    first 1000 iterations of train_seg) with the stochastic layers switched off
    (DropPath rate 0, Dropout2d p 0) so the comparison is deterministic."""
    from segmif_amd.core import Network3
    B, H, W = 2, 64, 96
    x = dw.det_input("tm_x", (B, 3, H, W))
    labels = dw.det_labels("tm_y", (B, H, W), 9)
    sd = _oracle_params(so.network3_shapes("mit_b1", 9))
    seg = so.network3_forward(sd, x.double(), "mit_b1", bn_training=True)
    ref_loss = F.cross_entropy(F.interpolate(seg, size=[H, W], mode="bilinear", align_corners=False), labels)
    ref_loss.backward()
    net = Network3("mit_b1", 9, pretrained=None)
    dw.load_det_weights(net, seed=0)
    net = net.cuda().train()
    net.denoise_net.encoder.reset_drop_path(0.0)
    net.denoise_net.decoder.dropout.p = 0.0
    loss = net._loss(x.cuda(), labels.cuda(), torch.nn.CrossEntropyLoss(ignore_index=255))
    assert abs(float(loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach())) < 1e-4
    loss.backward()
    assert _compare_param_grads(net, sd, tol=1e-3) > 150
    bn = net.denoise_net.decoder.linear_fuse.bn
    assert err(bn.running_mean, sd["denoise_net.decoder.linear_fuse.bn.running_mean"]) < 1e-5
    assert err(bn.running_var, sd["denoise_net.decoder.linear_fuse.bn.running_var"]) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    # stochastic layers: a DropPath'ed / Dropout2d'ed forward still runs and differs from the deterministic one
    net.denoise_net.encoder.reset_drop_path(0.5)
    net.denoise_net.decoder.dropout.p = 0.5
    torch.manual_seed(0)
    with torch.no_grad():
        a = net(x.cuda())[2]
        b = net(x.cuda())[2]
    assert torch.isfinite(a).all() and not torch.equal(a, b)
