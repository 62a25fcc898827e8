"""GPU tests of the f16x3 planes path (csrc/conv3x3_planes.hip, F16 = true; ops mode 'planes16'): half-pair activations,
three MFMA products per fp32 product.  Same yardsticks as the bf16x6 tests of test_gpu_kernels.py - the fp64 convolution
and the exact-fp32 MFMA kernel - plus the range guard: tensors outside the half's exponent range must be noticed and the
forward repeated on the bf16x6 kernels."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detweights as dw

pytestmark = pytest.mark.gpu

TOL = 1e-3
TIGHT = 1e-4


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from segmif_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def _sigma16():
    return torch.tensor([(j & 3) + 4 * (j >> 3) + 8 * ((j >> 2) & 1) for j in range(16)])


def _decode(pl, chunk0, nch):
    """f16x3 planes buffer -> (B, H, W, 16 * nch) float64 (x = p0 + 2^-11 p1) plus the raw padded array."""
    hp, wp = (pl.H + 7) // 8 * 8 + 4, (pl.W + 31) // 32 * 32 + 4
    raw = pl.data.view(torch.float16).view(pl.B, pl.chunks, hp, wp, 2, 16).double().cpu()
    val = (raw[..., 0, :] + raw[..., 1, :] / 2048.0)[:, chunk0:chunk0 + nch, 2:2 + pl.H, 2:2 + pl.W]
    out = torch.empty(pl.B, pl.H, pl.W, nch, 16, dtype=torch.float64)
    out[..., _sigma16()] = val.permute(0, 2, 3, 1, 4)
    return out.reshape(pl.B, pl.H, pl.W, nch * 16), raw


def test_planes16_roundtrip_border_and_amax(ops):
    """segmif_planes16_from_f32: x = p0 + 2^-11 p1 to within 2^-23 relative for 2^-12 <= |x| < 65504 and to 2^-35 absolute
    below (half subnormals), channel order sigma, border untouched, and the launch's max |x| in its guard slot."""
    B, H, W, C = 2, 13, 45, 64
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(B, H, W, 224, generator=g) * 2 - 1) * 10.0 ** (torch.rand(B, H, W, 224, generator=g) * 10 - 6)
    x = x.clamp(-6.0e4, 6.0e4)
    guard = ops.Planes16Guard("cuda")
    pl = ops.Planes(B, H, W, 6, "cuda", guard)
    assert pl.data.numel() == B * 6 * 20 * 68 * 64
    pl.data.fill_(0x7f)
    from segmif_amd import _lib
    _lib.check(_lib.load().segmif_planes16_zero_border(pl.data.data_ptr(), B, H, W, 6, None), "zero_border")
    pl.load_f32(x.cuda()[..., :C], chunk0=1)
    torch.cuda.synchronize()
    got, raw = _decode(pl, 1, C // 16)
    ref = x[..., :C].double()
    d = (got - ref).abs()
    big = ref.abs() >= 2.0 ** -12
    assert float((d[big] / ref.abs()[big]).max()) < 2.0 ** -23
    assert float(d[~big].max()) <= 2.0 ** -35
    mask = torch.ones_like(raw, dtype=torch.bool)
    mask[:, :, 2:2 + H, 2:2 + W] = False
    assert float(raw[mask].abs().max()) == 0.0
    m = guard.maxima()
    assert m.numel() == 1 and float(m[0]) == float(x[..., :C].abs().max().half())  # (slots hold the HIGH half's pattern)
    assert guard.ok()


PLANES_CASES = [  # B, H, W, Cin, dil
    (2, 20, 28, 64, 2), (1, 17, 45, 192, 2), (2, 8, 32, 96, 2), (1, 33, 70, 64, 1), (1, 9, 31, 128, 1),
]


@pytest.mark.parametrize("case", PLANES_CASES)
def test_conv3x3_planes_f16x3(ops, case):
    """The f16x3 kernel against the fp64 conv and the exact-fp32 MFMA kernel (tile 10): fp32-class accuracy (the bound the
    bf16x6 kernel is held to), result as fp32 rows and as two more half-pair chunks, the output's max in the guard."""
    B, H, W, Cin, d = case
    x, w, b = rnd(B, Cin, H, W, seed=13), rnd(32, Cin, 3, 3, seed=14), rnd(32, seed=15)
    w = w * (10.0 ** rnd(32, 1, 1, 1, seed=16, lo=-3, hi=1))  # rows of very different magnitude: the per-row scale
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=d, dilation=d)).permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    y32 = ops.conv2d(xh, ops.pack_weight(w.cuda()), 32, 3, pad=d, dil=d, bias=b.cuda(), act=1, tile=10)
    chunks = Cin // 16 + 2
    guard = ops.Planes16Guard("cuda")
    pl = ops.Planes(B, H, W, chunks, "cuda", guard).load_f32(xh)
    out = torch.full((B, H, W, 40), 7.0, device="cuda")
    ops.conv3x3_planes(pl, Cin, ops.pack_weight_planes16(w.cuda()), dil=d, bias=b.cuda(), act=1, out_chunk0=Cin // 16,
                       out=out[..., :32])
    e, e32 = err(out[..., :32], ref), err(y32, ref)
    assert e < TOL and e <= 2.0 * e32 + 1e-7, (e, e32)
    # elementwise against the conditioning of each output (rows differ by four orders of magnitude)
    cond = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=d, dilation=d).permute(0, 2, 3, 1)
    rel = float(((out[..., :32].double().cpu() - ref).abs() / cond).max())
    rel32 = float(((y32.double().cpu() - ref).abs() / cond).max())
    assert rel <= 2.0 * rel32 + 1e-7, (rel, rel32)
    assert float((out[..., 32:] - 7).abs().max()) == 0
    got, raw = _decode(pl, Cin // 16, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < TOL
    mask = torch.ones_like(raw, dtype=torch.bool)
    mask[:, :, 2:2 + H, 2:2 + W] = False
    assert float(raw[mask].abs().max()) == 0.0
    m = guard.maxima()
    assert m.numel() == 2 and abs(float(m[1]) - float(ref.abs().max())) <= 2.0 ** -11 * float(ref.abs().max())
    assert guard.ok()


def test_conv3x3_planes_f16x3_fused_tail_is_a_drdb(ops):
    """The fused DRDB tail on half pairs, wide-range operands (1e-4 .. 1e2), elementwise against its conditioning."""
    B, H, W, Cin = 2, 19, 37, 192
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(B, H, W, Cin, generator=g) * 2 - 1) * 10.0 ** (torch.rand(B, H, W, Cin, generator=g) * 6 - 4)
    w, b = rnd(32, Cin, 3, 3, seed=31) * 0.05, rnd(32, seed=32)
    w1, b1 = rnd(64, Cin + 32, seed=33) * 0.1, rnd(64, seed=34)
    xd = x.double().permute(0, 3, 1, 2)
    mid = F.relu(F.conv2d(xd, w.double(), b.double(), padding=2, dilation=2))
    cat = torch.cat((xd, mid), dim=1)
    pre = F.conv2d(cat, w1.double()[:, :, None, None], b1.double())
    ref = (xd[:, :64] + F.relu(pre)).permute(0, 2, 3, 1)
    xc = x.cuda()
    guard = ops.Planes16Guard("cuda")
    pl = ops.Planes(B, H, W, Cin // 16, "cuda", guard).load_f32(xc)
    out = torch.empty(B, H, W, 64, device="cuda")
    ops.conv3x3_planes(pl, Cin, ops.pack_weight_planes16(w.cuda()), dil=2, bias=b.cuda(), act=1,
                       tail=(ops.pack_weight_planes16(w1.cuda()), b1.cuda(), xc[..., :64], out, 1))
    assert err(out, ref) < TOL
    cond = F.conv2d(torch.cat((xd.abs(), mid.abs()), dim=1), w1.double().abs()[:, :, None, None]).permute(0, 2, 3, 1) \
        + x[..., :64].double().abs()
    rel = ((out.double().cpu() - ref).abs() / (cond + 1e-30)).max()
    assert float(rel) < 2e-6, float(rel)
    assert guard.ok() and guard.maxima().numel() == 2


def test_planes16_guard_sees_overflow_and_vanishing_tensors(ops):
    """Values past the half's range (inf in the planes) and tensors whose maximum is below 2^-13 trip the guard; the
    kernel must honour half subnormals (absolute error of a vanishing tensor stays at the 2^-35 level of its inputs)."""
    B, H, W, Cin = 1, 16, 32, 64
    x, w, b = rnd(B, H, W, Cin, seed=41), rnd(32, Cin, 3, 3, seed=42) * 0.1, torch.zeros(32)
    wt = ops.pack_weight_planes16(w.cuda())
    for scale, fine in ((1.0, True), (1.0e5, False), (1.0e-5, False)):
        guard = ops.Planes16Guard("cuda")
        pl = ops.Planes(B, H, W, 6, "cuda", guard).load_f32((x * scale).cuda())
        out = torch.empty(B, H, W, 32, device="cuda")
        ops.conv3x3_planes(pl, Cin, wt, dil=2, bias=b.cuda(), act=0, out_chunk0=4, out=out)
        assert guard.ok() == fine, (scale, guard.maxima())
        if scale < 1:
            ref = F.conv2d((x * scale).double().permute(0, 3, 1, 2), w.double(), padding=2, dilation=2).permute(0, 2, 3, 1)
            cond = F.conv2d(torch.full_like(x, 2.0 ** -35).double().permute(0, 3, 1, 2), w.double().abs(), padding=2,
                            dilation=2).permute(0, 2, 3, 1)
            assert float(((out.double().cpu() - ref).abs() / cond).max()) < 2.0, "half subnormals were flushed"


@pytest.mark.parametrize("cin,H,W", [(1, 37, 50), (16, 24, 40)])
def test_conv2d_f16x3_planes_copy_of_the_output(ops, cin, H, W):
    """ops.conv2d(planes=<f16x3 buffer>): the implicit-GEMM epilogue writes half pairs - byte for byte what
    segmif_planes16_from_f32 makes of the fp32 output - and folds max |output| into its guard slot."""
    B, N = 2, 64
    x = rnd(B, H, W, cin, seed=70).cuda()
    w = (rnd(N, cin, 3, 3, seed=71) * 0.3).cuda()
    b, slope = rnd(N, seed=72).cuda(), torch.tensor([0.2], device="cuda")
    pw = ops.pack_weight(w)
    for chunk0, chunks in ((0, 4), (2, 7)):
        guard = ops.Planes16Guard("cuda")
        pl = ops.Planes(B, H, W, chunks, "cuda", guard)
        pl.data.zero_()
        y = ops.conv2d(x, pw, N, 3, pad=1, bias=b, act=ops.ACT_PRELU, prelu=slope, planes=pl, planes_chunk0=chunk0)
        ref = ops.Planes(B, H, W, chunks, "cuda", ops.Planes16Guard("cuda"))
        ref.data.zero_()
        ref.load_f32(y, chunk0=chunk0)
        assert torch.equal(pl.data, ref.data)
        m = guard.maxima()
        assert m.numel() == 1 and float(m[0]) == float(y.abs().max().half())


def test_crosspath_tail_f16x3_planes_copy(ops):
    """The CrossPath tail's optional planes copy in the f16x3 format: byte for byte segmif_planes16_from_f32 of its fp32
    output (a token count that is not a multiple of 32), guard slot = max |out|, fp32 output as in the bf16 kernel."""
    B, H, W = 2, 13, 37
    N = H * W
    x3 = rnd(B, N, 64, seed=71).cuda()
    xi = rnd(B, N, 224, seed=72).cuda()[..., :64]
    w3, b3, wi, bi = rnd(64, 64, seed=73) * 0.3, rnd(64, seed=74) * 0.1, rnd(64, 64, seed=75) * 0.3, rnd(64, seed=76) * 0.1
    weff, bend = rnd(B, 64, 128, seed=77) * 0.2, rnd(64, seed=78) * 0.1
    gm, bt = rnd(64, seed=79, lo=0.5, hi=1.5), rnd(64, seed=80)
    args = (x3, xi, w3.cuda(), b3.cuda(), wi.cuda(), bi.cuda(), weff.cuda(), bend.cuda(), (gm.cuda(), bt.cuda(), 1e-5))
    plain = ops.crosspath_tail(*args)
    guard = ops.Planes16Guard("cuda")
    pl = ops.Planes(B, H, W, 6, "cuda", guard)
    pl.data.zero_()
    out = ops.crosspath_tail(*args, planes=pl, hw=(H, W))
    assert torch.equal(out, plain)
    ref = ops.Planes(B, H, W, 6, "cuda", ops.Planes16Guard("cuda"))
    ref.data.zero_()
    ref.load_f32(out.view(B, H, W, 64), chunk0=0)
    assert torch.equal(pl.data, ref.data)
    m = guard.maxima()
    assert m.numel() == 1 and float(m[0]) == float(out.abs().max().half())
    # planes_only (r4): the same planes, no fp32 tensor written at all
    pl2 = ops.Planes(B, H, W, 6, "cuda", ops.Planes16Guard("cuda"))
    pl2.data.zero_()
    assert ops.crosspath_tail(*args, planes=pl2, hw=(H, W), planes_only=True) is None
    assert torch.equal(pl2.data, ref.data)
    with pytest.raises(RuntimeError):
        ops.crosspath_tail(*args, planes_only=True)


def _act_ref(y, act):
    return {0: y, 1: F.relu(y), 3: F.gelu(y)}[act]


@pytest.mark.parametrize("M,N,K", [(4099, 320, 320), (2500, 1280, 320), (3000, 128, 256), (2048, 160, 64), (5000, 512, 2048)])
def test_gemm_split_f16x3(ops, M, N, K):
    """csrc/gemm_split.hip, F16: ops.linear_auto inside a guarded scope takes the half-pair kernel - the bf16x6 test's
    yardsticks (fp64; within 3x of the exact-fp32 MFMA GEMM's error; 1e-6 of each output's conditioning) with operands
    spanning 1e-4 .. 1e2 and weight rows of very different magnitude, max |A| in the guard slot."""
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.rand(M, K, generator=g) * 2 - 1) * 10.0 ** (torch.rand(M, K, generator=g) * 6 - 4)
    w, b, r = rnd(N, K, seed=2) * 0.1 * 10.0 ** rnd(N, 1, seed=5, lo=-2, hi=1), rnd(N, seed=3), rnd(M, N, seed=4)
    ref_lin = x.double() @ w.double().t() + b.double()
    packs = ops.pack_linear(w.cuda(), half=True)
    assert packs[1] is not None and packs[1].half is not None
    wide = torch.zeros(M, K + 32, device="cuda")
    wide[:, :K] = x.cuda()
    xv = wide[:, :K]
    assert ops.active_guard() is None
    guard = ops.Planes16Guard("cuda")
    ops.install_guard(guard)
    try:
        for act, use_res in ((0, False), (3, True), (1, False)):
            ref = _act_ref(ref_lin, act)
            if use_res:
                ref = ref + r.double()
            y = ops.linear_auto(xv, packs, N, bias=b.cuda(), act=act, res=r.cuda() if use_res else None)
            y32 = ops.linear(xv, packs[0], N, bias=b.cuda(), act=act, res=r.cuda() if use_res else None)
            e, e32 = err(y, ref), err(y32, ref)
            assert e < TOL and e <= 3.0 * e32 + 1e-7, (act, e, e32)
        cond = x.double().abs() @ w.double().abs().t() + b.double().abs()
        y = ops.linear_auto(xv, packs, N, bias=b.cuda())
        assert float(((y.double().cpu() - ref_lin).abs() / cond).max()) < 1e-6
    finally:
        ops.install_guard(None)
    m = guard.maxima()
    assert m.numel() == 4 and all(float(v) == float(x.abs().max().half()) for v in m)  # four launches of the half-pair kernel
    assert guard.ok()


def test_guarded_scope_repeats_on_bf16x6(ops):
    """ops.run_guarded: a GEMM whose input passes 65504 trips the guard and the scope's result is the bf16x6 kernel's, bit
    for bit; in range, the scope returns the f16x3 result and counts no fallback."""
    M, N, K = 4096, 256, 128
    x, w, b = rnd(M, K, seed=21).cuda(), (rnd(N, K, seed=22) * 0.1).cuda(), rnd(N, seed=23).cuda()
    packs = ops.pack_linear(w, half=True)
    plain = ops.linear_auto(x, packs, N, bias=b)                      # no scope: bf16x6
    before = ops.range_fallbacks()
    inside = ops.run_guarded(lambda: ops.linear_auto(x, packs, N, bias=b), "cuda")
    assert ops.range_fallbacks() == before and not torch.equal(inside, plain)
    assert float((inside - plain).abs().max()) < 1e-5 * float(plain.abs().max())
    big = x * 1.0e6
    plain_big = ops.linear_auto(big, packs, N, bias=b)
    calls = []

    def body():
        calls.append(ops.active_guard() is not None)
        return ops.run_guarded(lambda: ops.linear_auto(big, packs, N, bias=b), "cuda")  # a nested scope joins the outer one

    got = ops.run_guarded(body, "cuda")
    assert calls == [True, False] and ops.range_fallbacks() == before + 1
    assert torch.equal(got, plain_big)


def _build(cls, *a, **k):
    m = cls(*a, **k)
    dw.load_det_weights(m, seed=0)
    return m.cuda().eval()


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _rel(got, ref):
    got = torch.as_tensor(got).detach().double().cpu()
    ref = torch.as_tensor(ref).double()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def test_fusion_net_on_f16x3_planes_vs_reference_and_fallback(ops, golden_dir):
    """'planes16' mode against the reference's records (DRDB block, whole fusion net of the mit_b1 pair) and against the
    exact-fp32 mode; a DRDB fed activations of 1e6 must notice, fall back to bf16 triples and still be right."""
    import segmif_amd.core as core
    fus = _build(core.Fusion_Network3_ac)
    net = _build(core.Network3, "mit_b1", 9, pretrained=None)
    g = _load(golden_dir, "fusion_blocks.npz")
    gp = _load(golden_dir, "pair_b1_64x96.npz")
    ir, vis, mask = (torch.from_numpy(gp[k]).cuda() for k in ("ir", "vis", "mask"))
    outs = {}
    prev = ops.conv3x3_mode()
    try:
        with torch.no_grad():
            out0, out1 = net.denoise_net.encoder.forward_fusion(mask)
            for mode in ("planes16", "fp32"):
                ops.set_conv3x3_mode(mode)
                y = fus.DRDB1(torch.from_numpy(g["drdb_x"]).cuda())
                assert _rel(y, g["drdb_y"]) < TIGHT, mode
                yf = fus(ir, vis, out0, out1)
                assert _rel(yf, gp["y_fused"]) < 5 * TIGHT, mode
                outs[mode] = (y, yf)
            assert fus.planes16_fallbacks == 0
            assert _rel(outs["planes16"][0], outs["fp32"][0].cpu()) < 2e-6
            # (r6: 7.1e-6 since the CrossPath tail and conv1 run on f16x3 operands under planes16 as well; < 5e-6 before)
            assert _rel(outs["planes16"][1], outs["fp32"][1].cpu()) < 1.5e-5
            ops.set_conv3x3_mode("planes16")
            big = torch.from_numpy(g["drdb_x"]).cuda() * 1.0e6
            yb = fus.DRDB1(big)
            ops.set_conv3x3_mode("fp32")
            assert _rel(yb, fus.DRDB1(big).cpu()) < 2e-6
    finally:
        ops.set_conv3x3_mode(prev)


def test_full_size_b3_pair_on_f16x3_planes_vs_reference_checksum(ops, golden_dir):
    """The headline pair (mit_b3, 480x640) with the DRDBs on f16x3: sampled fused values and labels against the reference's
    record, no fallback taken."""
    import segmif_amd.core as core
    from segmif_amd.pipeline import PairForward
    g = _load(golden_dir, "pair_b3_480x640_checksum.npz")
    fus = _build(core.Fusion_Network3_ac)
    net = _build(core.Network3, "mit_b3", 9, pretrained=None)
    H, W = 480, 640
    ir = dw.det_input("b3_ir", (1, 1, H, W)).cuda()
    vis = dw.det_input("b3_vis", (1, 3, H, W)).cuda()
    mask = dw.det_input("b3_mask", (1, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    prev = ops.set_conv3x3_mode("planes16")
    before = ops.range_fallbacks()
    try:
        with torch.no_grad():
            fused, labels = PairForward(net, fus)(ir, vis, mask)
    finally:
        ops.set_conv3x3_mode(prev)
    assert ops.range_fallbacks() == before
    got = fused.contiguous().reshape(-1)[torch.from_numpy(g["fused_idx"]).cuda()].cpu()
    scale = max(abs(g["fused_stats"][2]), abs(g["fused_stats"][3]))
    e = float((got - torch.from_numpy(g["fused_val"])).abs().max()) / scale
    assert e < 5 * TIGHT, e
    ref_labels = torch.from_numpy(g["labels"]).long()
    stable = torch.from_numpy(g["margin_f16"].astype(np.float32)) > 1e-3
    assert torch.equal(labels.cpu().long().reshape(ref_labels.shape)[stable], ref_labels[stable])


def test_pair_forward_graph_replay_carries_its_own_guard(ops, golden_dir):
    """PairForward.capture with the f16x3 defaults: the graph owns a range guard (cleared by a memset node, filled by the
    recorded kernels), replay() returns what the eager guarded forward returns - bit for bit, also for new inputs copied into
    the static buffers - and an input that drives an activation past 65504 comes back as the eager bf16x6 result."""
    import segmif_amd.core as core
    from segmif_amd.pipeline import PairForward
    if not ops.f16x3_enabled():
        pytest.skip("f16x3 modes are switched off")
    fus = _build(core.Fusion_Network3_ac)
    net = _build(core.Network3, "mit_b1", 9, pretrained=None)
    gp = _load(golden_dir, "pair_b1_64x96.npz")
    ir, vis, mask = (torch.from_numpy(gp[k]).cuda() for k in ("ir", "vis", "mask"))

    def same(a, b):
        return bool(((a == b) | (a.isnan() & b.isnan())).all())

    pipe = PairForward(net, fus)
    with torch.no_grad():
        fused_e, labels_e = pipe.eager(ir, vis, mask)
    assert _rel(fused_e, gp["fused"]) < 5 * TIGHT
    pipe.capture(ir, vis, mask)
    assert pipe._graph_guard is not None and pipe._graph_guard.used > 0
    fused_g, labels_g = pipe(ir, vis, mask)
    assert same(fused_g, fused_e) and torch.equal(labels_g, labels_e)
    ir2, vis2, mask2 = (t.flip(-1).contiguous() for t in (ir, vis, mask))
    with torch.no_grad():
        fused_e2, labels_e2 = pipe.eager(ir2, vis2, mask2)
    fused_g2, labels_g2 = pipe(ir2, vis2, mask2)
    assert same(fused_g2, fused_e2) and torch.equal(labels_g2, labels_e2) and not same(fused_g2, fused_e)
    before = ops.range_fallbacks()
    big = ir * 1.0e6
    fused_gb, _ = pipe(big, vis, mask)
    fused_gb = fused_gb.clone()
    assert ops.range_fallbacks() == before + 1
    with torch.no_grad():
        fused_eb, _ = pipe.eager(big, vis, mask)
    assert ops.range_fallbacks() == before + 2 and same(fused_gb, fused_eb)


@pytest.mark.parametrize("case", [(2, 32, 40, 64, 2), (1, 16, 70, 192, 2), (1, 48, 33, 96, 1), (3, 16, 32, 128, 2)])
def test_conv3x3_planes_f16x3_four_subtiles(ops, case):
    """conv3x3_planes_kernel<.., SUB = 4> (16 x 32 patches, four sub-tiles per wave: the default for the plain f16x3 conv on
    heights that are whole 16-row patches): the same yardsticks as test_conv3x3_planes_f16x3, whose ragged heights keep
    covering the two-sub-tile kernel."""
    B, H, W, Cin, d = case
    x, w, b = rnd(B, Cin, H, W, seed=13), rnd(32, Cin, 3, 3, seed=14), rnd(32, seed=15)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=d, dilation=d)).permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    y32 = ops.conv2d(xh, ops.pack_weight(w.cuda()), 32, 3, pad=d, dil=d, bias=b.cuda(), act=1, tile=10)
    guard = ops.Planes16Guard("cuda")
    pl = ops.Planes(B, H, W, Cin // 16 + 2, "cuda", guard).load_f32(xh)
    out = torch.full((B, H, W, 40), 7.0, device="cuda")
    ops.conv3x3_planes(pl, Cin, ops.pack_weight_planes16(w.cuda()), dil=d, bias=b.cuda(), act=1, out_chunk0=Cin // 16,
                       out=out[..., :32])
    e, e32 = err(out[..., :32], ref), err(y32, ref)
    assert e < TOL and e <= 2.0 * e32 + 1e-7, (e, e32)
    assert float((out[..., 32:] - 7).abs().max()) == 0
    got, raw = _decode(pl, Cin // 16, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < TOL
    mask = torch.ones_like(raw, dtype=torch.bool)
    mask[:, :, 2:2 + H, 2:2 + W] = False
    assert float(raw[mask].abs().max()) == 0.0
    assert guard.ok()


@pytest.mark.parametrize("case", [(2, 32, 40, 64, 2, 1), (1, 16, 70, 160, 2, 1), (1, 48, 33, 96, 1, 2), (3, 16, 32, 128, 1, 0),
                                  (2, 16, 64, 32, 2, 2)])
def test_lean_plain_conv_equals_the_general_instantiation(ops, case):
    """(r6) conv3x3_planes_kernel<.., LEAN> - the instantiation an inference forward takes when no fp32 copy is asked for: halved
    constants, relu(t) = t' + |t'|, PReLU / none from the same halved value, plane stores deferred into the next LOAD phase - against
    the GENERAL instantiation (asking for the fp32 copy selects it) on the same planes: the written chunks must hold the same VALUES
    element for element (only the sign of a zero may differ), the border stays zero, the range slot reports the same maximum.
    Heights are whole 16-row patches (the four-sub-tile kernels); the last case has fewer chunks (2) than sub-tiles."""
    B, H, W, Cin, d, act = case
    x, w, b = rnd(B, Cin, H, W, seed=21), rnd(32, Cin, 3, 3, seed=22) * 0.3, rnd(32, seed=23)
    slope = torch.tensor([0.25], device="cuda") if act == 2 else None
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = ops.pack_weight_planes16(w.cuda())
    res = []
    for lean in (False, True):
        guard = ops.Planes16Guard("cuda")
        pl = ops.Planes(B, H, W, Cin // 16 + 2, "cuda", guard).load_f32(xh)
        kw = {} if lean else {"out": torch.empty(B, H, W, 32, device="cuda")}
        ops.conv3x3_planes(pl, Cin, wp, dil=d, bias=b.cuda(), act=act, prelu=slope, out_chunk0=Cin // 16, **kw)
        torch.cuda.synchronize()
        got, raw = _decode(pl, Cin // 16, 2)
        assert guard.ok()
        res.append((got, raw, guard.maxima().clone()))
    (g0, r0, m0), (g1, r1, m1) = res
    assert torch.equal(g0, g1)                      # (+0 == -0)
    assert torch.equal(r0.abs(), r1.abs())          # every half of the padded image, border included
    assert torch.equal(m0, m1)
    pre = F.conv2d(x.double(), w.double(), b.double(), padding=d, dilation=d)
    ref = (F.relu(pre) if act == 1 else torch.where(pre >= 0, pre, 0.25 * pre) if act == 2 else pre).permute(0, 2, 3, 1)
    assert float((g1 - ref).abs().max() / ref.abs().max()) < TOL


def test_lean_fused_tail_equals_the_general_instantiation(ops):
    """(r6) the LEAN fused tail (ReLU / ReLU, residual from the input planes, no planes out: 1x1 weights of the conv's own channels
    resident in LDS, residual pieces requested a sub-tile ahead, v_fma_mix_f32 reconstruction) against the general fused kernel, which
    asking for the conv's planes copy selects: out1 bit for bit (same products in the same order; only a zero's sign may differ)."""
    B, H, W, Cin = 2, 24, 70, 192
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, H, W, Cin, generator=g) * 2 - 1) * 10.0 ** (torch.rand(B, H, W, Cin, generator=g) * 4 - 3)
    w, b = rnd(32, Cin, 3, 3, seed=41) * 0.05, rnd(32, seed=42)
    w1, b1 = rnd(64, Cin + 32, seed=43) * 0.1, rnd(64, seed=44)
    wp, w1p = ops.pack_weight_planes16(w.cuda()), ops.pack_weight_planes16(w1.cuda())
    outs = []
    for lean in (False, True):
        guard = ops.Planes16Guard("cuda")
        pl = ops.Planes(B, H, W, Cin // 16 + 2, "cuda", guard).load_f32(x.cuda())
        out = torch.full((B, H, W, 64), 7.0, device="cuda")
        ops.conv3x3_planes(pl, Cin, wp, dil=2, bias=b.cuda(), act=1, out_chunk0=None if lean else Cin // 16,
                           tail=(w1p, b1.cuda(), None, out, 1, True))
        torch.cuda.synchronize()
        assert guard.ok()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    xd = x.double().permute(0, 3, 1, 2)
    mid = F.relu(F.conv2d(xd, w.double(), b.double(), padding=2, dilation=2))
    pre = F.conv2d(torch.cat((xd, mid), dim=1), w1.double()[:, :, None, None], b1.double())
    ref = (xd[:, :64] + F.relu(pre)).permute(0, 2, 3, 1)
    assert err(outs[1].cuda(), ref) < TOL


@pytest.mark.parametrize("act", [0, 1, 2])
def test_lean_conv_reports_overflow_nan_and_vanishing_outputs(ops, act):
    """(r6) the LEAN plain conv's compare-free activation (relu(t) = t' + |t'|, slope min(t, 0) = slope (t' - |t'|)) under the range
    contract: an output past the half's range (inf after the split), a NaN carried in from the input, an output of -inf (where
    t' + |t'| is NaN instead of the select's 0) and a vanishing tensor must all trip the guard; a healthy tensor must not."""
    B, H, W, Cin = 1, 16, 32, 64
    x, w, b = rnd(B, H, W, Cin, seed=51), rnd(32, Cin, 3, 3, seed=52) * 0.1, rnd(32, seed=53) * 0.1
    wt = ops.pack_weight_planes16(w.cuda())
    slope = torch.tensor([0.25], device="cuda") if act == 2 else None
    cases = {"healthy": (x, True), "overflow": (x * 3.0e4, False), "vanishing": (x * 1.0e-6, act == 1 and False)}
    xn = x.clone(); xn[0, 5, 7, 3] = float("nan")
    cases["nan"] = (xn, False)
    xi = x.clone(); xi[0, 9, 11, :] = -6.0e4          # (within the half's range on the way in; the sums leave it on the negative side)
    cases["negative overflow"] = (xi * 1.0, None)       # (trips unless the ReLU maps the whole region to exact zeros: checked below)
    for name, (xin, fine) in cases.items():
        guard = ops.Planes16Guard("cuda")
        pl = ops.Planes(B, H, W, 6, "cuda", guard).load_f32(xin.cuda())
        bias = b if name != "vanishing" else torch.zeros(32)
        ops.conv3x3_planes(pl, Cin, wt, dil=2, bias=bias.cuda(), act=act, prelu=slope, out_chunk0=4)
        torch.cuda.synchronize()
        if fine is None:
            got, _ = _decode(pl, 4, 2)
            assert guard.ok() == bool(torch.isfinite(got).all() and got.abs().max() < 65504), name
        else:
            assert guard.ok() == fine, (name, guard.maxima())
