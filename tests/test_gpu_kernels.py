"""GPU parity tests, kernel level: every C-ABI entry point of libsegmif_hip.so against an fp64
torch-CPU statement of the same aten op the reference calls.  Tolerance: the north-star bound is
1e-3 relative in fp32; the kernels are exact-fp32 (MFMA f32 / fmaf), so the gate here is 2e-5 of
the output's max magnitude."""
import itertools

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


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


def act_ref(y, act, slope=None):
    if act == 1:
        return F.relu(y)
    if act == 2:
        return torch.where(y >= 0, y, slope * y)
    if act == 3:
        return F.gelu(y)
    return y


TILES_ALL = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13]


@pytest.mark.parametrize("M,N,K", [(1000, 64, 64), (300, 512, 2048), (4099, 9, 256), (257, 128, 32), (77, 1, 64),
                                   (640, 256, 1024), (513, 320, 320)])
def test_igemm_dense_all_tiles(ops, M, N, K):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    ref_lin = x.double() @ w.double().t() + b.double()
    wt = ops.pack_weight(w.cuda())
    for tile in TILES_ALL:
        if tile in (1, 3, 5, 8, 11) and K % 32:
            continue
        for act, use_res in ((0, False), (1, True), (3, False)):
            ref = act_ref(ref_lin, act)
            if use_res:
                ref = ref + r.double()
            y = ops.linear(x.cuda(), wt, N, bias=b.cuda(), act=act, res=r.cuda() if use_res else None, tile=tile)
            assert err(y, ref) < TOL, (tile, act, use_res)


def test_igemm_dense_pitched_views_and_prelu(ops):
    M, K, N = 900, 64, 32
    buf = rnd(M, 224, seed=5).cuda()
    w, b = rnd(N, K, seed=6), rnd(N, seed=7)
    slope = torch.tensor([0.2], device="cuda")
    out = torch.full((M, 224), 7.0, device="cuda")
    ops.linear(buf[:, :K], ops.pack_weight(w.cuda()), N, bias=b.cuda(), act=2, prelu=slope, out=out[:, 64:96])
    ref = act_ref(buf[:, :K].cpu().double() @ w.double().t() + b.double(), 2, 0.2)
    assert err(out[:, 64:96], ref) < TOL
    assert float((out[:, :64] - 7).abs().max()) == 0 and float((out[:, 96:] - 7).abs().max()) == 0


def test_igemm_two_source_batched_weight(ops):
    B, n, C = 3, 700, 64
    p3, p1, x = rnd(B, n, 128, seed=8), rnd(B, n, 128, seed=9), rnd(B, n, C, seed=10)
    weff, bias = rnd(B, C, 128, seed=11), rnd(C, seed=12)
    y = ops.linear(p3.cuda()[..., :C], weff.cuda(), C, bias=bias.cuda(), res=x.cuda(), x2=p1.cuda()[..., C:],
                   batched_weight=True)
    a = torch.cat((p3[..., :C], p1[..., C:]), dim=-1).double()
    ref = x.double() + torch.einsum("bnk,bok->bno", a, weff.double()) + bias.double()
    assert err(y, ref) < TOL


def test_igemm_fused_layernorm_epilogue(ops):
    """out = LN_64(res + [a | b] @ W[z]^T + bias) in one launch (CrossPath end_proj + norm)."""
    B, n, C = 2, 1500, 64
    p3, p1, x = rnd(B, n, 128, seed=50), rnd(B, n, 128, seed=51), rnd(B, n, C, seed=52)
    weff, bias, gm, bt = rnd(B, C, 128, seed=53), rnd(C, seed=54), rnd(C, seed=55, lo=0.5, hi=1.5), rnd(C, seed=56)
    wide = torch.zeros(B, n, 224, device="cuda")
    ops.linear(p3.cuda()[..., :C], weff.cuda(), C, bias=bias.cuda(), res=x.cuda(), x2=p1.cuda()[..., C:],
               batched_weight=True, ln=(gm.cuda(), bt.cuda(), 1e-5), out=wide[..., :C])
    a = torch.cat((p3[..., :C], p1[..., C:]), dim=-1).double()
    pre = x.double() + torch.einsum("bnk,bok->bno", a, weff.double()) + bias.double()
    ref = F.layer_norm(pre, (C,), gm.double(), bt.double(), 1e-5)
    assert err(wide[..., :C], ref) < TOL
    assert float(wide[..., C:].abs().max()) == 0.0
    # plain dense with LN, rows not a multiple of the tile
    xx, w = rnd(777, 224, seed=57), rnd(64, 224, seed=58)
    y = ops.linear(xx.cuda(), ops.pack_weight(w.cuda()), 64, bias=bias.cuda(), ln=(gm.cuda(), bt.cuda(), 1e-6))
    ref = F.layer_norm(xx.double() @ w.double().t() + bias.double(), (64,), gm.double(), bt.double(), 1e-6)
    assert err(y, ref) < TOL


CONV_CASES = [
    # B, H, W, Cin, N, k, stride, pad, dil
    (2, 20, 28, 64, 32, 3, 1, 2, 2),  # DRDB dilated conv
    (2, 20, 28, 192, 32, 3, 1, 2, 2),
    (1, 17, 23, 64, 128, 3, 2, 1, 1),  # overlap patch embed, odd size
    (2, 37, 41, 3, 32, 7, 4, 3, 1),  # stage-1 patch embed (scalar gather path)
    (1, 9, 13, 128, 128, 4, 4, 0, 1),  # sr conv dropping the remainder
    (2, 16, 24, 1, 64, 3, 1, 1, 1),  # conv1_ir / conv1_vis
    (2, 16, 24, 32, 1, 3, 1, 1, 1),  # conv22
    (1, 33, 47, 128, 64, 3, 1, 1, 1),  # conv2
    (1, 8, 8, 320, 320, 2, 2, 0, 1),  # stage-3 sr conv
    (2, 20, 28, 32, 160, 3, 1, 2, 2),  # DRDB input gradient shape (N > 64: halo kernel with output-channel tiles)
    (1, 19, 37, 64, 96, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_igemm_conv(ops, case):
    B, H, W, Cin, N, k, s, p, d = case
    x, w, b = rnd(B, Cin, H, W, seed=13), rnd(N, Cin, k, k, seed=14), rnd(N, seed=15)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p, dilation=d).permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    wt = ops.pack_weight(w.cuda())
    tiles = [-1] if (Cin % 16) else list(TILES_ALL)
    if k == 3 and s == 1 and p == d and N <= 256 and Cin % 16 == 0:
        tiles += [9, 10]  # halo-tiled 3x3 variants (conv3x3.hip)
    for tile in tiles:
        if tile in (1, 3, 5, 8, 11) and Cin % 32:
            continue
        y = ops.conv2d(xh, wt, N, k, stride=s, pad=p, dil=d, bias=b.cuda(), tile=tile)
        assert err(y, ref) < TOL, tile


SPLIT_CASES = [c for c in CONV_CASES if c[5] == 3 and c[6] == 1 and c[3] % 16 == 0 and c[4] >= 16]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv3x3_split_bf16x6(ops, case):
    """Tile 14 (csrc/conv3x3_split.hip): bf16 MFMA with 3-way split operands must be as accurate as the
    exact-fp32 MFMA kernel — same fp64 reference, same tolerance, and an error within 2x of tile 10's."""
    B, H, W, Cin, N, k, s, p, d = case
    x, w, b = rnd(B, Cin, H, W, seed=13), rnd(N, Cin, k, k, seed=14), rnd(N, seed=15)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p, dilation=d).permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    y32 = ops.conv2d(xh, ops.pack_weight(w.cuda()), N, k, pad=p, dil=d, bias=b.cuda(), tile=10)
    y = ops.conv2d(xh, ops.pack_weight_split(w.cuda()), N, k, pad=p, dil=d, bias=b.cuda())
    e, e32 = err(y, ref), err(y32, ref)
    assert e < TOL and e <= 2.0 * e32 + 1e-7, (e, e32)


def test_conv3x3_split_epilogue_range_and_views(ops):
    """PReLU + residual + pitched in/out views on the split kernel, with operands spanning 1e-6..1e3
    (bf16 keeps the fp32 exponent range, so tiny and large values must both survive the split)."""
    B, H, W, Cin, N = 2, 21, 45, 96, 32
    g = torch.Generator().manual_seed(5)
    mag = 10.0 ** (torch.rand(B, H, W, Cin, generator=g) * 9 - 6)
    x = (torch.rand(B, H, W, Cin, generator=g) * 2 - 1) * mag
    w, b, res = rnd(N, Cin, 3, 3, seed=21) * 0.05, rnd(N, seed=22), rnd(B, H, W, N, seed=23)
    slope = torch.tensor([0.2])
    buf = torch.zeros(B, H, W, 224)
    buf[..., :Cin] = x
    buf = buf.cuda()
    ops.conv2d(buf[..., :Cin], ops.pack_weight_split(w.cuda()), N, 3, pad=2, dil=2, bias=b.cuda(), act=2,
               prelu=slope.cuda(), res=res.cuda(), out=buf[..., Cin:Cin + N])
    pre = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=2, dilation=2).permute(0, 2, 3, 1)
    ref = torch.where(pre >= 0, pre, 0.2 * pre) + res.double()
    assert err(buf[..., Cin:Cin + N], ref) < TOL
    assert float(buf[..., Cin + N:].abs().max()) == 0.0
    # elementwise: relative to the sum of |products| (the conditioning of each output), not the tensor max
    cond = F.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs(), padding=2, dilation=2).permute(0, 2, 3, 1)
    rel = ((buf[..., Cin:Cin + N].double().cpu() - ref).abs() / (cond + 1e-30)).max()
    assert float(rel) < 1e-6, float(rel)


def test_conv3x3_split_rejects_bad_geometry(ops):
    x = rnd(1, 8, 8, 32, seed=3).cuda()
    w = rnd(32, 32, 3, 3, seed=4).cuda()
    sw = ops.pack_weight_split(w)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, sw, 32, 3, stride=2, pad=1)  # tile 14 is stride-1 only
    with pytest.raises(RuntimeError):
        ops.pack_weight_split(rnd(32, 24, 3, 3, seed=5).cuda())  # Cin % 16


@pytest.mark.parametrize("M,N,K", [(4099, 320, 320), (2500, 1280, 320), (3000, 128, 256), (2048, 160, 64), (5000, 512, 2048)])
def test_gemm_split_bf16x6(ops, M, N, K):
    """csrc/gemm_split.hip: the encoder's Linear layers on the bf16 matrix pipe with 3-way split operands — same fp64
    yardstick and tolerance as the exact-fp32 MFMA GEMM and an error within 3x of it (the dropped cross terms are
    2^-23 of a product, one fp32 rounding; against heavy-tailed operands spanning 1e-4 .. 1e2 they show up at 2-2.5x the
    fp32 chain's own rounding, i.e. 5e-7 of the output scale); bias, GELU / ReLU, residual, ragged M / N, pitched views."""
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.rand(M, K, generator=g) * 2 - 1) * 10.0 ** (torch.rand(M, K, generator=g) * 6 - 4)
    w, b, r = rnd(N, K, seed=2) * 0.1, rnd(N, seed=3), rnd(M, N, seed=4)
    ref_lin = x.double() @ w.double().t() + b.double()
    packs = ops.pack_linear(w.cuda())
    assert packs[1] is not None
    wide = torch.zeros(M, K + 32, device="cuda")
    wide[:, :K] = x.cuda()
    xv = wide[:, :K]  # pitched rows view
    for act, use_res in ((0, False), (3, True), (1, False)):
        ref = act_ref(ref_lin, act)
        if use_res:
            ref = ref + r.double()
        y = ops.linear_auto(xv, packs, N, bias=b.cuda(), act=act, res=r.cuda() if use_res else None)
        y32 = ops.linear(xv, packs[0], N, bias=b.cuda(), act=act, res=r.cuda() if use_res else None)
        e, e32 = err(y, ref), err(y32, ref)
        assert e < TOL and e <= 3.0 * e32 + 1e-7, (act, e, e32)
    # elementwise, relative to each output's conditioning
    cond = x.double().abs() @ w.double().abs().t() + b.double().abs()
    y = ops.linear_auto(xv, packs, N, bias=b.cuda())
    assert float(((y.double().cpu() - ref_lin).abs() / cond).max()) < 1e-6
    # short problems keep the fp32 tiles (split-K); unsupported K falls back at pack time
    assert ops.pack_linear(rnd(64, 48, seed=9).cuda())[1] is None


@pytest.mark.parametrize("cin,H,W", [(1, 37, 50), (16, 24, 40)])
def test_conv2d_planes_copy_of_the_output(ops, cin, H, W):
    """ops.conv2d(planes=...): the implicit-GEMM epilogue also writes its (activated) result as planes chunks - byte for byte
    what segmif_planes_from_f32 makes of the fp32 output (conv1 of Fusion_Network3_ac feeds the first DRDB this way);
    ragged sizes, scalar-gather (Cin = 1) and vector modes, chunk offset."""
    B, N = 2, 64
    x = rnd(B, H, W, cin, seed=70).cuda()
    w = (rnd(N, cin, 3, 3, seed=71) * 0.3).cuda()
    b, slope = rnd(N, seed=72).cuda(), torch.tensor([0.2], device="cuda")
    pw = ops.pack_weight(w)
    for chunk0, chunks in ((0, 4), (2, 7)):
        pl = ops.Planes(B, H, W, chunks, "cuda")
        pl.data.zero_()
        y = ops.conv2d(x, pw, N, 3, pad=1, bias=b, act=ops.ACT_PRELU, prelu=slope, planes=pl, planes_chunk0=chunk0)
        y_plain = ops.conv2d(x, pw, N, 3, pad=1, bias=b, act=ops.ACT_PRELU, prelu=slope)
        assert err(y, y_plain.cpu()) < TOL  # (the plain call may take the halo-tiled kernel: another summation order)
        ref = ops.Planes(B, H, W, chunks, "cuda")
        ref.data.zero_()
        ref.load_f32(y, chunk0=chunk0)
        assert torch.equal(pl.data, ref.data)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pw, N, 3, pad=1, planes=ops.Planes(B, H, W, 3, "cuda"))  # 64 channels need four chunks


def test_conv2d_fused_layernorm_epilogue(ops):
    """OverlapPatchEmbed's conv (7x7, stride 4, Cin = 3 -> 64: mix_transformer.py:193-196) with bias + LayerNorm in
    the implicit GEMM's epilogue (scalar-gather mode, the 256 x 64 tile whose wave spans the row) against conv + LayerNorm in
    fp64; ragged output size; and a dense-mode case (Cin = 16)."""
    for cin, k, stride, H, W in ((3, 7, 4, 61, 83), (16, 3, 2, 40, 56)):
        B, N = 2, 64
        x = rnd(B, H, W, cin, seed=80).cuda()
        w = rnd(N, cin, k, k, seed=81) * 0.2
        b, g, t = rnd(N, seed=82), rnd(N, seed=83, lo=0.5, hi=1.5), rnd(N, seed=84)
        y = ops.conv2d(x, ops.pack_weight(w.cuda()), N, k, stride=stride, pad=k // 2, bias=b.cuda(), ln=(g.cuda(), t.cuda(), 1e-5))
        ref = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=k // 2)
        ref = torch.nn.functional.layer_norm(ref.permute(0, 2, 3, 1), (N,), g.double(), t.double(), 1e-5)
        assert tuple(y.shape) == tuple(ref.shape)
        assert err(y, ref) < TOL, (cin, err(y, ref))
    with pytest.raises(RuntimeError):
        ops.conv2d(x, ops.pack_weight(rnd(128, 16, 3, 3, seed=85).cuda()), 128, 3, pad=1, ln=(g.cuda(), t.cuda(), 1e-5))


def _sigma16():
    return torch.tensor([(j & 3) + 4 * (j >> 3) + 8 * ((j >> 2) & 1) for j in range(16)])


def _planes_decode(pl, chunk0, nch):
    """Planes buffer -> (B, H, W, 16 * nch) float64 plus the raw padded array (B, chunks, Hp, Wp, 3, 16)."""
    lib_hp = (pl.H + 7) // 8 * 8 + 4
    lib_wp = (pl.W + 31) // 32 * 32 + 4
    raw = pl.data.view(torch.bfloat16).view(pl.B, pl.chunks, lib_hp, lib_wp, 3, 16).double().cpu()
    val = raw.sum(dim=4)[:, chunk0:chunk0 + nch, 2:2 + pl.H, 2:2 + pl.W]  # (B, nch, H, W, 16 positions)
    out = torch.empty(pl.B, pl.H, pl.W, nch, 16, dtype=torch.float64)
    out[..., _sigma16()] = val.permute(0, 2, 3, 1, 4)  # position j holds channel sigma(j)
    return out.reshape(pl.B, pl.H, pl.W, nch * 16), raw


def test_planes_roundtrip_and_border(ops):
    """segmif_planes_from_f32: x = p0 + p1 + p2 to fp32 rounding for magnitudes 1e-6 .. 1e3, channel order
    sigma, zero border / round-up region untouched."""
    B, H, W, C = 2, 13, 45, 64
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(B, H, W, 224, generator=g) * 2 - 1) * 10.0 ** (torch.rand(B, H, W, 224, generator=g) * 9 - 6)
    pl = ops.Planes(B, H, W, 6, "cuda")
    pl.data.fill_(0x7f)  # poison, then clear the border the way the constructor does
    from segmif_amd import _lib
    _lib.check(_lib.load().segmif_planes_zero_border(pl.data.data_ptr(), B, H, W, 6, None), "zero_border")
    pl.load_f32(x.cuda()[..., :C], chunk0=1)
    torch.cuda.synchronize()
    got, raw = _planes_decode(pl, 1, C // 16)
    ref = x[..., :C].double()
    assert float(((got - ref).abs() / ref.abs()).max()) < 2.0 ** -23
    mask = torch.ones_like(raw, dtype=torch.bool)
    mask[:, :, 2:2 + H, 2:2 + W] = False
    assert float(raw[mask].abs().max()) == 0.0


PLANES_CASES = [  # B, H, W, Cin, dil
    (2, 20, 28, 64, 2), (1, 17, 45, 192, 2), (2, 8, 32, 96, 2), (1, 33, 70, 64, 1), (1, 9, 31, 128, 1),
]


@pytest.mark.parametrize("case", PLANES_CASES)
def test_conv3x3_planes_bf16x6(ops, case):
    """csrc/conv3x3_planes.hip against the fp64 conv and the exact-fp32 MFMA kernel (tile 10): fp32-class accuracy
    from pre-split operands, result delivered both as fp32 rows and as two more planes chunks."""
    B, H, W, Cin, d = case
    x, w, b = rnd(B, Cin, H, W, seed=13), rnd(32, Cin, 3, 3, seed=14), rnd(32, seed=15)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=d, dilation=d)).permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    y32 = ops.conv2d(xh, ops.pack_weight(w.cuda()), 32, 3, pad=d, dil=d, bias=b.cuda(), act=1, tile=10)
    chunks = Cin // 16 + 2
    pl = ops.Planes(B, H, W, chunks, "cuda").load_f32(xh)
    out = torch.full((B, H, W, 40), 7.0, device="cuda")
    ops.conv3x3_planes(pl, Cin, ops.pack_weight_planes(w.cuda()), dil=d, bias=b.cuda(), act=1, out_chunk0=Cin // 16,
                       out=out[..., :32])
    e, e32 = err(out[..., :32], ref), err(y32, ref)
    assert e < TOL and e <= 2.0 * e32 + 1e-7, (e, e32)
    assert float((out[..., 32:] - 7).abs().max()) == 0
    got, raw = _planes_decode(pl, Cin // 16, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < TOL
    mask = torch.ones_like(raw, dtype=torch.bool)
    mask[:, :, 2:2 + H, 2:2 + W] = False
    assert float(raw[mask].abs().max()) == 0.0  # nothing written outside the H x W interior


def test_conv3x3_planes_fused_tail_is_a_drdb(ops):
    """The fused tail: x + relu(W1 . [in | relu(conv(in))] + b1) — Dcov5 + cat + 1x1 + ReLU + residual of
    core/model_fusion.py:153-157 in one launch — with wide-range operands, elementwise against its conditioning."""
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
    pl = ops.Planes(B, H, W, Cin // 16, "cuda").load_f32(xc)
    out = torch.empty(B, H, W, 64, device="cuda")
    ops.conv3x3_planes(pl, Cin, ops.pack_weight_planes(w.cuda()), dil=2, bias=b.cuda(), act=1,
                       tail=(ops.pack_weight_planes(w1.cuda()), b1.cuda(), xc[..., :64], out, 1))
    assert err(out, ref) < TOL
    cond = F.conv2d(torch.cat((xd.abs(), mid.abs()), dim=1), w1.double().abs()[:, :, None, None]).permute(0, 2, 3, 1) \
        + x[..., :64].double().abs()
    rel = ((out.double().cpu() - ref).abs() / (cond + 1e-30)).max()
    assert float(rel) < 2e-6, float(rel)


def test_conv3x3_planes_rejects_bad_arguments(ops):
    pl = ops.Planes(1, 8, 32, 6, "cuda")
    w = ops.pack_weight_planes(rnd(32, 64, 3, 3, seed=4).cuda())
    with pytest.raises(RuntimeError):
        ops.conv3x3_planes(pl, 64, w, dil=3, out_chunk0=4)  # dilation 1 | 2 only
    with pytest.raises(RuntimeError):
        ops.conv3x3_planes(pl, 64, w, dil=2, out_chunk0=3)  # would overwrite its own input chunks
    with pytest.raises(RuntimeError):
        ops.conv3x3_planes(pl, 64, w, dil=2, out_chunk0=5)  # two output chunks do not fit
    with pytest.raises(RuntimeError):
        ops.pack_weight_planes(rnd(48, 64, 3, 3, seed=5).cuda())  # N % 32


def _gram_from_partials(part):
    p = part.double().cpu().sum(dim=1)  # (B, 3072)
    B = p.shape[0]
    G = torch.zeros(B, 64, 64, dtype=torch.float64)
    G[:, :32, :32] = p[:, :1024].view(B, 32, 32)
    G[:, :32, 32:] = p[:, 1024:2048].view(B, 32, 32)
    G[:, 32:, :32] = p[:, 1024:2048].view(B, 32, 32).transpose(1, 2)
    G[:, 32:, 32:] = p[:, 2048:].view(B, 32, 32)
    return G


@pytest.mark.parametrize("N", [1500, 70001])
def test_crosspath_gram_and_fold(ops, N):
    """csrc/crosspath.hip: G = sum relu(Wx+b) relu(Wx+b)^T against fp64 (N not a multiple of the 32-pixel tile), then
    softmax((Wk G Wv^T) scale) folded into end_proj against the fp64 statement of core/model_fusion.py:281-286 and
    against round 1's kv-projection kernels on the same data."""
    B = 2
    x = rnd(B, N, 224, seed=61).cuda()[..., 32:96]  # pitched rows view
    w, b = rnd(64, 64, seed=62) * 0.3, rnd(64, seed=63) * 0.1
    wkv, wend = rnd(128, 64, seed=64) * 0.05, rnd(64, 128, seed=65)
    y = F.relu(x.cpu().double() @ w.double().t() + b.double())
    part = ops.crosspath_gram(x, w.cuda(), b.cuda())
    G = _gram_from_partials(part)
    Gref = torch.einsum("bni,bnj->bij", y, y)
    assert float((G - Gref).abs().max() / Gref.abs().max()) < 1e-6
    scale = 8 ** -0.5
    weff = torch.zeros(B, 64, 128, device="cuda")
    ops.crosspath_fold(part, wkv.cuda(), wend.cuda(), weff, wofs=64, kofs=64, scale=scale)
    kv = y @ wkv.double().t()
    k, v = kv[..., :64].view(B, N, 8, 8), kv[..., 64:].view(B, N, 8, 8)
    ctx = torch.softmax(torch.einsum("bnhi,bnhj->bhij", k, v) * scale, dim=-2)
    ref = torch.einsum("bhij,nhj->bnhi", ctx, wend.double()[:, 64:].view(64, 8, 8)).reshape(B, 64, 64)
    assert err(weff[..., 64:], ref) < TOL
    assert float(weff[..., :64].abs().max()) == 0.0
    # round 1's path on the same tokens: projection GEMM, fused kv reduction, fold
    yf = ops.linear(x, ops.pack_weight(w.cuda()), 64, bias=b.cuda(), act=1)
    weff1 = torch.zeros(B, 64, 128, device="cuda")
    ops.linattn_fold(ops.linattn_kvpartial(yf, wkv.cuda()), wend.cuda(), weff1, wofs=64, kofs=64, scale=scale)
    assert err(weff[..., 64:], weff1[..., 64:].cpu()) < TOL


def test_crosspath_tail(ops):
    """out = LN(x_i + Weff_b [relu(W3 x_3 + b3) | relu(Wi x_i + bi)] + e) against fp64, pitched views, a token count
    that is not a multiple of 32, plus the planes copy of the result (next DRDB's input)."""
    B, H, W = 2, 13, 37
    N = H * W
    x3 = rnd(B, N, 64, seed=71).cuda()
    xi = rnd(B, N, 224, seed=72).cuda()[..., :64]
    w3, b3, wi, bi = rnd(64, 64, seed=73) * 0.3, rnd(64, seed=74) * 0.1, rnd(64, 64, seed=75) * 0.3, rnd(64, seed=76) * 0.1
    weff, bend = rnd(B, 64, 128, seed=77) * 0.2, rnd(64, seed=78) * 0.1
    gm, bt = rnd(64, seed=79, lo=0.5, hi=1.5), rnd(64, seed=80)
    t = torch.cat((F.relu(x3.cpu().double() @ w3.double().t() + b3.double()),
                   F.relu(xi.cpu().double() @ wi.double().t() + bi.double())), dim=-1)
    pre = xi.cpu().double() + torch.einsum("bnk,bok->bno", t, weff.double()) + bend.double()
    ref = F.layer_norm(pre, (64,), gm.double(), bt.double(), 1e-5)
    wide = torch.zeros(B, N, 128, device="cuda")
    pl = ops.Planes(B, H, W, 6, "cuda")
    out = ops.crosspath_tail(x3, xi, w3.cuda(), b3.cuda(), wi.cuda(), bi.cuda(), weff.cuda(), bend.cuda(),
                             (gm.cuda(), bt.cuda(), 1e-5), out=wide[..., :64], planes=pl, hw=(H, W))
    assert err(out, ref) < TOL
    assert float(wide[..., 64:].abs().max()) == 0.0
    got, raw = _planes_decode(pl, 0, 4)
    assert float((got.view(B, N, 64) - out.double().cpu()).abs().max()) < 2.0 ** -22 * float(ref.abs().max())
    mask = torch.ones_like(raw, dtype=torch.bool)
    mask[:, :4, 2:2 + H, 2:2 + W] = False
    assert float(raw[:, :4][mask[:, :4]].abs().max()) == 0.0  # border untouched (chunks 4, 5 are uninitialised scratch)


def test_drdb_concat_in_place(ops):
    """A conv reading the first Cin channels of a 224-wide buffer and writing its 32 channels in place."""
    B, H, W = 2, 14, 18
    buf = rnd(B, H, W, 224, seed=16).cuda()
    before = buf.clone()
    w, b = rnd(32, 96, 3, 3, seed=17), rnd(32, seed=18)
    ops.conv2d(buf[..., :96], ops.pack_weight(w.cuda()), 32, 3, pad=2, dil=2, bias=b.cuda(), act=1,
               out=buf[..., 96:128])
    xin = before[..., :96].cpu().double().permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(xin, w.double(), b.double(), padding=2, dilation=2)).permute(0, 2, 3, 1)
    assert err(buf[..., 96:128], ref) < TOL
    assert torch.equal(buf[..., :96], before[..., :96]) and torch.equal(buf[..., 128:], before[..., 128:])


@pytest.mark.parametrize("C", [32, 64, 128, 160, 256, 320, 512, 1024])
def test_layernorm(ops, C):
    rows = 1003
    x, g, b = rnd(rows, C, seed=19, lo=-3, hi=5), rnd(C, seed=20), rnd(C, seed=21)
    for eps in (1e-6, 1e-5):
        y = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), eps)
        assert err(y, F.layer_norm(x.double(), (C,), g.double(), b.double(), eps)) < TOL
    wide = torch.zeros(rows, C + 32, device="cuda")
    ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5, out=wide[:, :C])
    assert err(wide[:, :C], F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)) < TOL


@pytest.mark.parametrize("B,H,W,C", [(2, 9, 13, 128), (1, 30, 40, 1280), (1, 1, 1, 32), (2, 8, 12, 512), (1, 17, 5, 64),
                                     (2, 35, 17, 64), (1, 19, 16, 256), (1, 40, 21, 32)])  # W >= 16: the two-column kernel, odd widths
def test_dwconv_gelu(ops, B, H, W, C):
    x, w, b = rnd(B, H * W, C, seed=22, lo=-2, hi=2), rnd(C, 1, 3, 3, seed=23), rnd(C, seed=24)
    img = x.double().transpose(1, 2).reshape(B, C, H, W)
    ref = F.gelu(F.conv2d(img, w.double(), b.double(), padding=1, groups=C)).flatten(2).transpose(1, 2)
    y = ops.dwconv3x3_gelu(x.cuda(), ops.pack_dw_weight(w.cuda()), b.cuda(), H, W)
    assert err(y, ref) < TOL


@pytest.mark.parametrize("B,IH,IW,OH,OW,C", [(2, 16, 24, 64, 96, 64), (1, 15, 20, 120, 160, 256), (1, 5, 7, 18, 26, 32),
                                             (2, 18, 26, 72, 104, 9), (1, 30, 40, 30, 40, 8)])
def test_bilinear(ops, B, IH, IW, OH, OW, C):
    x = rnd(B, IH, IW, C, seed=25)
    ref = F.interpolate(x.double().permute(0, 3, 1, 2), size=[OH, OW], mode="bilinear", align_corners=False)
    y = ops.bilinear(x.cuda(), OH, OW)
    assert err(y, ref.permute(0, 2, 3, 1)) < TOL
    if C % 4 == 0:
        wide = torch.zeros(B, OH, OW, 3 * C, device="cuda")
        ops.bilinear(x.cuda(), OH, OW, out=wide[..., C:2 * C])
        assert err(wide[..., C:2 * C], ref.permute(0, 2, 3, 1)) < TOL


@pytest.mark.parametrize("B,heads,N,Nk,hd", [(2, 2, 96, 6, 64), (1, 1, 19200, 300, 64), (2, 5, 1200, 300, 64),
                                             (1, 8, 300, 300, 64), (1, 2, 1024, 64, 32), (1, 1, 130, 1, 64),
                                             (1, 1, 4096, 1024, 64), (1, 5, 35, 35, 32)])
@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "fp32"])
def test_sr_attention(ops, B, heads, N, Nk, hd, mode):
    """csrc/attention_split.hip (head_dim 64: f16 MFMA x 3 split products inside a guarded scope, bf16 MFMA x 6 outside) and
    csrc/attention.hip (fp32 MFMA) against fp64 softmax attention; head_dim 32 runs the fp32 kernel in every mode."""
    C = heads * hd
    q, kv = rnd(B, N, C, seed=26, lo=-2, hi=2), rnd(B, Nk, 2 * C, seed=27, lo=-2, hi=2)
    scale = hd ** -0.5
    qh = q.double().reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kvh = kv.double().reshape(B, Nk, 2, heads, hd)
    k, v = kvh[:, :, 0].permute(0, 2, 1, 3), kvh[:, :, 1].permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ k.transpose(-2, -1) * scale, -1) @ v).transpose(1, 2).reshape(B, N, C)
    prev = ops.attention_mode()
    ops.set_attention_mode(mode)
    guard = ops.Planes16Guard("cuda", B) if mode == "f16x3" else None
    pg = ops.install_guard(guard)
    try:
        y = ops.sr_attention(q.cuda(), kv.cuda(), heads, scale)
        y32 = y
        if mode == "f16x3":
            ops.install_guard(None)
            ops.set_attention_mode("fp32")
            y32 = ops.sr_attention(q.cuda(), kv.cuda(), heads, scale)
    finally:
        ops.install_guard(pg)
        ops.set_attention_mode(prev)
    assert err(y, ref) < TOL
    if mode == "f16x3" and hd == 64 and N >= 1024:
        assert err(y, ref) <= 3.0 * err(y32, ref) + 1e-7, (err(y, ref), err(y32, ref))
        m = guard.maxima()
        want = (q.abs().amax(dim=(1, 2)) * (scale * 1.4426950408889634)).half().float()
        assert m.shape == (1, B) and guard.ok() and all(abs(float(m[0, i]) - float(want[i])) <= 2.0 ** -10 * float(want[i]) for i in range(B))


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "fp32"])
def test_sr_attention_large_logits(ops, mode):
    """Online-softmax rescale path: one key dominates late in the sequence; f16x3: also a value tile 1000x larger than the
    others (the running sums change their power-of-two unit between tiles) and operands spanning 1e-3 .. 1e2."""
    B, heads, N, Nk, hd = 1, 1, (2048 if mode == "f16x3" else 64), 100, 64
    q, kv = rnd(B, N, hd, seed=28), rnd(B, Nk, 2 * hd, seed=29)
    kv[:, 77, :hd] = 40.0 * q[0, 5]  # spike: row 5's max jumps at tile 2
    if mode == "f16x3":
        kv[:, 32:64, hd:] *= 1000.0
        kv[:, :32, :hd] *= 1.0e-3
        kv[:, 64:, hd:] *= 10.0 ** rnd(1, 36, 1, seed=30, lo=-3, hi=2)
    ref = torch.softmax(q.double() @ kv[..., :hd].double().transpose(-2, -1) * 0.125, -1) @ kv[..., hd:].double()
    prev = ops.attention_mode()
    ops.set_attention_mode(mode)
    pg = ops.install_guard(ops.Planes16Guard("cuda", B) if mode == "f16x3" else None)
    try:
        y = ops.sr_attention(q.cuda(), kv.cuda(), heads, 0.125)
        ops.install_guard(None)
        ops.set_attention_mode("fp32")
        y32 = ops.sr_attention(q.cuda(), kv.cuda(), heads, 0.125)
    finally:
        ops.install_guard(pg)
        ops.set_attention_mode(prev)
    assert err(y, ref) < TOL and err(y, ref) <= 3.0 * err(y32, ref) + 1e-7, (err(y, ref), err(y32, ref))


@pytest.mark.parametrize("B,N", [(2, 3000), (1, 1024), (3, 37), (1, 70000)])
def test_linear_attention_context_and_fold(ops, B, N):
    heads, d, C = 8, 8, 64
    kv = rnd(B, N, 2 * C, seed=30)
    wend = rnd(C, 2 * C, seed=31)
    scale = d ** -0.5
    part = ops.linattn_partial(kv.cuda(), heads)
    kvh = kv.double().reshape(B, N, 2, heads, d)
    k, v = kvh[:, :, 0].permute(0, 2, 1, 3), kvh[:, :, 1].permute(0, 2, 1, 3)
    raw = k.transpose(-2, -1) @ v  # (B, h, d, d)
    assert err(part.sum(1).reshape(B, heads, d, d), raw) < 1e-6
    ctx = torch.softmax(raw * scale, dim=-2)
    weff = torch.zeros(B, C, 2 * C, device="cuda")
    ops.linattn_fold(part, wend.cuda(), weff, wofs=C, kofs=C, scale=scale)
    ops.linattn_fold(part, wend.cuda(), weff, wofs=0, kofs=0, scale=scale)
    # Weff[b][n][kofs + h*d + i] = sum_j ctx[b,h,i,j] * Wend[n][wofs + h*d + j]
    for ofs in (0, C):
        wpart = wend.double()[:, ofs:ofs + C].reshape(C, heads, d)
        ref = torch.einsum("bhij,nhj->bnhi", ctx, wpart).reshape(B, C, C)
        assert err(weff[:, :, ofs:ofs + C], ref) < TOL


@pytest.mark.parametrize("B,N", [(2, 3000), (1, 1024), (2, 129), (1, 70001)])
def test_fused_kv_projection_partial(ops, B, N):
    """segmif_linattn_kvpartial_f32 == kv Linear + segmif_linattn_partial_f32 (same per-head sums)."""
    C = 64
    p = rnd(B, N, 2 * C, seed=40, lo=0.0, hi=1.0)  # y is a ReLU output: non-negative
    wkv = rnd(2 * C, C, seed=41, lo=-0.3, hi=0.3)
    y = p.cuda()[..., :C]  # pitched view, as CrossPath passes it
    part = ops.linattn_kvpartial(y, wkv.cuda())
    kv = p[..., :C].double() @ wkv.double().t()
    kvh = kv.reshape(B, N, 2, 8, 8)
    k, v = kvh[:, :, 0].permute(0, 2, 1, 3), kvh[:, :, 1].permute(0, 2, 1, 3)
    raw = k.transpose(-2, -1) @ v
    assert err(part.sum(1).reshape(B, 8, 8, 8), raw) < 2e-6


def test_pointwise_and_layout(ops):
    import segmif_oracle as so
    x = rnd(2, 3, 11, 17, seed=32, lo=0, hi=1)
    y = ops.seg_normalize(x.cuda())
    mean = torch.tensor(so.SEG_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(so.SEG_STD).view(1, 3, 1, 1)
    assert err(y, ((x.double() * 255 - mean) / std).permute(0, 2, 3, 1)) < TOL
    t = rnd(2, 37, 5, 9, seed=33)
    assert torch.equal(ops.to_nhwc(t.cuda()).cpu(), t.permute(0, 2, 3, 1).contiguous())
    cl = t.cuda().contiguous(memory_format=torch.channels_last)
    assert ops.to_nhwc(cl).data_ptr() == cl.data_ptr()  # zero-copy for channels-last storage
    assert torch.equal(ops.to_nchw_contiguous(t.permute(0, 2, 3, 1).contiguous().cuda()).cpu(), t)
    vis, yf = rnd(2, 3, 11, 17, seed=34, lo=0, hi=1), rnd(2, 1, 11, 17, seed=35, lo=-0.2, hi=1.3)
    ycc = so.rgb2ycrcb(vis.double())
    ref = so.ycrcb2rgb(torch.cat((yf.double(), ycc[:, 1:2], ycc[:, 2:3]), 1)).clamp(0, 1)
    assert err(ops.fuse_ycrcb(vis.cuda(), yf.cuda()), ref) < TOL
    logits = rnd(3, 7, 9, 9, seed=36)
    assert torch.equal(ops.argmax_nhwc(logits.cuda()).cpu().long(), logits.argmax(-1))


def test_bad_arguments_fail_loudly(ops):
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 64), torch.zeros(8, 64).cuda(), 8)  # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 64).cuda(), torch.zeros(8, 48).cuda(), 8)  # wrong packed shape
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 30).cuda(), torch.ones(30).cuda(), torch.zeros(30).cuda(), 1e-5)  # C % 4


def test_confusion_matrix_and_quantize_on_device(ops):
    """SURVEY §8(f) N3 kernels against the fixture recorded from sklearn + the reference's util.py, and
    against the oracle's restatement of the uint8 write-out."""
    import os

    import numpy as np

    import detweights as dw
    import segmif_oracle as so
    from segmif_amd.utils import metrics
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
    pred, label = torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["label"]).cuda()
    conf = metrics.confusion_matrix(pred, label, 9)
    assert np.array_equal(conf.cpu().numpy(), g["conf"])
    metrics.confusion_matrix(pred, label, 9, out=conf)  # accumulates like conf_total += conf
    assert np.array_equal(conf.cpu().numpy(), 2 * g["conf"])
    lab2 = label.clone()
    lab2.view(-1)[:100] = 255  # outside the label set: ignored, as sklearn does with labels=[0..8]
    ref = so.confusion(lab2.cpu().numpy(), g["pred"], 9)
    assert np.array_equal(metrics.confusion_matrix(pred, lab2, 9).cpu().numpy(), ref)
    _, _, iou = metrics.compute_results(metrics.confusion_matrix(pred, label, 9))
    assert np.allclose(np.nan_to_num(iou), np.nan_to_num(g["iou"]), atol=1e-15)
    # large, ragged count
    big_p = dw.det_labels("cm_p", (1000003,), 9).to(torch.int32).cuda()
    big_l = dw.det_labels("cm_l", (1000003,), 9).cuda()
    assert np.array_equal(metrics.confusion_matrix(big_p, big_l, 9).cpu().numpy(),
                          so.confusion(big_l.cpu().numpy(), big_p.cpu().numpy(), 9))
    # uint8 write-out
    x = (dw.det_input("quant_gpu", (2, 3, 37, 53)) * 0.7 + 0.1)
    q = metrics.quantize_fused(x.cuda())
    assert q.dtype == torch.uint8 and np.array_equal(q.cpu().numpy(), so.quantize_fused_u8(x.numpy()))
    assert not metrics.quantize_fused(torch.full((1, 3, 5, 7), 0.5).cuda()).any()


def test_upsample_sum_act(ops):
    """segmif_upsum_act_nhwc_f32: act(base + bias + sum of bilinear resizes) against F.interpolate, pitched
    base / out, one to three sources, odd sizes."""
    B, OH, OW, C = 2, 30, 41, 64
    srcs = [rnd(B, 15, 21, C, seed=81), rnd(B, 8, 10, C, seed=82), rnd(B, 4, 5, C, seed=83)]
    base, bias = rnd(B, OH, OW, C, seed=84), rnd(C, seed=85)

    def up(t):
        return F.interpolate(t.double().permute(0, 3, 1, 2), size=(OH, OW), mode="bilinear",
                             align_corners=False).permute(0, 2, 3, 1)

    for n in (1, 2, 3):
        ref = F.relu(base.double() + bias.double() + sum(up(t) for t in srcs[:n]))
        wide_b = torch.zeros(B, OH, OW, 96)
        wide_b[..., 16:16 + C] = base
        wide_b = wide_b.cuda()
        out = torch.full((B, OH, OW, 80), 3.0, device="cuda")
        ops.upsum_act(wide_b[..., 16:16 + C], [t.cuda() for t in srcs[:n]], OH, OW, bias=bias.cuda(), act=1,
                      out=out[..., :C])
        assert err(out[..., :C], ref) < TOL, n
        assert float((out[..., C:] - 3).abs().max()) == 0
    y = ops.upsum_act(None, [srcs[0].cuda()], OH, OW)
    assert err(y, up(srcs[0])) < TOL
    with pytest.raises(RuntimeError):
        ops.upsum_act(None, [], OH, OW)
