"""GPU parity tests, module level: segmif_amd.core (HIP kernels through the C ABI) against
 (a) the golden vectors recorded from the real reference (tests/golden), and
 (b) the CPU oracle on fresh seeded inputs,
within the north-star tolerance of 1e-3 relative (observed ~1e-5) and with exact argmax labels
wherever the reference's own top-2 margin is above the stability threshold."""
import json
import os

import numpy as np
import pytest
import torch

import detweights as dw
import segmif_oracle as so

pytestmark = pytest.mark.gpu

TOL = 1e-3  # BASELINE.json north_star: 1e-3 rel fp32
TIGHT = 1e-4  # what the exact-fp32 kernels actually deliver; regressions show up here first


@pytest.fixture(scope="module")
def core():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import segmif_amd.core as c
    return c


def load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def rel(got, ref):
    got = torch.as_tensor(got).detach().double().cpu()
    ref = torch.as_tensor(ref).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def rel_elementwise(got, ref, floor=1e-2):
    """(r6) ELEMENT-WISE relative error |got - ref| / |ref| over the elements with |ref| above `floor` x the
    tensor's range -> (max, 99.9th percentile, RMS-relative over the whole tensor).  rel() above is a global max-norm, blind to
    damage confined to small-magnitude regions; this is the figure "1e-3 rel" reads as when taken element by element (below the
    floor a relative error is a statement about the rounding of numbers near zero, not about parity)."""
    got = torch.as_tensor(got).detach().double().cpu().reshape(-1)
    ref = torch.as_tensor(ref).double().reshape(-1)
    big = ref.abs() > floor * ref.abs().max()
    ew = ((got - ref).abs() / ref.abs().clamp_min(1e-300))[big]
    if ew.numel() > 4_000_000:  # (torch.quantile's input limit; a strided subsample keeps the percentile honest)
        ewq = ew[:: ew.numel() // 4_000_000 + 1]
    else:
        ewq = ew
    p999 = float(torch.quantile(ewq, 0.999)) if ewq.numel() else 0.0
    rms = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-300))
    return (float(ew.max()) if ew.numel() else 0.0), p999, rms


def build(core, cls, *a, **k):
    m = cls(*a, **k)
    dw.load_det_weights(m, seed=0)
    return m.cuda().eval()


@pytest.fixture(scope="module")
def fus(core):
    return build(core, core.Fusion_Network3_ac)


@pytest.fixture(scope="module")
def net_b1(core):
    return build(core, core.Network3, "mit_b1", 9, pretrained=None)


def test_state_dict_keys(core, golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for bb in ("mit_b0", "mit_b1", "mit_b2", "mit_b3", "mit_b4", "mit_b5"):
        n = core.Network3(bb, 9, pretrained=None)
        assert {k: list(v.shape) for k, v in n.state_dict().items()} == keys["Network3:" + bb], bb
    f = core.Fusion_Network3_ac()
    assert {k: list(v.shape) for k, v in f.state_dict().items()} == keys["Fusion_Network3_ac"]


def test_mit_b0_ragged_vs_reference(core, golden_dir):
    g = load(golden_dir, "mit_b0_72x104.npz")
    net = build(core, core.Network3, "mit_b0", 9, pretrained=None)
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        feats = net.denoise_net.encoder(x)
        o0, o1 = net.denoise_net.encoder.forward_fusion(x)
        a, b, seg = net(x)
    assert a is x and b is x  # Network3.forward returns its input object twice (ref :1088)
    assert [tuple(f.shape) for f in feats] == [(1, 32, 18, 26), (1, 64, 9, 13), (1, 160, 5, 7), (1, 256, 3, 4)]
    for i, f in enumerate(feats):
        assert rel(f, g[f"f{i + 1}"]) < TIGHT, i
    assert rel(o0[:, :, 1::5, 2::7], g["fus0_sample"]) < TIGHT
    assert rel(o1[:, :, 1::5, 2::7], g["fus1_sample"]) < TIGHT
    assert tuple(seg.shape) == (1, 9, 18, 26) and rel(seg, g["seg"]) < TIGHT


@pytest.mark.parametrize("bb", ["mit_b2", "mit_b4"])
def test_mit_b2_b4_ragged_vs_reference(core, golden_dir, bb):
    """The two constructors no other record exercises (core/mix_transformer.py:399-423), ragged 72x104: HIP encoder features,
    forward_fusion and logits vs the reference's."""
    g = load(golden_dir, bb + "_72x104.npz")
    net = build(core, core.Network3, bb, 9, pretrained=None)
    x = dw.det_input(bb + "_72x104", (1, 3, 72, 104)).cuda()
    with torch.no_grad():
        feats = net.denoise_net.encoder(x)
        o0, o1 = net.denoise_net.encoder.forward_fusion(x)
        _, _, seg = net(x)
    assert [tuple(f.shape) for f in feats] == [(1, 64, 18, 26), (1, 128, 9, 13), (1, 320, 5, 7), (1, 512, 3, 4)]
    for i, f in enumerate(feats):
        assert rel(f, g[f"f{i + 1}"]) < TIGHT, i
    assert rel(o0[:, :, 1::5, 2::7], g["fus0_sample"]) < TIGHT
    assert rel(o1[:, :, 1::5, 2::7], g["fus1_sample"]) < TIGHT
    assert tuple(seg.shape) == (1, 9, 18, 26) and rel(seg, g["seg"]) < TIGHT


def test_mit_blocks_vs_reference(net_b1, golden_dir):
    g = load(golden_dir, "mit_blocks.npz")
    enc = net_b1.denoise_net.encoder
    t = torch.from_numpy(g["tokens"]).cuda()
    blk = enc.block2[1]
    with torch.no_grad():
        assert rel(blk.attn(t, 8, 12), g["attn"]) < TIGHT
        assert rel(blk.mlp(t, 8, 12), g["ffn"]) < TIGHT
        keep = t.clone()
        assert rel(blk(t, 8, 12), g["block"]) < TIGHT
        assert torch.equal(t, keep)  # public Block.forward must not modify its input
        pt, ph, pw = enc.patch_embed2(torch.from_numpy(g["pe_x"]).cuda())
        assert (ph, pw) == (9, 12) and rel(pt, g["pe_tokens"]) < TIGHT
        assert rel(enc.block4[0].attn(torch.from_numpy(g["tokens4"]).cuda(), 2, 3), g["attn4"]) < TIGHT


def test_fusion_blocks_vs_reference(fus, golden_dir):
    g = load(golden_dir, "fusion_blocks.npz")
    with torch.no_grad():
        y = fus.DRDB1(torch.from_numpy(g["drdb_x"]).cuda())
        assert rel(y, g["drdb_y"]) < TIGHT
        o1, o2 = fus.ffm(*(torch.from_numpy(g[k]).cuda() for k in ("ffm_x1", "ffm_x2", "ffm_seg")))
        assert rel(o1, g["ffm_o1"]) < TIGHT and rel(o2, g["ffm_o2"]) < TIGHT


def test_crosspath_in_both_modes(net_b1, fus, golden_dir):
    """FeatureFusionModule / the whole fusion net on the Gram-matrix CrossPath kernels (default) and on round 1's
    GEMM + kv-reduction path: both against the reference fixtures, and against each other."""
    from segmif_amd import ops
    g = load(golden_dir, "fusion_blocks.npz")
    gp = load(golden_dir, "pair_b1_64x96.npz")
    ir, vis, mask = (torch.from_numpy(gp[k]).cuda() for k in ("ir", "vis", "mask"))
    outs = {}
    prev = ops.crosspath_mode()
    try:
        with torch.no_grad():
            out0, out1 = net_b1.denoise_net.encoder.forward_fusion(mask)
            for mode in ("gram", "gemm"):
                ops.set_crosspath_mode(mode)
                o1, o2 = fus.ffm(*(torch.from_numpy(g[k]).cuda() for k in ("ffm_x1", "ffm_x2", "ffm_seg")))
                assert rel(o1, g["ffm_o1"]) < TIGHT and rel(o2, g["ffm_o2"]) < TIGHT, mode
                yf = fus(ir, vis, out0, out1)
                assert rel(yf, gp["y_fused"]) < 5 * TIGHT, mode
                outs[mode] = (o1, yf)
    finally:
        ops.set_crosspath_mode(prev)
    assert rel(outs["gram"][0], outs["gemm"][0].cpu()) < 5e-6
    # (the whole net: two formulations of the context sums behind two softmaxes in series - 6.6e-6 observed on the f16x3 features
    # forward_fusion now returns when called on its own (r5); each is within 5 x TIGHT of the reference above)
    assert rel(outs["gram"][1], outs["gemm"][1].cpu()) < 2e-5


def test_fusion_net_in_all_conv3x3_modes(net_b1, fus, golden_dir):
    """The 3x3 convs have four modes (planes16, default: the DRDBs and closing convs on pre-split half pairs, three f16
    products per MAC, range-guarded; planes: the same on bf16 triples, six products; bf16x6: split operands made on the
    fly; fp32: exact-fp32 MFMA): all must meet the same reference fixtures, and agree with each other far inside the
    parity tolerance."""
    from segmif_amd import ops
    g = load(golden_dir, "fusion_blocks.npz")
    gp = load(golden_dir, "pair_b1_64x96.npz")
    ir, vis, mask = (torch.from_numpy(gp[k]).cuda() for k in ("ir", "vis", "mask"))
    outs = {}
    prev = ops.conv3x3_mode()
    try:
        with torch.no_grad():
            out0, out1 = net_b1.denoise_net.encoder.forward_fusion(mask)
            for mode in ("planes16", "planes", "bf16x6", "fp32"):
                ops.set_conv3x3_mode(mode)
                y = fus.DRDB1(torch.from_numpy(g["drdb_x"]).cuda())
                assert rel(y, g["drdb_y"]) < TIGHT, mode
                yf = fus(ir, vis, out0, out1)
                assert rel(yf, gp["y_fused"]) < 5 * TIGHT, mode
                outs[mode] = (y, yf)
    finally:
        ops.set_conv3x3_mode(prev)
    for mode in ("planes16", "planes", "bf16x6"):
        assert rel(outs[mode][0], outs["fp32"][0].cpu()) < 2e-6, mode
        # (r6: under planes16 the CrossPath tail and conv1 run on f16x3 operands too - 22-23 significand bits in two more
        # contractions - and the whole net lands 7.1e-6 from the exact-fp32 convs instead of < 5e-6; the bound that matters,
        # 5 x TIGHT against the reference record, is asserted per mode above)
        assert rel(outs[mode][1], outs["fp32"][1].cpu()) < (1.5e-5 if mode == "planes16" else 5e-6), mode


def test_conv3_conv4_commute_with_the_resize(net_b1, fus, golden_dir):
    """forward_from_features (1x1 conv3 / conv4 at feature resolution, then bilinear)
    is the same function as forward on the up-sampled features — checked against the reference fixture
    and against the textbook order, through the pair pipeline too."""
    from segmif_amd.pipeline import PairForward
    gp = load(golden_dir, "pair_b1_64x96.npz")
    ir, vis, mask = (torch.from_numpy(gp[k]).cuda() for k in ("ir", "vis", "mask"))
    enc = net_b1.denoise_net.encoder
    with torch.no_grad():
        a = fus(ir, vis, *enc.forward_fusion(mask))
        b = fus.forward_from_features(ir, vis, *enc.forward_fusion_features(mask))
        assert rel(b, gp["y_fused"]) < 5 * TIGHT
        assert rel(b, a.cpu()) < 1e-5
        fused_c, labels_c = PairForward(net_b1, fus, commute_resize=True)(ir, vis, mask)
        fused_t, labels_t = PairForward(net_b1, fus, commute_resize=False)(ir, vis, mask)
    assert rel(fused_c, gp["fused"]) < 5 * TIGHT and rel(fused_c, fused_t.cpu()) < 1e-5
    stable = torch.from_numpy(gp["margin"]) > 1e-3
    ref = torch.from_numpy(gp["labels"]).long()
    assert torch.equal(labels_c.cpu().long()[stable], ref[stable])
    assert torch.equal(labels_t.cpu().long()[stable], ref[stable])
    with pytest.raises(RuntimeError):
        fus.forward_from_features(ir, vis, torch.zeros(1, 16, 24, 32).cuda(), torch.zeros(1, 8, 12, 64).cuda())


def test_pair_forward_uint8_roundtrip_flag(net_b1, fus, golden_dir):
    """with uint8_roundtrip the segmentation net is fed what test_segmentation.py reads back from the PNGs
    test_fusion.py:112-120 wrote (uint8(255 x) -> batch-global min-max -> uint8, then float32 / 255 as
    TaskFusion_dataset2.py:84-88 loads it); emulated here in numpy from the fp32 fused image, bit for bit."""
    from segmif_amd.pipeline import PairForward
    gp = load(golden_dir, "pair_b1_64x96.npz")
    ir, vis, mask = (torch.from_numpy(gp[k]).cuda() for k in ("ir", "vis", "mask"))
    fused32, labels32 = PairForward(net_b1, fus)(ir, vis, mask)
    fused_q, labels_q = PairForward(net_b1, fus, uint8_roundtrip=True)(ir, vis, mask)
    u = np.uint8(255.0 * fused32.cpu().numpy()).transpose((0, 2, 3, 1))
    u = (u - np.min(u)) / (np.max(u) - np.min(u))
    u = np.uint8(255.0 * u)
    back = np.asarray(u, dtype=np.float32).transpose((0, 3, 1, 2)) / 255.0
    assert np.array_equal(fused_q.cpu().numpy(), back)
    with torch.no_grad():
        direct = net_b1.predict_labels(torch.from_numpy(back).cuda(), vis.shape[2:])
    assert torch.equal(labels_q, direct)
    assert labels_q.shape == labels32.shape and float((labels_q == labels32).float().mean()) > 0.9


def test_head_fuse_commutes_with_the_resize(net_b1, golden_dir):
    """SURVEY §8(f) N4, second half: SegFormerHead in eval mode applies linear_fuse per scale before the
    bilinear resize (composed with the scale's Linear, BatchNorm folded); same function as the textbook
    order (concat -> fuse), and both meet the reference fixture through Network3."""
    gp = load(golden_dir, "pair_b1_64x96.npz")
    mask = torch.from_numpy(gp["mask"]).cuda()
    head = net_b1.denoise_net.decoder
    assert head.commute_resize
    with torch.no_grad():
        feats = net_b1.denoise_net.encoder.forward_features_nhwc(mask)
        a = head.forward_nhwc(feats)
        head.commute_resize = False
        try:
            b = head.forward_nhwc(feats)
        finally:
            head.commute_resize = True
    assert rel(a, b.cpu()) < 1e-5
    fused = torch.from_numpy(gp["fused"]).cuda()
    with torch.no_grad():
        _, _, seg = net_b1(fused)
    assert rel(seg, gp["seg"]) < 5 * TIGHT


def assert_miou_parity(ref_labels, hip_labels, gt_name):
    """North star: seg mIoU within +-0.1 of the reference on fixed synthetic inputs.  mIoU is taken
    against seeded synthetic ground truth with the reference's own formula (util/util.py:31-55)."""
    gt = dw.det_labels(gt_name, ref_labels.shape, 9).numpy()
    m_ref, _ = so.miou(so.confusion(gt, ref_labels))
    m_hip, _ = so.miou(so.confusion(gt, hip_labels))
    assert abs(m_ref - m_hip) < 1e-4, (m_ref, m_hip)
    assert float((np.asarray(ref_labels) == np.asarray(hip_labels)).mean()) > 0.999


def pair_forward_hip(core, seg_net, fus_net, ir, vis, mask3):
    """The measured unit of work on the HIP path."""
    out0, out1 = seg_net.denoise_net.encoder.forward_fusion(mask3)
    y_f = fus_net(ir, vis, out0, out1)
    fused = core.fuse_to_rgb(vis, y_f)
    _, _, seg = seg_net(fused)
    from segmif_amd import ops
    logits = ops.bilinear(ops.to_nhwc(seg), vis.shape[2], vis.shape[3])
    labels = ops.argmax_nhwc(logits)
    return dict(out0=out0, out1=out1, y_fused=y_f, fused=fused, seg=seg, logits=ops.as_nchw(logits), labels=labels)


def test_pair_b1_vs_reference(core, net_b1, fus, golden_dir):
    g = load(golden_dir, "pair_b1_64x96.npz")
    t64 = load(golden_dir, "pair_b1_64x96_fp64.npz")
    ir, vis, mask = (torch.from_numpy(g[k]).cuda() for k in ("ir", "vis", "mask"))
    with torch.no_grad():
        feats = net_b1.denoise_net.encoder(mask)
        r = pair_forward_hip(core, net_b1, fus, ir, vis, mask)
    for i, f in enumerate(feats):
        assert rel(f, g[f"f{i + 1}"]) < TIGHT
    assert rel(r["out0"][:, :, 1::5, 2::7], g["out0_sample"]) < TIGHT
    assert rel(r["out1"][:, :, 1::5, 2::7], g["out1_sample"]) < TIGHT
    for k in ("y_fused", "fused", "seg", "logits"):
        assert rel(r[k], g[k]) < TOL, k
        assert rel(r[k], g[k]) < 5 * TIGHT, k
        # (r6) ELEMENT by element.  Read that way float32 itself is not 1e-3 accurate on these tensors: the
        # reference's own float32 record sits 5.3e-3 (y_fused; p99.9 3.0e-3) .. 8.6e-3 (fused) from the reference evaluated in
        # float64 over the elements above 1 % of the range (tests/golden/pair_b1_64x96_fp64.npz, oracle/make_golden_r6_truth.py:
        # the real reference's modules cast to double).  So the element-wise gate is against that TRUTH: the HIP result may be no
        # further from it than 1.5 x the reference's float32 result is - maximum, 99.9th percentile and RMS-relative alike.
        ew_max, ew_p999, rms = rel_elementwise(r[k], t64[k])
        ref_max, ref_p999, ref_rms = (float(t64[f"ref32_{k}_{q}"]) for q in ("ew_max", "ew_p999", "rms"))
        vs_ref = rel_elementwise(r[k], g[k])
        from _observed import observed
        observed(f"elementwise[pair_b1_64x96:{k}]", {"hip_vs_fp64": {"max_above_1pct": ew_max, "p999_above_1pct": ew_p999, "rms_rel": rms},
                                                     "reference_fp32_vs_fp64": {"max_above_1pct": ref_max, "p999_above_1pct": ref_p999, "rms_rel": ref_rms},
                                                     "hip_vs_reference_fp32": {"max_above_1pct": vs_ref[0], "p999_above_1pct": vs_ref[1], "rms_rel": vs_ref[2]},
                                                     "max_norm_vs_reference_fp32": rel(r[k], g[k])})
        assert ew_max <= max(TOL, 1.5 * ref_max) and ew_p999 <= max(TOL, 1.5 * ref_p999) and rms <= max(TIGHT, 1.5 * ref_rms), \
            (k, (ew_max, ew_p999, rms), (ref_max, ref_p999, ref_rms))
    stable = torch.from_numpy(g["margin"]) > 1e-3
    got = r["labels"].cpu().long()
    assert torch.equal(got[stable], torch.from_numpy(g["labels"]).long()[stable])
    assert_miou_parity(g["labels"], got.numpy(), "b1_gt")


def test_f2_and_cpu_inputs_fail_loudly(core, fus):
    x = torch.zeros(1, 3, 32, 32).cuda()
    with pytest.raises(RuntimeError):  # mit_b0 widths (32/64) do not fit conv3/conv4
        fus(x[:, :1], x, torch.zeros(1, 32, 32, 32).cuda(), torch.zeros(1, 64, 32, 32).cuda())
    with pytest.raises(RuntimeError):  # no CPU fallback
        fus(x.cpu()[:, :1], x.cpu(), torch.zeros(1, 64, 32, 32), torch.zeros(1, 128, 32, 32))


def test_nchw_contiguous_inputs_are_accepted(core, net_b1, fus):
    """The boundary takes ordinary NCHW tensors (as the reference's callers pass) as well as the
    channels-last views our own modules emit."""
    ir = dw.det_input("t_ir", (1, 1, 32, 48)).cuda()
    vis = dw.det_input("t_vis", (1, 3, 32, 48)).cuda()
    with torch.no_grad():
        o0, o1 = net_b1.denoise_net.encoder.forward_fusion(vis)
        a = fus(ir, vis, o0, o1)
        b = fus(ir, vis, o0.contiguous(), o1.contiguous())
    assert torch.equal(a, b)


def test_oracle_parity_fresh_inputs_b1(core, net_b1, fus):
    """HIP path vs the CPU oracle on inputs that are not in any fixture (odd batch / size)."""
    B, H, W = 3, 40, 56
    ir = dw.det_input("o_ir", (B, 1, H, W))
    vis = dw.det_input("o_vis", (B, 3, H, W))
    mask = dw.det_input("o_mask", (B, 1, H, W)).repeat(1, 3, 1, 1)
    sd_seg = dw.det_state_dict(so.network3_shapes("mit_b1", 9), seed=0)
    sd_fus = dw.det_state_dict(so.fusion_shapes(), seed=0)
    with torch.no_grad():
        ref = so.pair_forward(sd_seg, sd_fus, ir, vis, mask, "mit_b1", return_all=True)
        sd64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()} for sd in (sd_seg, sd_fus)]
        ref64 = so.pair_forward(sd64[0], sd64[1], ir.double(), vis.double(), mask.double(), "mit_b1", return_all=True)
        r = pair_forward_hip(core, net_b1, fus, ir.cuda(), vis.cuda(), mask.cuda())
    for k in ("out0", "out1", "y_fused", "fused", "seg", "logits"):
        assert rel(r[k], ref[k]) < 5 * TIGHT, k
        # (r6) element by element above 1 % of the range, against the oracle evaluated in float64, with the oracle's float32
        # result (= the reference's arithmetic) as the yardstick - see test_pair_b1_vs_reference
        hip, ref32 = rel_elementwise(r[k], ref64[k]), rel_elementwise(ref[k], ref64[k])
        assert all(a <= max(lim, 1.5 * b) for a, b, lim in zip(hip, ref32, (TOL, TOL, TIGHT))), (k, hip, ref32)
    stable = so.top2_margin(ref["logits"]) > 1e-3
    assert torch.equal(r["labels"].cpu().long()[stable], ref["labels"][stable])


def test_full_size_b3_vs_reference_checksum(core, fus, golden_dir):
    """mit_b3, 480x640 (the headline configuration): HIP pair forward vs the record the real
    reference produced — sampled values of every stage, exact labels above the margin threshold,
    mIoU of HIP labels vs reference labels."""
    g = load(golden_dir, "pair_b3_480x640_checksum.npz")
    net = build(core, core.Network3, "mit_b3", 9, pretrained=None)
    H, W = 480, 640
    ir = dw.det_input("b3_ir", (1, 1, H, W)).cuda()
    vis = dw.det_input("b3_vis", (1, 3, H, W)).cuda()
    mask = dw.det_input("b3_mask", (1, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    with torch.no_grad():
        r = pair_forward_hip(core, net, fus, ir, vis, mask)
    # sampled values of every stage: max-norm gate + (r6) the element-wise relative gate above 1 % of the range (_check_samples)
    labels, ref_labels = _check_samples(r, g, ("out0", "out1", "y_fused", "fused", "seg", "logits"))
    stable = torch.from_numpy(g["margin_f16"].astype(np.float32)) > 1e-3
    assert torch.equal(labels[stable], ref_labels[stable])
    mismatches = int((labels != ref_labels).sum())
    assert mismatches <= int((~stable).sum())
    assert_miou_parity(ref_labels.numpy(), labels.numpy(), "b3_gt")
    # the measured pipeline (segmif_amd.pipeline.PairForward: conv3 / conv4 before the resize) on the same record
    from segmif_amd.pipeline import PairForward
    fused_p, labels_p = PairForward(net, fus)(ir, vis, mask)
    got = fused_p.contiguous().reshape(-1)[torch.from_numpy(g["fused_idx"]).cuda()].cpu()
    scale = max(abs(g["fused_stats"][2]), abs(g["fused_stats"][3]))
    assert float((got - torch.from_numpy(g["fused_val"])).abs().max()) / scale < 5 * TIGHT
    labels_p = labels_p.cpu().long().reshape(ref_labels.shape)
    assert torch.equal(labels_p[stable], ref_labels[stable])
    assert_miou_parity(ref_labels.numpy(), labels_p.numpy(), "b3_gt")


def _check_samples(r, g, names):
    for name in names:
        got = r[name].contiguous().reshape(-1)[torch.from_numpy(g[name + "_idx"]).cuda()].cpu()
        scale = max(abs(g[name + "_stats"][2]), abs(g[name + "_stats"][3]))
        want = torch.from_numpy(g[name + "_val"])
        e = float((got - want).abs().max()) / scale
        # (r6) beside the max-norm figure: the ELEMENT-WISE relative error over the sampled elements above 1 % of
        # the tensor's range - its maximum gates below, its 99.9th percentile and the RMS-relative error are recorded
        # (profiles/r06_parity_observed/) - so that damage confined to small-magnitude regions cannot hide behind the largest value
        big = want.abs() > 1e-2 * scale
        ew = ((got - want).abs() / want.abs())[big].double()
        ew_max = float(ew.max()) if ew.numel() else 0.0
        import inspect
        from _observed import observed
        observed(f"fullsize_samples[{inspect.stack()[1].function}:{name}]",
                 {"max_norm": e, "p999_elementwise_rel_above_1pct": float(torch.quantile(ew, 0.999)) if ew.numel() else 0.0,
                  "max_elementwise_rel_above_1pct": ew_max,
                  "rms_rel": float((got - want).double().pow(2).mean().sqrt() / want.double().pow(2).mean().sqrt().clamp_min(1e-30)),
                  "samples": int(want.numel()), "samples_above_1pct": int(big.sum())})
        assert e < TOL and e < 5 * TIGHT, (name, e)
        # (r6) gating as well: every sampled element above 1 % of the tensor's range is within 1e-3 of the reference's value,
        # element-wise relative (observed: <= 1.3e-4 on the b3 / b5 full-size records)
        assert ew_max < TOL, (name, ew_max)
    labels = r["labels"].cpu().long().reshape(g["labels"].shape)
    ref_labels = torch.from_numpy(g["labels"]).long()
    stable = torch.from_numpy(g["margin_f16"].astype(np.float32)) > 1e-3
    assert torch.equal(labels[stable], ref_labels[stable])
    assert int((labels != ref_labels).sum()) <= int((~stable).sum())
    return labels, ref_labels


def _direct_hip(net, x, g):
    """test_segmentation.py:169-174 on the HIP path with the fixture's re-centred linear_pred.bias."""
    from segmif_amd import ops
    with torch.no_grad():
        net.denoise_net.decoder.linear_pred.bias.copy_(torch.from_numpy(g["pred_bias"]).cuda())
        _, _, seg = net(x)
        logits = ops.bilinear(ops.to_nhwc(seg), x.shape[2], x.shape[3])
        return dict(seg=seg, logits=ops.as_nchw(logits), labels=ops.argmax_nhwc(logits))


def test_direct_segmentation_b3_all_nine_classes_and_miou(core, golden_dir):
    """The mIoU gate on a record in which the reference predicts all nine classes (>= 24 000 pixels each): sampled
    logits, exact labels above the margin, mIoU within 1e-4 of the reference's (north star: +-0.1)."""
    g = load(golden_dir, "seg_b3_480x640_direct.npz")
    assert int((g["label_hist"] >= 20000).sum()) == 9
    net = build(core, core.Network3, "mit_b3", 9, pretrained=None)
    r = _direct_hip(net, dw.det_input("b3_direct", (1, 3, 480, 640)).cuda(), g)
    labels, ref_labels = _check_samples(r, g, ("seg", "logits"))
    assert_miou_parity(ref_labels.numpy(), labels.numpy(), "b3_direct_gt")


def test_config4_mit_b5_1024_vs_reference_checksums(core, fus, golden_dir):
    """BASELINE config[4]: mit_b5 at 1024x1024 — Network3 forward and the whole pair forward against the records of
    the real reference (Nk = 1024 keys per attention block, 65 536 stage-1 tokens), plus the config's batch of 2 as a
    batch-consistency property (batch 2 == two batches of 1 to fp32 rounding)."""
    net = build(core, core.Network3, "mit_b5", 9, pretrained=None)
    H = W = 1024
    g = load(golden_dir, "seg_b5_1024_direct.npz")
    saved = net.denoise_net.decoder.linear_pred.bias.detach().clone()
    x = dw.det_input("b5_direct", (1, 3, H, W)).cuda()
    labels, ref_labels = _check_samples(_direct_hip(net, x, g), g, ("seg", "logits"))
    assert_miou_parity(ref_labels.numpy(), labels.numpy(), "b5_direct_gt")
    with torch.no_grad():
        x2 = torch.cat((x, dw.det_input("b5_second", (1, 3, H, W)).cuda()))
        both = net(x2)[2]
        # (not bitwise: small grids run split-K, whose split — hence summation order — depends on the row count)
        for i in range(2):
            assert rel(both[i:i + 1], net(x2[i:i + 1])[2].cpu()) < TIGHT, i
        net.denoise_net.decoder.linear_pred.bias.copy_(saved)
    g = load(golden_dir, "pair_b5_1024_checksum.npz")
    ir = dw.det_input("b5_ir", (1, 1, H, W)).cuda()
    vis = dw.det_input("b5_vis", (1, 3, H, W)).cuda()
    mask = dw.det_input("b5_mask", (1, 1, H, W)).repeat(1, 3, 1, 1).cuda()
    with torch.no_grad():
        r = pair_forward_hip(core, net, fus, ir, vis, mask)
    _check_samples(r, g, ("out0", "out1", "y_fused", "fused", "seg", "logits"))
    from segmif_amd.pipeline import PairForward
    fused_p, labels_p = PairForward(net, fus)(ir, vis, mask)
    _check_samples(dict(fused=fused_p, labels=labels_p), g, ("fused",))


def test_batch_consistency_full_size(core, fus):
    """Size-independent property at the bench size: a batch of 2 equals two batches of 1, bitwise
    (no cross-sample leakage, deterministic reductions)."""
    H, W = 480, 640
    ir = dw.det_input("bc_ir", (2, 1, H, W)).cuda()
    vis = dw.det_input("bc_vis", (2, 3, H, W)).cuda()
    o1 = dw.det_input("bc_o1", (2, 64, H, W), lo=-1, hi=1).cuda()
    o2 = dw.det_input("bc_o2", (2, 128, H, W), lo=-1, hi=1).cuda()
    with torch.no_grad():
        both = fus(ir, vis, o1, o2)
        one = torch.cat([fus(ir[i:i + 1], vis[i:i + 1], o1[i:i + 1], o2[i:i + 1]) for i in range(2)])
    assert torch.equal(both, one)
