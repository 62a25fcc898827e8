"""Generate tests/golden/* by running the REAL upstream reference on CPU.

Runs only in the build container (needs /root/reference).  The fixtures it writes are
data only: seeded inputs and the reference's outputs.  The GPU box and the CPU test
suite read the fixtures, never the reference.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--full]

--full additionally records the full-size (mit_b3, 480x640) pair-forward checksum
record (about one minute of CPU time) and a direct segmentation forward on a U[0,1) image whose
labels populate all nine classes (the mIoU gate's fixture).
--full5 records the BASELINE config[4] checksums: mit_b5 at 1024x1024, batch 1 — Network3 forward on a
U[0,1) image and the whole pair forward (several minutes of CPU time, ~20 GB of RSS).
"""
import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detweights as dw  # noqa: E402
import refload  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
NUM_CLASSES = 9


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npy(t):
    return t.detach().cpu().numpy()


def ref_rgb2ycrcb(x):
    # test_fusion.py:155-172 is the CPU-capable copy of core/model_fusion.py:69-91; re-stated
    # inline because both upstream copies hard-wire a CUDA device.
    flat = x.transpose(1, 3).transpose(1, 2).reshape(-1, 3)
    R, G, B = flat[:, 0], flat[:, 1], flat[:, 2]
    Y = 0.299 * R + 0.587 * G + 0.114 * B
    Cr = (R - Y) * 0.713 + 0.5
    Cb = (B - Y) * 0.564 + 0.5
    temp = torch.cat((Y[:, None], Cr[:, None], Cb[:, None]), dim=1)
    return temp.reshape(x.size(0), x.size(2), x.size(3), 3).transpose(1, 3).transpose(2, 3)


def ref_ycrcb2rgb(x):
    flat = x.transpose(1, 3).transpose(1, 2).reshape(-1, 3)
    mat = torch.tensor([[1.0, 1.0, 1.0], [1.403, -0.714, 0.0], [0.0, -0.344, 1.773]])
    bias = torch.tensor([0.0 / 255, -0.5, -0.5])
    temp = (flat + bias).mm(mat)
    return temp.reshape(x.size(0), x.size(2), x.size(3), 3).transpose(1, 3).transpose(2, 3)


def ref_pair_forward(seg, fus, ir, vis, mask3):
    """test_fusion.py:100-111 then test_segmentation.py:169-174 on the reference modules."""
    out0, out1 = seg.denoise_net.encoder.forward_fusion(mask3)
    y_f = fus(ir, vis, out0, out1)
    ycc = ref_rgb2ycrcb(vis)
    fused = ref_ycrcb2rgb(torch.cat((y_f, ycc[:, 1:2], ycc[:, 2:]), dim=1))
    fused = torch.where(fused > 1, torch.ones_like(fused), fused)
    fused = torch.where(fused < 0, torch.zeros_like(fused), fused)
    fused = fused.contiguous()
    _, _, seg1 = seg.forward(fused)
    logits = F.interpolate(seg1, size=vis.shape[2:], mode="bilinear", align_corners=False)
    return dict(out0=out0, out1=out1, y_fused=y_f, fused=fused, seg=seg1, logits=logits,
                labels=logits.argmax(1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--full5", action="store_true")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    mt, sh, mf = refload.load_reference()
    meta = {"torch": torch.__version__, "seed_weights": 0, "seed_inputs": 1}

    # ---- 1. state_dict keys / shapes (checkpoint compatibility) ---------------
    keys = {}
    for bb in ("mit_b0", "mit_b1", "mit_b2", "mit_b3", "mit_b4", "mit_b5"):
        net = quiet(mf.Network3, bb, NUM_CLASSES, pretrained=None)
        keys["Network3:" + bb] = {k: list(v.shape) for k, v in net.state_dict().items()}
    fus = quiet(mf.Fusion_Network3_ac)
    keys["Fusion_Network3_ac"] = {k: list(v.shape) for k, v in fus.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    # ---- 2. mit_b0, ragged 72x104 (sr-conv drops remainders at every stage) ------------------
    net0 = quiet(mf.Network3, "mit_b0", NUM_CLASSES, pretrained=None).eval()
    dw.load_det_weights(net0, seed=0)
    x = dw.det_input("b0_72x104", (1, 3, 72, 104))
    feats = net0.denoise_net.encoder(x)
    o0, o1 = net0.denoise_net.encoder.forward_fusion(x)
    _, _, seg = net0.forward(x.clone())
    np.savez_compressed(
        os.path.join(OUT, "mit_b0_72x104.npz"), x=npy(x),
        f1=npy(feats[0]), f2=npy(feats[1]), f3=npy(feats[2]), f4=npy(feats[3]),
        fus0_sample=npy(o0[:, :, 1::5, 2::7]), fus1_sample=npy(o1[:, :, 1::5, 2::7]),
        fus0_mean=np.float64(o0.double().mean()), fus1_mean=np.float64(o1.double().mean()),
        seg=npy(seg))

    # ---- 2b. mit_b2 / mit_b4 (depths 3-4-6-3 / 3-8-27-3), ragged 72x104: the two constructors no other record exercises --
    for bb in ("mit_b2", "mit_b4"):
        netx = quiet(mf.Network3, bb, NUM_CLASSES, pretrained=None).eval()
        dw.load_det_weights(netx, seed=0)
        xx = dw.det_input(bb + "_72x104", (1, 3, 72, 104))
        fx = netx.denoise_net.encoder(xx)
        q0, q1 = netx.denoise_net.encoder.forward_fusion(xx)
        _, _, segx = netx.forward(xx.clone())
        np.savez_compressed(
            os.path.join(OUT, bb + "_72x104.npz"),
            f1=npy(fx[0]), f2=npy(fx[1]), f3=npy(fx[2]), f4=npy(fx[3]),
            fus0_sample=npy(q0[:, :, 1::5, 2::7]), fus1_sample=npy(q1[:, :, 1::5, 2::7]),
            fus0_mean=np.float64(q0.double().mean()), fus1_mean=np.float64(q1.double().mean()),
            seg=npy(segx))
        del netx

    # F2: the fusion net cannot consume mit_b0 features (64/128 channels hard-coded)
    fus = quiet(mf.Fusion_Network3_ac).eval()
    dw.load_det_weights(fus, seed=0)
    ir = dw.det_input("f2_ir", (1, 1, 72, 104))
    try:
        fus(ir, x, o0, o1)
        meta["F2_mit_b0_fusion_raises"] = False
    except RuntimeError:
        meta["F2_mit_b0_fusion_raises"] = True

    # ---- 3. mit_b1, batch 2, non-square 64x96: encoder, head, fusion net, full pair ----------
    net1 = quiet(mf.Network3, "mit_b1", NUM_CLASSES, pretrained=None).eval()
    dw.load_det_weights(net1, seed=0)
    B, H, W = 2, 64, 96
    ir = dw.det_input("b1_ir", (B, 1, H, W))
    vis = dw.det_input("b1_vis", (B, 3, H, W))
    mask = dw.det_input("b1_mask", (B, 1, H, W)).repeat(1, 3, 1, 1)
    feats = net1.denoise_net.encoder(mask)
    r = ref_pair_forward(net1, fus, ir, vis, mask)
    margin = torch.topk(r["logits"], 2, dim=1).values
    margin = margin[:, 0] - margin[:, 1]
    np.savez_compressed(
        os.path.join(OUT, "pair_b1_64x96.npz"), ir=npy(ir), vis=npy(vis), mask=npy(mask),
        f1=npy(feats[0]), f2=npy(feats[1]), f3=npy(feats[2]), f4=npy(feats[3]),
        out0_sample=npy(r["out0"][:, :, 1::5, 2::7]), out1_sample=npy(r["out1"][:, :, 1::5, 2::7]),
        y_fused=npy(r["y_fused"]), fused=npy(r["fused"]), seg=npy(r["seg"]), logits=npy(r["logits"]),
        labels=npy(r["labels"]).astype(np.uint8), margin=npy(margin))

    # ---- 4. fusion-net building blocks in isolation ------------------------------------------
    xd = dw.det_input("drdb_x", (2, 64, 20, 28), lo=-1.0, hi=1.0)
    yd = fus.DRDB1(xd)
    x1 = dw.det_input("ffm_x1", (2, 64, 12, 16), lo=-1.0, hi=1.0)
    x2 = dw.det_input("ffm_x2", (2, 64, 12, 16), lo=-1.0, hi=1.0)
    x3 = dw.det_input("ffm_seg", (2, 64, 12, 16), lo=-1.0, hi=1.0)
    f1, f2 = fus.ffm(x1, x2, x3)
    np.savez_compressed(os.path.join(OUT, "fusion_blocks.npz"), drdb_x=npy(xd), drdb_y=npy(yd),
                        ffm_x1=npy(x1), ffm_x2=npy(x2), ffm_seg=npy(x3), ffm_o1=npy(f1), ffm_o2=npy(f2))

    # ---- 5. MiT building blocks in isolation (stage-2 geometry of mit_b1: C=128, 2 heads, sr 4)
    enc = net1.denoise_net.encoder
    t = dw.det_input("blk_tokens", (2, 8 * 12, 128), lo=-1.0, hi=1.0)
    blk = enc.block2[1]
    att = blk.attn(t, 8, 12)
    ffn = blk.mlp(t, 8, 12)
    full = blk(t, 8, 12)
    pe_x = dw.det_input("pe_x", (2, 64, 17, 23), lo=-1.0, hi=1.0)
    pe_t, pe_h, pe_w = enc.patch_embed2(pe_x)
    t4 = dw.det_input("blk4_tokens", (2, 2 * 3, 512), lo=-1.0, hi=1.0)
    att4 = enc.block4[0].attn(t4, 2, 3)  # sr_ratio 1: no reduction conv
    np.savez_compressed(os.path.join(OUT, "mit_blocks.npz"), tokens=npy(t), attn=npy(att), ffn=npy(ffn),
                        block=npy(full), pe_x=npy(pe_x), pe_tokens=npy(pe_t), pe_hw=np.array([pe_h, pe_w]),
                        tokens4=npy(t4), attn4=npy(att4))

    # ---- 5b. gradients from the reference's own autograd (training path parity pin) -----------
    def grad_record(module, prefix=""):
        rec = {}
        for name, p in module.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach()
            rec[prefix + name + "|norm"] = np.float64(g.double().norm())
            flat = g.reshape(-1)
            rec[prefix + name + "|head"] = npy(flat[:2048])  # first 2048 entries (whole tensor when smaller)
        return rec

    torch.set_grad_enabled(True)
    B, H, W = 2, 64, 96
    xs = dw.det_input("tr_x", (B, 3, H, W))
    ys = dw.det_labels("tr_y", (B, H, W), 9)
    ys[0, 5:9, 7:30] = 255
    net1.zero_grad()
    net1.eval()
    loss = net1._loss(xs.clone(), ys, torch.nn.CrossEntropyLoss(ignore_index=255))  # model_fusion.py:1090-1097
    loss.backward()
    rec = grad_record(net1)
    rec["loss"] = np.float64(loss.detach())
    np.savez_compressed(os.path.join(OUT, "grads_seg_b1_64x96.npz"), **rec)
    net1.zero_grad()

    Bf, Hf, Wf = 2, 24, 40
    irf, visf = dw.det_input("g_ir", (Bf, 1, Hf, Wf)), dw.det_input("g_vis", (Bf, 3, Hf, Wf))
    g = torch.Generator().manual_seed(4242)
    o1f = torch.rand(Bf, 64, Hf, Wf, generator=g) * 2 - 1
    o2f = torch.rand(Bf, 128, Hf, Wf, generator=g) * 2 - 1
    cot = torch.rand(Bf, 1, Hf, Wf, generator=g) * 2 - 1
    fus.zero_grad()
    outf = fus(irf, visf, o1f, o2f)
    (outf * cot).sum().backward()
    rec = grad_record(fus)
    rec.update(o1=npy(o1f), o2=npy(o2f), cot=npy(cot), out=npy(outf))
    np.savez_compressed(os.path.join(OUT, "grads_fusion_24x40.npz"), **rec)
    fus.zero_grad()
    torch.set_grad_enabled(False)

    # ---- 6. full-size checksum record: mit_b3, 480x640, B=1 -----------------------------------
    if args.full:
        net3 = quiet(mf.Network3, "mit_b3", NUM_CLASSES, pretrained=None).eval()
        dw.load_det_weights(net3, seed=0)
        H, W = 480, 640
        ir = dw.det_input("b3_ir", (1, 1, H, W))
        vis = dw.det_input("b3_vis", (1, 3, H, W))
        mask = dw.det_input("b3_mask", (1, 1, H, W)).repeat(1, 3, 1, 1)
        r = ref_pair_forward(net3, fus, ir, vis, mask)
        margin = torch.topk(r["logits"], 2, dim=1).values
        margin = margin[:, 0] - margin[:, 1]
        g = np.random.Generator(np.random.PCG64(1234))
        rec = {"labels": npy(r["labels"]).astype(np.uint8),
               "margin_f16": npy(margin).astype(np.float16)}
        for name in ("out0", "out1", "y_fused", "fused", "seg", "logits"):
            t = r[name].double()
            flat = r[name].reshape(-1)
            idx = g.integers(0, flat.numel(), size=4096)
            rec[name + "_idx"] = idx
            rec[name + "_val"] = npy(flat[torch.from_numpy(idx)])
            rec[name + "_stats"] = np.array([t.mean().item(), t.abs().mean().item(), t.min().item(), t.max().item()])
        rec["label_hist"] = np.bincount(rec["labels"].reshape(-1), minlength=NUM_CLASSES)
        np.savez_compressed(os.path.join(OUT, "pair_b3_480x640_checksum.npz"), **rec)
        meta["full_label_hist"] = rec["label_hist"].tolist()
        meta["full_margin_min"] = float(margin.min())

    def checksum_record(r, names, seed):
        g = np.random.Generator(np.random.PCG64(seed))
        margin = torch.topk(r["logits"], 2, dim=1).values
        margin = margin[:, 0] - margin[:, 1]
        rec = {"labels": npy(r["labels"]).astype(np.uint8), "margin_f16": npy(margin).astype(np.float16)}
        for name in names:
            t = r[name].double()
            flat = r[name].reshape(-1)
            idx = g.integers(0, flat.numel(), size=4096)
            rec[name + "_idx"] = idx
            rec[name + "_val"] = npy(flat[torch.from_numpy(idx)])
            rec[name + "_stats"] = np.array([t.mean().item(), t.abs().mean().item(), t.min().item(), t.max().item()])
        rec["label_hist"] = np.bincount(rec["labels"].reshape(-1), minlength=NUM_CLASSES)
        return rec, float(margin.min())

    def seg_direct(net, x):
        """test_segmentation.py:169-174 on a given RGB image: logits -> x4 bilinear -> argmax.  With the seeded weights
        one or two classes win everywhere, which would make an mIoU comparison vacuous: linear_pred.bias (a parameter
        like any other) is first re-centred by the per-class mean logit of this very image, so that all nine classes
        are predicted; the bias actually used is part of the fixture."""
        pred = net.denoise_net.decoder.linear_pred
        _, _, seg0 = net.forward(x.clone())
        saved = pred.bias.detach().clone()
        pred.bias.copy_(saved - seg0.mean(dim=(0, 2, 3)))
        _, _, seg1 = net.forward(x.clone())
        logits = F.interpolate(seg1, size=x.shape[2:], mode="bilinear", align_corners=False)
        out = dict(seg=seg1, logits=logits, labels=logits.argmax(1), pred_bias=pred.bias.detach().clone())
        pred.bias.copy_(saved)
        return out

    # ---- 7. direct segmentation forward, mit_b3 480x640: all nine classes predicted (mIoU gate) ----
    if args.full:
        x = dw.det_input("b3_direct", (1, 3, 480, 640))
        r = seg_direct(net3, x)
        rec, mmin = checksum_record(r, ("seg", "logits"), 4321)
        rec["pred_bias"] = npy(r["pred_bias"])
        np.savez_compressed(os.path.join(OUT, "seg_b3_480x640_direct.npz"), **rec)
        meta["direct_b3_label_hist"] = rec["label_hist"].tolist()
        del net3

    # ---- 8. BASELINE config[4]: mit_b5, 1024x1024, batch 1 ----------------------------------------
    if args.full5:
        net5 = quiet(mf.Network3, "mit_b5", NUM_CLASSES, pretrained=None).eval()
        dw.load_det_weights(net5, seed=0)
        H = W = 1024
        x = dw.det_input("b5_direct", (1, 3, H, W))
        r = seg_direct(net5, x)
        rec, _ = checksum_record(r, ("seg", "logits"), 555)
        rec["pred_bias"] = npy(r["pred_bias"])
        np.savez_compressed(os.path.join(OUT, "seg_b5_1024_direct.npz"), **rec)
        meta["direct_b5_label_hist"] = rec["label_hist"].tolist()
        ir = dw.det_input("b5_ir", (1, 1, H, W))
        vis = dw.det_input("b5_vis", (1, 3, H, W))
        mask = dw.det_input("b5_mask", (1, 1, H, W)).repeat(1, 3, 1, 1)
        r = ref_pair_forward(net5, fus, ir, vis, mask)
        rec, mmin = checksum_record(r, ("out0", "out1", "y_fused", "fused", "seg", "logits"), 556)
        np.savez_compressed(os.path.join(OUT, "pair_b5_1024_checksum.npz"), **rec)
        meta["full5_label_hist"] = rec["label_hist"].tolist()
        meta["full5_margin_min"] = mmin

    meta_path = os.path.join(OUT, "meta.json")
    if os.path.exists(meta_path):  # a run without --full / --full5 keeps the entries those legs recorded earlier
        with open(meta_path) as f:
            meta = {**json.load(f), **meta}
    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote fixtures to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn:40s} {os.path.getsize(os.path.join(OUT, fn)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
