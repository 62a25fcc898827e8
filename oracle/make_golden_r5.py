"""Round-5 fixture: a FULL-SIZE gradient record from the REAL upstream reference.

  grads_fusion_step_b3_480x640.npz   one iteration of train.py:351-385 (iter_ = 2: Fusionloss_grad3 + cross-entropy through the
                                     segmentation net) on Network3('mit_b3') + Fusion_Network3_ac at 1 x 480 x 640 - BASELINE
                                     config[2]'s backbone and image size, one sample: the three loss values and, for every
                                     parameter of the fusion net, the gradient's norm and its first 2048 entries, taken after
                                     seg_loss.backward() and before the optimizer step.

Until now gradients were pinned to the reference at 24 x 40 .. 64 x 96 only, and at full size against themselves (batch 8 vs
batch 2).  Modules run in eval() mode for the reasons given in make_golden_train.py; `.cuda()` calls in the reference's loss /
colour code are identity functions during generation (device placement is not arithmetic).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r5.py
Container-only (needs /root/reference); the fixture is data (inputs are re-derived from detweights names, outputs recorded)."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detweights as dw  # noqa: E402
import make_golden_train as mgt  # noqa: E402
import refload  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    torch.manual_seed(0)
    mt, sh, mf = refload.load_reference()
    with mgt.cuda_is_identity():
        _, loss_mod = mgt.load_reference_losses()
    B, H, W, iter_ = 1, 480, 640, 2
    net = mgt.quiet(mf.Network3, "mit_b3", mgt.NUM_CLASSES, pretrained=None).eval()
    dw.load_det_weights(net, seed=0)
    fus = mgt.quiet(mf.Fusion_Network3_ac).eval()
    dw.load_det_weights(fus, seed=0)
    with mgt.cuda_is_identity():
        floss = loss_mod.Fusionloss_grad3()
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    ir3 = dw.det_input("r5g_ir", (B, 1, H, W)).repeat(1, 3, 1, 1)
    vis3 = dw.det_input("r5g_vis", (B, 3, H, W))
    mask3 = dw.det_input("r5g_mask", (B, 1, H, W)).repeat(1, 3, 1, 1)
    labels = dw.det_labels("r5g_lab", (B, H, W), mgt.NUM_CLASSES)
    labels[0, 11:40, 100:300] = 255
    ir = ir3[:, 0:1]
    t0 = time.time()
    with mgt.cuda_is_identity():
        vis = mf.RGB2YCrCb(vis3)
        with torch.no_grad():
            out0, out1 = net.denoise_net.encoder.forward_fusion(mask3)
        fusion = fus(ir, vis, out0, out1)
        ycc = vis.clone()
        ycc[:, 0:1] = fusion
        rgb = mf.YCrCb2RGB(ycc)
        loss1 = floss(ir, vis, fusion, mask3)
        loss2 = net._loss(rgb, labels, crit)
    seg_loss = (0.4 / iter_) * loss1 + 0.8 * loss2
    fus.zero_grad()
    seg_loss.backward()
    rec = {"loss1": np.float64(loss1.detach()), "loss2": np.float64(loss2.detach()), "total": np.float64(seg_loss.detach()),
           "fusion_stats": np.array([float(fusion.min()), float(fusion.max()), float(fusion.double().mean())])}
    for name, p in fus.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        rec[name + "|norm"] = np.float64(g.double().norm())
        rec[name + "|head"] = mgt.npy(g.reshape(-1)[:2048])
    rec["no_grad_params"] = np.array([n for n, p in fus.named_parameters() if p.grad is None])
    # Scalar parameters (the shared PReLU slope): their gradient is ONE fp32 sum over every pixel of every layer that uses them -
    # 2e8 terms in whatever order torch's CPU reductions take - so the float32 record above carries a summation error of its own.
    # The same step with the reference's modules in float64 gives the value both implementations are approximating.
    small = [n for n, p in fus.named_parameters() if p.grad is not None and p.numel() <= 4]
    if small:
        net64, fus64 = net.double(), fus.double()
        torch.set_default_dtype(torch.float64)  # (the reference's colour / loss code builds its constant matrices with torch.tensor(..))
        with mgt.cuda_is_identity():
            floss64 = loss_mod.Fusionloss_grad3()
            for m in floss64.modules():
                m.double()
            for k, v in list(vars(floss64).items()):
                if torch.is_tensor(v) and v.is_floating_point():
                    setattr(floss64, k, v.double())
            vis64 = mf.RGB2YCrCb(vis3.double())
            with torch.no_grad():
                o0, o1 = net64.denoise_net.encoder.forward_fusion(mask3.double())
            fusion64 = fus64(ir.double(), vis64, o0, o1)
            ycc64 = vis64.clone()
            ycc64[:, 0:1] = fusion64
            rgb64 = mf.YCrCb2RGB(ycc64)
            l1 = floss64(ir.double(), vis64, fusion64, mask3.double())
            l2 = net64._loss(rgb64, labels, crit)
        fus64.zero_grad()
        ((0.4 / iter_) * l1 + 0.8 * l2).backward()
        for name, p in fus64.named_parameters():
            if name in small:
                rec[name + "|f64"] = mgt.npy(p.grad.detach().reshape(-1))
            elif p.grad is not None:
                # (r6) every other tensor too: the gradient is an ill-conditioned function of the forward (it runs back through the
                # CrossPath context softmaxes), so "distance to the float32 record" is a noisy yardstick at the 2e-3 level - a
                # last-bit change in one forward kernel moved it from 1.7e-3 to 2.1e-3.  The float64 run is what both approximate.
                rec[name + "|f64head"] = mgt.npy(p.grad.detach().reshape(-1)[:2048])
                rec[name + "|f64norm"] = np.float64(p.grad.detach().norm())
        rec["loss1_f64"], rec["loss2_f64"] = np.float64(l1.detach()), np.float64(l2.detach())
        torch.set_default_dtype(torch.float32)
    path = os.path.join(OUT, "grads_fusion_step_b3_480x640.npz")
    np.savez_compressed(path, **rec)
    print(f"  {os.path.basename(path)}: {os.path.getsize(path) / 1024:.1f} KiB, {time.time() - t0:.0f} s of reference CPU time, "
          f"losses {float(loss1):.6f} {float(loss2):.6f} {float(seg_loss):.6f}")


if __name__ == "__main__":
    main()
