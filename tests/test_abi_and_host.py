"""CPU-only checks: the C-ABI library loads and exports every symbol include/segmif_hip.h
declares, the ctypes struct matches the C layout, host-side module logic (state_dict keys, cache
invalidation, loud failure on CPU tensors).  No kernel is launched here."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from segmif_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "segmif_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(segmif_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from segmif_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in segmif_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.segmif_abi_version() == 3
    assert lib.segmif_igemm_num_tiles() == 15
    assert lib.segmif_linattn_num_blocks(307200) == 300


def test_struct_layout_matches_c(lib, tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    from segmif_amd._lib import SegmifIgemm
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "segmif_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(SegmifIgemm),offsetof(SegmifIgemm,M),offsetof(SegmifIgemm,act),'
                   'offsetof(SegmifIgemm,in_zstride),offsetof(SegmifIgemm,tile));return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(SegmifIgemm), SegmifIgemm.M.offset, SegmifIgemm.act.offset,
            SegmifIgemm.in_zstride.offset, SegmifIgemm.tile.offset]
    assert got == want


def test_struct_layout_of_the_f16x3_fields_matches_c(lib, tmp_path):
    """The descriptors that grew f16x3 fields (planes_f16 / planes_amax): size and the new fields' offsets against gcc."""
    from segmif_amd._lib import SegmifConvPlanes, SegmifCrossTail, SegmifGemmSplit, SegmifIgemm
    src = tmp_path / "layout16.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "segmif_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(SegmifIgemm),offsetof(SegmifIgemm,planes_f16),offsetof(SegmifIgemm,planes_amax),'
                   'sizeof(SegmifCrossTail),offsetof(SegmifCrossTail,planes_f16),offsetof(SegmifCrossTail,planes_amax),'
                   'offsetof(SegmifIgemm,planes_amax_images),offsetof(SegmifCrossTail,planes_amax_images),'
                   'sizeof(SegmifConvPlanes),sizeof(SegmifGemmSplit));return 0;}')
    exe = tmp_path / "layout16"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(SegmifIgemm), SegmifIgemm.planes_f16.offset, SegmifIgemm.planes_amax.offset,
            ctypes.sizeof(SegmifCrossTail), SegmifCrossTail.planes_f16.offset, SegmifCrossTail.planes_amax.offset,
            SegmifIgemm.planes_amax_images.offset, SegmifCrossTail.planes_amax_images.offset,
            ctypes.sizeof(SegmifConvPlanes), ctypes.sizeof(SegmifGemmSplit)]
    assert got == want


def test_f16x3_entry_points_size_rules_and_rejections_without_a_gpu(lib):
    """segmif_planes16_* / segmif_gemm_split16_*: buffer sizes (4 bytes per activation instead of 6; weight images = the bf16
    images + one float per output row) and EINVAL on null / malformed arguments - no kernel is launched."""
    from segmif_amd._lib import SegmifConvPlanes, SegmifGemmSplit
    assert lib.segmif_planes16_bytes(2, 13, 45, 6) * 3 == lib.segmif_planes_bytes(2, 13, 45, 6) * 2
    assert lib.segmif_planes16_bytes(2, 13, 45, 6) == 2 * 6 * 20 * 68 * 64
    assert lib.segmif_planes16_bytes(0, 13, 45, 6) == 0
    assert lib.segmif_planes16_weight_bytes(32, 64, 9) == lib.segmif_planes_weight_bytes(32, 64, 9) + 4 * 32
    assert lib.segmif_planes16_weight_bytes(48, 64, 9) == 0 and lib.segmif_planes16_weight_bytes(32, 24, 9) == 0
    assert lib.segmif_gemm_split16_weight_bytes(320, 320) == lib.segmif_gemm_split_weight_bytes(320, 320) + 4 * 384
    assert lib.segmif_gemm_split16_weight_bytes(320, 48) == 0
    assert lib.segmif_planes16_zero_border(None, 1, 8, 32, 4, None) == -22
    assert lib.segmif_planes16_from_f32(None, 64, None, 1, 8, 32, 4, 0, 4, None, 1, None) == -22
    assert lib.segmif_planes16_pack_weight(None, 32, 64, 9, 576, None, None) == -22
    assert lib.segmif_gemm_split16_pack(None, 320, 320, 320, None, None) == -22
    assert lib.segmif_conv3x3_planes_f16x3(ctypes.byref(SegmifConvPlanes()), None, 1, None) == -22
    assert lib.segmif_gemm_split16_f32(ctypes.byref(SegmifGemmSplit()), None, 1, None) == -22


def test_guarded_scope_logic_on_the_host():
    """ops.run_guarded without a GPU (the guard's slots are an ordinary tensor): a scope whose slots stay inside
    [2^-13, 65504) returns its first result; one whose producer reports an overflow, a vanishing tensor or a NaN is run again
    with no guard active (nested scopes included) and counted; nested scopes join the outer one; all-zero tensors pass."""
    from segmif_amd import ops

    def producer(value):  # what a kernel does with its slot: atomic max on the bit pattern
        g = ops.active_guard()
        if g is None:
            return "bf16x6"
        ptr, nimg = g.slot()
        assert nimg == 1
        g.amax.view(-1)[(ptr - g.amax.data_ptr()) // 4] = int(torch.tensor([value], dtype=torch.float32).view(torch.int32))
        return "f16x3"

    base = ops.range_fallbacks()
    assert ops.run_guarded(lambda: producer(1.0), "cpu", enabled=True) == "f16x3"
    assert ops.run_guarded(lambda: producer(0.0), "cpu", enabled=True) == "f16x3"            # an all-zero tensor
    assert ops.run_guarded(lambda: producer(1.0), "cpu", enabled=False) == "bf16x6"          # modes that never ask
    assert ops.range_fallbacks() == base and ops.active_guard() is None
    for k, bad in enumerate((65504.0, 7.0e4, float("inf"), float("nan"), 2.0 ** -14, 1.0e-30)):
        calls = []

        def body(bad=bad, calls=calls):
            calls.append(ops.active_guard() is not None)
            first = ops.run_guarded(lambda: producer(3.0), "cpu", enabled=True)              # nested: joins / is suppressed
            return first, producer(bad)

        assert ops.run_guarded(body, "cpu", enabled=True) == ("bf16x6", "bf16x6"), bad
        assert calls == [True, False] and ops.range_fallbacks() == base + k + 1 and ops.active_guard() is None
    with pytest.raises(ZeroDivisionError):  # an exception inside a scope must not leave a guard behind
        ops.run_guarded(lambda: 1 / 0, "cpu", enabled=True)
    assert ops.active_guard() is None
    g = ops.Planes16Guard("cpu")
    slots = {g.slot()[0] for _ in range(g.SLOTS + 5)}
    assert len(slots) == g.SLOTS and g.used == g.SLOTS  # past the end the last slot is shared


def test_guarded_scope_repeats_only_the_images_that_tripped():
    """Per-image range slots on the host: a producer that indexes its row by image marks single images; run_guarded hands
    exactly those to `redo` (with the f16x3 kernels switched off) and keeps the others' first results; a launch that does
    not index by image stands for the whole batch; when every image tripped the scope runs again as a whole."""
    from segmif_amd import ops
    B = 5

    def bits(v):
        return int(torch.tensor([v], dtype=torch.float32).view(torch.int32))

    def producer(values, per_image=True):
        g = ops.active_guard()
        if g is None:
            return ["bf16x6"] * B
        ptr, nimg = g.slot(B if per_image else None)
        row = (ptr - g.amax.data_ptr()) // (4 * g.images)
        assert nimg == (B if per_image else 1) and g.images == B
        for b, v in enumerate(values if per_image else values[:1]):
            g.amax[row, b] = bits(v)
        return ["f16x3"] * B

    seen = []

    def redo(out, idx):
        assert ops.active_guard() is None
        seen.append(idx.tolist())
        for i in idx.tolist():
            out[i] = "bf16x6"
        return out

    s0 = ops.range_stats()
    out = ops.run_guarded(lambda: producer([1.0, 7.0e4, 0.5, float("nan"), 0.0]), "cpu", enabled=True, images=B, redo=redo)
    assert out == ["f16x3", "bf16x6", "f16x3", "bf16x6", "f16x3"] and seen == [[1, 3]]
    out = ops.run_guarded(lambda: producer([1.0, 2.0, 3.0, 4.0, 5.0]), "cpu", enabled=True, images=B, redo=redo)
    assert out == ["f16x3"] * B and len(seen) == 1
    out = ops.run_guarded(lambda: producer([1.0e-9], per_image=False), "cpu", enabled=True, images=B, redo=redo)
    assert out == ["bf16x6"] * B and len(seen) == 1  # a whole-batch row: everything again, as one forward
    out = ops.run_guarded(lambda: producer([float("inf")] * B), "cpu", enabled=True, images=B, redo=redo)
    assert out == ["bf16x6"] * B and len(seen) == 1
    s1 = ops.range_stats()
    assert s1["scopes"] - s0["scopes"] == 4 and s1["fallbacks"] - s0["fallbacks"] == 3
    assert s1["images"] - s0["images"] == 4 * B and s1["images_repeated"] - s0["images_repeated"] == 2 + B + B
    assert 0.0 < s1["trip_rate"] <= 1.0


def test_guarded_scope_repeats_ill_conditioned_images_with_exact_convs():
    """(r5) The conditioning half of the guard on the host: crosspath_fold raises one word per image and interaction to the
    softmax's kappa; an image whose estimate COND_EPS (k1 + k2 + k1 k2) passes COND_BOUND (or is NaN) and that stayed in range is
    handed to `redo` with the 3x3 convs switched to exact fp32 and the f16x3 kernels off; ONE large kappa alone does not trip; a
    range trip of the same image wins (bf16x6 repeat only); the statistics count the two kinds separately."""
    from segmif_amd import ops
    B = 4

    def bits(v):
        return int(torch.tensor([v], dtype=torch.float32).view(torch.int32))

    def producer(ranges, k1, k2):
        g = ops.active_guard()
        if g is None:
            return [ops.conv3x3_mode()] * B
        ptr, _ = g.slot(B)
        row = (ptr - g.amax.data_ptr()) // (4 * g.images)
        base = g.amax.data_ptr() + 4 * g.images * g.SLOTS
        assert g.cond_slot(B + 1) is None
        g.next_interaction()
        assert g.cond_slot(B) == base
        g.next_interaction()
        assert g.cond_slot(B) == base + 4 * g.images
        g.next_interaction()
        assert g.cond_slot(B) == base + 4 * g.images      # every later interaction shares the second row
        for b in range(B):
            g.amax[row, b] = bits(ranges[b])
            g.amax[g.SLOTS, b] = bits(k1[b])
            g.amax[g.SLOTS + 1, b] = bits(k2[b])
        return ["f16x3"] * B

    seen = []

    def redo(out, idx):
        assert ops.active_guard() is None
        seen.append((ops.conv3x3_mode(), idx.tolist()))
        for i in idx.tolist():
            out[i] = ops.conv3x3_mode()
        return out

    G = ops.Planes16Guard
    big = (G.COND_BOUND / G.COND_EPS) ** 0.5 * 2.0   # k1 = k2 = big: estimate ~ 4 x the bound
    alone = 0.5 * G.COND_BOUND / G.COND_EPS          # one interaction alone at half the bound
    mode0 = ops.conv3x3_mode()
    s0 = ops.range_stats()
    out = ops.run_guarded(lambda: producer([1.0, 1.0, 7.0e4, 1.0], [alone, big, big, float("nan")], [0.0, big, big, 1.0]), "cpu",
                          enabled=True, images=B, redo=redo)
    assert out == ["f16x3", "fp32", mode0, "fp32"] and seen == [(mode0, [2]), ("fp32", [1, 3])]
    assert ops.conv3x3_mode() == mode0
    out = ops.run_guarded(lambda: producer([1.0] * B, [0.0, alone, 3.0, 0.0], [alone, 0.0, 3.0, 0.0]), "cpu", enabled=True, images=B, redo=redo)
    assert out == ["f16x3"] * B and len(seen) == 2
    out = ops.run_guarded(lambda: producer([1.0] * B, [big] * B, [big] * B), "cpu", enabled=True, images=B, redo=redo)
    assert out == ["fp32"] * B and len(seen) == 2                       # every image: one whole repeat, exact convs
    out = ops.run_guarded(lambda: producer([1.0] * B, [big] * B, [big] * B), "cpu", enabled=True, images=B)
    assert out == ["fp32"] * B and ops.conv3x3_mode() == mode0          # no redo given: the same
    s1 = ops.range_stats()
    assert s1["images_repeated"] - s0["images_repeated"] == 1 and s1["images_repeated_fp32conv"] - s0["images_repeated_fp32conv"] == 2 + B + B
    assert s1["cond_repeat_rate"] > 0.0


def test_guard_state_is_per_thread():
    """Two threads in guarded scopes at once each see their own guard (the state used to be module globals)."""
    import threading
    from segmif_amd import ops
    seen, gate = {}, threading.Barrier(2)

    def work(name):
        def body():
            gate.wait(timeout=30)
            seen[name] = id(ops.active_guard())
            gate.wait(timeout=30)
            return name
        assert ops.run_guarded(body, "cpu", enabled=True) == name

    ts = [threading.Thread(target=work, args=(n,)) for n in ("a", "b")]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert len(seen) == 2 and seen["a"] != seen["b"] and ops.active_guard() is None


def test_f16x3_operand_format_in_numpy():
    """The f16x3 format's definition restated in numpy (the kernels' split: planes16.h; the weight scale:
    planes16_pack_weight / gemm_split16_pack): a half pair carries an fp32 value to 2^-23 relative for 2^-12 <= |x| < 65504
    and to 2^-36 absolute below; three products with fp32 accumulation stay within 1.5x of a plain fp32 chain's error
    against fp64 over a K = 1008 contraction (relative to each output's conditioning)."""
    import numpy as np
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-7, 4.5, 200000)).astype(np.float32)
    x = x[np.abs(x) < 65504]
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    back = hi.astype(np.float64) + lo.astype(np.float64) / 2048
    err = np.abs(back - x.astype(np.float64))
    big = np.abs(x) >= 2.0 ** -12
    assert (err[big] / np.abs(x[big])).max() <= 2.0 ** -23 and err[~big].max() <= 2.0 ** -36

    def mm32(a, b):
        return a.astype(np.float32) @ b.astype(np.float32)

    M, K, N = 512, 1008, 32
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (0.03 * rng.standard_normal((K, N)) * 10.0 ** rng.uniform(-2, 1, (1, N))).astype(np.float32)
    e = 14 - np.floor(np.log2(np.abs(w).max(axis=0, keepdims=True)))  # the pack kernels: 14 - ilogb(max |w[n][.]|)
    W = (w * np.exp2(e)).astype(np.float32)
    assert (np.abs(W).max(axis=0) >= 2.0 ** 14).all() and (np.abs(W).max(axis=0) < 2.0 ** 15).all()
    W0 = W.astype(np.float16)
    Wl = (W - W0.astype(np.float32)).astype(np.float16)
    W0s = (W0.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
    a0 = a.astype(np.float16)
    al = ((a - a0.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    y = ((mm32(al, W0s) + mm32(a0, Wl)) + mm32(a0, W0)) * np.exp2(-e).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    cond = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
    e16 = (np.abs(y - ref) / cond).max()
    e32 = (np.abs(mm32(a, w) - ref) / cond).max()
    assert e16 <= 1.5 * e32 and e16 < 4e-7, (e16, e32)


def test_invalid_descriptors_are_rejected_without_a_gpu(lib):
    from segmif_amd._lib import SegmifIgemm
    d = SegmifIgemm()
    assert lib.segmif_igemm_f32(ctypes.byref(d), None) == -22  # SEGMIF_EINVAL: null pointers
    assert lib.segmif_layernorm_f32(None, None, None, None, 4, 64, 64, 64, 1e-5, None) == -22
    assert lib.segmif_sr_attention_f32(None, None, None, None, 1, 1, 1, 1, 64, 64, 128, 64, 0.125, None) == -22


def test_missing_library_is_loud(monkeypatch):
    from segmif_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsegmif_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_module_mirror_keys_and_cpu_refusal(golden_dir):
    from segmif_amd.core import Fusion_Network3_ac, Network3
    import segmif_amd.core as core
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for bb in ("mit_b0", "mit_b1"):
        n = Network3(bb, 9, pretrained=None)
        assert {k: list(v.shape) for k, v in n.state_dict().items()} == keys["Network3:" + bb]
    f = Fusion_Network3_ac()
    assert {k: list(v.shape) for k, v in f.state_dict().items()} == keys["Fusion_Network3_ac"]
    assert core.Network is core.Network3  # the reference's phantom export resolves here
    groups = n.denoise_net.get_param_groups()
    assert [len(g) for g in groups] == [len([1 for k, _ in n.denoise_net.encoder.named_parameters() if "norm" not in k]),
                                        len([1 for k, _ in n.denoise_net.encoder.named_parameters() if "norm" in k]),
                                        len(list(n.denoise_net.decoder.parameters())) + 1]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f(torch.zeros(1, 1, 8, 8), torch.zeros(1, 3, 8, 8), torch.zeros(1, 64, 8, 8), torch.zeros(1, 128, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        n(torch.zeros(1, 3, 32, 32))


def test_packed_cache_invalidation():
    from segmif_amd.core._util import PackedCache
    calls = []
    p = torch.nn.Parameter(torch.ones(3))
    c = PackedCache()
    f = lambda t: calls.append(1) or (t.detach() * 2)
    a = c.get("w", p, f)
    assert c.get("w", p, f) is a and len(calls) == 1
    with torch.no_grad():
        p.add_(1.0)  # optimizer-style in-place update bumps _version
    b = c.get("w", p, f)
    assert len(calls) == 2 and torch.equal(b, torch.full((3,), 4.0))
    import copy
    assert copy.deepcopy(c)._entries == {}


def test_oracle_is_not_imported_by_the_product():
    """The product package must never reach into oracle/ (judge's rule; grep-level check)."""
    pkg = os.path.join(ROOT, "segmif_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                text = open(os.path.join(dirpath, fn)).read()
                assert "segmif_oracle" not in text and "detweights" not in text and "oracle" not in text.lower().replace(
                    "cpu oracle", "").replace("the oracle", ""), os.path.join(dirpath, fn)


def test_reference_import_lines_resolve():
    """train.py:18,111-112 / test_fusion.py:9 import lines work against the mirror package."""
    import importlib
    import segmif_amd.core as core
    for name in ("core", "core.model_fusion", "core.mix_transformer", "core.segformer_head", "core.loss"):
        sys.modules.pop(name, None)
    sys.modules["core"] = core
    sys.modules["core.model_fusion"] = core.model_fusion
    try:
        ns = {}
        exec("from core.model_fusion import Fusion_Network3_ac, Network3, Mean\n"
             "from core import Total_fusion_loss, Total_fusion_loss2, RGB2YCrCb, Fusionloss, Fusionloss_add, "
             "Fusionloss2, Fusionloss3, Fusionloss4, Fusionloss_grad3\n"
             "from core import SegFormerHead, WeTr, mit_b3", ns)
        assert ns["Network3"] is core.Network3 and callable(ns["Fusionloss_grad3"])
        with pytest.raises(NotImplementedError):
            ns["Fusionloss4"]()(None)
    finally:
        sys.modules.pop("core", None)
        sys.modules.pop("core.model_fusion", None)


def test_losses_match_reference_formulas_on_cpu():
    """Fusionloss3 / Fusionloss_grad3 restated (segmif_amd/losses.py) vs the reference formulas written
    with library convolutions (core/loss.py:459-476, 506-517, 634-650; pytorch_ssim/__init__.py:19-43)."""
    import torch.nn.functional as F
    from segmif_amd import losses
    g = torch.Generator().manual_seed(3)
    a = torch.rand(2, 1, 21, 33, generator=g, dtype=torch.float64)
    m = torch.rand(2, 3, 21, 33, generator=g, dtype=torch.float64)
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]], dtype=torch.float64)[None, None]
    ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]], dtype=torch.float64)[None, None]
    sob = lambda t: F.conv2d(t, kx, padding=1).abs() + F.conv2d(t, ky, padding=1).abs()
    assert torch.allclose(losses.sobel_xy(a), sob(a), atol=1e-12)
    ref3 = F.l1_loss(m[:, :1], a) + F.l1_loss(sob(m[:, :1]), sob(a))
    assert abs(float(losses.fusion_loss3(a, m)) - float(ref3)) < 1e-12
    assert 0.0 < float(losses.ssim(a, m[:, :1])) < 1.0 and abs(float(losses.ssim(a, a)) - 1.0) < 1e-9


def test_compute_results_host_port_matches_reference(golden_dir):
    """segmif_amd.utils.metrics.compute_results (host arithmetic on the K x K matrix) against the
    values util/util.py:31-55 produced, NaN pattern included; device entry points refuse CPU tensors."""
    import numpy as np
    from segmif_amd.utils import metrics
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for conf in (g["conf"], torch.from_numpy(g["conf"])):
        for got, name in zip(metrics.compute_results(conf), ("precision", "recall", "iou")):
            assert np.array_equal(np.isnan(got), np.isnan(g[name])), name
            assert np.allclose(np.nan_to_num(got), np.nan_to_num(g[name]), rtol=0, atol=1e-15), name
    with pytest.raises(RuntimeError):
        metrics.confusion_matrix(torch.zeros(4, dtype=torch.int32), torch.zeros(4, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        metrics.quantize_fused(torch.zeros(1, 3, 4, 4))


def test_struct_layout_of_the_round4_igemm_fields_matches_c(lib, tmp_path):
    """SegmifIgemm's round-4 tail (relu_mask .. mask_zstride: the DRDB / CrossPath backward epilogues and the f16x3 training convs'
    range slots): size and every new field's offset against gcc."""
    from segmif_amd._lib import SegmifIgemm
    names = ["relu_mask", "ld_mask", "split_f16", "split_in_amax", "split_in_amax_n", "split_out_amax", "split_out_amax_n",
             "wgrad_dy_amax", "wgrad_dy_amax_n", "mask_zstride"]
    src = tmp_path / "layout_r4.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "segmif_hip.h"\nint main(){printf("%zu' + " %zu" * len(names) + '\\n",sizeof(SegmifIgemm)'
                   + "".join(f",offsetof(SegmifIgemm,{n})" for n in names) + ");return 0;}")
    exe = tmp_path / "layout_r4"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(SegmifIgemm)] + [getattr(SegmifIgemm, n).offset for n in names]


def test_round4_training_entry_points_size_rules_and_rejections_without_a_gpu(lib):
    """segmif_conv3x3_split16_* / segmif_amax_f32 / the residual LayerNorm pair / act_bwd2 / prelu_bwd_rows: buffer sizes and
    EINVAL on null / malformed arguments - no kernel is launched."""
    # f16x3 image = the bf16x6 image's 96-byte rows + one float per PADDED output channel (32- or 64-wide column tiles)
    assert lib.segmif_conv3x3_split16_weight_bytes(32, 64) == lib.segmif_conv3x3_split_weight_bytes(32, 64) + 4 * 32
    assert lib.segmif_conv3x3_split16_weight_bytes(48, 64) == lib.segmif_conv3x3_split_weight_bytes(48, 64) + 4 * 64
    assert lib.segmif_conv3x3_split16_weight_bytes(32, 24) == 0
    assert lib.segmif_conv3x3_split16_pack(None, 32, 64, 576, None, None) == -22
    assert lib.segmif_amax_f32(None, 10, 64, 64, None, 8, None) == -22
    assert lib.segmif_add_layernorm_f32(None, None, None, 0, None, None, None, None, 4, 64, 64, 64, 64, 64, 1e-5, None) == -22
    assert lib.segmif_layernorm_bwd_add_f32(None, None, None, None, 0, None, 0, None, None, 0, None, 4, 64, 64, 64, 64, 1e-5, None) == -22
    assert lib.segmif_act_bwd2_f32(None, None, None, None, 4, 64, 64, 64, 64, 64, 1, None, None) == -22
    assert lib.segmif_prelu_bwd_rows_f32(None, 64, None, None, None, None, None, 4, 64, None) == -22


def test_device_made_range_arithmetic_emulated_in_numpy():
    """The training convs' f16x3 arithmetic (csrc/conv3x3_split.hip F16, csrc/wgrad.hip F16; planes16.h range_scale) restated in
    numpy: the operand is scaled by the power of two that puts its maximum in [2^13, 2^14), split into half pairs, multiplied
    in three products with fp32 accumulation and descaled - for activations of order 1, gradients of order 1e-7 and values of
    order 1e4 the result is as close to fp64 as plain fp32 arithmetic is, where unscaled halves of the small tensor are ~100x off."""
    rng = np.random.default_rng(0)

    def range_scale(x):  # planes16.h: exponent E of max |x| -> 2^(13 - E); all zero -> 1
        mx = np.abs(x).max()
        return np.float32(1.0) if mx == 0 else np.float32(2.0 ** (13 - int(np.floor(np.log2(mx)))))

    def mm32(a, b):
        return a.astype(np.float32) @ b.astype(np.float32)

    def conv_form(x, w, scaled=True):  # activations: hi + 2^-11 lo'; weights: rows scaled to 2^14, planes W0 | Wl | 2^-11 W0
        s = range_scale(x) if scaled else np.float32(1.0)
        xs = (x * s).astype(np.float32)
        hi = xs.astype(np.float16)
        lo = ((xs - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        e = 14 - np.floor(np.log2(np.abs(w).max(axis=0, keepdims=True)))
        ws = (w * np.exp2(e)).astype(np.float32)
        w0 = ws.astype(np.float16)
        wl = (ws - w0.astype(np.float32)).astype(np.float16)
        w0s = (w0.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
        acc = mm32(lo, w0s) + mm32(hi, wl) + mm32(hi, w0)
        return acc * np.exp2(-e).astype(np.float32) / s

    def wgrad_form(dy, x):  # both operands scaled, low halves UNscaled, products hi.hi + hi.lo + lo.hi in one accumulator
        sy, sx = range_scale(dy), range_scale(x)
        a, b = (dy * sy).astype(np.float32), (x * sx).astype(np.float32)
        ah, bh = a.astype(np.float16), b.astype(np.float16)
        al, bl = (a - ah.astype(np.float32)).astype(np.float16), (b - bh.astype(np.float32)).astype(np.float16)
        acc = mm32(al.T, bh) + mm32(ah.T, bl) + mm32(ah.T, bh)
        return acc / sy / sx

    M, K, N = 4096, 288, 32
    w = (0.05 * rng.standard_normal((K, N)) * np.logspace(-3, 1, N)[None, :]).astype(np.float32)
    for scale in (1.0, 3e-7, 2e4):
        x = (scale * rng.standard_normal((M, K))).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64)
        yard = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)  # each output's conditioning
        e16 = (np.abs(conv_form(x, w) - ref) / yard).max()
        e32 = (np.abs(mm32(x, w) - ref) / yard).max()
        assert e16 < 4 * e32 + 1e-7, (scale, e16, e32)
        if scale < 1e-3:  # what the scale is for: the high halves of a 1e-7 tensor without it are subnormal (two or three bits)
            assert (np.abs(conv_form(x, w, scaled=False) - ref) / yard).max() > max(20 * e16, 1e-6)
    dy = (3e-7 * rng.standard_normal((M, N))).astype(np.float32)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ref = dy.astype(np.float64).T @ x.astype(np.float64)
    yard = np.abs(dy).astype(np.float64).T @ np.abs(x).astype(np.float64)
    e16 = (np.abs(wgrad_form(dy, x) - ref) / yard).max()
    e32 = (np.abs(mm32(dy.T, x) - ref) / yard).max()
    assert e16 < 4 * e32 + 1e-7, (e16, e32)
    assert range_scale(np.zeros(4, np.float32)) == 1.0 and 2 ** 13 <= 0.3 * range_scale(np.float32([0.3])) < 2 ** 14


def test_join_node_and_gradient_sink_host_logic():
    """Host-side pieces of the training path that involve no kernel: JoinFn (producers wrote channel slices of one buffer; the node
    hands on the buffer and splits its gradient into views - nothing is copied) and ProjSink (which gradient buffer a consumer's
    GEMM epilogue writes a masked half into, and how CrossProjFn recognises it)."""
    from segmif_amd import autograd as ag
    whole = torch.zeros(2, 3, 5, 12)
    a = (whole[..., :4].detach() + 0).requires_grad_()   # stand-ins for producers' outputs ...
    b = (whole[..., 4:].detach() + 0).requires_grad_()
    with pytest.raises(RuntimeError, match="not the expected channel slice"):
        ag.join(ag.Out(whole), a, b)                      # ... that do NOT live in the buffer: refused

    class Place(torch.autograd.Function):                 # a producer that writes into its placement, like LayerNormFn(out=)
        @staticmethod
        def forward(ctx, x, out):
            out.t.copy_(x)
            return out.t

        @staticmethod
        def backward(ctx, g):
            Place.seen.append((g.data_ptr(), tuple(g.stride())))
            return g, None
    Place.seen = []
    xa, xb = torch.randn(2, 3, 5, 4, requires_grad=True), torch.randn(2, 3, 5, 8, requires_grad=True)
    pa, pb = Place.apply(xa, ag.Out(whole[..., :4])), Place.apply(xb, ag.Out(whole[..., 4:]))
    j = ag.join(ag.Out(whole), pa, pb)
    assert j.data_ptr() == whole.data_ptr() and torch.equal(j[..., :4], xa.detach()) and torch.equal(j[..., 4:], xb.detach())
    g = torch.randn(2, 3, 5, 12)
    j.backward(g)
    assert torch.equal(xa.grad, g[..., :4]) and torch.equal(xb.grad, g[..., 4:])
    assert sorted(p for p, _ in Place.seen) == sorted([g.data_ptr(), g[..., 4:].data_ptr()])  # views of the gradient, no copies
    with pytest.raises(RuntimeError, match="do not cover"):
        ag.join(ag.Out(whole), Place.apply(xa.detach(), ag.Out(whole[..., :4])))

    sink = ag.ProjSink()
    sink.p = [torch.rand(2, 7, 128) for _ in range(3)]
    out, mask = sink.slot(1, 1)
    assert out.shape == (2, 7, 64) and out.data_ptr() == sink.dz[1][..., 64:].data_ptr() and mask.data_ptr() == sink.p[1][..., 64:].data_ptr()
    assert not sink.holds(1, 1, out)            # not marked yet
    sink.done.add((1, 1))
    assert sink.holds(1, 1, out) and not sink.holds(1, 0, out) and not sink.holds(1, 1, out.clone()) and not sink.holds(1, 1, None)
    assert sink.slot(1, 0)[0].data_ptr() == sink.dz[1].data_ptr()  # the other half lives in the same buffer
    sink.reset()
    assert sink.dz == [None] * 3 and not sink.done




def test_lazy_gram_geometry_holds_for_every_admitted_size():
    """csrc/crosspath.hip, crosspath_gram_lazy_kernel: an aligned group of four output pixels reads source columns c0, c0 + 1,
    c0 + 2 (clamped at the right edge), c0 = the first pixel's left tap, and a pixel takes (c0, c0 + 1) or (c0 + 1, c0 + 2).  The C
    entry point admits W % 4 == 0 and 3 iw <= W.  Here the kernel's fp32 arithmetic (bilinear_kernel's: src = max(s (x + 0.5)
    - 0.5, 0), s = iw / W in fp32) is replayed in numpy for every admitted (iw, W) up to W = 1024: the pixel's own taps
    (x0, min(x0 + 1, iw - 1)) are always the pair the group scheme selects.  (The tested sizes on the GPU are a handful; this is
    the statement for all of them.  The same scan up to W = 16 384 finds no exception either: the real-arithmetic margin to an
    integer boundary is a multiple of 1 / (2 W), far above fp32's error in the coordinate at these widths.)"""
    import numpy as np
    f32 = np.float32
    checked = 0
    for W in range(4, 1025, 4):
        xx = np.arange(W, dtype=np.int64)
        lead = xx & ~3
        for iw in sorted({1, 2, 3, W // 8, W // 4, W // 3, max(1, W // 3 - 1), max(1, W // 5), max(1, (W * 3) // 10)}):
            if iw < 1 or 3 * iw > W:
                continue
            s = f32(iw) / f32(W)
            fx = np.maximum(s * (xx.astype(f32) + f32(0.5)) - f32(0.5), f32(0)).astype(f32)
            fl = np.maximum(s * (lead.astype(f32) + f32(0.5)) - f32(0.5), f32(0)).astype(f32)
            x0, c0 = fx.astype(np.int64), fl.astype(np.int64)
            x1 = np.minimum(x0 + 1, iw - 1)
            ck = x0 - c0
            assert ck.min() >= 0 and ck.max() <= 1, (W, iw, ck.min(), ck.max())
            cols = np.stack([c0, np.minimum(c0 + 1, iw - 1), np.minimum(c0 + 2, iw - 1)])  # what the group loads
            left = np.where(ck == 1, cols[1], cols[0])
            right = np.where(ck == 1, cols[2], cols[1])
            assert np.array_equal(left, x0) and np.array_equal(right, x1), (W, iw)
            assert x0.max() <= iw - 1
            checked += 1
    assert checked > 1500


def test_resize_commutes_with_the_linears_the_lazy_path_moves_across_it():
    """What ops.LazySeg rests on (core/mix_transformer.py:364-373 followed by core/model_fusion.py:351-353): a bilinear resize is a
    convex combination per channel (weights sum to 1), so conv3 / conv4 (1 x 1, bias) and channel_proj3 (Linear, bias) give the
    same tensor before or after it - in float64 to rounding, at x 4, x 8 and a non-integer enlargement.  The ReLU does NOT commute,
    which is why the kernels resize the projected rows and apply it afterwards."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7)
    for (ih, iw, H, W) in ((6, 8, 24, 32), (5, 7, 40, 56), (7, 9, 30, 52)):
        x = torch.randn(2, 64, ih, iw, generator=g, dtype=torch.float64)
        w = torch.randn(128, 64, generator=g, dtype=torch.float64) * 0.2
        b = torch.randn(128, generator=g, dtype=torch.float64)
        up = lambda t: F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
        lin = lambda t: F.conv2d(t, w[:, :, None, None], b)
        a, c = lin(up(x)), up(lin(x))
        assert float((a - c).abs().max()) < 1e-12 * float(a.abs().max()) + 1e-13
        assert float((torch.relu(a) - up(torch.relu(lin(x)))).abs().max()) > 1e-3  # (the ReLU has to come after the resize)
