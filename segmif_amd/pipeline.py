"""The measured unit of work as one callable: IR + visible pair forward
(test_fusion.py:100-111 followed by test_segmentation.py:169-174, in memory), optionally
captured once into a hipGraph and replayed (removes ~700 host launches per step; matters for
small batches, where the step is launch-bound)."""
import torch

from . import ops
from .core.model_fusion import fuse_to_rgb
from .utils.metrics import dequantize_fused, quantize_fused


class PairForward:
    def __init__(self, seg_net, fusion_net, commute_resize=True, uint8_roundtrip=False):
        """uint8_roundtrip: the reference's scripted flow hands the fused image from test_fusion.py to
        test_segmentation.py through uint8 PNG files (uint8(255 x), global min-max rescale over the batch, uint8 -
        test_fusion.py:112-120; read back as float32 / 255 - TaskFusion_dataset2.py:84-88).  True reproduces that
        quantisation in memory: the segmentation net then sees exactly the pixels the script's PNGs hold
        and the returned `fused` is that de-quantised image.  False (default) keeps the fp32 image."""
        self.seg, self.fus = seg_net, fusion_net
        self.commute_resize = commute_resize
        self.uint8_roundtrip = uint8_roundtrip
        self._graph = None
        self._graph_guard = None
        self._static = None

    def eager(self, ir, vis, mask3):
        """-> (fused RGB (B,3,H,W), labels int32 (B,H,W)).  One guarded scope around the whole pair forward: the encoder's
        tall GEMMs and the fusion net's 3x3 convs run on f16x3 operands, their range slots - one per pair - are read back once
        at the end and exactly the pairs whose activations left the half's exponent range are computed again on the bf16x6
        kernels (a pair's result does not depend on what else is in the batch); (r5) pairs whose CrossPath context softmax
        report ill-conditioned columns in both interactions (Planes16Guard.cond_estimate > COND_BOUND) are computed again with the 3x3 convs in exact fp32."""
        return ops.run_guarded(lambda: self._eager_body(ir, vis, mask3), ir.device, images=ir.shape[0],
                               redo=lambda out, idx: self._redo(out, idx, ir, vis, mask3))

    def _redo(self, out, idx, ir, vis, mask3):
        """Recompute the pairs `idx` (run_guarded has switched the f16x3 kernels off) and patch them into `out`."""
        fused, labels = self._eager_body(ir.index_select(0, idx), vis.index_select(0, idx), mask3.index_select(0, idx))
        out[0].index_copy_(0, idx, fused)
        out[1].index_copy_(0, idx, labels)
        return out

    def _eager_body(self, ir, vis, mask3):
        enc = self.seg.denoise_net.encoder
        if self.commute_resize:  # conv3 / conv4 (1x1) before the bilinear resize: same function
            y_f = self.fus.forward_from_features(ir, vis, *enc.forward_fusion_features(mask3))
        else:
            out0, out1 = enc.forward_fusion(mask3)
            y_f = self.fus(ir, vis, out0, out1)
        fused = fuse_to_rgb(vis, y_f)
        if self.uint8_roundtrip:
            fused = dequantize_fused(quantize_fused(fused))
        return fused, self.seg.predict_labels(fused, vis.shape[2:])

    def capture(self, ir, vis, mask3, warmup=2):
        """Capture one step on static copies of the inputs; later calls to replay() copy new inputs
        into the static buffers and launch the graph."""
        if ops.launch_timer_active():
            raise RuntimeError("disable the launch timer before graph capture")
        self._static = [t.clone() for t in (ir, vis, mask3)]
        # The f16x3 kernels need a range guard, and a captured graph cannot hold run_guarded's host read-back: the graph gets
        # a guard of its own whose slots are cleared by a memset node at its start and filled by its kernels; replay() reads
        # them back after the launch and computes the pair again, eagerly on the bf16x6 kernels, if one tripped.
        guard = ops.Planes16Guard(ir.device, ir.shape[0]) if ops.f16x3_enabled() else None

        def body():
            if guard is None:
                return self._eager_body(*self._static)
            guard.reset()
            prev = ops.install_guard(guard)
            try:
                return self._eager_body(*self._static)
            finally:
                ops.install_guard(prev)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # packs weights, raises LDS limits, warms the allocator
                body()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._out = body()
        self._graph_guard = guard
        return self

    def replay(self, ir=None, vis=None, mask3=None):
        if self._graph is None:
            raise RuntimeError("call capture() first")
        for dst, src in zip(self._static, (ir, vis, mask3)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self._graph.replay()
        if self._graph_guard is not None:  # (one small read-back per replay)
            bad, sat = self._graph_guard.verdict()
            if bool(bad.any() or sat.any()):
                with torch.no_grad():
                    dev = self._static[0].device
                    return ops.finish_guarded(self._out, bad, sat, lambda: self._eager_body(*self._static), dev,
                                              redo=lambda out, idx: self._redo(out, idx, *self._static))
            ops.finish_guarded(None, bad, sat, None, None)  # (statistics only)
        return self._out

    def __call__(self, ir, vis, mask3):
        with torch.no_grad():
            return self.replay(ir, vis, mask3) if self._graph is not None else self.eager(ir, vis, mask3)
