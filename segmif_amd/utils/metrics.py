"""Evaluation metrics and image write-out on the device.

  confusion_matrix   test_segmentation.py:176-177 (sklearn confusion_matrix(labels=[0..K-1]), accumulated)
  compute_results    util/util.py:31-55 (per-class precision / recall / IoU, NaN for empty classes)
  quantize_fused     test_fusion.py:112-120 (uint8(255 x) -> global min-max rescale -> uint8, NHWC)
"""
import ctypes

import numpy as np
import torch

from .. import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"segmif_amd: {name} must be a tensor on the MI355X device (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"segmif_amd: {name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def confusion_matrix(pred, label, n_class=9, out=None):
    """pred: int32 labels (segmif_amd.ops.argmax_nhwc / Network3.predict_labels), label: int64 ground
    truth, same number of elements.  Returns / accumulates into an (n_class, n_class) int64 device tensor:
    rows = ground truth, columns = prediction."""
    pred, label = _dev(pred, "pred", torch.int32), _dev(label, "label", torch.int64)
    if pred.numel() != label.numel():
        raise RuntimeError(f"pred has {pred.numel()} elements, label {label.numel()}")
    if out is None:
        out = torch.zeros((n_class, n_class), device=pred.device, dtype=torch.int64)
    elif tuple(out.shape) != (n_class, n_class) or out.dtype != torch.int64 or not out.is_cuda or not out.is_contiguous():
        raise RuntimeError("out must be a contiguous (n_class, n_class) int64 device tensor")
    _lib.check(_lib.load().segmif_confusion_i32(pred.data_ptr(), label.data_ptr(), out.data_ptr(), pred.numel(), n_class,
                                                _stream()), "segmif_confusion_i32")
    return out


def compute_results(conf_total):
    """-> (precision, recall, iou) per class as float64 arrays; a class with an empty denominator gets NaN.
    Host-side arithmetic on the K x K matrix (accepts a device tensor or an array)."""
    conf = conf_total.detach().cpu().numpy() if isinstance(conf_total, torch.Tensor) else np.asarray(conf_total)
    conf = conf.astype(np.float64)
    tp = np.diag(conf)
    predicted = conf.sum(axis=0)  # column sums: everything predicted as the class
    actual = conf.sum(axis=1)  # row sums: everything that is the class
    with np.errstate(invalid="ignore", divide="ignore"):
        precision = np.where(predicted == 0, np.nan, tp / predicted)
        recall = np.where(actual == 0, np.nan, tp / actual)
        union = actual + predicted - tp
        iou = np.where(union == 0, np.nan, tp / union)
    return precision, recall, iou


def quantize_fused(fused):
    """fused: (B, C, H, W) fp32 in [0, 1] (core.fuse_to_rgb output) -> (B, H, W, C) uint8 exactly as the
    reference's script writes it to disk (global min / max over the batch)."""
    fused = _dev(fused, "fused", torch.float32)
    if fused.dim() != 4:
        raise RuntimeError("quantize_fused expects (B, C, H, W)")
    B, C, H, W = fused.shape
    out = torch.empty((B, H, W, C), device=fused.device, dtype=torch.uint8)
    mm = torch.empty((2,), device=fused.device, dtype=torch.int32)
    _lib.check(_lib.load().segmif_quantize_u8(fused.data_ptr(), out.data_ptr(), mm.data_ptr(), B, C, H * W, _stream()),
               "segmif_quantize_u8")
    return out


def dequantize_fused(q):
    """(B, H, W, C) uint8 (quantize_fused output = the PNG pixels) -> (B, C, H, W) fp32 = u / 255, the image
    test_segmentation.py's loader hands to the network (TaskFusion_dataset2.py:84-88)."""
    q = _dev(q, "q", torch.uint8)
    if q.dim() != 4 or not q.is_contiguous():
        raise RuntimeError("dequantize_fused expects a contiguous (B, H, W, C) uint8 tensor")
    B, H, W, C = q.shape
    out = torch.empty((B, C, H, W), device=q.device, dtype=torch.float32)
    _lib.check(_lib.load().segmif_dequantize_u8(q.data_ptr(), out.data_ptr(), B, C, H * W, _stream()), "segmif_dequantize_u8")
    return out
