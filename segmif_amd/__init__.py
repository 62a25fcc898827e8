"""segmif_amd — MI355X (gfx950) native implementation of SegMiF's fusion + segmentation hot path.

The arithmetic lives in hand-written HIP kernels behind a C ABI (include/segmif_hip.h, built by
segmif_amd.build into segmif_amd/lib/libsegmif_hip.so); `segmif_amd.core` mirrors the reference's
`core` package (same nn.Module names, constructor/forward signatures and state_dict keys) on top
of them.  PyTorch is used for device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
