"""Weight-derived tensors of the training path, built as ONE launch per step (EXPERIMENTAL, off by default:
SEGMIF_WEIGHT_PREP=1; round 4: bitwise equal to the one-by-one path on the GPU and NO faster - the segmentation step is bound by its
kernels, not by launches, and the naive gather gives back what the launches saved: profiles/r04_weight_prep_ab.txt, DESIGN.md
section 7).

Every training step re-derives small tensors from the parameters it is about to differentiate: W^T for the input-gradient
GEMM of each Linear (core/mix_transformer.py's q / kv / proj / fc1 / fc2 ...), the tap-major [9][C] form of each depthwise
weight and its flipped twin, the (ky, kx, c)-major form of the spatial-reduction convs' weights.  One by one they are ~300
launches of a few microseconds per segmentation step - what is left of torch's own kernels there.  Each of them is a strided
view of the parameter made contiguous, so all of them can be one gather launch (csrc/backward.hip gather_copy_kernel).

How it stays correct without any call discipline: an entry is served only while the parameter is the SAME object at the SAME
address with the SAME `_version` it was built from (FusedAdamW bumps the counters after its raw-pointer update, an in-place
torch op bumps them itself); anything else is a miss, and a miss simply builds the tensor the old way and remembers the request
for the next `begin_step()`.  So the first step runs exactly as before, and a step that forgets `begin_step()` is merely slow.
"""
import ctypes
import os
import weakref

import torch

_CHUNK = 16384


class _GatherEntry(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("n", ctypes.c_int64), ("shape", ctypes.c_int32 * 4),
                ("stride", ctypes.c_int64 * 4)]


def _hip_launch(entries, chunk_entry, chunk_off, device, items):
    """One gather_copy launch over `entries` (list of (src_ptr, dst_ptr, n, shape4, stride4)); `items` (the same requests as
    (param, dst tensor, shape, strides, offset)) is for launchers that cannot follow raw pointers - the CPU test's."""
    from . import _lib
    lib = _lib.load()
    assert lib.segmif_gather_entry_bytes() == ctypes.sizeof(_GatherEntry)
    table = (_GatherEntry * len(entries))()
    for i, (src, dst, n, shape, stride) in enumerate(entries):
        e = table[i]
        e.src, e.dst, e.n = src, dst, n
        for d in range(4):
            e.shape[d], e.stride[d] = shape[d], stride[d]
    t = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(device)
    ce = torch.tensor(chunk_entry, dtype=torch.int32, device=device)
    co = torch.tensor(chunk_off, dtype=torch.int64, device=device)
    _lib.check(lib.segmif_gather_copy_f32(t.data_ptr(), ce.data_ptr(), co.data_ptr(), len(chunk_entry), _CHUNK,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "segmif_gather_copy_f32")
    return t, ce, co  # kept alive by the caller until the next batch


def spec_of_view(view):
    """(shape, element strides, storage offset relative to the parameter's first element) of a strided VIEW of a parameter."""
    return tuple(view.shape), tuple(view.stride()), view.storage_offset()


class WeightPrep:
    def __init__(self, enabled=None, launcher=_hip_launch):
        self.enabled = (os.environ.get("SEGMIF_WEIGHT_PREP") == "1") if enabled is None else enabled
        self._launcher = launcher
        self._specs = {}   # (id(param), kind) -> (weakref(param), shape, strides, offset)
        self._cache = {}   # (id(param), kind) -> (tensor, weakref(param), version, data_ptr)
        self._keep = None
        self.hits = self.misses = self.batches = 0

    # ---- what the autograd Functions call --------------------------------------------------------------------------------
    def lookup(self, param, kind, spec_fn):
        """The contiguous derived tensor of `param` if this step's batch built it from its current value, else None - the caller
        then builds it as it always did, and the request (spec_fn(param) -> (shape, strides, offset), at most 4 dims, strides
        in elements relative to the parameter's own first element, negative allowed) is remembered for the next begin_step()."""
        if not self.enabled or not isinstance(param, torch.nn.Parameter) or param.dtype != torch.float32:
            return None
        key = (id(param), kind)
        hit = self._cache.get(key)
        if hit is not None:
            t, ref, version, ptr = hit
            if ref() is param and version == param._version and ptr == param.data_ptr():
                self.hits += 1
                return t
        self.misses += 1
        spec = self._specs.get(key)
        if spec is None or spec[0]() is not param:
            shape, strides, offset = spec_fn(param)
            if len(shape) > 4 or len(shape) != len(strides):
                raise RuntimeError(f"WeightPrep: a derived tensor of at most 4 dims expected, got shape {tuple(shape)}")
            self._specs[key] = (weakref.ref(param), tuple(int(v) for v in shape), tuple(int(v) for v in strides), int(offset))
        return None

    # ---- once per step, before the forward ----------------------------------------------------------------------------------
    def begin_step(self):
        """Build every remembered derived tensor from the parameters' CURRENT values: one launch per device."""
        if not self.enabled or not self._specs:
            return
        by_dev = {}
        for key, (ref, shape, strides, offset) in list(self._specs.items()):
            p = ref()
            if p is None:
                del self._specs[key]
                self._cache.pop(key, None)
                continue
            hit = self._cache.get(key)
            if hit is not None and hit[1]() is p and hit[2] == p._version and hit[3] == p.data_ptr():
                continue  # still current (a step that did not touch this parameter)
            by_dev.setdefault(p.device, []).append((key, p, shape, strides, offset))
        keep = []
        for dev, items in by_dev.items():
            total = sum(_pad(_numel(shape)) for _, _, shape, _, _ in items)
            flat = torch.empty((total,), device=dev, dtype=torch.float32)
            entries, chunk_entry, chunk_off, described = [], [], [], []
            pos = 0
            for i, (key, p, shape, strides, offset) in enumerate(items):
                n = _numel(shape)
                dst = flat[pos:pos + n].view(shape)
                pad = 4 - len(shape)
                entries.append((p.data_ptr() + 4 * offset, dst.data_ptr(), n, (1,) * pad + shape, (0,) * pad + strides))
                described.append((p, dst, shape, strides, offset))
                for off in range(0, n, _CHUNK):
                    chunk_entry.append(i)
                    chunk_off.append(off)
                self._cache[key] = (dst, weakref.ref(p), p._version, p.data_ptr())
                pos += _pad(n)  # every derived tensor starts on a 256-byte boundary (the kernels' 16-byte weight loads)
            keep.append((flat, self._launcher(entries, chunk_entry, chunk_off, dev, described)))
            self.batches += 1
        if keep:
            self._keep = keep


def _pad(n):
    return (n + 63) // 64 * 64


def _numel(shape):
    n = 1
    for v in shape:
        n *= v
    return n


PREP = WeightPrep()
