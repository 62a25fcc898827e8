"""AdamW with warm-up + polynomial decay, on one multi-tensor HIP kernel.

Counterpart of the reference's utils/optimizer.py (PolyWarmupAdamW :3-33, PolyWarmupAdamW_seg
:36-66): same constructor arguments, same in-`step()` learning-rate schedule (per-group initial LR
times a warm-up or polynomial factor, written back into `param_groups`), same arithmetic as
torch.optim.AdamW(eps=1e-8) — decoupled weight decay, bias correction, parameters whose `.grad` is
None are skipped.  One kernel launch updates every tensor (586 of them in the
segmentation step) instead of one aten foreach chain per group.
"""
import ctypes

import torch

from .. import _lib

_CHUNK = 65536


def _bump_versions(params):
    """The kernel writes parameters through raw pointers, which autograd's version counters do not see; every cache
    keyed on `tensor._version` (core/_util.PackedCache: packed weights, folded BatchNorm, fused head matrices) would
    keep serving pre-step copies to the next eval / no_grad forward.  Bump the counters without touching the data."""
    if not params:
        return
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        try:
            setter(params, [p._version + 1 for p in params])
            return
        except (TypeError, RuntimeError):
            pass
    torch._foreach_add_(params, 0.0)


class _AdamEntry(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("lr", ctypes.c_float), ("wd", ctypes.c_float)]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._layout = None  # (signature, chunk_entry, chunk_off) cached while the grad set is unchanged

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        assert lib.segmif_adamw_entry_bytes() == ctypes.sizeof(_AdamEntry)
        by_hyper = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters must be contiguous fp32 tensors on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1  # a torch.optim.AdamW checkpoint stores `step` as a tensor
                key = (st["step"], group["betas"], group["eps"], p.device)
                by_hyper.setdefault(key, []).append((p, p.grad.contiguous(), st, group["lr"], group["weight_decay"]))
        for (step, betas, eps, dev), items in by_hyper.items():
            entries = (_AdamEntry * len(items))()
            chunk_entry, chunk_off = [], []
            keep = []
            for i, (p, g, st, lr, wd) in enumerate(items):
                keep.append(g)
                e = entries[i]
                e.p, e.g, e.m, e.v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                e.n, e.lr, e.wd = p.numel(), lr, wd
                for off in range(0, p.numel(), _CHUNK):
                    chunk_entry.append(i)
                    chunk_off.append(off)
            table = torch.frombuffer(bytearray(bytes(entries)), dtype=torch.uint8).to(dev)
            ce = torch.tensor(chunk_entry, dtype=torch.int32, device=dev)
            co = torch.tensor(chunk_off, dtype=torch.int64, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            # bias corrections in double on the host, like torch.optim.AdamW (fp32 powf is ~3e-5 off in early steps)
            bc1 = 1.0 - float(betas[0]) ** step
            bc2_sqrt = (1.0 - float(betas[1]) ** step) ** 0.5
            _lib.check(lib.segmif_adamw_f32(table.data_ptr(), ce.data_ptr(), co.data_ptr(), len(chunk_entry), _CHUNK,
                                            betas[0], betas[1], eps, bc1, bc2_sqrt, stream), "segmif_adamw_f32")
            self._keepalive = (table, ce, co, keep)
            _bump_versions([p for p, *_ in items])
        return loss


class _PolyWarmupMixin:
    def _init_schedule(self, global_step, warmup_iter, max_iter, warmup_ratio, power):
        self.global_step = global_step
        self.warmup_iter, self.warmup_ratio, self.max_iter, self.power = warmup_iter, warmup_ratio, max_iter, power
        self._init_lr = [g["lr"] for g in self.param_groups]

    def _apply_schedule(self):
        mult = None
        if self.global_step < self.warmup_iter:
            mult = 1 - (1 - self.global_step / self.warmup_iter) * (1 - self.warmup_ratio)
        elif self.global_step < self.max_iter:
            mult = (1 - self.global_step / self.max_iter) ** self.power
        if mult is not None:
            for g, lr0 in zip(self.param_groups, self._init_lr):
                g["lr"] = lr0 * mult


class PolyWarmupAdamW(_PolyWarmupMixin, FusedAdamW):
    def __init__(self, params, lr, weight_decay, betas, warmup_iter=None, max_iter=None, warmup_ratio=None, power=None):
        FusedAdamW.__init__(self, params, lr=lr, betas=tuple(betas), weight_decay=weight_decay, eps=1e-8)
        self._init_schedule(0, warmup_iter, max_iter, warmup_ratio, power)

    def step(self, closure=None):
        self._apply_schedule()
        out = FusedAdamW.step(self, closure)
        self.global_step += 1
        return out


class PolyWarmupAdamW_seg(_PolyWarmupMixin, FusedAdamW):
    def __init__(self, params, lr, weight_decay, betas, iter_curr, warmup_iter=None, max_iter=None, warmup_ratio=None,
                 power=None):
        FusedAdamW.__init__(self, params, lr=lr, betas=tuple(betas), weight_decay=weight_decay, eps=1e-8)
        self._init_schedule(iter_curr, warmup_iter, max_iter, warmup_ratio, power)

    def step(self, closure=None):
        self._apply_schedule()
        out = FusedAdamW.step(self, closure)
        self.global_step += 1
        return out
