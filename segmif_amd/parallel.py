"""Data-parallel gradient exchange for the training steps: one process per GPU,
sum-all-reduce of fp32 gradients over RCCL (backend "nccl" on ROCm; "gloo" on CPU for tests),
bucketed in reverse parameter order and launched from autograd hooks as soon as a bucket's last
gradient has been produced, so the collective overlaps the rest of backward.

The reference has no distributed code; semantics follow an 8x larger single-GPU batch:
gradients are averaged over ranks, parameters whose grad is None are skipped on every rank
(ffm2.*, classifier.weight: SURVEY F7), BatchNorm statistics stay per rank.

xGMI note: a ring all-reduce is bound by one ~153 GB/s link; the seg step's 178 MB of gradients in
~25 MB buckets take a few ms in total and hide under a >100 ms backward.
"""
import time

import torch
import torch.distributed as dist


def plan_buckets(params, bucket_bytes):
    """Reverse-order greedy packing of parameters into buckets of at most `bucket_bytes` (a single larger parameter gets a
    bucket of its own) -> list of lists.  The one layout rule of the exchange: the hook path, the static path and
    bench.py --dry-collective all use it."""
    buckets, cur, size = [], [], 0
    for p in reversed(list(params)):
        nbytes = p.numel() * p.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


class GradAllReducer:
    def __init__(self, params, bucket_mb=25.0, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self._buckets = None  # list of lists of params (reverse order), built after the first backward
        self._flat = None
        self._pending = None
        self._handles = []
        self._hooks = []
        self._stream = None
        self.time_exposed_wait = False  # measure the non-overlapped part of the all-reduce in finish() (adds device syncs)
        self.exposed_wait_s = 0.0
        self.copies = 0  # per-parameter gradient copies into the buckets made by the hook (0 after step 1 under train._zero_grads)
        self.exposed_wait_per_bucket_s = []  # (launch order = reverse parameter order) how long each bucket's wait blocked

    # ---- bucket layout ------------------------------------------------------------------------
    def _build(self, active, hooks=True):
        buckets = plan_buckets(active, self.bucket_bytes)
        self._buckets = buckets
        self._flat = [torch.empty(sum(p.numel() for p in b), device=b[0].device, dtype=b[0].dtype) for b in buckets]
        self._where = {}
        for bi, b in enumerate(buckets):
            off = 0
            for p in b:
                self._where[p] = (bi, off)
                off += p.numel()
        for h in self._hooks:
            h.remove()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in active] if hooks else []
        self._reset()

    def _reset(self):
        self._pending = [len(b) for b in self._buckets]
        self._handles = []

    # ---- hook: gradient of `p` is final ----------------------------------------------------------
    def _on_grad(self, p):
        bi, off = self._where[p]
        if self._pending[bi] <= 0:
            # a second backward before finish() (gradient accumulation, an extra loss.backward()) would overwrite a
            # bucket whose all-reduce is already in flight: not supported, say so instead of corrupting gradients
            raise RuntimeError("GradAllReducer: a parameter received a second gradient before finish(); call "
                               "finish() after every backward (gradient accumulation is not supported)")
        dst = self._flat[bi][off:off + p.numel()]
        if p.grad.data_ptr() != dst.data_ptr():
            self.copies += 1  # (segmif_amd.train zeroes the buckets in place instead - zero_buckets() - and never gets here)
            # (after finish() p.grad IS this slice; with optimizer.zero_grad(set_to_none=False) autograd then accumulates
            # straight into the bucket and there is nothing to copy.  set_to_none=True - what segmif_amd.train uses - gives
            # a fresh gradient tensor every step, copied here.)
            dst.copy_(p.grad.reshape(-1))
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def zero_buckets(self):
        """Zero every gradient this reducer manages, in place (one fill per flat bucket); the parameters' .grad stay the bucket
        views finish() left, so the next backward accumulates into the buckets directly.  -> False before the buckets exist
        (first step) or when a .grad no longer is its bucket view (someone called zero_grad(set_to_none=True)): the caller then
        clears the gradients its own way and the hook copies, as before."""
        if self._buckets is None:
            return False
        for b, flat in zip(self._buckets, self._flat):
            off = 0
            for p in b:
                if p.grad is None or p.grad.data_ptr() != flat[off:off + p.numel()].data_ptr():
                    return False
                off += p.numel()
        for flat in self._flat:
            flat.zero_()
        return True

    def _launch(self, bi):
        flat = self._flat[bi]
        if dist.is_initialized():  # also with one rank: the collective path is then exercised end to end
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- call after loss.backward() ---------------------------------------------------------------
    def finish(self):
        """Wait for the exchange and leave the rank-averaged gradients in p.grad (each p.grad becomes a view of its flat
        bucket: correct under both zero_grad(set_to_none=True) and in-place zeroing, see _on_grad)."""
        self.exposed_wait_s = 0.0
        if self._buckets is None:
            # first step: discover which parameters actually receive gradients, exchange without overlap
            active = [p for p in self.params if p.grad is not None]
            self._build(active)
            for p in active:
                bi, off = self._where[p]
                self._flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
            for bi in range(len(self._buckets)):
                self._launch(bi)
        else:
            missing = [bi for bi, n in enumerate(self._pending) if n != 0]
            if missing:
                raise RuntimeError(f"gradient buckets {missing} were not completed by backward: the set of "
                                   "parameters receiving gradients changed; rebuild the GradAllReducer")
        if self._handles and self.time_exposed_wait:
            # how long the host-visible stream still has to wait for the collectives once backward has been issued: the
            # part of the exchange that did NOT hide under backward (two device syncs; bench.py's train leg only)
            cur = torch.cuda.current_stream()
            cur.synchronize()  # backward's kernels only: RCCL runs on its own stream and keeps going
            t0 = time.perf_counter()
            per = []
            for h in self._handles:
                t1 = time.perf_counter()
                h.wait()  # the compute stream now depends on this collective
                cur.synchronize()
                per.append(time.perf_counter() - t1)
            self.exposed_wait_s = time.perf_counter() - t0
            self.exposed_wait_per_bucket_s = per  # (a bucket that finished under backward costs ~0: the first ones launched)
        else:
            for h in self._handles:
                h.wait()
        inv = 1.0 / self.world
        for b, flat in zip(self._buckets, self._flat):
            if self.world > 1:
                flat.mul_(inv)
            off = 0
            for p in b:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self._reset()

    def close(self):
        """Detach from the parameters (removes the autograd hooks); the object must not be used afterwards."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def allreduce_static(self):
        """Exchange for a backward that autograd did not run this step (a replayed hipGraph: no hooks fire, and the
        gradients are the graph's static tensors, so p.grad must keep pointing at them): pack, all-reduce every bucket,
        average, copy back in place.  Not overlapped with backward.  A reducer used this way has the same bucket layout as
        the hook path but no hooks: calling finish() on it afterwards raises ("buckets were not completed")."""
        if self._buckets is None:
            self._build([p for p in self.params if p.grad is not None], hooks=False)
        for b in self._buckets:
            for p in b:
                if not p.grad.is_contiguous():  # view(-1) below must alias the gradient, not a temporary copy of it
                    raise RuntimeError("GradAllReducer.allreduce_static needs contiguous gradients")
        handles = []
        for b, flat in zip(self._buckets, self._flat):
            torch._foreach_copy_(list(flat.split([p.numel() for p in b])), [p.grad.view(-1) for p in b])
            if dist.is_initialized():
                handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        inv = 1.0 / self.world
        for b, flat in zip(self._buckets, self._flat):
            if self.world > 1:
                flat.mul_(inv)
            torch._foreach_copy_([p.grad.view(-1) for p in b], list(flat.split([p.numel() for p in b])))

    def gradient_bytes(self):
        return sum(f.numel() * f.element_size() for f in (self._flat or []))


def allreduce_scalar_mean(value, process_group=None):
    """Average a python float over ranks (the two losses behind train.py:369-374's dynamic weights must
    be identical on every rank)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return float(value)
    dev = "cuda" if dist.get_backend(process_group) == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, group=process_group)
    return float(t.item()) / dist.get_world_size(process_group)
