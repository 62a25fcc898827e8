"""torch.autograd.Function wrappers: forward AND backward run in the HIP kernels (C ABI); autograd
only records the graph.  Layout conventions as in ops.py (NHWC / tokens, fp32).

Backward building blocks
  input gradients  : segmif_igemm_f32 with transposed (linear) / rotated+swapped (stride-1 conv)
                     weights; segmif_conv_dgrad_strided_f32 for strided convs
  weight gradients : segmif_wgrad_f32 (fp32 MFMA over the row dimension, deterministic 2-pass)
  bias gradients   : segmif_colsum_f32
  the rest         : LayerNorm / dwconv+GELU / bilinear / row-softmax / cross-entropy kernels
"""
import ctypes

import torch

from . import _lib, ops
from .ops import ACT_GELU, ACT_NONE, ACT_PRELU, ACT_RELU, _req, _stream, rows_view


# weight-derived tensors of the backward pass (built per call: the weights change every step)
def _w_transposed(w, N, K):
    """(K, N) contiguous transpose of a Linear / 1x1-conv weight stored as (N, K[, 1, 1])."""
    return w.detach().reshape(N, K).t().contiguous()


def _w_taps(w, N, cin, k):
    """[(ky, kx, c)][n] form of an OIHW conv weight: the "weights" of the cols = dY W^T GEMM of a strided conv's input gradient."""
    return w.detach().permute(2, 3, 1, 0).reshape(k * k * cin, N).contiguous()


def _dw9(w, flipped=False):
    """[9][C] tap-major form of a (C, 1, 3, 3) depthwise weight (flipped: taps reversed - the input gradient's kernel)."""
    if flipped:
        return ops.pack_dw_weight(w).flip(0).contiguous()
    return ops.pack_dw_weight(w)


# ------------------------------------------------------------------------------------------------
# raw helpers over the backward ABI
# ------------------------------------------------------------------------------------------------
def colsum(x2d, out=None, accumulate=False):
    rows, N, ld = rows_view(x2d, "x")
    lib = _lib.load()
    if out is None:
        out = torch.empty((N,), device=x2d.device, dtype=torch.float32)
    ws = torch.empty((lib.segmif_colsum_blocks(rows) * N,), device=x2d.device, dtype=torch.float64)
    _lib.check(lib.segmif_colsum_f32(x2d.data_ptr(), out.data_ptr(), ws.data_ptr(), rows, N, ld, int(accumulate),
                                     _stream()), "segmif_colsum_f32")
    return out


class Out:
    """An output placement (a rows view of a wider buffer: a DRDB's concat buffer, the two halves of conv2's input) handed to a
    Function as a NON-tensor argument: the kernel writes there and the Function returns it as its (fresh) output."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t.detach()


def _rows(t):
    """t itself when the kernels can address it in place (a rows view on 16-byte boundaries), else a contiguous copy."""
    try:
        _, C, ld = rows_view(t, "gradient")
    except RuntimeError:
        return t.contiguous()
    # (an expanded gradient - the ones of loss.sum() - passes rows_view with pitch 0: rows must not overlap)
    return t if ld >= C and ops.aligned16(t) else t.contiguous()


def act_bwd(dy, ref, act, slope=None, out=None, ref2=None):
    """dx = dy * act'(.) with the activation output `ref` (ref - ref2 when ref2 is given: the output sits under a residual)."""
    rows, C, ldy = rows_view(dy, "dy")
    _, _, ldr = rows_view(ref, "ref")
    dx = torch.empty(dy.shape, device=dy.device, dtype=torch.float32) if out is None else out
    orow, oc, ldx = rows_view(dx, "dx")
    if (orow, oc) != (rows, C):
        raise RuntimeError("act_bwd out shape mismatch")
    sl = slope.data_ptr() if slope is not None else None
    if ref2 is not None:
        _, _, ldr2 = rows_view(ref2, "ref2")
        _lib.check(_lib.load().segmif_act_bwd2_f32(dy.data_ptr(), ref.data_ptr(), ref2.data_ptr(), dx.data_ptr(), rows, C, ldy, ldr,
                                                   ldr2, ldx, act, sl, _stream()), "segmif_act_bwd2_f32")
    else:
        _lib.check(_lib.load().segmif_act_bwd_f32(dy.data_ptr(), ref.data_ptr(), dx.data_ptr(), rows, C, ldy, ldr, ldx, act,
                                                  sl, _stream()), "segmif_act_bwd_f32")
    return dx


def _wgrad(desc, dy, ldy, dw, dy_zstride=0, sn=0, sk=1, nz=1, db=None):
    lib = _lib.load()
    ws = torch.empty((lib.segmif_wgrad_workspace_size(desc.M, desc.N, desc.K) * max(nz, 1),), device=dw.device,
                     dtype=torch.float32)
    _lib.check(lib.segmif_wgrad_f32(ctypes.byref(desc), dy.data_ptr(), ldy, dy_zstride, dw.data_ptr(), sn, sk,
                                    db.data_ptr() if db is not None else None, ws.data_ptr(), 0, _stream()),
               "segmif_wgrad_f32")
    return dw


def _bias_out(want, N, like):
    return torch.empty((N,), device=like.device, dtype=torch.float32) if want else None


def linear_wgrad(x, dy, N, want_bias=False):
    """dW (N, K) = dy^T x (and db = column sums of dy, fused).  x: rows view (..., K); dy: rows view (..., N)."""
    rows, K, lda = rows_view(x, "x")
    rows2, n2, ldy = rows_view(dy, "dy")
    assert rows == rows2 and n2 == N
    d = _lib.SegmifIgemm()
    d.in_ = x.data_ptr()
    d.M, d.N, d.K, d.lda = rows, N, K, lda
    d.H = d.W = d.OH = d.OW = 1
    d.Cin = K
    d.KH = d.KW = d.stride = d.dil = 1
    dw = torch.empty((N, K), device=x.device, dtype=torch.float32)
    db = _bias_out(want_bias, N, x)
    _wgrad(d, dy, ldy, dw, sn=K, sk=1, db=db)
    return (dw, db) if want_bias else dw


def conv_wgrad(x, dy, w_shape, k, stride, pad, dil, want_bias=False, amax=None):
    """dW in OIHW (and db, fused). x: (B,H,W,Cin) rows view; dy: (B,OH,OW,N) rows view.
    amax = (x's range slots, dy's range slots): the f16x3 form of the 3x3 stride-1 kernel (int32 device tensors)."""
    N, cin = w_shape[0], w_shape[1]
    B, H, W, _ = x.shape
    _, _, lda = rows_view(x, "x")
    rows, n2, ldy = rows_view(dy, "dy")
    d = _lib.SegmifIgemm()
    d.in_ = x.data_ptr()
    d.M, d.N, d.K, d.lda = rows, N, k * k * cin, lda
    d.H, d.W, d.Cin, d.KH, d.KW = H, W, cin, k, k
    d.stride, d.pad, d.dil, d.OH, d.OW = stride, pad, dil, dy.shape[1], dy.shape[2]
    if amax is not None:
        xs, ys = amax
        d.split_f16 = 1
        d.split_in_amax, d.split_in_amax_n = xs.data_ptr(), xs.numel()
        d.wgrad_dy_amax, d.wgrad_dy_amax_n = ys.data_ptr(), ys.numel()
    dw = torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32)
    db = _bias_out(want_bias, N, x)
    _wgrad(d, dy, ldy, dw, db=db)
    return (dw, db) if want_bias else dw


TRAIN_SPLIT_MIN_ROWS = 500000


def _gemm(x, w_nk, N, bias=None, act=ACT_NONE, res=None, out=None):
    """x @ w_nk^T (+ bias, act) for a plain (N, K) weight matrix: the bf16x6 GEMM for tall problems (same size rule as
    inference, ops.linear_auto), the fp32 tiles otherwise.  Training re-packs per call: the weights change every step."""
    rows, K, _ = rows_view(x, "x")
    # (measured at 8 images per step: with the per-call weight packing the bf16x6 GEMM only pays for the full-resolution
    # problems of the fusion net; the encoder's Linears - at most 153 600 rows - are 2.5 ms per step faster on the fp32 tiles)
    if rows >= TRAIN_SPLIT_MIN_ROWS and ops.linear_wants_split(rows, N, K):
        return ops.linear_auto(x, ops.pack_linear(w_nk, half=False), N, bias=bias, act=act, res=res, out=out)
    wt = w_nk if (K % 16 == 0 and w_nk.is_contiguous()) else ops.pack_weight(w_nk)
    return ops.linear(x, wt, N, bias=bias, act=act, res=res, out=out)


def _no_prelu(act):
    if act == ACT_PRELU:
        raise RuntimeError("segmif_amd.autograd: the shared PReLU is its own node on the training path (ag.conv2d / ag.linear "
                           "compose it after an un-activated conv): its backward needs the pre-activation, which a fused "
                           "epilogue does not keep")


# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x @ w^T + b); w is the raw (N, K) Linear weight or a (N, K, 1, 1) conv weight."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope, out=None):
        _no_prelu(act)
        N = w.shape[0]
        w2 = w.reshape(N, -1)
        y = _gemm(x, w2.detach().contiguous(), N, bias=b, act=act, out=out.t if out is not None else None)
        ctx.act = act
        ctx.save_for_backward(x, w, y if act == ACT_RELU else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N = w.shape[0]
        w2 = w.reshape(N, -1)
        K = w2.shape[1]
        dy = _rows(dy)  # (a channel slice of a concatenated gradient is read in place)
        dz = act_bwd(dy, y, ACT_RELU) if ctx.act == ACT_RELU else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm(dz, _w_transposed(w, N, K), K)  # (K, N): "weights" of the input-gradient GEMM
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            r = linear_wgrad(x, dz, N, want_bias=want_b)
            dw, db = (r if want_b else (r, None))
            dw = dw.reshape(w.shape)
        elif want_b:
            db = colsum(dz)
        return dx, dw, db, None, None, None


def _pack_conv(w, k, stride, pad, dil):
    """Kernel-layout weights for a conv: the split-bf16 image for stride-1 'same' 3x3 convs when
    ops.conv3x3_mode() allows it (csrc/conv3x3_split.hip), the fp32 packing otherwise."""
    if k == 3 and stride == 1 and pad == dil and dil in (1, 2):
        return ops.pack_conv3x3(w)
    return ops.pack_weight(w)


class ConvFn(torch.autograd.Function):
    """NHWC convolution y = act(conv(x, w) + b), w in OIHW."""

    @staticmethod
    def _f16(k, stride, pad, dil, cin, N):
        """3x3 stride-1 'same' convs with both channel counts in the split kernel's range run on f16x3 (ops.train_conv_f16):
        forward, input gradient (the roles of cin and N swapped) and weight gradient."""
        return k == 3 and stride == 1 and pad == dil and dil in (1, 2) and ops.train_conv_f16() and cin % 16 == 0 and N % 16 == 0 \
            and 16 <= N <= 256 and 16 <= cin <= 256

    @staticmethod
    def forward(ctx, x, w, b, k, stride, pad, dil, act, slope):
        _no_prelu(act)
        N = w.shape[0]
        xslot = None
        if ConvFn._f16(k, stride, pad, dil, x.shape[-1], N) and ops.aligned16(x):
            xslot = ops.range_slots(1, x.device).view(-1)  # max |x|: the kernel scales its staged input from it
            ops.amax_rows(x, xslot)
            y = ops.conv2d(x, ops.pack_weight_split16(w), N, k, stride=stride, pad=pad, dil=dil, bias=b, act=act, prelu=slope,
                           in_amax=xslot)
        else:
            y = ops.conv2d(x, _pack_conv(w, k, stride, pad, dil), N, k, stride=stride, pad=pad, dil=dil, bias=b, act=act,
                           prelu=slope)
        ctx.geom = (k, stride, pad, dil, act)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act == ACT_RELU else None, xslot)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, xslot = ctx.saved_tensors
        k, stride, pad, dil, act = ctx.geom
        N, cin = w.shape[0], w.shape[1]
        dy = dy.contiguous()
        dz = act_bwd(dy, y, ACT_RELU) if act == ACT_RELU else dy
        dx = dw = db = None
        zslot = None
        if xslot is not None:  # the f16x3 kernels: dz's range slot serves the input gradient and the weight gradient
            zslot = ops.range_slots(1, x.device).view(-1)
            ops.amax_rows(dz, zslot)
        if ctx.needs_input_grad[0]:
            B, H, W, _ = x.shape
            if stride == 1:
                # full correlation with the 180-degree rotated, in/out-swapped kernel
                wr = w.flip(2, 3).transpose(0, 1).contiguous()
                if zslot is not None:
                    dx = ops.conv2d(dz, ops.pack_weight_split16(wr), cin, k, stride=1, pad=dil * (k - 1) - pad, dil=dil, in_amax=zslot)
                else:
                    dx = ops.conv2d(dz, _pack_conv(wr, k, 1, dil * (k - 1) - pad, dil), cin, k, stride=1,
                                    pad=dil * (k - 1) - pad, dil=dil)
            elif stride == k and pad == 0 and dil == 1:
                # non-overlapping patches (sr conv): every input pixel belongs to exactly one patch, so
                # dX is one dense GEMM dY (M', N) @ W (N, k*k*Cin) followed by a patch -> image permutation
                OH, OW = dz.shape[1], dz.shape[2]
                wt = _w_taps(w, N, cin, k)  # [(ky,kx,c)][n]
                wt = wt if N % 16 == 0 else ops.pack_weight(wt)
                cols = ops.linear(dz.view(B, OH * OW, N), wt, k * k * cin)
                # patch -> image gather (rows / columns the forward conv dropped receive zeros)
                dx = torch.empty((B, H, W, cin), device=x.device, dtype=torch.float32)
                _lib.check(_lib.load().segmif_col2im_f32(cols.data_ptr(), dx.data_ptr(), B, H, W, cin, k, k, 0, OH, OW, _stream()),
                           "segmif_col2im_f32")
            elif dil == 1:
                # overlapping strided conv (patch embeds): cols = dY W^T on the matrix pipe, then a gather (col2im)
                OH, OW = dz.shape[1], dz.shape[2]
                wt = _w_taps(w, N, cin, k)  # [(ky, kx, c)][n]
                cols = _gemm(dz.view(B, OH * OW, N), wt, k * k * cin)
                dx = torch.empty((B, H, W, cin), device=x.device, dtype=torch.float32)
                _lib.check(_lib.load().segmif_col2im_f32(cols.data_ptr(), dx.data_ptr(), B, H, W, cin, k, stride, pad, OH, OW,
                                                         _stream()), "segmif_col2im_f32")
            else:
                wd = w.permute(2, 3, 0, 1).contiguous()  # [ky][kx][n][c]
                dx = torch.empty((B, H, W, cin), device=x.device, dtype=torch.float32)
                _lib.check(_lib.load().segmif_conv_dgrad_strided_f32(
                    dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, H, W, cin, N, k, k, stride, pad, dz.shape[1],
                    dz.shape[2], N, cin, _stream()), "segmif_conv_dgrad_strided_f32")
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            r = conv_wgrad(x, dz, w.shape, k, stride, pad, dil, want_bias=want_b, amax=(xslot, zslot) if zslot is not None else None)
            dw, db = (r if want_b else (r, None))
        elif want_b:
            db = colsum(dz)
        return dx, dw, db, None, None, None, None, None, None


class PReluFn(torch.autograd.Function):
    """y = z > 0 ? z : a z with the fusion net's shared scalar slope `a` (core/model_fusion.py:1038).  A node of its own on
    the training path: the backward branches on the saved PRE-activation, so it is exact for any slope (nn.PReLU trains
    through a <= 0; AdamW's weight decay can take it there), and d loss / d a comes out of the same kernel.
    out: optional Out placement (a DRDB buffer's first channels: the block then starts without a copy)."""

    @staticmethod
    def forward(ctx, z, slope, out=None):
        z = z.contiguous()
        if out is None:
            y = torch.empty_like(z)
            _lib.check(_lib.load().segmif_prelu_f32(z.data_ptr(), _req(slope, "slope").data_ptr(), y.data_ptr(), z.numel(), _stream()),
                       "segmif_prelu_f32")
        else:  # rows-view form of the same map: z * (z >= 0 ? 1 : a)
            y = act_bwd(z, z, ACT_PRELU, slope=_req(slope, "slope"), out=out.t)
        ctx.save_for_backward(z, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, slope = ctx.saved_tensors
        dy = _rows(dy)
        lib = _lib.load()
        n = z.numel()
        dz = torch.empty_like(z)
        part = torch.empty((2 * (lib.segmif_prelu_bwd_blocks(n) + 1),), device=z.device, dtype=torch.float64)
        dslope = torch.empty((1,), device=z.device, dtype=torch.float32)
        if dy.is_contiguous():
            _lib.check(lib.segmif_prelu_bwd_f32(dy.data_ptr(), z.data_ptr(), slope.data_ptr(), dz.data_ptr(), part.data_ptr(),
                                                dslope.data_ptr(), n, _stream()), "segmif_prelu_bwd_f32")
        else:  # a channel slice of a wider gradient buffer (the DRDB's), read in place
            rows, C, ldy = rows_view(dy, "dy")
            _lib.check(lib.segmif_prelu_bwd_rows_f32(dy.data_ptr(), ldy, z.data_ptr(), slope.data_ptr(), dz.data_ptr(),
                                                     part.data_ptr(), dslope.data_ptr(), rows, C, _stream()),
                       "segmif_prelu_bwd_rows_f32")
        return dz, (dslope.reshape(slope.shape) if ctx.needs_input_grad[1] else None), None


def _ln_param_grads(partial, C, want):
    if not want:
        return None, None
    gb = colsum(partial)
    return gb[:C], gb[C:]


class LayerNormFn(torch.autograd.Function):
    """out: optional Out placement for the result (a rows view of a wider buffer)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out=None):
        y = ops.layernorm(x, gamma, beta, eps, out=out.t if out is not None else None)
        ctx.eps = eps
        ctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = _rows(dy)  # (a channel slice of a concatenated gradient is read in place)
        rows, C, ldx = rows_view(x, "x")
        _, _, ldy = rows_view(dy, "dy")
        lib = _lib.load()
        nblk = lib.segmif_layernorm_bwd_blocks(rows, C)
        partial = torch.empty((nblk, 2 * C), device=x.device, dtype=torch.float32)
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        _lib.check(lib.segmif_layernorm_bwd_f32(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                                                partial.data_ptr(), rows, C, ldx, ldy, C, float(ctx.eps), _stream()),
                   "segmif_layernorm_bwd_f32")
        dg, db = _ln_param_grads(partial, C, ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dg, db, None, None


class AddLayerNormFn(torch.autograd.Function):
    """(x, branch, scale) -> (s, n):  s = x + scale[b] * branch,  n = LayerNorm(s)      (core/mix_transformer.py:171-177)

    The residual add, the per-sample DropPath factor (scale (B,) = mask / keep, or None) and the NEXT LayerNorm of a
    transformer block as one node and one kernel each way.  With branch = None it is a LayerNorm that also hands its input
    on (s is x).  Why a node with two outputs: s feeds the next residual and n feeds the next branch, so in the backward the
    gradient of the sum (ds) and the LayerNorm's input gradient meet HERE - one kernel forms dx = ds + LN'(dn) (and
    dbranch = scale * dx) where autograd would otherwise run an accumulation pass (and a multiply) per residual connection:
    112 + 108 elementwise launches per mit_b3 segmentation step in round 3."""

    @staticmethod
    def forward(ctx, x, branch, scale, gamma, beta, eps):
        ctx.set_materialize_grads(False)
        ctx.eps = eps
        ctx.has_branch = branch is not None
        if branch is None:
            n = ops.layernorm(x, gamma, beta, eps)
            ctx.save_for_backward(x, gamma, None)
            return x, n
        if x.shape != branch.shape or x.dim() != 3:
            raise RuntimeError(f"AddLayerNormFn: token tensors (B, N, C) of one shape expected, got {tuple(x.shape)} / {tuple(branch.shape)}")
        x, branch = x.contiguous(), branch.contiguous()
        B, N, C = x.shape
        if scale is not None:
            scale = _req(scale, "scale").contiguous()
            if scale.numel() != B:
                raise RuntimeError("AddLayerNormFn: one scale per image")
        if not ops.aligned16(gamma, beta):
            gamma, beta = gamma.detach().clone(), beta.detach().clone()
        s = torch.empty_like(x)
        n = torch.empty_like(x)
        _lib.check(_lib.load().segmif_add_layernorm_f32(x.data_ptr(), branch.data_ptr(), scale.data_ptr() if scale is not None else None,
                                                        N, _req(gamma).data_ptr(), _req(beta).data_ptr(), s.data_ptr(), n.data_ptr(),
                                                        B * N, C, C, C, C, C, float(eps), _stream()), "segmif_add_layernorm_f32")
        ctx.save_for_backward(s, gamma, scale)
        return s, n

    @staticmethod
    def backward(ctx, ds, dn):
        s, gamma, scale = ctx.saved_tensors
        if dn is None:  # the normalised output went nowhere: only the sum's gradient passes through
            if ds is None:
                return None, None, None, None, None, None
            db = None
            if ctx.has_branch:
                db = ds if scale is None else ds * scale.view(-1, 1, 1)
            return ds, db, None, None, None, None
        dn = _rows(dn)
        rows, C, lds = rows_view(s, "s")
        _, _, ldn = rows_view(dn, "dn")
        if ds is not None:
            ds = _rows(ds)
        lib = _lib.load()
        nblk = lib.segmif_layernorm_bwd_blocks(rows, C)
        partial = torch.empty((nblk, 2 * C), device=s.device, dtype=torch.float32)
        dx = torch.empty(s.shape, device=s.device, dtype=torch.float32)
        scaled = ctx.has_branch and scale is not None
        dbr = torch.empty(s.shape, device=s.device, dtype=torch.float32) if scaled else None
        _lib.check(lib.segmif_layernorm_bwd_add_f32(
            s.data_ptr(), dn.data_ptr(), gamma.data_ptr(), ds.data_ptr() if ds is not None else None,
            rows_view(ds, "ds")[2] if ds is not None else 0, scale.data_ptr() if scaled else None, s.shape[1] if scaled else 0,
            dx.data_ptr(), dbr.data_ptr() if scaled else None, C, partial.data_ptr(), rows, C, lds, ldn, C, float(ctx.eps),
            _stream()), "segmif_layernorm_bwd_add_f32")
        dg, db = _ln_param_grads(partial, C, ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        return dx, ((dbr if scaled else dx) if ctx.has_branch else None), None, dg, db, None


class DwconvGeluFn(torch.autograd.Function):
    """tokens (B, H*W, C) -> gelu(dwconv3x3(tokens as image) + b); w is the (C,1,3,3) depthwise weight."""

    @staticmethod
    def forward(ctx, h, w, b, H, W):
        y = ops.dwconv3x3_gelu(h, _dw9(w), b, H, W)
        ctx.hw = (H, W)
        ctx.save_for_backward(h, w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, b = ctx.saved_tensors
        H, W = ctx.hw
        B, _, C = h.shape
        dy = dy.contiguous()
        lib = _lib.load()
        w9 = _dw9(w)
        prow = lib.segmif_dwconv_bwd_partial_rows(B, H, W)
        partial = torch.empty((prow, 10 * C), device=h.device, dtype=torch.float32)
        dz = torch.empty_like(h)
        _lib.check(lib.segmif_dwconv3x3_gelu_bwd_f32(h.data_ptr(), w9.data_ptr(), b.data_ptr(), dy.data_ptr(),
                                                     dz.data_ptr(), partial.data_ptr(), B, H, W, C, _stream()),
                   "segmif_dwconv3x3_gelu_bwd_f32")
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            sums = colsum(partial).view(10, C)
            dw = sums[:9].t().reshape(C, 1, 3, 3)
            db = sums[9]
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(h)
            w9f = _dw9(w, flipped=True)
            _lib.check(lib.segmif_dwconv3x3_plain_f32(dz.data_ptr(), w9f.data_ptr(), dh.data_ptr(), B, H, W, C,
                                                      _stream()), "segmif_dwconv3x3_plain_f32")
        return dh, dw, db, None, None


class DwconvFn(torch.autograd.Function):
    """DWConv.forward on its own: tokens (B, H*W, C) -> dwconv3x3(tokens as image) + b (core/mix_transformer.py:381-387)."""

    @staticmethod
    def forward(ctx, h, w, b, H, W):
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and h.shape[2] % 128:
            # (the forward kernel takes any C % 4 == 0; say it here instead of an EINVAL from the backward launch)
            raise RuntimeError(f"DWConv under autograd: the parameter-gradient kernel needs a channel count that is a multiple of "
                               f"128 (every MiT hidden width is), got {h.shape[2]}")
        ctx.hw = (H, W)
        ctx.save_for_backward(h, w)
        return ops.dwconv3x3_bias(h, _dw9(w), b, H, W)

    @staticmethod
    def backward(ctx, dy):
        h, w = ctx.saved_tensors
        H, W = ctx.hw
        B, _, C = h.shape
        dy = dy.contiguous()
        lib = _lib.load()
        w9 = _dw9(w)
        dw = db = dh = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            partial = torch.empty((lib.segmif_dwconv_bwd_partial_rows(B, H, W), 10 * C), device=h.device, dtype=torch.float32)
            _lib.check(lib.segmif_dwconv3x3_bias_bwd_f32(h.data_ptr(), w9.data_ptr(), dy.data_ptr(), partial.data_ptr(), B, H, W,
                                                         C, _stream()), "segmif_dwconv3x3_bias_bwd_f32")
            sums = colsum(partial).view(10, C)
            dw = sums[:9].t().reshape(C, 1, 3, 3)
            db = sums[9]
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(h)
            _lib.check(lib.segmif_dwconv3x3_plain_f32(dy.data_ptr(), _dw9(w, flipped=True).data_ptr(), dh.data_ptr(), B, H, W,
                                                      C, _stream()), "segmif_dwconv3x3_plain_f32")
        return dh, dw, db, None, None


class BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW, out=None):
        ctx.in_hw = (x.shape[1], x.shape[2])
        return ops.bilinear(x, OH, OW, out=out.t if out is not None else None)

    @staticmethod
    def backward(ctx, dy):
        dy = _rows(dy)  # (a channel slice of the decoder's concatenated gradient is read in place)
        B, OH, OW, C = dy.shape
        IH, IW = ctx.in_hw
        dx = torch.empty((B, IH, IW, C), device=dy.device, dtype=torch.float32)
        _lib.check(_lib.load().segmif_bilinear_nhwc_bwd_f32(dy.data_ptr(), dx.data_ptr(), B, IH, IW, OH, OW, C,
                                                            rows_view(dy, "dy")[2], C, _stream()), "segmif_bilinear_nhwc_bwd_f32")
        return dx, None, None, None


def _batched_heads_gemm(a, a_ld, a_bs, a_hs, w, w_ld, w_bs, w_hs, out, o_ld, o_bs, o_hs, B, heads, M, N, K):
    """out[b,h] (M x N) = a[b,h] (M x K) @ w[b,h]^T (N x K); every operand addressed by
    (batch stride, head stride, row pitch) so q/k/v slices of the fused projections are used in place."""
    d = _lib.SegmifIgemm()
    d.in_, d.wt, d.out = a, w, out
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldo, d.ldw = a_ld, o_ld, w_ld
    d.H = d.W = d.OH = d.OW = 1
    d.Cin = K
    d.KH = d.KW = d.stride = d.dil = 1
    d.nz, d.nz2 = B, heads
    d.in_zstride, d.wt_zstride, d.out_zstride = a_bs, w_bs, o_bs
    d.in_zstride2, d.wt_zstride2, d.out_zstride2 = a_hs, w_hs, o_hs
    d.tile = -1
    _lib.check(_lib.load().segmif_igemm_f32(ctypes.byref(d), _stream()), "segmif_igemm_f32")


class SrAttentionFn(torch.autograd.Function):
    """softmax(q k^T scale) v per (batch, head); forward = the fused kernel.  Backward (r5): head_dim 64 - the fused
    flash-style kernels of csrc/attention_bwd.hip (scores recomputed per tile in registers, nothing of size N x Nk in HBM;
    SEGMIF_ATTN_BWD=materialize restores round 4's); other head sizes - score recomputation with batched fp32-MFMA GEMMs +
    row-softmax kernels over materialised (B, heads, N, Nk) tensors."""
    FUSED = __import__("os").environ.get("SEGMIF_ATTN_BWD", "fused") != "materialize"

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        ctx.heads, ctx.scale = heads, scale
        out = ops.sr_attention(q, kv, heads, scale)
        ctx.save_for_backward(q, kv, out)  # (the output: D_i = dO_i . O_i of the fused backward; alive anyway - proj's input)
        return out

    @staticmethod
    def backward(ctx, do):
        q, kv, out = ctx.saved_tensors
        heads, scale = ctx.heads, ctx.scale
        do = do.contiguous()
        B, N, C = q.shape
        if SrAttentionFn.FUSED and C == heads * 64 and q.is_contiguous() and kv.is_contiguous() and out.is_contiguous() \
                and ops.aligned16(q, kv, out, do):  # (r6, ADVICE r5: unaligned storage takes the materialising path below instead of EINVAL)
            dq, dkv = ops.sr_attention_bwd(q, kv, out, do, heads, scale)
            return dq, dkv, None, None
        Nk = kv.shape[1]
        hd = C // heads
        dev = q.device
        lib = _lib.load()
        Lp = (Nk + 15) // 16 * 16  # score row pitch (zero padded so it can be a GEMM K dimension)
        # (the padding columns [Nk, Lp) are zeroed by the row-softmax kernels: no fill pass over the two score tensors)
        P = torch.empty((B, heads, N, Lp), device=dev, dtype=torch.float32)
        dP = torch.empty((B, heads, N, Lp), device=dev, dtype=torch.float32)
        kptr, vptr = kv.data_ptr(), kv.data_ptr() + 4 * C
        # S = q k^T ; dP = do v^T    (weights = k / v slices of kv, pitch 2C)
        _batched_heads_gemm(q.data_ptr(), C, N * C, hd, kptr, 2 * C, Nk * 2 * C, hd, P.data_ptr(), Lp,
                            heads * N * Lp, N * Lp, B, heads, N, Nk, hd)
        _batched_heads_gemm(do.data_ptr(), C, N * C, hd, vptr, 2 * C, Nk * 2 * C, hd, dP.data_ptr(), Lp,
                            heads * N * Lp, N * Lp, B, heads, N, Nk, hd)
        rows = B * heads * N
        _lib.check(lib.segmif_row_softmax_f32(P.data_ptr(), rows, Nk, Lp, float(scale), _stream()), "row_softmax")
        _lib.check(lib.segmif_row_softmax_bwd_f32(P.data_ptr(), dP.data_ptr(), rows, Nk, Lp, float(scale), _stream()),
                   "row_softmax_bwd")
        dS = dP  # in place
        # dq = dS k : weights = k^T per (b, h) as [hd][Lp]
        kT = torch.zeros((B, heads, hd, Lp), device=dev, dtype=torch.float32)
        kT[..., :Nk] = kv[..., :C].reshape(B, Nk, heads, hd).permute(0, 2, 3, 1)
        dq = torch.empty_like(q)
        _batched_heads_gemm(dS.data_ptr(), Lp, heads * N * Lp, N * Lp, kT.data_ptr(), Lp, heads * hd * Lp, hd * Lp,
                            dq.data_ptr(), C, N * C, hd, B, heads, N, hd, Lp)
        # dk^T[d][key] = sum_n q[n][d] dS[n][key] ; dv^T[d][key] = sum_n do[n][d] P[n][key]  (wgrad form).
        # Written with key stride 2C into a (B, Lp, 2C) buffer: rows >= Nk are padding and are sliced away.
        # (key count a multiple of 4: the contraction runs over the Nk real keys only - the kernel pads its own tiles with
        # zeros - and dkv is exactly (B, Nk, 2C); otherwise over the padded pitch, rows >= Nk sliced away afterwards)
        Kk = Nk if Nk % 4 == 0 else Lp
        dkv = torch.empty((B, Kk, 2 * C), device=dev, dtype=torch.float32)
        ws = torch.empty((lib.segmif_wgrad_workspace_size(N, hd, Lp) * B * heads,), device=dev, dtype=torch.float32)
        for src, probs, col0 in ((q, dS, 0), (do, P, C)):
            # every (image, head) in one launch: batch z = b * heads + h
            d = _lib.SegmifIgemm()
            d.in_ = probs.data_ptr()
            d.M, d.N, d.K, d.lda = N, hd, Kk, Lp
            d.H = d.W = d.OH = d.OW = 1
            d.Cin = Kk
            d.KH = d.KW = d.stride = d.dil = 1
            d.nz, d.nz2 = B, heads
            d.in_zstride, d.in_zstride2 = heads * N * Lp, N * Lp
            d.out_zstride, d.out_zstride2 = Kk * 2 * C, hd
            # element (n = d, k = key) of (b, h) -> dkv[b][key][col0 + h*hd + d]
            _lib.check(lib.segmif_wgrad_batched2_f32(ctypes.byref(d), src.data_ptr(), C, N * C, hd, dkv.data_ptr() + 4 * col0, 1,
                                                     2 * C, ws.data_ptr(), 0, _stream()), "segmif_wgrad_batched2_f32")
        return dq, (dkv if Kk == Nk else dkv[:, :Nk]), None, None


class SoftmaxCEFn(torch.autograd.Function):
    """mean cross-entropy over non-ignored pixels of NHWC logits (B,H,W,C) vs labels (B,H,W) int64."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        rows, C, ld = rows_view(logits, "logits")
        labels = labels.contiguous()
        lib = _lib.load()
        nblk = lib.segmif_softmax_ce_blocks(rows)
        partial = torch.empty((nblk, 2), device=logits.device, dtype=torch.float64)
        dlog = torch.empty(logits.shape, device=logits.device, dtype=torch.float32)
        _lib.check(lib.segmif_softmax_ce_f32(logits.data_ptr(), labels.data_ptr(), dlog.data_ptr(), partial.data_ptr(),
                                             rows, C, ld, C, int(ignore_index), _stream()), "segmif_softmax_ce_f32")
        tot = partial.sum(0)
        ctx.save_for_backward(dlog, tot[1:2])
        return (tot[0] / tot[1]).float()

    @staticmethod
    def backward(ctx, g):
        dlog, cnt = ctx.saved_tensors
        return dlog * (g / cnt.float()), None, None


class DRDBFn(torch.autograd.Function):
    """Dense dilated block (core/model_fusion.py:134-157) as ONE autograd node.  Forward is the
    inference path (five dilated convs writing in place into a 224-channel buffer, 1x1 conv + ReLU +
    residual); only that buffer and the output are kept for backward (every intermediate concat of
    the reference is a channel prefix of it)."""

    @staticmethod
    def forward(ctx, x, home, *params):  # params = (w1, b1, ..., w5, b5, w6, b6)
        """home: Out holding the (B, H, W, total) buffer whose first C0 channels x already IS (its producer wrote there:
        PReluFn / LayerNormFn with out=), or None (x is copied in)."""
        B, H, W, C0 = x.shape
        growth = params[0].shape[0]
        total = C0 + 5 * growth
        if home is not None and tuple(home.t.shape) == (B, H, W, total) and home.t.is_contiguous() \
                and x.data_ptr() == home.t.data_ptr() and x.stride() == home.t[..., :C0].stride():
            buf = home.t
        else:
            buf = torch.empty((B, H, W, total), device=x.device, dtype=torch.float32)
            buf[..., :C0].copy_(x)
        ch = C0
        # f16x3 convs (ops.train_conv_f16): range slots [x | out1 .. out5] - x's from one reduction pass, the others from the
        # producing conv's epilogue; conv i scales its staged input by the maximum over slots 0 .. i
        f16 = ops.train_conv_f16() and C0 % 16 == 0 and growth % 16 == 0
        slots = ops.range_slots(6, x.device) if f16 else None
        if f16:
            ops.amax_rows(buf[..., :C0], slots[0])
        for i in range(5):
            w, b = params[2 * i], params[2 * i + 1]
            if f16:
                ops.conv2d(buf[..., :ch], ops.pack_weight_split16(w), growth, 3, pad=2, dil=2, bias=b, act=ACT_RELU,
                           out=buf[..., ch:ch + growth], in_amax=slots[:i + 1].view(-1), out_amax=slots[i + 1])
            else:
                ops.conv2d(buf[..., :ch], ops.pack_conv3x3(w), growth, 3, pad=2, dil=2, bias=b, act=ACT_RELU,
                           out=buf[..., ch:ch + growth])
            ch += growth
        w6, b6 = params[10], params[11]
        out = ops.linear(buf, ops.pack_weight(w6), C0, bias=b6, act=ACT_RELU, res=buf[..., :C0])
        ctx.save_for_backward(buf, out, slots, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        buf, out, fslots = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        dout = _rows(dout)
        B, H, W, total = buf.shape
        C0 = out.shape[-1]
        growth = params[0].shape[0]
        grads = [None] * 12
        w6 = params[10].reshape(C0, total)
        dz6 = act_bwd(dout, out, ACT_RELU, ref2=buf[..., :C0])  # mask source: the 1x1 branch's relu output = out - x
        dbuf = torch.empty_like(buf)
        w6t = w6.t().contiguous()  # (total, C0): input-gradient weights, rows = concat channels
        ops.linear(dz6, w6t[:C0].contiguous(), C0, res=dout, out=dbuf[..., :C0])  # + residual path
        ops.linear(dz6, w6t[C0:].contiguous(), total - C0, out=dbuf[..., C0:])
        g10, grads[11] = linear_wgrad(buf, dz6, C0, want_bias=True)
        grads[10] = g10.reshape(params[10].shape)
        del dz6
        # Input gradients in GATHER form.  Conv i scatters into every channel below its own block
        # (N = 64..192 outputs from 32 inputs: two-chunk workgroups, the halo re-read per output tile);
        # instead, once dy_5 .. dy_{i+1} are known, the block just below conv i+1's outputs receives all
        # of its contributions in ONE conv over their concatenation — Cin = 32 * (5 - i) inputs, 32 (64
        # for the x block) outputs: the shape of the forward convs, same total FLOPs.
        dz = torch.empty((B, H, W, 5 * growth), device=buf.device, dtype=torch.float32)  # [dy5 | dy4 | .. | dy1]
        ch = total - growth
        # block 5's gradient is complete after the 1x1 conv's input gradient: through its ReLU into the first dz slot
        act_bwd(dbuf[..., ch:ch + growth], buf[..., ch:ch + growth], ACT_RELU, out=dz[..., :growth])
        f16 = ops.train_conv_f16() and C0 % 16 == 0 and growth % 16 == 0
        slots = None
        if f16:  # range slots of the dz blocks, in the order they are produced (gradients: 1e-7 and below - scaled, not guarded)
            slots = ops.range_slots(5, buf.device)
            ops.amax_rows(dz[..., :growth], slots[0])
        for i in range(4, -1, -1):
            k = 4 - i
            dy = dz[..., k * growth:(k + 1) * growth]
            # (conv i's input = x and the outputs of convs 1 .. i: forward slots 0 .. i; dy = dz block k: slot k)
            rng_w = (fslots[:i + 1].view(-1), slots[k]) if f16 and fslots is not None else None
            grads[2 * i], grads[2 * i + 1] = conv_wgrad(buf[..., :ch], dy, params[2 * i].shape, 3, 1, 2, 2, want_bias=True, amax=rng_w)
            lo = ch - growth if i > 0 else 0
            # rows = the channels of the receiving block, columns = (tap rotated by 180 deg, dz channel)
            wcat = torch.cat([params[2 * q][:, lo:ch].flip(2, 3).transpose(0, 1) for q in range(4, i - 1, -1)], dim=1)
            packed = ops.pack_weight_split16(wcat.contiguous()) if f16 else ops.pack_conv3x3(wcat.contiguous())
            src = dz[..., :(k + 1) * growth]
            # (the last conv's result - the gradient of the block's input - has no f16x3 consumer: no report)
            rng = dict(in_amax=slots[:k + 1].view(-1), **(dict(out_amax=slots[k + 1]) if i > 0 else {})) if f16 else {}
            if i == 0:  # the block's input x: no activation between it and the convs
                ops.conv2d(src, packed, ch - lo, 3, pad=2, dil=2, res=dbuf[..., lo:ch], out=dbuf[..., lo:ch], **rng)
            elif isinstance(packed, ops.SplitWeight):
                # (r4) this conv completes the gradient of block i - 1: its epilogue adds the 1x1 conv's part (res) and writes
                # the sum through that block's ReLU mask straight into the next dz slot - no mask pass of its own
                ops.conv2d(src, packed, ch - lo, 3, pad=2, dil=2, res=dbuf[..., lo:ch], out=dz[..., (k + 1) * growth:(k + 2) * growth],
                           mask=buf[..., lo:ch], **rng)
            else:
                ops.conv2d(src, packed, ch - lo, 3, pad=2, dil=2, res=dbuf[..., lo:ch], out=dbuf[..., lo:ch])
                act_bwd(dbuf[..., lo:ch], buf[..., lo:ch], ACT_RELU, out=dz[..., (k + 1) * growth:(k + 2) * growth])
            ch -= growth
        # (a rows view of the gradient buffer: the consumers - PReluFn / LayerNormFn backward - read it in place)
        dx = dbuf[..., :C0] if ctx.needs_input_grad[0] else None
        return (dx, None, *grads)


class BatchedLinearFn(torch.autograd.Function):
    """y[b] = x[b] @ w[b]^T + bias with one weight matrix per image (the folded end_proj of CrossPath)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        B, n, K = x.shape
        N = w.shape[1]
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.linear(x, w.contiguous(), N, bias=bias, batched_weight=True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, n, K = x.shape
        N = w.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = w.transpose(1, 2)  # (B, K, N): the input-gradient GEMM contracts over N
            if N % 16:  # (its packed form pads the contraction to a multiple of 16: linear_pred's 9 classes)
                wt = torch.nn.functional.pad(wt, (0, (N + 15) // 16 * 16 - N))
            dx = ops.linear(dy, wt.contiguous(), K, batched_weight=True)
        if ctx.needs_input_grad[1]:
            d = _lib.SegmifIgemm()
            d.in_ = x.data_ptr()
            d.M, d.N, d.K, d.lda = n, N, K, x.stride(1)
            d.H = d.W = d.OH = d.OW = 1
            d.Cin = K
            d.KH = d.KW = d.stride = d.dil = 1
            d.nz = B
            d.in_zstride = x.stride(0)
            d.out_zstride = N * K
            dw = torch.empty((B, N, K), device=x.device, dtype=torch.float32)
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            db = _bias_out(want_b, N, x)
            _wgrad(d, dy, N, dw, dy_zstride=n * N, sn=K, sk=1, nz=B, db=db)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)
        return dx, dw, db


class BatchedLinear2Fn(torch.autograd.Function):
    """y[b] = res[b] + [xa[b] | xb[b]] @ w[b]^T + bias: CrossPath's `x + end_proj(cat(z, v))` (core/model_fusion.py:357-360
    with the contexts folded into a per-image weight) as ONE node - two-source GEMM with the residual in its epilogue, so
    neither the 128-wide concatenation nor the separate add exists (they were 4 + 4 full-resolution aten passes per fusion
    training step).  xa, xb: (B, n, Ka / Kb) rows views (channel slices of wider tensors are fine)."""

    @staticmethod
    def forward(ctx, xa, xb, w, bias, res):
        N = w.shape[1]
        ctx.save_for_backward(xa, xb, w)
        ctx.has_bias = bias is not None
        return ops.linear(xa, w.contiguous(), N, bias=bias, res=res, x2=xb, batched_weight=True)

    @staticmethod
    def backward(ctx, dy):
        xa, xb, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, n, Ka = xa.shape
        Kb = xb.shape[2]
        K, N = Ka + Kb, w.shape[1]
        dxa = dxb = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = ops.linear(dy, w.transpose(1, 2).contiguous(), K, batched_weight=True)
            dxa, dxb = dx[..., :Ka], dx[..., Ka:]
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2]:
            dw = torch.empty((B, N, K), device=dy.device, dtype=torch.float32)
            db = _bias_out(want_b, N, dy)
            for src, k0, kk, bias_out in ((xa, 0, Ka, db), (xb, Ka, Kb, None)):
                d = _lib.SegmifIgemm()
                d.in_ = src.data_ptr()
                d.M, d.N, d.K, d.lda = n, N, kk, src.stride(1)
                d.H = d.W = d.OH = d.OW = 1
                d.Cin = kk
                d.KH = d.KW = d.stride = d.dil = 1
                d.nz = B
                d.in_zstride = src.stride(0)
                d.out_zstride = N * K
                _wgrad(d, dy, N, dw[:, :, k0:], dy_zstride=n * N, sn=K, sk=1, nz=B, db=bias_out)
        elif want_b:
            db = colsum(dy)
        return dxa, dxb, dw, db, (dy if ctx.needs_input_grad[4] else None)


def _batched_wgrad(src, dy, dw, k0, n, N, K, db=None):
    """dw[b][:, k0:k0+kk] = dy[b]^T src[b] per image (src (B, n, kk) rows view, dy (B, n, N) contiguous, dw (B, N, K))."""
    B, kk = src.shape[0], src.shape[2]
    d = _lib.SegmifIgemm()
    d.in_ = src.data_ptr()
    d.M, d.N, d.K, d.lda = n, N, kk, src.stride(1)
    d.H = d.W = d.OH = d.OW = 1
    d.Cin = kk
    d.KH = d.KW = d.stride = d.dil = 1
    d.nz = B
    d.in_zstride = src.stride(0)
    d.out_zstride = N * K
    _wgrad(d, dy, N, dw[:, :, k0:], dy_zstride=n * N, sn=K, sk=1, nz=B, db=db)


class ProjSink:
    """Shared by CrossProjFn and the nodes that consume its halves (TailPairFn, KvContextFn).  In the backward a consumer's
    last GEMM writes the gradient of a 64-channel half THROUGH that half's ReLU mask straight into CrossProjFn's dz buffer
    (ops.linear(mask=): the mask lives in the GEMM's epilogue) and marks it; CrossProjFn recognises its own buffer and skips
    the separate mask pass (twelve full-resolution read-read-write passes per fusion training step)."""

    def __init__(self):
        self.p = None          # the three relu(channel_proj) tensors (B, n, 2C), set by CrossProjFn.forward
        self.dz = [None] * 3   # their gradient buffers, allocated on first use in the backward
        self.done = set()

    def slot(self, i, half):
        """-> (where the masked gradient of half `half` of tensor i goes, that half's forward activation = its ReLU mask)."""
        p = self.p[i]
        if self.dz[i] is None:
            self.dz[i] = torch.empty(p.shape, device=p.device, dtype=torch.float32)
        h = p.shape[-1] // 2
        sl = slice(half * h, half * h + h)
        return self.dz[i][..., sl], p[..., sl]

    def holds(self, i, half, g):
        """True when g IS the marked slot (its consumer wrote it masked, in place)."""
        if (i, half) not in self.done or self.dz[i] is None or g is None:
            return False
        h = self.p[i].shape[-1] // 2
        ref = self.dz[i][..., half * h:half * h + h]
        return g.data_ptr() == ref.data_ptr() and g.shape == ref.shape and g.stride() == ref.stride()

    def reset(self):
        self.dz = [None] * 3
        self.done = set()


class CrossProjFn(torch.autograd.Function):
    """The three channel_proj + ReLU of CrossPath (core/model_fusion.py:351-353) as ONE node with every half its own output:
        (x1, x2, x3) -> (y1, u1, y2, u2, y3, u3, x1, x2),   [y_i | u_i] = relu(x_i W_i^T + b_i)
    y_i / u_i are the two 64-channel halves of one (B, n, 128) buffer (rows views, nothing copied), x1 / x2 are handed on for
    the residual of the closing projection.  Every output then has exactly ONE consumer (context reduction or TailPairFn), and
    the gradients of the halves arrive separately and are written - through the ReLU mask - into the two halves of one dz
    buffer; the residual's gradient joins the input gradient in that GEMM's epilogue.  The round-3 graph (a Linear node,
    slices, a separate residual edge) cost 34 GB of autograd accumulation passes and 20 GB of zero-filled slice gradients per
    8-image step."""

    @staticmethod
    def forward(ctx, x1, x2, x3, w1, b1, w2, b2, w3, b3, sink=None):
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        outs, ps = [], []
        for x, w, b in ((x1, w1, b1), (x2, w2, b2), (x3, w3, b3)):
            N = w.shape[0]
            p = _gemm(x, w.detach().contiguous(), N, bias=b, act=ACT_RELU)
            ps.append(p)
            outs += [p[..., :N // 2], p[..., N // 2:]]
        ctx.save_for_backward(x1, x2, x3, w1, w2, w3, *ps)
        ctx.has_bias = tuple(b is not None for b in (b1, b2, b3))
        if sink is not None:
            sink.p = ps
            sink.reset()
        return (*outs, x1, x2)

    @staticmethod
    def backward(ctx, *g):
        xs, ws, ps = ctx.saved_tensors[:3], ctx.saved_tensors[3:6], ctx.saved_tensors[6:9]
        gres = (g[6], g[7], None)
        grads = [None] * 10
        sink = ctx.sink
        for i in range(3):
            x, w, p = xs[i], ws[i], ps[i]
            N, K = w.shape
            h = N // 2
            dz = sink.dz[i] if sink is not None and sink.dz[i] is not None else torch.empty(p.shape, device=p.device, dtype=torch.float32)
            for half, gy in ((0, g[2 * i]), (1, g[2 * i + 1])):
                sl = slice(half * h, half * h + h)
                if gy is None:
                    dz[..., sl].zero_()
                elif sink is not None and sink.holds(i, half, gy):
                    pass  # the consumer's GEMM epilogue already wrote this half, masked, where it belongs
                else:
                    act_bwd(_rows(gy), p[..., sl], ACT_RELU, out=dz[..., sl])
            if ctx.needs_input_grad[i]:
                r = _rows(gres[i]) if gres[i] is not None else None
                grads[i] = _gemm(dz, w.detach().t().contiguous(), K, res=r)  # + the residual's gradient, in the epilogue
            want_b = ctx.has_bias[i] and ctx.needs_input_grad[4 + 2 * i]
            if ctx.needs_input_grad[3 + 2 * i]:
                r = linear_wgrad(x, dz, N, want_bias=want_b)
                grads[3 + 2 * i], grads[4 + 2 * i] = r if want_b else (r, None)
            elif want_b:
                grads[4 + 2 * i] = colsum(dz)
        if sink is not None:
            sink.reset()
        return tuple(grads)


class TailPairFn(torch.autograd.Function):
    """Both closing projections of CrossPath (core/model_fusion.py:357-360 with the contexts folded into per-image weights):
        t_i = x_i + [y3 | u_i] @ weff_i^T + b_i        i = 1, 2
    as one node (two-source GEMMs, residual in the epilogue).  One node for the pair because y3 feeds both: its two gradient
    contributions are summed by the second GEMM's epilogue instead of an autograd accumulation pass."""

    @staticmethod
    def forward(ctx, y3, u1, u2, weff1, weff2, b1, b2, x1, x2, sink=None):
        """sink: the ProjSink of the CrossProjFn whose halves y3 = (2, 0), u1 = (0, 1), u2 = (1, 1) are (or None)."""
        N = weff1.shape[1]
        ctx.sink = sink
        ctx.save_for_backward(y3, u1, u2, weff1, weff2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        t1 = ops.linear(y3, weff1.contiguous(), N, bias=b1, res=x1, x2=u1, batched_weight=True)
        t2 = ops.linear(y3, weff2.contiguous(), N, bias=b2, res=x2, x2=u2, batched_weight=True)
        return t1, t2

    @staticmethod
    def backward(ctx, dt1, dt2):
        y3, u1, u2, weff1, weff2 = ctx.saved_tensors
        dt = (dt1.contiguous(), dt2.contiguous())
        B, n, Ka = y3.shape
        Kb = u1.shape[2]
        K, N = Ka + Kb, weff1.shape[1]
        need = ctx.needs_input_grad
        dy3 = None
        du = [None, None]
        dw = [None, None]
        db = [None, None]
        sink = ctx.sink if ctx.sink is not None and ctx.sink.p is not None else None
        for i, (u, weff) in enumerate(((u1, weff1), (u2, weff2))):
            wt = weff.detach().transpose(1, 2)  # (B, K, N): rows = the channels of [y3 | u_i]
            if need[0]:  # the second image of y3's gradient accumulates onto the first
                if i == 1 and sink is not None:  # ... and goes through y3's ReLU mask into CrossProjFn's buffer
                    out, mk = sink.slot(2, 0)
                    dy3 = ops.linear(dt[i], wt[:, :Ka].contiguous(), Ka, res=dy3, out=out, mask=mk, batched_weight=True)
                    sink.done.add((2, 0))
                else:
                    dy3 = ops.linear(dt[i], wt[:, :Ka].contiguous(), Ka, res=dy3, out=dy3, batched_weight=True)
            if need[1 + i]:
                if sink is not None:
                    out, mk = sink.slot(i, 1)
                    du[i] = ops.linear(dt[i], wt[:, Ka:].contiguous(), Kb, out=out, mask=mk, batched_weight=True)
                    sink.done.add((i, 1))
                else:
                    du[i] = ops.linear(dt[i], wt[:, Ka:].contiguous(), Kb, batched_weight=True)
            want_b = ctx.has_bias[i] and need[5 + i]
            if need[3 + i]:
                dw[i] = torch.empty((B, N, K), device=y3.device, dtype=torch.float32)
                db[i] = _bias_out(want_b, N, y3)
                _batched_wgrad(y3, dt[i], dw[i], 0, n, N, K, db=db[i])
                _batched_wgrad(u, dt[i], dw[i], Ka, n, N, K)
            elif want_b:
                db[i] = colsum(dt[i])
        # (the residuals' gradients are dt_i themselves: CrossProjFn adds them to its input gradients)
        return dy3, du[0], du[1], dw[0], dw[1], db[0], db[1], (dt[0] if need[7] else None), (dt[1] if need[8] else None), None


class JoinFn(torch.autograd.Function):
    """Tensors that ARE consecutive channel slices of one buffer (their producers wrote there: out= placements) -> that buffer:
    torch.cat(parts, -1) without the copy, and without the slice-gradient copies of its backward."""

    @staticmethod
    def forward(ctx, whole, *parts):
        w = whole.t
        c0 = 0
        ctx.cuts = []
        for t in parts:
            c1 = c0 + t.shape[-1]
            ref = w[..., c0:c1]
            if t.data_ptr() != ref.data_ptr() or t.shape != ref.shape or t.stride() != ref.stride():
                raise RuntimeError("JoinFn: an input is not the expected channel slice of the buffer")
            ctx.cuts.append((c0, c1))
            c0 = c1
        if c0 != w.shape[-1]:
            raise RuntimeError("JoinFn: the inputs do not cover the buffer")
        return w

    @staticmethod
    def backward(ctx, dw):
        return (None, *(dw[..., a:b] for a, b in ctx.cuts))


class KvContextFn(torch.autograd.Function):
    """ctx_raw[b][h] = K^T V per head with [K | V] = y @ wkv^T (no bias): the N-reduction of the linear
    cross attention (core/model_fusion.py:281, 316-318).  Forward = fused projection + reduction kernel
    (kv never hits HBM); backward recomputes kv with one GEMM."""

    @staticmethod
    def forward(ctx, y, wkv, sink=None, which=None):
        """sink / which = (i, half): y is that half of a CrossProjFn output - its gradient is written through the ReLU mask."""
        ctx.sink, ctx.which = sink, which
        ctx.save_for_backward(y, wkv)
        part = ops.linattn_kvpartial(y, wkv.contiguous())
        B = y.shape[0]
        return part.sum(1).view(B, 8, 8, 8)  # fp64: these logits feed a saturated softmax

    @staticmethod
    def backward(ctx, dctx):
        y, wkv = ctx.saved_tensors
        B, n, C = y.shape
        kv = ops.linear(y, wkv.contiguous(), 2 * C)  # (B, n, 128)
        dctx = dctx.float().contiguous()
        # dk = v @ D1^T-form, dv = k @ D2^T-form with block-diagonal (B, 64, 64) weights [out][in]
        # block-diagonal weights, one broadcast product each (not 8 slice assignments):
        #   wk[b][h8+i][h8+j] = dctx[b][h][i][j]   dk[.., h8+i] = sum_j dctx[h][i][j] v[.., h8+j]
        #   wv[b][h8+j][h8+i] = dctx[b][h][i][j]   dv[.., h8+j] = sum_i dctx[h][i][j] k[.., h8+i]
        eye = torch.eye(8, device=y.device, dtype=torch.float32).view(1, 8, 1, 8, 1)
        wk = (dctx.view(B, 8, 8, 1, 8) * eye).reshape(B, C, C)
        wv = (dctx.transpose(2, 3).reshape(B, 8, 8, 1, 8) * eye).reshape(B, C, C)
        dkv = torch.empty_like(kv)
        ops.linear(kv[..., C:], wk, C, out=dkv[..., :C], batched_weight=True)
        ops.linear(kv[..., :C], wv, C, out=dkv[..., C:], batched_weight=True)
        dy = None
        if ctx.needs_input_grad[0]:
            sink = ctx.sink if ctx.sink is not None and ctx.sink.p is not None else None
            if sink is not None:
                out, mk = sink.slot(*ctx.which)
                dy = ops.linear(dkv, wkv.t().contiguous(), C, out=out, mask=mk)
                sink.done.add(tuple(ctx.which))
            else:
                dy = ops.linear(dkv, wkv.t().contiguous(), C)
        dw = linear_wgrad(y, dkv, 2 * C) if ctx.needs_input_grad[1] else None
        return dy, dw, None, None


class ContextFoldFn(torch.autograd.Function):
    """(r6) CrossPath's context fold on the training path as ONE node with HIP kernels on both sides:
        Weff[b][n][0:64]   = sum_j softmax_i(ktv_a scale_a)[h][i][j] Wend[n][8h + j]        (the modality's own context: z half)
        Weff[b][n][64:128] = sum_j softmax_i(ktv_3 scale_3)[h][i][j] Wend[n][64 + 8h + j]   (the segmentation context: v half)
    so that cat(z_i, v_i) @ Wend^T == [y3 | u_i] @ Weff^T (core/model_fusion.py:281-286, :316-326, :357-360).  Forward = two
    segmif_linattn_fold_f32 launches on the fp64 K^T V (one "partial" per image), backward = segmif_linattn_fold_bwd_f32 per half
    + one column sum over the images for d end_proj - no torch softmax / einsum / cat."""

    @staticmethod
    def forward(ctx, ktv_a, ktv_3, wend, scale_a, scale_3):
        B = ktv_a.shape[0]
        ka, k3 = ktv_a.contiguous(), ktv_3.contiguous()
        w = wend.contiguous()
        weff = torch.empty((B, w.shape[0], 128), device=w.device, dtype=torch.float32)
        ops.linattn_fold(ka.view(B, 1, 512), w, weff, wofs=0, kofs=0, scale=scale_a)
        ops.linattn_fold(k3.view(B, 1, 512), w, weff, wofs=64, kofs=64, scale=scale_3)
        ctx.save_for_backward(ka, k3, w)
        ctx.scales = (float(scale_a), float(scale_3))
        return weff

    @staticmethod
    def backward(ctx, dweff):
        ka, k3, w = ctx.saved_tensors
        B, Nout = ka.shape[0], w.shape[0]
        dweff = dweff.contiguous()
        dka, dk3 = torch.empty_like(ka), torch.empty_like(k3)
        part = torch.empty((B, Nout, 128), device=w.device, dtype=torch.float32)
        lib = _lib.load()
        for k, dk, ofs, sc in ((ka, dka, 0, ctx.scales[0]), (k3, dk3, 64, ctx.scales[1])):
            _lib.check(lib.segmif_linattn_fold_bwd_f32(k.data_ptr(), w.data_ptr(), 128, ofs, dweff.data_ptr(), 128, ofs, sc, dk.data_ptr(),
                                                       part.data_ptr(), 128, B, Nout, _stream()), "segmif_linattn_fold_bwd_f32")
        dw = colsum(part.view(B, Nout * 128)).view(Nout, 128) if ctx.needs_input_grad[2] else None
        return dka, dk3, dw, None, None


class BatchNormReluFn(torch.autograd.Function):
    """Train-mode BatchNorm (batch statistics, biased variance) + ReLU over NHWC rows (rows, C).
    Returns (y, batch_mean, batch_var_biased); the caller updates the running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        rows, C, _ = rows_view(x, "x")
        lib = _lib.load()
        mean = colsum(x.view(rows, C)) / rows
        nblk = (rows + 255) // 256
        part = torch.empty((nblk * C,), device=x.device, dtype=torch.float64)
        ssq = torch.empty((C,), device=x.device, dtype=torch.float64)
        _lib.check(lib.segmif_bn_colstats_f32(x.data_ptr(), None, mean.data_ptr(), None, part.data_ptr(), ssq.data_ptr(),
                                              rows, C, 0, _stream()), "segmif_bn_colstats_f32")
        var = (ssq / rows).float()
        rstd = torch.rsqrt(var + eps)
        scale = (gamma * rstd).contiguous()
        shift = (beta - mean * scale).contiguous()
        y = torch.empty_like(x)
        _lib.check(lib.segmif_bn_apply_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), rows, C, 1,
                                           _stream()), "segmif_bn_apply_f32")
        ctx.save_for_backward(x, y, mean, rstd, gamma)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, y, mean, rstd, gamma = ctx.saved_tensors
        rows, C, _ = rows_view(x, "x")
        lib = _lib.load()
        dz = act_bwd(dy.contiguous(), y, ACT_RELU)
        nblk = (rows + 255) // 256
        part = torch.empty((nblk * 2 * C,), device=x.device, dtype=torch.float64)
        sums = torch.empty((2 * C,), device=x.device, dtype=torch.float64)
        _lib.check(lib.segmif_bn_colstats_f32(x.data_ptr(), dz.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                              part.data_ptr(), sums.data_ptr(), rows, C, 1, _stream()),
                   "segmif_bn_colstats_f32")
        dbeta = sums[:C].float()
        dgamma = sums[C:].float()
        a = (dbeta / rows).contiguous()
        b = (dgamma / rows).contiguous()
        scale = (gamma * rstd).contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.segmif_bn_bwd_apply_f32(x.data_ptr(), dz.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               scale.data_ptr(), a.data_ptr(), b.data_ptr(), dx.data_ptr(), rows, C,
                                               _stream()), "segmif_bn_bwd_apply_f32")
        return dx, dgamma, dbeta, None


class GaussBlurFn(torch.autograd.Function):
    """11x11 Gaussian "same" blur (sigma 1.5) of (B, C, H, W) images; symmetric => backward = blur."""

    _taps = None

    @staticmethod
    def taps():
        if GaussBlurFn._taps is None:
            import math
            g = torch.tensor([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
            g = (g / g.sum()).tolist()
            GaussBlurFn._taps = (ctypes.c_float * 11)(*g)
        return GaussBlurFn._taps

    @staticmethod
    def _run(x):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        _lib.check(_lib.load().segmif_gauss_blur11_f32(x.data_ptr(), y.data_ptr(), B * C, H, W, GaussBlurFn.taps(),
                                                       _stream()), "segmif_gauss_blur11_f32")
        return y

    @staticmethod
    def forward(ctx, x):
        return GaussBlurFn._run(x)

    @staticmethod
    def backward(ctx, g):
        return GaussBlurFn._run(g)


def _blur_planes(x, planes, H, W):
    y = torch.empty_like(x)
    _lib.check(_lib.load().segmif_gauss_blur11_f32(x.data_ptr(), y.data_ptr(), planes, H, W, GaussBlurFn.taps(), _stream()),
               "segmif_gauss_blur11_f32")
    return y


class FusionLossGrad3Fn(torch.autograd.Function):
    """MSE(mask, gen) + 1.1 (1 - SSIM(gen, mask)) on single-channel images (core/loss.py:506-517), csrc/losses.hip:
    product planes -> separable Gaussian blur -> SSIM map + reductions (+ derivative planes); backward = blur of the
    three derivative planes + one assembly kernel.  No gradient w.r.t. the mask (the reference's is a data tensor)."""

    @staticmethod
    def forward(ctx, gen, mask):
        lib = _lib.load()
        g, m = gen.contiguous(), mask.contiguous()
        B, C, H, W = g.shape
        n = g.numel()
        stack = torch.empty((5, n), device=g.device, dtype=torch.float32)
        _lib.check(lib.segmif_ssim_prep_f32(g.data_ptr(), m.data_ptr(), stack.data_ptr(), n, _stream()), "segmif_ssim_prep_f32")
        bl = _blur_planes(stack, 5 * B * C, H, W)
        need = ctx.needs_input_grad[0]
        der = torch.empty((3, n), device=g.device, dtype=torch.float32) if need else None
        part = torch.empty((2 * lib.segmif_loss_blocks(n),), device=g.device, dtype=torch.float64)
        sums = torch.empty((2,), device=g.device, dtype=torch.float64)
        _lib.check(lib.segmif_ssim_map_f32(bl.data_ptr(), g.data_ptr(), m.data_ptr(), der.data_ptr() if need else None,
                                           part.data_ptr(), sums.data_ptr(), n, _stream()), "segmif_ssim_map_f32")
        ctx.save_for_backward(g, m, der)
        ctx.geom = (B * C, H, W, n)
        return (sums[1] / n + 1.1 * (1.0 - sums[0] / n)).float()

    @staticmethod
    def backward(ctx, dloss):
        g, m, der = ctx.saved_tensors
        planes, H, W, n = ctx.geom
        bd = _blur_planes(der, 3 * planes, H, W)
        grad = torch.empty_like(g)
        up = dloss.reshape(1).float().contiguous()
        _lib.check(_lib.load().segmif_ssim_grad_f32(bd.data_ptr(), g.data_ptr(), m.data_ptr(), grad.data_ptr(), n, up.data_ptr(),
                                                    -1.1 / n, 2.0 / n, _stream()), "segmif_ssim_grad_f32")
        return grad, None


class FusionLoss3Fn(torch.autograd.Function):
    """L1(mask, gen) + L1(Sobelxy(mask), Sobelxy(gen)) (core/loss.py:459-476, :634-650), csrc/losses.hip."""

    @staticmethod
    def forward(ctx, gen, mask):
        lib = _lib.load()
        g, m = gen.contiguous(), mask.contiguous()
        B, C, H, W = g.shape
        n = g.numel()
        need = ctx.needs_input_grad[0]
        pxy = torch.empty((2, n), device=g.device, dtype=torch.float32) if need else None
        part = torch.empty((2 * lib.segmif_loss_blocks(n),), device=g.device, dtype=torch.float64)
        sums = torch.empty((2,), device=g.device, dtype=torch.float64)
        _lib.check(lib.segmif_sobel_l1_f32(g.data_ptr(), m.data_ptr(), pxy.data_ptr() if need else None, part.data_ptr(),
                                           sums.data_ptr(), B * C, H, W, _stream()), "segmif_sobel_l1_f32")
        ctx.save_for_backward(g, m, pxy)
        ctx.geom = (B * C, H, W)
        return ((sums[0] + sums[1]) / n).float()

    @staticmethod
    def backward(ctx, dloss):
        g, m, pxy = ctx.saved_tensors
        planes, H, W = ctx.geom
        grad = torch.empty_like(g)
        up = dloss.reshape(1).float().contiguous()
        _lib.check(_lib.load().segmif_sobel_l1_bwd_f32(pxy.data_ptr(), g.data_ptr(), m.data_ptr(), grad.data_ptr(), planes, H, W,
                                                       up.data_ptr(), _stream()), "segmif_sobel_l1_bwd_f32")
        return grad, None


class LapLoss2Fn(torch.autograd.Function):
    """LapLoss2(gen, ir, vis) (lap_loss.py:100-118) on single-channel images, csrc/losses.hip: one pass over the 7 x 7
    neighbourhood for the three Gaussian levels of the three images, sign planes kept for the backward."""

    @staticmethod
    def forward(ctx, gen, ir, vis):
        lib = _lib.load()
        g, a, b = gen.contiguous(), ir.contiguous(), vis.contiguous()
        B, C, H, W = g.shape
        n = g.numel()
        need = ctx.needs_input_grad[0]
        sign3 = torch.empty((3, n), device=g.device, dtype=torch.float32) if need else None
        part = torch.empty((2 * lib.segmif_loss_blocks(n),), device=g.device, dtype=torch.float64)
        sums = torch.empty((2,), device=g.device, dtype=torch.float64)
        _lib.check(lib.segmif_laploss2_f32(g.data_ptr(), a.data_ptr(), b.data_ptr(), sign3.data_ptr() if need else None,
                                           part.data_ptr(), sums.data_ptr(), B * C, H, W, _stream()), "segmif_laploss2_f32")
        ctx.save_for_backward(sign3)
        ctx.geom = (B * C, H, W, g.shape)
        return (sums[0] / n).float()

    @staticmethod
    def backward(ctx, dloss):
        (sign3,) = ctx.saved_tensors
        planes, H, W, shape = ctx.geom
        grad = torch.empty(shape, device=sign3.device, dtype=torch.float32)
        up = dloss.reshape(1).float().contiguous()
        _lib.check(_lib.load().segmif_laploss2_bwd_f32(sign3.data_ptr(), grad.data_ptr(), planes, H, W, up.data_ptr(), _stream()),
                   "segmif_laploss2_bwd_f32")
        return grad, None, None


# functional front-ends ------------------------------------------------------------------------------
def _color3(x, mode, ysrc=None, nout=3):
    """segmif_color3_f32 on a contiguous (B, 3, H, W) fp32 device tensor (ysrc: (B, 1, H, W))."""
    ops._req(x, "colour transform input")
    if x.dim() != 4 or x.shape[1] != 3:
        raise RuntimeError(f"colour transform expects (B, 3, H, W), got {tuple(x.shape)}")
    x = x.contiguous()
    B, _, H, W = x.shape
    y = None
    if ysrc is not None:
        y = ops._req(ysrc, "Y source").contiguous()
        if tuple(y.shape) != (B, 1, H, W):
            raise RuntimeError(f"Y source must be ({B}, 1, {H}, {W}), got {tuple(y.shape)}")
    out = torch.empty((B, nout, H, W), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().segmif_color3_f32(x.data_ptr(), y.data_ptr() if y is not None else None, out.data_ptr(), B, H * W,
                                             mode, nout, _stream()), "segmif_color3_f32")
    return out


class Rgb2YCrCbFn(torch.autograd.Function):
    """RGB2YCrCb (core/model_fusion.py:69-91): one pointwise kernel each way (the map is affine: its backward is the
    transposed matrix)."""

    @staticmethod
    def forward(ctx, rgb):
        return _color3(rgb, 0)

    @staticmethod
    def backward(ctx, dy):
        return _color3(dy, 2)


class YCrCb2RgbFn(torch.autograd.Function):
    """YCrCb2RGB (core/model_fusion.py:93-111) of [y | ycc[:, 1:]] when y is given (train.py:362-365: the fused luminance
    takes the place of the visible image's - no clone, no slice assignment, no cat), of ycc itself otherwise."""

    @staticmethod
    def forward(ctx, ycc, y):
        ctx.has_y = y is not None
        return _color3(ycc, 1, ysrc=y)

    @staticmethod
    def backward(ctx, dy):
        g_ycc = g_y = None
        if ctx.has_y:
            if ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]:
                return None, _color3(dy, 3, nout=1)
            g = _color3(dy, 3)
            if ctx.needs_input_grad[1]:
                g_y = g[:, 0:1].contiguous()
            if ctx.needs_input_grad[0]:
                g_ycc = g
                g_ycc[:, 0:1] = 0  # channel 0 of ycc was not read
        elif ctx.needs_input_grad[0]:
            g_ycc = _color3(dy, 3)
        return g_ycc, g_y


def rgb2ycrcb(rgb):
    return Rgb2YCrCbFn.apply(rgb)


def ycrcb2rgb(ycc, y=None):
    return YCrCb2RgbFn.apply(ycc, y)


def linear(x, w, b=None, act=ACT_NONE, slope=None, out=None):
    if act == ACT_PRELU:  # the shared PReLU is a node of its own: its backward reads the pre-activation (PReluFn)
        return PReluFn.apply(LinearFn.apply(x, w, b, ACT_NONE, None), slope, out)
    return LinearFn.apply(x, w, b, act, slope, out)


def conv2d(x, w, b=None, k=3, stride=1, pad=0, dil=1, act=ACT_NONE, slope=None, out=None):
    """out (Out placement) is honoured for act == PReLU only (the activation node writes it)."""
    if act == ACT_PRELU:
        return PReluFn.apply(ConvFn.apply(x, w, b, k, stride, pad, dil, ACT_NONE, None), slope, out)
    if out is not None:
        raise RuntimeError("ag.conv2d: out= needs act == ACT_PRELU")
    return ConvFn.apply(x, w, b, k, stride, pad, dil, act, slope)


def prelu(z, slope, out=None):
    return PReluFn.apply(z, slope, out)


def layernorm(x, gamma, beta, eps, out=None):
    return LayerNormFn.apply(x, gamma, beta, eps, out)


def add_layernorm(x, branch, scale, gamma, beta, eps):
    """-> (s, n) = (x + scale[b] * branch, LayerNorm(s)); branch None: (x, LayerNorm(x))."""
    return AddLayerNormFn.apply(x, branch, scale, gamma, beta, eps)


def dwconv(h, w, b, H, W):
    return DwconvFn.apply(h, w, b, H, W)


def dwconv_gelu(h, w, b, H, W):
    return DwconvGeluFn.apply(h, w, b, H, W)


def bilinear(x, OH, OW, out=None):
    return BilinearFn.apply(x, OH, OW, out)


def sr_attention(q, kv, heads, scale):
    return SrAttentionFn.apply(q, kv, heads, scale)


def softmax_ce(logits_nhwc, labels, ignore_index=255):
    return SoftmaxCEFn.apply(logits_nhwc, labels, ignore_index)


def drdb(x, params, home=None):
    return DRDBFn.apply(x, home, *params)


def batched_linear(x, w, bias=None):
    return BatchedLinearFn.apply(x, w, bias)


def batched_linear2(xa, xb, w, bias=None, res=None):
    return BatchedLinear2Fn.apply(xa, xb, w, bias, res)


def kv_context(y, wkv, sink=None, which=None):
    return KvContextFn.apply(y, wkv, sink, which)


def context_fold(ktv_a, ktv_3, wend, scale_a, scale_3):
    return ContextFoldFn.apply(ktv_a, ktv_3, wend, scale_a, scale_3)


def cross_proj(x1, x2, x3, w1, b1, w2, b2, w3, b3, sink=None):
    return CrossProjFn.apply(x1, x2, x3, w1, b1, w2, b2, w3, b3, sink)


def tail_pair(y3, u1, u2, weff1, weff2, b1, b2, x1, x2, sink=None):
    return TailPairFn.apply(y3, u1, u2, weff1, weff2, b1, b2, x1, x2, sink)


def join(whole, *parts):
    """whole: Out of the buffer; parts: tensors already living in its consecutive channel slices."""
    return JoinFn.apply(whole, *parts)


def gauss_blur11(x):
    return GaussBlurFn.apply(x)


def batchnorm_relu_train(x, gamma, beta, eps):
    return BatchNormReluFn.apply(x, gamma, beta, eps)
