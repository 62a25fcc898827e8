"""numpy emulation of the split-operand schemes against fp64 (K = 1008 contraction, fp32 accumulation): plain fp32, bf16x6
(3-way bf16 split, six products), f16x3 (half pairs with the per-row weight scale, three products; one- and two-accumulator
forms).  Errors are relative to sum |x||w| (each output's conditioning).  No GPU needed.
    python tools/f16x3_numerics_study.py > profiles/r03_f16x3_numerics_study.txt"""
import numpy as np
rng = np.random.default_rng(0)
def bf16_rn(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split_bf16x3(x):
    x0 = bf16_rn(x); r = (x - x0).astype(np.float32); x1 = bf16_rn(r); r2 = (r - x1).astype(np.float32); x2 = bf16_rn(r2)
    return x0, x1, x2
def mm32(a, b):  # fp32 accumulate (numpy uses blocked sums in fp32)
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)
def bf16x6(x, w):
    x0, x1, x2 = split_bf16x3(x); w0, w1, w2 = split_bf16x3(w)
    acc = mm32(x2, w0) + mm32(x0, w2); acc = acc + mm32(x1, w1); acc = acc + mm32(x1, w0) + mm32(x0, w1); acc = acc + mm32(x0, w0)
    return acc
def fp16x3(x, w):
    # weights: per output channel power-of-two scale so max|W| ~ 2^14
    mx = np.abs(w).max(axis=0, keepdims=True); s = (14 - np.floor(np.log2(np.maximum(mx, 1e-30)))).clip(-60, 60)
    W = (w * np.exp2(s)).astype(np.float32)
    W0 = W.astype(np.float16); Wl = (W - W0.astype(np.float32)).astype(np.float16); W0s = (W0.astype(np.float32) * 2.0**-11).astype(np.float16)
    x0 = x.astype(np.float16); l = ((x - x0.astype(np.float32)) * 2048.0).astype(np.float16)
    acc = mm32(x0, Wl) + mm32(l, W0s); acc = acc + mm32(x0, W0)
    return (acc * np.exp2(-s)).astype(np.float32)
def fp16x3_two_acc(x, w):
    mx = np.abs(w).max(axis=0, keepdims=True); s = (14 - np.floor(np.log2(np.maximum(mx, 1e-30)))).clip(-60, 60)
    W = (w * np.exp2(s)).astype(np.float32)
    W0 = W.astype(np.float16); Wl = ((W - W0.astype(np.float32)) * 2048.0).astype(np.float16)
    x0 = x.astype(np.float16); l = ((x - x0.astype(np.float32)) * 2048.0).astype(np.float16)
    lo = mm32(x0, Wl) + mm32(l, W0); hi = mm32(x0, W0)
    return ((hi + lo * np.float32(2.0**-11)) * np.exp2(-s)).astype(np.float32)
def report(name, x, w):
    ref = x.astype(np.float64) @ w.astype(np.float64)
    scale = (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64))  # conditioning-relative yardstick
    out = {}
    for nm, f in (("fp32", mm32), ("bf16x6", bf16x6), ("fp16x3", fp16x3), ("fp16x3_2acc", fp16x3_two_acc)):
        y = f(x, w).astype(np.float64); e = np.abs(y - ref) / scale
        out[nm] = (e.max(), np.sqrt((e**2).mean()))
    print(f"{name:34s} " + "  ".join(f"{k}: max {v[0]:.2e} rms {v[1]:.2e}" for k, v in out.items()))
M, K, N = 2048, 1008, 32
relu = lambda a: np.maximum(a, 0)
report("relu N(0,1) x N(0,.03)", relu(rng.standard_normal((M, K))).astype(np.float32), (0.03 * rng.standard_normal((K, N))).astype(np.float32))
report("N(0,1) x N(0,.03)", rng.standard_normal((M, K)).astype(np.float32), (0.03 * rng.standard_normal((K, N))).astype(np.float32))
report("lognormal(s=3) x N(0,.03)", np.exp(3 * rng.standard_normal((M, K))).clip(0, 6e4).astype(np.float32), (0.03 * rng.standard_normal((K, N))).astype(np.float32))
report("1e-5*N x N(0,.03)", (1e-5 * rng.standard_normal((M, K))).astype(np.float32), (0.03 * rng.standard_normal((K, N))).astype(np.float32))
report("1e-7*N x N(0,.03)", (1e-7 * rng.standard_normal((M, K))).astype(np.float32), (0.03 * rng.standard_normal((K, N))).astype(np.float32))
report("1e4*N x lognormal w", (1e4 * rng.standard_normal((M, K))).astype(np.float32), (np.exp(3 * rng.standard_normal((K, N))) * rng.choice([-1, 1], (K, N))).astype(np.float32))
report("uniform[0,1] x N(0,.1) K=64", rng.random((M, 64)).astype(np.float32), (0.1 * rng.standard_normal((64, N))).astype(np.float32))
report("K=9 (cin=1)", rng.random((M, 9)).astype(np.float32), (0.1 * rng.standard_normal((9, N))).astype(np.float32))
