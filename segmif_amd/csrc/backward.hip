// Backward kernels of the SegMiF training step that are not contractions (those are igemm.hip with
// transposed weights for dgrad and wgrad.hip for weight gradients): LayerNorm, Mix-FFN depthwise conv +
// GELU, bilinear resize, attention row softmax, softmax cross-entropy, strided-conv input gradient,
// and a multi-tensor AdamW.  All reductions are two-pass and deterministic (no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "segmif_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = (x - mu) * rstd * gamma + beta (mu, rstd recomputed from x):
//   g = dy * gamma ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
//   dgamma = sum_rows dy * xhat ; dbeta = sum_rows dy    (per-block partials -> segmif_colsum_f32)
// ---------------------------------------------------------------------------------------------------
template <int G, int IT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ partial, long long rows, int C, int ldx,
                                                            int ldy, int lddx, float eps, int rows_per_block,
                                                            const float* __restrict__ dres, int lddres,
                                                            const float* __restrict__ scale, long long rpi,
                                                            float* __restrict__ dbr, int lddbr) {
  constexpr int SLOTS = 256 / G;
  __shared__ float red[256 * IT * 4];
  const int tid = threadIdx.x, sub = tid % G, slot = tid / G;
  const int nvec = C >> 2;
  const long long r_begin = (long long)blockIdx.x * rows_per_block;
  f32x4 dg[IT], db[IT], gm[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    dg[it] = db[it] = gm[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int u = sub + it * G;
    if (u < nvec) gm[it] = *reinterpret_cast<const f32x4*>(gamma + 4 * u);
  }
  for (int rr = slot; rr < rows_per_block; rr += SLOTS) {
    const long long row = r_begin + rr;
    const bool ok = row < rows;
    f32x4 xv[IT], gy[IT];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = sub + it * G;
      xv[it] = gy[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok && u < nvec) {
        xv[it] = *reinterpret_cast<const f32x4*>(x + row * ldx + 4 * u);
        gy[it] = *reinterpret_cast<const f32x4*>(dy + row * ldy + 4 * u);
      }
      sum += (xv[it][0] + xv[it][1]) + (xv[it][2] + xv[it][3]);
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it)
      if (sub + it * G < nvec)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xv[it][e] - mean;
          sq += d * d;
        }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;  // sum g, sum g * xhat
#pragma unroll
    for (int it = 0; it < IT; ++it)
      if (sub + it * G < nvec)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[it][e] - mean) * rstd;
          const float g = gy[it][e] * gm[it][e];
          s1 += g;
          s2 += g * xh;
          dg[it][e] += gy[it][e] * xh;
          db[it][e] += gy[it][e];
        }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = sub + it * G;
      if (ok && u < nvec) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[it][e] - mean) * rstd;
          o[e] = rstd * (gy[it][e] * gm[it][e] - m1 - xh * m2);
        }
        if (dres) {  // residual form: the gradient arriving at the sum itself joins here (no separate accumulation pass)
          const f32x4 rr4 = *reinterpret_cast<const f32x4*>(dres + row * lddres + 4 * u);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rr4[e];
        }
        *reinterpret_cast<f32x4*>(dx + row * lddx + 4 * u) = o;
        if (dbr) {  // ... and the branch's gradient is the same row times its per-sample stochastic-depth factor
          const float sc = scale[row / rpi];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= sc;
          *reinterpret_cast<f32x4*>(dbr + row * lddbr + 4 * u) = o;
        }
      }
    }
  }
  // block reduction of dgamma / dbeta over the row slots, then one partial row [dgamma | dbeta]
  float* out = partial + (long long)blockIdx.x * 2 * C;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(slot * IT + it) * G * 4 + sub * 4 + e] = pass == 0 ? dg[it][e] : db[it][e];
    __syncthreads();
    if (slot == 0) {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int u = sub + it * G;
        if (u < nvec) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float s = 0.f;
            for (int sl = 0; sl < SLOTS; ++sl) s += red[(sl * IT + it) * G * 4 + sub * 4 + e];
            out[pass * C + 4 * u + e] = s;
          }
        }
      }
    }
  }
}

struct LnBwdAdd {  // residual form (autograd.AddLayerNormFn): dx += dres; dbr = scale[row / rpi] * dx.  All zero: plain backward.
  const float* dres = nullptr; int lddres = 0; const float* scale = nullptr; long long rpi = 1; float* dbr = nullptr; int lddbr = 0;
};

template <int G, int IT>
int launch_ln_bwd(const float* x, const float* dy, const float* g, float* dx, float* partial, long long rows, int C,
                  int ldx, int ldy, int lddx, float eps, int rpb, int nblk, hipStream_t s, const LnBwdAdd& a) {
  hipLaunchKernelGGL((layernorm_bwd_kernel<G, IT>), dim3((unsigned)nblk), dim3(256), 0, s, x, dy, g, dx, partial, rows,
                     C, ldx, ldy, lddx, eps, rpb, a.dres, a.lddres, a.scale, a.rpi, a.dbr, a.lddbr);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Mix-FFN middle backward, part 1: z = dwconv(h) + b recomputed, dz = dy * gelu'(z) stored, and
// per-block partial sums of dbias (sum dz) and dw9[t] (sum dz * h shifted by tap t).
// Block = 32 channel quads (128 channels) x 8 x-positions, strip of DWB_TY rows.
// partial row layout: [10][C]  (index 0..8 = taps, 9 = bias)
// ---------------------------------------------------------------------------------------------------
constexpr int DWB_TY = 8;

template <bool GELU>
__global__ __launch_bounds__(256) void dwconv_gelu_bwd_kernel(const float* __restrict__ hin, const float* __restrict__ w9,
                                                              const float* __restrict__ bias, const float* __restrict__ dy,
                                                              float* __restrict__ dz, float* __restrict__ partial, int H,
                                                              int W, int C, int xtiles) {
  __shared__ f32x4 red[8][32];
  const int tid = threadIdx.x, cq = tid & 31, px = tid >> 5;
  const int ctile = blockIdx.x / xtiles, xt = blockIdx.x - ctile * xtiles;
  const int c = (ctile * 32 + cq) * 4;
  const int xo = xt * 8 + px;
  const int y0 = blockIdx.y * DWB_TY;
  const long long img = (long long)blockIdx.z * H * W;
  const bool active = xo < W && c < C;
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  f32x4 wv[9], acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = zero;
  f32x4 bv = zero;
  if (active) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
    bv = *reinterpret_cast<const f32x4*>(bias + c);
  }
  auto load_row = [&](int yy, f32x4& l, f32x4& m, f32x4& r) {
    l = m = r = zero;
    if (active && (unsigned)yy < (unsigned)H) {
      const float* p = hin + (img + (long long)yy * W + xo) * C + c;
      m = *reinterpret_cast<const f32x4*>(p);
      if (xo > 0) l = *reinterpret_cast<const f32x4*>(p - C);
      if (xo + 1 < W) r = *reinterpret_cast<const f32x4*>(p + C);
    }
  };
  f32x4 win[3][3];
  load_row(y0 - 1, win[0][0], win[0][1], win[0][2]);
  load_row(y0, win[1][0], win[1][1], win[1][2]);
  for (int dyy = 0; dyy < DWB_TY; ++dyy) {
    const int yo = y0 + dyy;
    if (yo >= H) break;
    load_row(yo + 1, win[2][0], win[2][1], win[2][2]);
    if (active) {
      f32x4 z = bv;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) z += win[ky][kx] * wv[ky * 3 + kx];
      const long long o = (img + (long long)yo * W + xo) * C + c;
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
      f32x4 d = g;
      if (GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = z[e];
          const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
          const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
          d[e] = g[e] * (cdf + v * pdf);
        }
        *reinterpret_cast<f32x4*>(dz + o) = d;
      }
      acc[9] += d;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += d * win[ky][kx];
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      win[0][kx] = win[1][kx];
      win[1][kx] = win[2][kx];
    }
  }
  const long long prow = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * xtiles + xt;
  float* out = partial + prow * 10 * C;
  for (int t = 0; t < 10; ++t) {
    __syncthreads();
    red[px][cq] = acc[t];
    __syncthreads();
    if (px == 0 && c < C) {
      f32x4 s = red[0][cq];
#pragma unroll
      for (int q = 1; q < 8; ++q) s += red[q][cq];
      *reinterpret_cast<f32x4*>(out + t * C + c) = s;
    }
  }
}

// plain depthwise 3x3 (no bias, no activation): dh = dwconv(dz, flipped taps)
__global__ __launch_bounds__(256) void dwconv3x3_plain_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                              float* __restrict__ y, int H, int W, int C) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)W * c4n) return;
  const int xo = (int)(idx / c4n), c = (int)(idx - (long long)xo * c4n) * 4;
  const int y0 = blockIdx.y * DWB_TY;
  const long long img = (long long)blockIdx.z * H * W;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  auto load_row = [&](int yy, f32x4& l, f32x4& m, f32x4& r) {
    l = m = r = zero;
    if ((unsigned)yy < (unsigned)H) {
      const float* p = x + (img + (long long)yy * W + xo) * C + c;
      m = *reinterpret_cast<const f32x4*>(p);
      if (xo > 0) l = *reinterpret_cast<const f32x4*>(p - C);
      if (xo + 1 < W) r = *reinterpret_cast<const f32x4*>(p + C);
    }
  };
  f32x4 win[3][3];
  load_row(y0 - 1, win[0][0], win[0][1], win[0][2]);
  load_row(y0, win[1][0], win[1][1], win[1][2]);
  for (int dyy = 0; dyy < DWB_TY; ++dyy) {
    const int yo = y0 + dyy;
    if (yo >= H) break;
    load_row(yo + 1, win[2][0], win[2][1], win[2][2]);
    f32x4 acc = zero;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc += win[ky][kx] * wv[ky * 3 + kx];
    *reinterpret_cast<f32x4*>(y + (img + (long long)yo * W + xo) * C + c) = acc;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      win[0][kx] = win[1][kx];
      win[1][kx] = win[2][kx];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Bilinear resize backward (adjoint), gather form: every input pixel sums the output pixels whose
// 2x2 footprint contains it, with the forward weights recomputed exactly.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bl_axis(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
  const float f = fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = (int)f;
  i1 = min(i0 + 1, in_size - 1);
  l1 = f - (float)i0;
}

template <int V>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int IH,
                                                           int IW, int OH, int OW, int C, int lddy, int lddx, float sy,
                                                           float sx, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cvn = C / V;
  const int c = (int)(idx % cvn) * V;
  long long pix = idx / cvn;
  const int ix = (int)(pix % IW);
  pix /= IW;
  const int iy = (int)(pix % IH);
  const long long b = pix / IH;
  // conservative candidate ranges in the output
  const int oy_lo = max(0, (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1);
  const int oy_hi = min(OH - 1, (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1);
  const int ox_lo = max(0, (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1);
  const int ox_hi = min(OW - 1, (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1);
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    int y0, y1;
    float ly;
    bl_axis(oy, sy, IH, y0, y1, ly);
    float wy = 0.f;
    if (y0 == iy) wy += 1.f - ly;
    if (y1 == iy) wy += ly;
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      int x0, x1;
      float lx;
      bl_axis(ox, sx, IW, x0, x1, lx);
      float wx = 0.f;
      if (x0 == ix) wx += 1.f - lx;
      if (x1 == ix) wx += lx;
      if (wx == 0.f) continue;
      const float* g = dy + ((b * OH + oy) * OW + ox) * lddy + c;
      const float w = wy * wx;
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += w * g[e];
    }
  }
  float* o = dx + ((b * IH + iy) * IW + ix) * lddx + c;
#pragma unroll
  for (int e = 0; e < V; ++e) o[e] = acc[e];
}

// ---------------------------------------------------------------------------------------------------
// Row softmax (attention training path): p = softmax(s * scale) in place; backward
// ds = p * (dp - sum(p * dp)) * scale in place on dp.  One wave per row, L <= 1024.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ s, long long rows, int L, int ld, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* p = s + row * ld;
  float v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = lane + 64 * i;
    v[i] = j < L ? p[j] * scale : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = expf(v[i] - mx);
    sum += v[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = lane + 64 * i;
    if (j < L) p[j] = v[i] * inv;
    else if (j < ld) p[j] = 0.f;  // pitch padding: a later GEMM contracts over the whole pitch
  }
}

__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                              long long rows, int L, int ld, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* pr = p + row * ld;
  float* dr = dp + row * ld;
  float pv[16], dv[16];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = lane + 64 * i;
    pv[i] = j < L ? pr[j] : 0.f;
    dv[i] = j < L ? dr[j] : 0.f;
    dot += pv[i] * dv[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = lane + 64 * i;
    if (j < L) dr[j] = pv[i] * (dv[i] - dot) * scale;
    else if (j < ld) dr[j] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------
// Softmax cross-entropy with ignore_index over NHWC logits (rows x C, C <= 32):
// per-block partial (loss sum, valid count) and UNNORMALISED gradient (softmax - onehot, 0 if ignored).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                         float* __restrict__ dlogits, double* __restrict__ partial,
                                                         long long rows, int C, int ld, int ldd, int ignore_index) {
  __shared__ double red[2][256];
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  double loss = 0.0, cnt = 0.0;
  if (r < rows) {
    const float* x = logits + r * ld;
    const long long lab = labels[r];
    float v[32];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
      v[c] = x[c];
      mx = fmaxf(mx, v[c]);
    }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
      v[c] = expf(v[c] - mx);
      sum += v[c];
    }
    const bool valid = lab != ignore_index && lab >= 0 && lab < C;
    if (valid) {
      loss = (double)(logf(sum) + mx - x[lab]);
      cnt = 1.0;
    }
    if (dlogits) {
      const float inv = 1.0f / sum;
      for (int c = 0; c < C; ++c) dlogits[r * ldd + c] = valid ? (v[c] * inv - (c == lab ? 1.f : 0.f)) : 0.f;
    }
  }
  red[0][threadIdx.x] = loss;
  red[1][threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * (long long)blockIdx.x] = red[0][0];
    partial[2 * (long long)blockIdx.x + 1] = red[1][0];
  }
}

// ---------------------------------------------------------------------------------------------------
// Input gradient of a STRIDED convolution (overlap patch embed k7s4 / k3s2, sr conv k = s):
//   dx[b][iy][ix][c] = sum over taps (ky,kx) with (iy + pad - ky) % s == 0, (ix + pad - kx) % s == 0
//                      of sum_n dy[b][oy][ox][n] * W[n][c][ky][kx]
// wd is the weight repacked to [tap][n][c].  VALU kernel: these layers are ~3 % of the FLOPs.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_dgrad_strided_kernel(const float* __restrict__ dy, const float* __restrict__ wd,
                                                                 float* __restrict__ dx, int H, int W, int Cin, int N, int KH,
                                                                 int KW, int stride, int pad, int OH, int OW, int lddy,
                                                                 int lddx, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % Cin);
  long long pix = idx / Cin;
  const int ix = (int)(pix % W);
  pix /= W;
  const int iy = (int)(pix % H);
  const long long b = pix / H;
  float acc = 0.f;
  for (int ky = 0; ky < KH; ++ky) {
    const int ty = iy + pad - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= OH) continue;
    for (int kx = 0; kx < KW; ++kx) {
      const int tx = ix + pad - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= OW) continue;
      const float* g = dy + ((b * OH + oy) * OW + ox) * lddy;
      const float* w = wd + ((long long)(ky * KW + kx) * N) * Cin + c;
      for (int n = 0; n < N; ++n) acc = fmaf(g[n], w[(long long)n * Cin], acc);
    }
  }
  dx[((b * H + iy) * W + ix) * lddx + c] = acc;
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d on NHWC rows (the SegFormer head's linear_fuse.bn, segformer_head.py:50-55), train mode.
//   bn_colstats<MODE>: per-column partial sums over 256-row blocks (fp64 partial rows, summed by
//                      segmif_colsum-style final pass on the host side):
//        MODE 0: sum (x - mu)^2            (mu = per-column mean, for the biased batch variance)
//        MODE 1: [sum dz | sum dz * xhat]  (backward reductions; dz already masked by the ReLU)
//   bn_apply:        y = relu(x * s[c] + t[c])
//   bn_bwd_apply:    dx = s[c] * (dz - a[c] - xhat * b[c]),  a = mean dz, b = mean dz*xhat
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                          const float* __restrict__ mu, const float* __restrict__ rstd,
                                                          double* __restrict__ partial, long long rows, int C) {
  __shared__ float red[2][4][64];
  const int c = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * 256;
  const long long r1 = r0 + 256 < rows ? r0 + 256 : rows;
  const int W = MODE == 0 ? C : 2 * C;
  for (int n0 = 0; n0 < C; n0 += 64) {
    const int n = n0 + c;
    float s0 = 0.f, s1 = 0.f;
    if (n < C) {
      const float m = mu[n], rs = MODE == 1 ? rstd[n] : 0.f;
      for (long long r = r0 + slot; r < r1; r += 4) {
        const float xv = x[r * C + n];
        if (MODE == 0) {
          const float d = xv - m;
          s0 += d * d;
        } else {
          const float g = dz[r * C + n];
          s0 += g;
          s1 += g * (xv - m) * rs;
        }
      }
    }
    red[0][slot][c] = s0;
    red[1][slot][c] = s1;
    __syncthreads();
    if (slot == 0 && n < C) {
      partial[(long long)blockIdx.x * W + n] = ((double)red[0][0][c] + red[0][1][c]) + ((double)red[0][2][c] + red[0][3][c]);
      if (MODE == 1)
        partial[(long long)blockIdx.x * W + C + n] = ((double)red[1][0][c] + red[1][1][c]) + ((double)red[1][2][c] + red[1][3][c]);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void colsum_f64_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                                         int nblk, int N) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  double s = 0.0;
  if (n < N)
    for (int b = slot; b < nblk; b += 4) s += partial[(long long)b * N + n];
  red[slot][c] = s;
  __syncthreads();
  if (slot == 0 && n < N) out[n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       const float* __restrict__ t, float* __restrict__ y, long long total4,
                                                       int C, int relu) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c = (int)((i * 4) % C);
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4);
  const f32x4 sv = *reinterpret_cast<const f32x4*>(s + c), tv = *reinterpret_cast<const f32x4*>(t + c);
  f32x4 o = xv * sv + tv;
  if (relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
  }
  *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                           const float* __restrict__ mu, const float* __restrict__ rstd,
                                                           const float* __restrict__ s, const float* __restrict__ a,
                                                           const float* __restrict__ b, float* __restrict__ dx,
                                                           long long total4, int C) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c = (int)((i * 4) % C);
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4), g = *reinterpret_cast<const f32x4*>(dz + i * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float xh = (xv[e] - mu[c + e]) * rstd[c + e];
    o[e] = s[c + e] * (g[e] - a[c + e] - xh * b[c + e]);
  }
  *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// ---------------------------------------------------------------------------------------------------
// Multi-tensor AdamW (decoupled weight decay), torch.optim.AdamW arithmetic:
//   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// One launch walks a table of (param, grad, m, v, numel, lr, wd) entries; grads that are null are skipped.
// ---------------------------------------------------------------------------------------------------
struct AdamEntry {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
  float lr, wd;
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamEntry* __restrict__ table, const int* __restrict__ chunk_entry,
                                                    const long long* __restrict__ chunk_off, float b1, float b2, float eps,
                                                    float bc1, float bc2_sqrt, int chunk_elems) {
  const AdamEntry e = table[chunk_entry[blockIdx.x]];
  if (!e.g) return;
  const long long base = chunk_off[blockIdx.x];
  const float step = e.lr / bc1, decay = 1.0f - e.lr * e.wd;
  for (int i = threadIdx.x; i < chunk_elems; i += 256) {
    const long long j = base + i;
    if (j >= e.n) break;
    const float g = e.g[j];
    const float m = b1 * e.m[j] + (1.0f - b1) * g;
    const float v = b2 * e.v[j] + (1.0f - b2) * g * g;
    e.m[j] = m;
    e.v[j] = v;
    e.p[j] = e.p[j] * decay - step * m / (sqrtf(v) / bc2_sqrt + eps);
  }
}

// Input gradient of a strided convolution, second half.  First half: cols = dY (rows = output pixels) x W^T as ONE dense GEMM
// on the matrix pipe, cols[b][oy][ox][(ky, kx, c)]; this kernel gathers them back: input pixel (y, x) receives the taps
// with ky = (y + pad) mod s (+ s, + 2 s ..) from output pixel ((y + pad - ky) / s, ..).  The scalar gather kernel it replaces
// (conv_dgrad_strided_kernel) did the contraction itself on the vector ALU: 7 TFLOP/s on the stage-2 patch embed.
template <int V>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ cols, float* __restrict__ dx, int H, int W, int C,
                                                     int k, int s, int pad, int OH, int OW, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cv = C / V;
  const int c = (int)(idx % cv) * V;
  const long long px = idx / cv;
  const int x = (int)(px % W), y = (int)((px / W) % H);
  const long long b = px / ((long long)W * H);
  const long long K9 = (long long)k * k * C;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  for (int ky = (y + pad) % s; ky < k; ky += s) {
    const int ty = y + pad - ky;
    if (ty < 0) break;
    const int oy = ty / s;
    if (oy >= OH) continue;
    for (int kx = (x + pad) % s; kx < k; kx += s) {
      const int tx = x + pad - kx;
      if (tx < 0) break;
      const int ox = tx / s;
      if (ox >= OW) continue;
      const float* src = cols + ((b * OH + oy) * OW + ox) * K9 + (long long)(ky * k + kx) * C + c;
      if (V == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
      } else acc[0] += src[0];
    }
  }
  float* dst = dx + px * C + c;
  if (V == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  else dst[0] = acc[0];
}

}  // namespace

extern "C" int segmif_col2im_f32(const float* cols, float* dx, int B, int H, int W, int C, int k, int stride, int pad, int OH,
                                 int OW, void* stream) {
  if (!cols || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || OH <= 0 || OW <= 0)
    return SEGMIF_EINVAL;
  const bool vec = !(C & 3) && !(((uintptr_t)cols | (uintptr_t)dx) & 15);
  const long long total = (long long)B * H * W * (vec ? C / 4 : C);
  if (vec)
    hipLaunchKernelGGL(col2im_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cols, dx, H, W, C,
                       k, stride, pad, OH, OW, total);
  else
    hipLaunchKernelGGL(col2im_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cols, dx, H, W, C,
                       k, stride, pad, OH, OW, total);
  return (int)hipGetLastError();
}

extern "C" int segmif_layernorm_bwd_blocks(int64_t rows, int C) {
  const int nvec = C >> 2;
  const int G = nvec <= 8 ? 8 : nvec <= 16 ? 16 : nvec <= 32 ? 32 : 64;
  const int slots = 256 / G;
  long long rpb = (rows + 1023) / 1024;
  rpb = (rpb + slots - 1) / slots * slots;
  if (rpb < slots) rpb = slots;
  return (int)((rows + rpb - 1) / rpb);
}

static int ln_bwd_dispatch(const float* x, const float* dy, const float* gamma, float* dx, float* partial, int64_t rows, int C,
                           int ldx, int ldy, int lddx, float eps, void* stream, const LnBwdAdd& a) {
  if (!x || !dy || !gamma || !dx || !partial || rows <= 0 || C <= 0 || (C & 3) || C > 1024 || (ldx & 3) || (ldy & 3) ||
      (lddx & 3))
    return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma) & 15) return SEGMIF_EINVAL;
  const int nblk = segmif_layernorm_bwd_blocks(rows, C);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  const int nvec = C >> 2;
  const int G = nvec <= 8 ? 8 : nvec <= 16 ? 16 : nvec <= 32 ? 32 : 64;
  const int slots = 256 / G;
  const int rpb_al = (rpb + slots - 1) / slots * slots;
  const int nblk2 = (int)((rows + rpb_al - 1) / rpb_al);
  if (nblk2 > nblk) return SEGMIF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // partial rows beyond nblk2 (if any) must not be read: zero them
  if (nblk2 < nblk) hipMemsetAsync(partial + (long long)nblk2 * 2 * C, 0, (size_t)(nblk - nblk2) * 2 * C * sizeof(float), s);
  if (nvec <= 8) return launch_ln_bwd<8, 1>(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, rpb_al, nblk2, s, a);
  if (nvec <= 16) return launch_ln_bwd<16, 1>(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, rpb_al, nblk2, s, a);
  if (nvec <= 32) return launch_ln_bwd<32, 1>(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, rpb_al, nblk2, s, a);
  if (nvec <= 64) return launch_ln_bwd<64, 1>(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, rpb_al, nblk2, s, a);
  if (nvec <= 128) return launch_ln_bwd<64, 2>(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, rpb_al, nblk2, s, a);
  return launch_ln_bwd<64, 4>(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, rpb_al, nblk2, s, a);
}

extern "C" int segmif_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx, float* partial,
                                        int64_t rows, int C, int ldx, int ldy, int lddx, float eps, void* stream) {
  return ln_bwd_dispatch(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, stream, LnBwdAdd());
}

extern "C" int segmif_layernorm_bwd_add_f32(const float* x, const float* dy, const float* gamma, const float* dres, int lddres,
                                            const float* scale, int64_t rows_per_image, float* dx, float* dbranch, int lddbr,
                                            float* partial, int64_t rows, int C, int ldx, int ldy, int lddx, float eps,
                                            void* stream) {
  if ((dres && ((lddres & 3) || ((uintptr_t)dres & 15))) || (dbranch && (!scale || (lddbr & 3) || ((uintptr_t)dbranch & 15))))
    return SEGMIF_EINVAL;
  if (scale && (rows_per_image <= 0 || rows % rows_per_image)) return SEGMIF_EINVAL;
  LnBwdAdd a;
  a.dres = dres; a.lddres = lddres; a.scale = scale; a.rpi = scale ? rows_per_image : 1; a.dbr = dbranch; a.lddbr = lddbr;
  return ln_bwd_dispatch(x, dy, gamma, dx, partial, rows, C, ldx, ldy, lddx, eps, stream, a);
}

extern "C" int64_t segmif_dwconv_bwd_partial_rows(int B, int H, int W) {
  return (int64_t)B * ((H + DWB_TY - 1) / DWB_TY) * ((W + 7) / 8);
}

extern "C" int segmif_dwconv3x3_gelu_bwd_f32(const float* h, const float* w9, const float* bias, const float* dy, float* dz,
                                             float* partial, int B, int H, int W, int C, void* stream) {
  if (!h || !w9 || !bias || !dy || !dz || !partial || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 128)) return SEGMIF_EINVAL;
  const int xtiles = (W + 7) / 8;
  dim3 grid((unsigned)((C / 128) * xtiles), (unsigned)((H + DWB_TY - 1) / DWB_TY), (unsigned)B);
  hipLaunchKernelGGL(dwconv_gelu_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, h, w9, bias, dy, dz, partial, H, W,
                     C, xtiles);
  return (int)hipGetLastError();
}

// parameter-gradient partials of a bare depthwise 3x3 + bias (DWConv.forward without the GELU): same partial layout,
// dz == dy so nothing else is written
extern "C" int segmif_dwconv3x3_bias_bwd_f32(const float* h, const float* w9, const float* dy, float* partial, int B, int H,
                                             int W, int C, void* stream) {
  if (!h || !w9 || !dy || !partial || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 128)) return SEGMIF_EINVAL;
  const int xtiles = (W + 7) / 8;
  dim3 grid((unsigned)((C / 128) * xtiles), (unsigned)((H + DWB_TY - 1) / DWB_TY), (unsigned)B);
  hipLaunchKernelGGL(dwconv_gelu_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, h, w9, w9 /* bias unused */, dy,
                     (float*)nullptr, partial, H, W, C, xtiles);
  return (int)hipGetLastError();
}

extern "C" int segmif_dwconv3x3_plain_f32(const float* x, const float* w9, float* y, int B, int H, int W, int C,
                                          void* stream) {
  if (!x || !w9 || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return SEGMIF_EINVAL;
  const long long per_row = (long long)W * (C >> 2);
  dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)((H + DWB_TY - 1) / DWB_TY), (unsigned)B);
  hipLaunchKernelGGL(dwconv3x3_plain_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w9, y, H, W, C);
  return (int)hipGetLastError();
}

extern "C" int segmif_bilinear_nhwc_bwd_f32(const float* dy, float* dx, int B, int IH, int IW, int OH, int OW, int C,
                                            int lddy, int lddx, void* stream) {
  if (!dy || !dx || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || lddy < C || lddx < C) return SEGMIF_EINVAL;
  const bool vec = !(C & 3);
  const long long total = (long long)B * IH * IW * (vec ? (C >> 2) : C);
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (vec)
    hipLaunchKernelGGL(bilinear_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, IH, IW, OH, OW, C, lddy,
                       lddx, sy, sx, total);
  else
    hipLaunchKernelGGL(bilinear_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, IH, IW, OH, OW, C, lddy,
                       lddx, sy, sx, total);
  return (int)hipGetLastError();
}

extern "C" int segmif_row_softmax_f32(float* s, int64_t rows, int L, int ld, float scale, void* stream) {
  if (!s || rows <= 0 || L <= 0 || ld > 1024 || ld < L) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s,
                     (long long)rows, L, ld, scale);
  return (int)hipGetLastError();
}

extern "C" int segmif_row_softmax_bwd_f32(const float* p, float* dp, int64_t rows, int L, int ld, float scale, void* stream) {
  if (!p || !dp || rows <= 0 || L <= 0 || ld > 1024 || ld < L) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp,
                     (long long)rows, L, ld, scale);
  return (int)hipGetLastError();
}

extern "C" int segmif_softmax_ce_blocks(int64_t rows) { return (int)((rows + 255) / 256); }

extern "C" int segmif_softmax_ce_f32(const float* logits, const int64_t* labels, float* dlogits, double* partial,
                                     int64_t rows, int C, int ld, int ldd, int ignore_index, void* stream) {
  if (!logits || !labels || !partial || rows <= 0 || C <= 0 || C > 32 || ld < C || (dlogits && ldd < C)) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)segmif_softmax_ce_blocks(rows)), dim3(256), 0, (hipStream_t)stream,
                     logits, (const long long*)labels, dlogits, partial, (long long)rows, C, ld, ldd, ignore_index);
  return (int)hipGetLastError();
}

extern "C" int segmif_conv_dgrad_strided_f32(const float* dy, const float* wd, float* dx, int B, int H, int W, int Cin, int N,
                                             int KH, int KW, int stride, int pad, int OH, int OW, int lddy, int lddx,
                                             void* stream) {
  if (!dy || !wd || !dx || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || N <= 0 || stride <= 0) return SEGMIF_EINVAL;
  const long long total = (long long)B * H * W * Cin;
  hipLaunchKernelGGL(conv_dgrad_strided_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dy, wd, dx, H, W, Cin, N, KH, KW, stride, pad, OH, OW, lddy, lddx, total);
  return (int)hipGetLastError();
}

extern "C" int segmif_adamw_entry_bytes(void) { return (int)sizeof(AdamEntry); }

extern "C" int segmif_adamw_f32(const void* table, const int32_t* chunk_entry, const int64_t* chunk_off, int nchunks,
                                int chunk_elems, float beta1, float beta2, float eps, float bc1, float bc2s, void* stream) {
  if (!table || !chunk_entry || !chunk_off || nchunks <= 0 || chunk_elems <= 0 || !(bc1 > 0.f) || !(bc2s > 0.f)) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)table,
                     (const int*)chunk_entry, (const long long*)chunk_off, beta1, beta2, eps, bc1, bc2s, chunk_elems);
  return (int)hipGetLastError();
}

extern "C" int segmif_bn_colstats_f32(const float* x, const float* dz, const float* mu, const float* rstd, double* partial,
                                      double* out, int64_t rows, int C, int mode, void* stream) {
  // mode 0: out[C] = sum (x - mu)^2 ; mode 1: out[2C] = [sum dz | sum dz * xhat].  partial: ceil(rows/256) * (mode ? 2C : C) doubles
  if (!x || !mu || !partial || !out || rows <= 0 || C <= 0 || (mode == 1 && (!dz || !rstd))) return SEGMIF_EINVAL;
  const int nblk = (int)((rows + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  const int Wd = mode == 0 ? C : 2 * C;
  if (mode == 0) hipLaunchKernelGGL(bn_colstats_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, s, x, dz, mu, rstd, partial, (long long)rows, C);
  else hipLaunchKernelGGL(bn_colstats_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, s, x, dz, mu, rstd, partial, (long long)rows, C);
  hipLaunchKernelGGL(colsum_f64_kernel, dim3((unsigned)((Wd + 63) / 64)), dim3(256), 0, s, partial, out, nblk, Wd);
  return (int)hipGetLastError();
}

extern "C" int segmif_bn_apply_f32(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C,
                                   int relu, void* stream) {
  if (!x || !scale || !shift || !y || rows <= 0 || C <= 0 || (C & 3)) return SEGMIF_EINVAL;
  const long long total4 = (long long)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale,
                     shift, y, total4, C, relu);
  return (int)hipGetLastError();
}

extern "C" int segmif_bn_bwd_apply_f32(const float* x, const float* dz, const float* mu, const float* rstd,
                                       const float* scale, const float* a, const float* b, float* dx, int64_t rows, int C,
                                       void* stream) {
  if (!x || !dz || !mu || !rstd || !scale || !a || !b || !dx || rows <= 0 || C <= 0 || (C & 3)) return SEGMIF_EINVAL;
  const long long total4 = (long long)rows * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dz,
                     mu, rstd, scale, a, b, dx, total4, C);
  return (int)hipGetLastError();
}
