// Bandwidth-bound kernels of the SegMiF hot path (gfx950, wave64): LayerNorm, the Mix-FFN
// depthwise 3x3 + GELU, bilinear resize, layout transposes and the per-pixel colour / normalise /
// argmax helpers.  All are NHWC (== token layout) with 16-byte vector accesses; the roofline for
// every kernel here is HBM (~6.3 TB/s achievable), so the rule is: one read, one write, float4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "planes16.h"
#include "segmif_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm: G lanes cooperate on one row (G = 8..64, power of two), values stay in registers,
// two-pass mean / biased variance like aten's (core/mix_transformer.py:152-153 etc.).
// ---------------------------------------------------------------------------------------------
// Residual form (training path, autograd.AddLayerNormFn): br != NULL normalises s = x + scale[row / rpi] * br (scale NULL = 1:
// rpi rows per image, the per-sample stochastic-depth factor) and also stores s - the residual add, the DropPath multiply and
// the LayerNorm of a transformer block in one pass.
template <int G, int IT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long long rows, int C, int ldx, int ldy, float eps,
                                                        const float* __restrict__ br, int ldb, const float* __restrict__ scale,
                                                        long long rpi, float* __restrict__ sum_out, int lds) {
  constexpr int RPB = 256 / G;  // rows per block
  const int tid = threadIdx.x;
  const int sub = tid % G;
  const long long row = (long long)blockIdx.x * RPB + tid / G;
  const bool row_ok = row < rows;
  const int nvec = C >> 2;
  f32x4 v[IT];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int u = sub + it * G;
    v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row_ok && u < nvec) {
      v[it] = *reinterpret_cast<const f32x4*>(x + row * ldx + 4 * u);
      if (br) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(br + row * ldb + 4 * u);
        const float sc = scale ? scale[row / rpi] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[it][e] = fmaf(sc, b4[e], v[it][e]);
        *reinterpret_cast<f32x4*>(sum_out + row * lds + 4 * u) = v[it];
      }
    }
    sum += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int u = sub + it * G;
    if (u < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[it][e] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int u = sub + it * G;
    if (row_ok && u < nvec) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * u);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + 4 * u);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * g[e] + bb[e];
      *reinterpret_cast<f32x4*>(y + row * ldy + 4 * u) = o;
    }
  }
}

// (r5) LayerNorm writing PAIRS rows (gemm_pairs.hip: 16-channel groups of [16 hi | 16 lo] halves; the byte count is fp32's) with
// max |y| folded into the range slot of each row's image.  Same arithmetic and launch shape as layernorm_kernel.  Its waves live
// for a row or two, so (planes16.h, fold_pat_async) they report with a fire-and-forget atomic - no look first, whose latency
// every wave would end on - and the launch is given `nsub` ROWS of slots (a power of two, `sub_stride` words apart), workgroup i
// reporting to row i % nsub: the 76 800 atomics of a stage-3 launch meet on nsub x images addresses instead of `images`.
template <int G, int IT>
__global__ __launch_bounds__(1024) void layernorm_pairs_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, unsigned char* __restrict__ y,
                                                              long long rows, int C, int ldx, long long ldy_bytes, float eps,
                                                              uint32_t* __restrict__ amax, long long amax_rows, int nsub,
                                                              long long sub_stride) {
  constexpr int RPB = 1024 / G;  // 16 waves per workgroup: ONE range report per workgroup (below)
  const int tid = threadIdx.x;
  const int sub = tid % G;
  const long long row = (long long)blockIdx.x * RPB + tid / G;
  const bool row_ok = row < rows;
  const int nvec = C >> 2;
  f32x4 v[IT];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int u = sub + it * G;
    v[it] = (row_ok && u < nvec) ? *reinterpret_cast<const f32x4*>(x + row * ldx + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    sum += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int u = sub + it * G;
    if (u < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[it][e] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  uint32_t amx = 0u;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int u = sub + it * G;
    if (row_ok && u < nvec) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * u);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + 4 * u);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * g[e] + bb[e];
      uint32_t ha, la, hb, lb;
      segmif::p16::split2(o[0], o[1], ha, la);
      segmif::p16::split2(o[2], o[3], hb, lb);
      // (the quad u & ~3 .. u | 3 = one 16-channel group is active as a whole: C % 16 == 0)
      *reinterpret_cast<segmif::p16::u4*>(y + row * ldy_bytes + (u >> 2) * 64 + (u & 3) * 16) = segmif::p16::quad_piece(ha, hb, la, lb);
      amx = segmif::p16::absmax_pk(segmif::p16::absmax_pk(amx, ha, la), hb, lb);
    }
  }
  if (amax) {
    // The workgroup's RPB consecutive rows normally belong to one image: its 16 waves meet in LDS and one lane reports with a
    // fire-and-forget atomic (planes16.h, fold_pat_block).  Slot words of neighbouring images share a cache line and atomics on one
    // line queue up (~10 ns each): with one report per WAVE a stage-2 launch queued 9 600 of them per line and took 118 us
    // instead of 31.  A workgroup that straddles two images reports per wave.
    uint32_t* slots = amax + (long long)(blockIdx.x & (unsigned)(nsub - 1)) * sub_stride;
    const long long first = (long long)blockIdx.x * RPB;
    long long last = first + RPB - 1;
    if (last >= rows) last = rows - 1;
    const int b0 = (int)(first / amax_rows), b1 = (int)(last / amax_rows);
    if (b0 == b1) {
      segmif::p16::fold_pat_block(slots, b0, amx);
    } else {
      const long long r0 = first + (tid & ~63) / G;
      long long r1 = r0 + 64 / G - 1;
      if (r1 >= rows) r1 = rows - 1;
      if (r0 < rows) segmif::p16::fold_pat_async(slots, (int)(r0 / amax_rows), (int)(r1 / amax_rows), amx);
    }
  }
}

template <int G, int IT>
int launch_ln_pairs(const float* x, const float* g, const float* b, void* y, long long rows, int C, int ldx, int ldy, float eps,
                    uint32_t* amax, long long amax_rows, int nsub, long long sub_stride, hipStream_t s) {
  constexpr int RPB = 1024 / G;
  hipLaunchKernelGGL((layernorm_pairs_kernel<G, IT>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(1024), 0, s, x, g, b,
                     reinterpret_cast<unsigned char*>(y), rows, C, ldx, (long long)ldy * 4, eps, amax, amax_rows, nsub, sub_stride);
  return (int)hipGetLastError();
}

struct LnAdd {  // the residual form's extra operands (all zero: plain LayerNorm)
  const float* br = nullptr; int ldb = 0; const float* scale = nullptr; long long rpi = 1; float* sum_out = nullptr; int lds = 0;
};

template <int G, int IT>
int launch_ln(const float* x, const float* g, const float* b, float* y, long long rows, int C, int ldx, int ldy,
              float eps, hipStream_t s, const LnAdd& a = LnAdd()) {
  constexpr int RPB = 256 / G;
  hipLaunchKernelGGL((layernorm_kernel<G, IT>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, s, x, g, b, y,
                     rows, C, ldx, ldy, eps, a.br, a.ldb, a.scale, a.rpi, a.sum_out, a.lds);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3 + bias + exact GELU.  A thread owns (x, 4 channels) and slides down TY rows with
// a 3x3 register window, so each input row is fetched once per strip (+ the 2-row halo) and the
// x-neighbours come from adjacent lanes' cache lines.
// ---------------------------------------------------------------------------------------------
constexpr int DW_TY = 8;

__device__ __forceinline__ float gelu_exact(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// (r6) The erf GELU as csrc/mixffn.hip computes it for stages 1-2 (Abramowitz & Stegun 7.1.26: |error of erf| <= 1.5e-7, i.e.
// <= 7.5e-8 |x| on GELU - about one fp32 ulp of x; 12 instructions instead of erff's ~35), for the PAIRS producer of stages 3-4:
// the same function on both halves of the encoder, inside the same guarded inference scope; the fp32 kernel (training, and the
// path a tripped pair is repeated on) keeps erff.
__device__ __forceinline__ float gelu_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float q = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);  // (coefficients carry the 1/2 of erfc / 2)
  q = fmaf(q, t, 0.5f * 1.421413741f);
  q = fmaf(q, t, 0.5f * -0.284496736f);
  q = fmaf(q, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2 / 2)
  const float half_erfc = q * t * e;
  const float phi = x >= 0.f ? 1.0f - half_erfc : half_erfc;
  return x * phi;
}

template <bool GELU>
__global__ __launch_bounds__(256) void dwconv3x3_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int H, int W, int C) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)W * c4n) return;
  const int xo = (int)(idx / c4n), c = (int)(idx - (long long)xo * c4n) * 4;
  const int y0 = blockIdx.y * DW_TY;
  const long long img = (long long)blockIdx.z * H * W;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
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
#pragma unroll
  for (int dy = 0; dy < DW_TY; ++dy) {
    const int yo = y0 + dy;
    if (yo >= H) break;
    load_row(yo + 1, win[2][0], win[2][1], win[2][2]);
    f32x4 acc = bv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc += win[ky][kx] * wv[ky * 3 + kx];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = GELU ? gelu_exact(acc[e]) : acc[e];
    *reinterpret_cast<f32x4*>(y + (img + (long long)yo * W + xo) * C + c) = o;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      win[0][kx] = win[1][kx];
      win[1][kx] = win[2][kx];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Bilinear resize, align_corners=False (aten area_pixel_compute_source_index semantics):
//   src = max(0, scale * (dst + 0.5) - 0.5), scale = in / out (fp32), i1 = min(i0 + 1, in - 1).
// ---------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int IH,
                                                       int IW, int OH, int OW, int C, int ldx, int ldo, float sy,
                                                       float sx) {
  // grid = (units of an output row, output row, image): the row terms are wave-uniform and the only division left is a
  // 32-bit one (the first version spent three 64-bit divisions per 16-byte store: 2.8 TB/s on the forward_fusion resizes)
  const int cvn = C / V;
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (unsigned)(OW * cvn)) return;
  const int ox = (int)(t / (unsigned)cvn);
  const int c = (int)(t - (unsigned)ox * cvn) * V;
  const int oy = blockIdx.y;
  const long long b = blockIdx.z;
  const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, IH - 1), x1 = min(x0 + 1, IW - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* base = x + b * IH * IW * ldx + c;
  const float* p00 = base + ((long long)y0 * IW + x0) * ldx;
  const float* p01 = base + ((long long)y0 * IW + x1) * ldx;
  const float* p10 = base + ((long long)y1 * IW + x0) * ldx;
  const float* p11 = base + ((long long)y1 * IW + x1) * ldx;
  float* dst = y + ((b * OH + oy) * OW + ox) * ldo + c;
  if (V == 4) {
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(p00), v01 = *reinterpret_cast<const f32x4*>(p01);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(p10), v11 = *reinterpret_cast<const f32x4*>(p11);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(dst) = o;
  } else {
    *dst = hy * (hx * *p00 + lx * *p01) + ly * (hx * *p10 + lx * *p11);
  }
}

// ---------------------------------------------------------------------------------------------
// Batched 2-D transpose through LDS: in (B, R, Cc) pitch ldi -> out (B, Cc, R) pitch ldo.
// NCHW -> NHWC is R = C, Cc = HW; NHWC -> NCHW is R = HW, Cc = C.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, long long R,
                                                        long long Cc, long long ldi, long long ldo,
                                                        long long in_bs, long long out_bs) {
  __shared__ float tile[32][33];
  const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
  const float* xi = x + (long long)blockIdx.z * in_bs;
  float* yo = y + (long long)blockIdx.z * out_bs;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const long long r = r0 + ty + j, c = c0 + tx;
    if (r < R && c < Cc) tile[ty + j][tx] = xi[r * ldi + c];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const long long c = c0 + ty + j, r = r0 + tx;
    if (r < R && c < Cc) yo[c * ldo + r] = tile[tx][ty + j];
  }
}

// (x*255 - mean_c) / std_c, NCHW (B,3,H,W) -> NHWC (B,H,W,3)   core/model_fusion.py:1079-1085
__global__ __launch_bounds__(256) void seg_normalize_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            long long HW, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long b = idx / HW, p = idx - b * HW;
  const float mean[3] = {123.675f, 116.28f, 103.53f};
  const float stdv[3] = {58.395f, 57.12f, 57.375f};
#pragma unroll
  for (int c = 0; c < 3; ++c) y[idx * 3 + c] = (x[(b * 3 + c) * HW + p] * 255.f - mean[c]) / stdv[c];
}

// fused = clamp01(YCrCb2RGB([Yf, Cr(vis), Cb(vis)]))  core/model_fusion.py:69-111, test_fusion.py:102-111
__global__ __launch_bounds__(256) void fuse_ycrcb_kernel(const float* __restrict__ vis, const float* __restrict__ yf,
                                                         float* __restrict__ out, long long HW, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long b = idx / HW, p = idx - b * HW;
  const float R = vis[(b * 3 + 0) * HW + p], G = vis[(b * 3 + 1) * HW + p], Bc = vis[(b * 3 + 2) * HW + p];
  const float Y = 0.299f * R + 0.587f * G + 0.114f * Bc;
  const float Cr = (R - Y) * 0.713f + 0.5f;
  const float Cb = (Bc - Y) * 0.564f + 0.5f;
  const float t0 = yf[idx] + 0.0f, t1 = Cr + -0.5f, t2 = Cb + -0.5f;
  // row-vector times the 3x3 matrix of model_fusion.py:96-98, accumulated in aten mm order
  float r = t0 * 1.0f + t1 * 1.403f + t2 * 0.0f;
  float g = t0 * 1.0f + t1 * -0.714f + t2 * -0.344f;
  float bl = t0 * 1.0f + t1 * 0.0f + t2 * 1.773f;
  out[(b * 3 + 0) * HW + p] = fminf(fmaxf(r, 0.f), 1.f);
  out[(b * 3 + 1) * HW + p] = fminf(fmaxf(g, 0.f), 1.f);
  out[(b * 3 + 2) * HW + p] = fminf(fmaxf(bl, 0.f), 1.f);
}

// (r6) Two-source pointwise pass over rows views (the ablation networks' glue, core/model_fusion.py:714-820):
//   MODE 0  a + b                       Fusion_Network3_Add  (:745-746, :751-752)
//   MODE 1  silu(a) + silu(b)           Fusion_Network3_Average: att_i(x) + att_j(seg), AttentionModule = sigmoid(z) z  (:769-770)
//   MODE 2  silu(a)                     AttentionModule alone
__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
template <int MODE>
__global__ __launch_bounds__(256) void pointwise2_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                         float* __restrict__ y, int ldy, long long rows, int C4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * C4) return;
  const long long r = idx / C4;
  const int c = (int)(idx - r * C4) * 4;
  float4 va = *reinterpret_cast<const float4*>(a + r * lda + c);
  if (MODE == 0) {
    const float4 vb = *reinterpret_cast<const float4*>(b + r * ldb + c);
    va = make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w);
  } else if (MODE == 1) {
    const float4 vb = *reinterpret_cast<const float4*>(b + r * ldb + c);
    va = make_float4(silu_f(va.x) + silu_f(vb.x), silu_f(va.y) + silu_f(vb.y), silu_f(va.z) + silu_f(vb.z), silu_f(va.w) + silu_f(vb.w));
  } else {
    va = make_float4(silu_f(va.x), silu_f(va.y), silu_f(va.z), silu_f(va.w));
  }
  *reinterpret_cast<float4*>(y + r * ldy + c) = va;
}

// RGB2YCrCb / YCrCb2RGB (core/model_fusion.py:69-91, :93-111) and their backward, on planar (B, 3, HW) images.
//   MODE 0  RGB -> YCrCb          Y = .299 R + .587 G + .114 B, Cr = (R - Y) .713 + .5, Cb = (B - Y) .564 + .5
//   MODE 1  YCrCb -> RGB          ([Y, Cr, Cb] + [0, -.5, -.5]) M, M = [[1, 1, 1], [1.403, -.714, 0], [0, -.344, 1.773]], summed
//                                 in aten mm order; ysrc != null supplies channel 0 from a (B, 1, HW) tensor instead
//                                 (train.py:362-364: fused_ycbcr = vis.clone(); fused_ycbcr[:, 0:1] = fusion) - no clone, no cat
//   MODE 2  backward of 0: in = d/dYCrCb -> d/dRGB          (both maps are affine: the backward is the transposed matrix)
//   MODE 3  backward of 1: in = d/dRGB -> d/d[Y, Cr, Cb]; nout = 1 writes d/dY only (the gradient of `fusion`)
template <int MODE>
__global__ __launch_bounds__(256) void color3_kernel(const float* __restrict__ in, const float* __restrict__ ysrc,
                                                     float* __restrict__ out, long long HW, long long total, int nout) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long b = idx / HW, p = idx - b * HW;
  const float a = (MODE == 1 && ysrc) ? ysrc[idx] : in[(b * 3 + 0) * HW + p];
  const float c1 = in[(b * 3 + 1) * HW + p], c2 = in[(b * 3 + 2) * HW + p];
  float o0, o1, o2;
  if (MODE == 0) {
    o0 = 0.299f * a + 0.587f * c1 + 0.114f * c2;
    o1 = (a - o0) * 0.713f + 0.5f;
    o2 = (c2 - o0) * 0.564f + 0.5f;
  } else if (MODE == 1) {
    const float t0 = a + 0.0f, t1 = c1 + -0.5f, t2 = c2 + -0.5f;
    o0 = t0 * 1.0f + t1 * 1.403f + t2 * 0.0f;
    o1 = t0 * 1.0f + t1 * -0.714f + t2 * -0.344f;
    o2 = t0 * 1.0f + t1 * 0.0f + t2 * 1.773f;
  } else if (MODE == 2) {  // (a, c1, c2) = (dY, dCr, dCb); dCr reaches R directly and everything through Y
    const float gy = a - 0.713f * c1 - 0.564f * c2;  // total gradient arriving at Y
    o0 = 0.299f * gy + 0.713f * c1;
    o1 = 0.587f * gy;
    o2 = 0.114f * gy + 0.564f * c2;
  } else {  // (a, c1, c2) = (dR, dG, dB): d t_i = sum_j d out_j M[i][j]
    o0 = a + c1 + c2;
    o1 = 1.403f * a - 0.714f * c1;
    o2 = -0.344f * c1 + 1.773f * c2;
  }
  if (nout == 1) {
    out[idx] = o0;
  } else {
    out[(b * 3 + 0) * HW + p] = o0;
    out[(b * 3 + 1) * HW + p] = o1;
    out[(b * 3 + 2) * HW + p] = o2;
  }
}

// ---------------------------------------------------------------------------------------------
// 11-tap separable Gaussian blur with zero padding over (planes, H, W) images: the window of
// pytorch_ssim (pytorch_ssim/__init__.py:8-17; an outer product, so the 2-D "same" convolution
// factorises exactly).  32x32 output tile, (32+10)^2 halo in LDS, row pass then column pass.
// The operator is symmetric, so the same kernel is its own adjoint (SSIM backward).
// ---------------------------------------------------------------------------------------------
struct GaussTaps {
  float g[11];
};

__global__ __launch_bounds__(256) void gauss_blur11_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                           int W, GaussTaps taps) {
  __shared__ float tile[42][43];
  __shared__ float rowp[42][33];
  const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
  const float* xp = x + (long long)blockIdx.z * H * W;
  float* yp = y + (long long)blockIdx.z * H * W;
  for (int i = threadIdx.x; i < 42 * 42; i += 256) {
    const int r = i / 42, c = i - r * 42;
    const int gy = ty0 - 5 + r, gx = tx0 - 5 + c;
    tile[r][c] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? xp[(long long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 42 * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) s = fmaf(taps.g[t], tile[r][c + t], s);
    rowp[r][c] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    const int gy = ty0 + r, gx = tx0 + c;
    if (gy < H && gx < W) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 11; ++t) s = fmaf(taps.g[t], rowp[r + t][c], s);
      yp[(long long)gy * W + gx] = s;
    }
  }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, int32_t* __restrict__ labels,
                                                     long long rows, int C, int ldx) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* p = x + r * ldx;
  float best = p[0];
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    const float v = p[c];
    if (v > best) {
      best = v;
      bi = c;
    }
  }
  labels[r] = bi;
}

// (r6) logits -> bilinear resize -> argmax in one pass (test_segmentation.py:169-174: F.interpolate(seg, size=labels.shape[1:],
// mode='bilinear') then .argmax(1)): the (B, OH, OW, C) resized logits - 708 MB per 64 images at 480 x 640, written by
// bilinear_kernel<1> at 1.4 TB/s (C = 9: scalar accesses) and read back by argmax_kernel - never exist.  One thread per output
// pixel; the interpolation is bilinear_kernel's expression, the comparison argmax_kernel's (ties -> lowest index), so the labels
// are those of the two-pass form.
__global__ __launch_bounds__(256) void bilinear_argmax_kernel(const float* __restrict__ x, int32_t* __restrict__ labels, int IH, int IW,
                                                              int OH, int OW, int C, int ldx, float sy, float sx) {
  const int ox = blockIdx.x * 256 + threadIdx.x;
  if (ox >= OW) return;
  const int oy = blockIdx.y;
  const long long b = blockIdx.z;
  const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, IH - 1), x1 = min(x0 + 1, IW - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* base = x + b * IH * IW * ldx;
  const float* p00 = base + ((long long)y0 * IW + x0) * ldx;
  const float* p01 = base + ((long long)y0 * IW + x1) * ldx;
  const float* p10 = base + ((long long)y1 * IW + x0) * ldx;
  const float* p11 = base + ((long long)y1 * IW + x1) * ldx;
  float best = hy * (hx * p00[0] + lx * p01[0]) + ly * (hx * p10[0] + lx * p11[0]);
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    const float v = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
    if (v > best) {
      best = v;
      bi = c;
    }
  }
  labels[(b * OH + oy) * OW + ox] = bi;
}

// out = act(base + bias + sum_s bilinear(x_s -> OH x OW)): the SegFormer head with linear_fuse applied per
// scale BEFORE the resize (segformer_head.py:67-77 commuted) needs the sum of three
// up-sampled maps, the full-resolution term, the folded BatchNorm shift and the ReLU in one pass.
struct UpSrc {
  const float* x;
  int ih, iw;
  float sy, sx;
};

__device__ __forceinline__ f32x4 bilinear_tap4(const UpSrc& s, long long b, int oy, int ox, int c, int C) {
  const float fy = fmaxf(s.sy * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(s.sx * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, s.ih - 1), x1 = min(x0 + 1, s.iw - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* base = s.x + b * s.ih * s.iw * C + c;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * s.iw + x0) * C);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * s.iw + x1) * C);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * s.iw + x0) * C);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * s.iw + x1) * C);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
  return o;
}

__global__ __launch_bounds__(256) void upsum_act_kernel(const float* __restrict__ basep, int ldb, UpSrc s0, UpSrc s1, UpSrc s2,
                                                        int nsrc, const float* __restrict__ bias, float* __restrict__ out,
                                                        int ldo, int OH, int OW, int C, int act, long long total) {
  // (r6) grid = (units of an output row, output row, image), like bilinear_kernel: the row terms are wave-uniform and the only
  // division left is a 32-bit one (this kernel spent three 64-bit divisions per 16-byte store: 0.93 ms for 2.5 GB = 2.7 TB/s)
  const int c4n = C / 4;
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (unsigned)(OW * c4n)) return;
  const int ox = (int)(t / (unsigned)c4n);
  const int c = (int)(t - (unsigned)ox * c4n) * 4;
  const int oy = blockIdx.y;
  const long long b = blockIdx.z;
  const long long row = (b * OH + oy) * OW + ox;
  f32x4 y = {0.f, 0.f, 0.f, 0.f};
  if (basep) y = *reinterpret_cast<const f32x4*>(basep + row * ldb + c);
  if (bias) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] += bv[e];
  }
  if (nsrc > 0) {
    const f32x4 t = bilinear_tap4(s0, b, oy, ox, c, C);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] += t[e];
  }
  if (nsrc > 1) {
    const f32x4 t = bilinear_tap4(s1, b, oy, ox, c, C);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] += t[e];
  }
  if (nsrc > 2) {
    const f32x4 t = bilinear_tap4(s2, b, oy, ox, c, C);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] += t[e];
  }
  if (act == SEGMIF_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
  }
  *reinterpret_cast<f32x4*>(out + row * ldo + c) = y;
}

}  // namespace

extern "C" int segmif_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                                    int C, int ldx, int ldy, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0 || (C & 3) || C > 1024 || (ldx & 3) || (ldy & 3))
    return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return SEGMIF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nvec = C >> 2;
  if (nvec <= 8) return launch_ln<8, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s);
  if (nvec <= 16) return launch_ln<16, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s);
  if (nvec <= 32) return launch_ln<32, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s);
  if (nvec <= 64) return launch_ln<64, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s);
  if (nvec <= 128) return launch_ln<64, 2>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s);
  return launch_ln<64, 4>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s);
}

extern "C" int segmif_layernorm_pairs_f32(const float* x, const float* gamma, const float* beta, void* y, int64_t rows, int C, int ldx,
                                          int ldy, float eps, uint32_t* amax, int amax_images, int amax_sub, int amax_pitch, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0 || (C & 15) || C > 1024 || (ldx & 3) || (ldy & 3) || ldy < C)
    return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return SEGMIF_EINVAL;
  if (amax && (amax_images < 1 || rows % amax_images || amax_sub < 1 || (amax_sub & (amax_sub - 1)) || amax_pitch < amax_images))
    return SEGMIF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long long ar = rows / (amax ? amax_images : 1);
  const int ns = amax ? amax_sub : 1;
  // (r6) the pitch of the slot rows is the GUARD's image count, given explicitly: a launch whose batch is not the
  // guard's reports to column 0 (amax_images = 1) of rows that still lie guard.images words apart
  const long long st = amax ? amax_pitch : 1;
  const int nvec = C >> 2;
  if (nvec <= 8) return launch_ln_pairs<8, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, amax, ar, ns, st, s);
  if (nvec <= 16) return launch_ln_pairs<16, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, amax, ar, ns, st, s);
  if (nvec <= 32) return launch_ln_pairs<32, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, amax, ar, ns, st, s);
  if (nvec <= 64) return launch_ln_pairs<64, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, amax, ar, ns, st, s);
  if (nvec <= 128) return launch_ln_pairs<64, 2>(x, gamma, beta, y, rows, C, ldx, ldy, eps, amax, ar, ns, st, s);
  return launch_ln_pairs<64, 4>(x, gamma, beta, y, rows, C, ldx, ldy, eps, amax, ar, ns, st, s);
}

extern "C" int segmif_add_layernorm_f32(const float* x, const float* branch, const float* scale, int64_t rows_per_image,
                                        const float* gamma, const float* beta, float* sum, float* y, int64_t rows, int C, int ldx,
                                        int ldb, int lds, int ldy, float eps, void* stream) {
  if (!x || !branch || !sum || !gamma || !beta || !y || rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ((ldx | ldb | lds | ldy) & 3))
    return SEGMIF_EINVAL;
  if (scale && (rows_per_image <= 0 || rows % rows_per_image)) return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)branch | (uintptr_t)sum | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return SEGMIF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  LnAdd a;
  a.br = branch; a.ldb = ldb; a.scale = scale; a.rpi = scale ? rows_per_image : 1; a.sum_out = sum; a.lds = lds;
  const int nvec = C >> 2;
  if (nvec <= 8) return launch_ln<8, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s, a);
  if (nvec <= 16) return launch_ln<16, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s, a);
  if (nvec <= 32) return launch_ln<32, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s, a);
  if (nvec <= 64) return launch_ln<64, 1>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s, a);
  if (nvec <= 128) return launch_ln<64, 2>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s, a);
  return launch_ln<64, 4>(x, gamma, beta, y, rows, C, ldx, ldy, eps, s, a);
}

// XT output columns per thread (x0 = XT xp ..) and a 3 x (XT + 2) register window: XT + 2 16-byte loads per row serve XT
// outputs (the one-column kernel above spends three per output: every element came out of L2 3.75 times), strips of 16
// rows (halo rows 1.125x instead of 1.25x).  Loads are unconditional from clamped addresses; out-of-image taps are zeroed
// by select.  Round 3: 3.67 -> 4.4 TB/s of algorithmic traffic at XT = 2 over the encoder's shapes.
constexpr int DW2_TY = 16;
template <bool GELU, int XT, bool PAIRS = false>
__global__ __launch_bounds__(256) void dwconv3x3_xt_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int H, int W, int C, uint32_t* __restrict__ amax, int amax_images) {
  // PAIRS (r5): y receives the tokens in PAIRS format (gemm_pairs.hip; same byte count), max |y| goes to the image's range slot
  const int c4n = C >> 2;
  const int wp = (W + XT - 1) / XT;
  unsigned idx = blockIdx.x * 256 + threadIdx.x;
  uint32_t amx = 0u;
  bool live = true;
  if (idx >= (unsigned)(wp * c4n)) {
    if constexpr (!PAIRS) return;
    // PAIRS: the workgroup's range report ends in barriers, which every wave must reach the same number of times: a thread
    // past the end recomputes the last valid column and stores nothing
    live = false;
    idx = (unsigned)(wp * c4n) - 1;
  }
  const int xp = (int)(idx / (unsigned)c4n), c = (int)(idx - (unsigned)xp * c4n) * 4;
  const int x0 = XT * xp;
  const int y0 = blockIdx.y * DW2_TY;
  const long long img = (long long)blockIdx.z * H * W;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  auto load_row = [&](int yy, f32x4* v) {  // columns x0 - 1 .. x0 + XT
    const bool in = (unsigned)yy < (unsigned)H;
    const float* row = x + (img + (long long)(in ? yy : 0) * W) * C + c;
#pragma unroll
    for (int k = 0; k < XT + 2; ++k) {
      const int xx = x0 - 1 + k;
      const bool ok = in && (unsigned)xx < (unsigned)W;
      const f32x4 t = *reinterpret_cast<const f32x4*>(row + (long long)min(max(xx, 0), W - 1) * C);
      v[k] = ok ? t : zero;
    }
  };
  f32x4 win[3][XT + 2];
  load_row(y0 - 1, win[0]);
  load_row(y0, win[1]);
#pragma unroll 2
  for (int dy = 0; dy < DW2_TY; ++dy) {
    const int yo = y0 + dy;
    if (yo >= H) break;
    load_row(yo + 1, win[2]);
    float* dst = y + (img + (long long)yo * W + x0) * C + c;
#pragma unroll
    for (int j = 0; j < XT; ++j) {
      f32x4 a = bv;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) a += win[ky][kx + j] * wv[ky * 3 + kx];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = GELU ? (PAIRS ? gelu_as(a[e]) : gelu_exact(a[e])) : a[e];
      if constexpr (PAIRS) {
        if (live && x0 + j < W) {
          uint32_t ha, la, hb, lb;
          segmif::p16::split2(o[0], o[1], ha, la);
          segmif::p16::split2(o[2], o[3], hb, lb);
          // (threads c, c + 4, c + 8, c + 12 of one 16-channel group are four consecutive lanes with the same x0, j, live)
          unsigned char* d16 = reinterpret_cast<unsigned char*>(dst + (long long)j * C - c) + (c >> 4) * 64 + ((c >> 2) & 3) * 16;
          *reinterpret_cast<segmif::p16::u4*>(d16) = segmif::p16::quad_piece(ha, hb, la, lb);
          amx = segmif::p16::absmax_pk(segmif::p16::absmax_pk(amx, ha, la), hb, lb);
        }
      } else {
        if (x0 + j < W) *reinterpret_cast<f32x4*>(dst + (long long)j * C) = o;
      }
    }
#pragma unroll
    for (int k = 0; k < XT + 2; ++k) {
      win[0][k] = win[1][k];
      win[1][k] = win[2][k];
    }
  }
  if constexpr (PAIRS) {
    if (amax) segmif::p16::fold_pat_block(amax, amax_images > 1 ? blockIdx.z : 0, amx);  // one slot access per workgroup
  }
}

// dwconv + bias + GELU with the result in PAIRS format (the A operand of fc2 on gemm_pairs): the two-column kernel only
static int launch_dwconv_pairs(const float* x, const float* w9, const float* bias, void* y, int B, int H, int W, int C, uint32_t* amax,
                               int amax_images, hipStream_t s) {
  const long long per_row = (long long)((W + 1) / 2) * (C >> 2);
  dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)((H + DW2_TY - 1) / DW2_TY), (unsigned)B);
  hipLaunchKernelGGL((dwconv3x3_xt_kernel<true, 2, true>), grid, dim3(256), 0, s, x, w9, bias, reinterpret_cast<float*>(y), H, W, C, amax,
                     amax_images);
  return (int)hipGetLastError();
}

template <bool GELU>
static int launch_dwconv(const float* x, const float* w9, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s) {
  static const int xt_env = [] { const char* e = getenv("SEGMIF_DWCONV_XT"); return e ? atoi(e) : 0; }();  // diagnosis (read once): 1 = the one-column kernel, 2 / 4 = columns per thread
  const int xt = xt_env ? xt_env : (W >= 16 ? 2 : 1);
  if (xt >= 2) {
    const int XT = xt >= 4 ? 4 : 2;
    const long long per_row = (long long)((W + XT - 1) / XT) * (C >> 2);
    dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)((H + DW2_TY - 1) / DW2_TY), (unsigned)B);
    if (XT == 4) hipLaunchKernelGGL((dwconv3x3_xt_kernel<GELU, 4>), grid, dim3(256), 0, s, x, w9, bias, y, H, W, C, (uint32_t*)nullptr, 1);
    else hipLaunchKernelGGL((dwconv3x3_xt_kernel<GELU, 2>), grid, dim3(256), 0, s, x, w9, bias, y, H, W, C, (uint32_t*)nullptr, 1);
  } else {
    const long long per_row = (long long)W * (C >> 2);
    dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)((H + DW_TY - 1) / DW_TY), (unsigned)B);
    hipLaunchKernelGGL(dwconv3x3_gelu_kernel<GELU>, grid, dim3(256), 0, s, x, w9, bias, y, H, W, C);
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 3x3 "same" convolution from 32 channels to ONE (conv22 of Fusion_Network3_ac, core/model_fusion.py:1065) + bias + act.
// On the matrix pipe this layer pads its single output channel to a 32-wide tile (3.5 ms per 64-image step for 11 GFLOP);
// here it is what it is: a bandwidth-bound stencil.  Eight lanes share a pixel (one channel quad each), a thread slides
// down TY rows with a 3x3 register window of float4s (the dwconv scheme), and the eight partial dot products are combined
// with three xor-shuffles.  wt: [9][32] tap-major (the segmif_pack_conv_weight image of the (1, 32, 3, 3) weight).
// ---------------------------------------------------------------------------------------------
constexpr int C1_TY = 8;
__global__ __launch_bounds__(256) void conv3x3_c32to1_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ wt,
                                                             const float* __restrict__ bias, const float* __restrict__ prelu,
                                                             int act, float* __restrict__ y, int H, int W) {
  const int q = threadIdx.x & 7;
  const int xo = blockIdx.x * 32 + (threadIdx.x >> 3);
  const int y0 = blockIdx.y * C1_TY;
  const long long img = (long long)blockIdx.z * H * W;
  const bool live = xo < W;
  const int xc = live ? xo : W - 1;  // clamped column: loads stay unconditional, dead lanes just do not store
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(wt + t * 32 + 4 * q);
  const float b0 = bias ? bias[0] : 0.f;
  const float slope = act == SEGMIF_ACT_PRELU ? prelu[0] : 0.f;
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  auto load_row = [&](int yy, f32x4& l, f32x4& m, f32x4& r) {
    const bool in = (unsigned)yy < (unsigned)H;
    const float* p = x + (img + (long long)(in ? yy : 0) * W + xc) * ldx + 4 * q;
    const f32x4 lm = *reinterpret_cast<const f32x4*>(xc > 0 ? p - ldx : p);
    const f32x4 mm = *reinterpret_cast<const f32x4*>(p);
    const f32x4 rm = *reinterpret_cast<const f32x4*>(xc + 1 < W ? p + ldx : p);
    l = (in && xc > 0) ? lm : zero;
    m = in ? mm : zero;
    r = (in && xc + 1 < W) ? rm : zero;
  };
  f32x4 win[3][3];
  load_row(y0 - 1, win[0][0], win[0][1], win[0][2]);
  load_row(y0, win[1][0], win[1][1], win[1][2]);
#pragma unroll
  for (int dy = 0; dy < C1_TY; ++dy) {
    const int yo = y0 + dy;
    if (yo >= H) break;
    load_row(yo + 1, win[2][0], win[2][1], win[2][2]);
    f32x4 acc = zero;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc += win[ky][kx] * wv[ky * 3 + kx];
    float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (q == 0 && live) {
      float v = s + b0;
      if (act == SEGMIF_ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == SEGMIF_ACT_PRELU) v = v >= 0.f ? v : slope * v;
      y[img + (long long)yo * W + xo] = v;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      win[0][kx] = win[1][kx];
      win[1][kx] = win[2][kx];
    }
  }
}

extern "C" int segmif_conv3x3_c32to1_f32(const float* x, int ldx, const float* wt, const float* bias, const float* prelu, int act,
                                         float* y, int B, int H, int W, void* stream) {
  if (!x || !wt || !y || B <= 0 || H <= 0 || W <= 0 || ldx < 32 || (ldx & 3)) return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wt) & 15) return SEGMIF_EINVAL;
  if (act != SEGMIF_ACT_NONE && act != SEGMIF_ACT_RELU && act != SEGMIF_ACT_PRELU) return SEGMIF_EINVAL;
  if (act == SEGMIF_ACT_PRELU && !prelu) return SEGMIF_EINVAL;
  dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + C1_TY - 1) / C1_TY), (unsigned)B);
  hipLaunchKernelGGL(conv3x3_c32to1_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, wt, bias, prelu, act, y, H, W);
  return (int)hipGetLastError();
}

extern "C" int segmif_dwconv3x3_gelu_f32(const float* x, const float* w9, const float* bias, float* y, int B, int H,
                                         int W, int C, void* stream) {
  if (!x || !w9 || !bias || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w9 | (uintptr_t)bias) & 15) return SEGMIF_EINVAL;
  return launch_dwconv<true>(x, w9, bias, y, B, H, W, C, (hipStream_t)stream);
}

// (r5) the same with the result in PAIRS format (gemm_pairs.hip) and max |y| reported per image: C % 16 == 0
extern "C" int segmif_dwconv3x3_gelu_pairs_f32(const float* x, const float* w9, const float* bias, void* y, int B, int H, int W, int C,
                                               uint32_t* amax, int amax_images, void* stream) {
  if (!x || !w9 || !bias || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 15)) return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w9 | (uintptr_t)bias) & 15) return SEGMIF_EINVAL;
  if (amax && amax_images != 1 && amax_images != B) return SEGMIF_EINVAL;
  return launch_dwconv_pairs(x, w9, bias, y, B, H, W, C, amax, amax_images, (hipStream_t)stream);
}

// DWConv.forward on its own (core/mix_transformer.py:381-387): depthwise 3x3 + bias, no activation
extern "C" int segmif_dwconv3x3_bias_f32(const float* x, const float* w9, const float* bias, float* y, int B, int H,
                                         int W, int C, void* stream) {
  if (!x || !w9 || !bias || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return SEGMIF_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w9 | (uintptr_t)bias) & 15) return SEGMIF_EINVAL;
  return launch_dwconv<false>(x, w9, bias, y, B, H, W, C, (hipStream_t)stream);
}

extern "C" int segmif_bilinear_nhwc_f32(const float* x, float* y, int B, int IH, int IW, int OH, int OW, int C,
                                        int ldx, int ldo, void* stream) {
  if (!x || !y || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || ldx < C || ldo < C)
    return SEGMIF_EINVAL;
  const bool vec = !((C | ldx | ldo) & 3) && !(((uintptr_t)x | (uintptr_t)y) & 15);
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  if (OH > 65535 || B > 65535 || (long long)OW * C >= (1ll << 31)) return SEGMIF_EINVAL;
  const dim3 grid((unsigned)(((long long)OW * (vec ? C / 4 : C) + 255) / 256), (unsigned)OH, (unsigned)B);
  if (vec)
    hipLaunchKernelGGL(bilinear_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, y, IH, IW, OH, OW, C, ldx, ldo, sy, sx);
  else
    hipLaunchKernelGGL(bilinear_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, y, IH, IW, OH, OW, C, ldx, ldo, sy, sx);
  return (int)hipGetLastError();
}

static int launch_transpose(const float* x, float* y, int B, long long R, long long Cc, long long ldi, long long ldo,
                            long long in_bs, long long out_bs, hipStream_t s) {
  dim3 grid((unsigned)((Cc + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, x, y, R, Cc, ldi, ldo, in_bs, out_bs);
  return (int)hipGetLastError();
}

extern "C" int segmif_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int64_t HW, int ldo, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || ldo < C) return SEGMIF_EINVAL;
  return launch_transpose(x, y, B, C, HW, HW, ldo, (long long)C * HW, (long long)HW * ldo, (hipStream_t)stream);
}

extern "C" int segmif_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int64_t HW, int ldx, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || ldx < C) return SEGMIF_EINVAL;
  return launch_transpose(x, y, B, HW, C, ldx, HW, (long long)HW * ldx, (long long)C * HW, (hipStream_t)stream);
}

extern "C" int segmif_seg_normalize_f32(const float* x, float* y, int B, int H, int W, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0) return SEGMIF_EINVAL;
  const long long HW = (long long)H * W, total = HW * B;
  hipLaunchKernelGGL(seg_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     y, HW, total);
  return (int)hipGetLastError();
}

extern "C" int segmif_pointwise2_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int64_t rows, int C, int mode,
                                     void* stream) {
  if (!a || !y || rows <= 0 || C <= 0 || (C & 3) || mode < 0 || mode > 2 || (mode != 2 && !b)) return SEGMIF_EINVAL;
  if ((lda & 3) || (ldy & 3) || (b && (ldb & 3)) || (((uintptr_t)a | (uintptr_t)y | (uintptr_t)b) & 15)) return SEGMIF_EINVAL;
  const long long total = (long long)rows * (C / 4);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(pointwise2_kernel<0>, grid, block, 0, s, a, lda, b, ldb, y, ldy, (long long)rows, C / 4);
  else if (mode == 1) hipLaunchKernelGGL(pointwise2_kernel<1>, grid, block, 0, s, a, lda, b, ldb, y, ldy, (long long)rows, C / 4);
  else hipLaunchKernelGGL(pointwise2_kernel<2>, grid, block, 0, s, a, lda, b, ldb, y, ldy, (long long)rows, C / 4);
  return (int)hipGetLastError();
}

extern "C" int segmif_fuse_ycrcb_f32(const float* vis, const float* yf, float* out, int B, int64_t HW, void* stream) {
  if (!vis || !yf || !out || B <= 0 || HW <= 0) return SEGMIF_EINVAL;
  const long long total = (long long)HW * B;
  hipLaunchKernelGGL(fuse_ycrcb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vis,
                     yf, out, (long long)HW, total);
  return (int)hipGetLastError();
}

extern "C" int segmif_color3_f32(const float* in, const float* ysrc, float* out, int B, int64_t HW, int mode, int nout, void* stream) {
  if (!in || !out || B <= 0 || HW <= 0 || mode < 0 || mode > 3 || (nout != 3 && !(nout == 1 && mode == 3)) || (ysrc && mode != 1))
    return SEGMIF_EINVAL;
  const long long total = (long long)HW * B;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(color3_kernel<0>, grid, block, 0, s, in, ysrc, out, (long long)HW, total, nout);
  else if (mode == 1) hipLaunchKernelGGL(color3_kernel<1>, grid, block, 0, s, in, ysrc, out, (long long)HW, total, nout);
  else if (mode == 2) hipLaunchKernelGGL(color3_kernel<2>, grid, block, 0, s, in, ysrc, out, (long long)HW, total, nout);
  else hipLaunchKernelGGL(color3_kernel<3>, grid, block, 0, s, in, ysrc, out, (long long)HW, total, nout);
  return (int)hipGetLastError();
}

extern "C" int segmif_argmax_nhwc_i32(const float* x, int32_t* labels, int64_t rows, int C, int ldx, void* stream) {
  if (!x || !labels || rows <= 0 || C <= 0 || ldx < C) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, labels,
                     (long long)rows, C, ldx);
  return (int)hipGetLastError();
}

extern "C" int segmif_bilinear_argmax_i32(const float* x, int32_t* labels, int B, int IH, int IW, int OH, int OW, int C, int ldx,
                                          void* stream) {
  if (!x || !labels || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || ldx < C) return SEGMIF_EINVAL;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;  // (as segmif_bilinear_nhwc_f32 forms them)
  dim3 grid((unsigned)((OW + 255) / 256), (unsigned)OH, (unsigned)B);
  hipLaunchKernelGGL(bilinear_argmax_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, labels, IH, IW, OH, OW, C, ldx, sy, sx);
  return (int)hipGetLastError();
}

extern "C" int segmif_gauss_blur11_f32(const float* x, float* y, int planes, int H, int W, const float* taps11,
                                       void* stream) {
  if (!x || !y || !taps11 || planes <= 0 || H <= 0 || W <= 0) return SEGMIF_EINVAL;
  GaussTaps t;
  for (int i = 0; i < 11; ++i) t.g[i] = taps11[i];  // host pointer: 11 window weights
  dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 31) / 32), (unsigned)planes);
  hipLaunchKernelGGL(gauss_blur11_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, H, W, t);
  return (int)hipGetLastError();
}

extern "C" int segmif_upsum_act_nhwc_f32(const float* base, int ldb, const float* x0, int ih0, int iw0, const float* x1,
                                         int ih1, int iw1, const float* x2, int ih2, int iw2, const float* bias, float* out,
                                         int ldo, int B, int OH, int OW, int C, int act, void* stream) {
  if (!out || B <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 3) || ldo < C || (ldo & 3) || (base && (ldb < C || (ldb & 3))))
    return SEGMIF_EINVAL;
  if (act != SEGMIF_ACT_NONE && act != SEGMIF_ACT_RELU) return SEGMIF_EINVAL;
  const float* xs[3] = {x0, x1, x2};
  const int ihs[3] = {ih0, ih1, ih2}, iws[3] = {iw0, iw1, iw2};
  UpSrc src[3] = {};
  int n = 0;
  for (int i = 0; i < 3; ++i) {
    if (!xs[i]) continue;
    if (ihs[i] <= 0 || iws[i] <= 0 || ((uintptr_t)xs[i] & 15)) return SEGMIF_EINVAL;
    src[n++] = UpSrc{xs[i], ihs[i], iws[i], (float)ihs[i] / (float)OH, (float)iws[i] / (float)OW};
  }
  if ((((uintptr_t)out | (uintptr_t)(base ? base : out) | (uintptr_t)(bias ? bias : out)) & 15)) return SEGMIF_EINVAL;
  const long long total = (long long)B * OH * OW * (C / 4);
  if ((long long)OW * (C / 4) >= (1ll << 31) || OH > 65535 || B > 65535) return SEGMIF_EINVAL;
  const dim3 grid((unsigned)(((long long)OW * (C / 4) + 255) / 256), (unsigned)OH, (unsigned)B);
  hipLaunchKernelGGL(upsum_act_kernel, grid, dim3(256), 0, (hipStream_t)stream, base, ldb,
                     src[0], src[1], src[2], n, bias, out, ldo, OH, OW, C, act, total);
  return (int)hipGetLastError();
}
