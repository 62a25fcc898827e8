// CrossPath (core/model_fusion.py:329-361) in inference, restructured around what its linear attention needs.
//
// Reference data flow per modality i (tokens x_i, segmentation tokens x_3, all (B, N, 64), N = H*W):
//     [y_i | u_i] = ReLU(channel_proj_i(x_i))                      three 64 -> 128 Linears          (:351-353)
//     ctx_3 = softmax((K^T V) d^-1/2),  [K | V] = kv3(u_3)        cross_attn,  8 heads of 8        (:281-286)
//     ctx_i = softmax((K^T V) d^-1/2),  [K | V] = kv_i(y_i)       cross_attn2                      (:316-326)
//     out_i = LayerNorm(x_i + end_proj_i([y_3 @ ctx_i | u_i @ ctx_3]))                               (:357-360)
// Round 1 ran this as 3 GEMMs (channel_proj, 128-wide outputs written to HBM), 3 fused kv-projection + K^T V
// reductions and 2 two-source GEMMs with the contexts folded into a per-image end_proj weight: 5.1 KB of HBM
// traffic per pixel per call, HBM-bound at 3-4 TB/s, 19.7 ms per call at 32 x 480 x 640.  Two observations remove
// most of it:
//   * K^T V = Wk (Y^T Y) Wv^T: the reduction over N only needs the 64 x 64 Gram matrix of the projected tokens
//     (Y = y_i or u_3); the kv projection of 9.8 M tokens disappears into two 64 x 64 matrix products per image.
//   * each consumer needs only ONE 64-wide half of a channel_proj output, so recomputing that half where it is
//     consumed costs the same FLOPs as producing both halves once - and the 128-wide tensors never exist.
// crosspath_gram_kernel:  G_b = sum_n relu(W x_n + c) relu(W x_n + c)^T     (reads x: 256 B per pixel)
// crosspath_tail_kernel:  out = LN(x_i + Weff_b [relu(W3 x_3 + c3) | relu(Wi x_i + ci)] + e)
//                                                                            (reads x_3, x_i, writes out: 768 B)
// Both keep a wave's 32 pixels in registers from load to store.  Arithmetic: every fp32 operand is split in registers
// into three bf16 planes (x = x0 + x1 + x2, round-to-nearest at each step) and every product runs as six
// v_mfma_f32_32x32x16_bf16 (the same fp32-class scheme as csrc/conv3x3_planes.hip: 2.65x the fp32 matrix pipe's rate;
// the first version of these kernels sat at 70 % of that pipe: profiles/r02_bench_kernel_stats_v3.txt).  The accumulator
// of one product is, register by register, the K operand of the next: a lane holds 16 rows of a 32 x 32 result, rows
// (v&3) + 8(v>>2) + 4h, and registers 8s .. 8s+7 are exactly the 8 K-slots a lane feeds to one MFMA - the K order inside
// a contraction is free as long as both operands agree, so the weights are staged in LDS in that order and neither the
// 64-channel intermediate nor its transpose ever goes through LDS.  There is no barrier in the pixel loop.
// Accuracy: the Gram sums run in fp32 over 32-pixel runs (inside the MFMA accumulator) and in fp64 across runs.
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>

#include "planes16.h"
#include "segmif_hip.h"

namespace p16 = segmif::p16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

// LDS image of a weight matrix with 64 (or 128) input columns: row n = [plane 0..2][position 0..63(127)] bf16 + 16 B.
// Position 16 s + 8 h + j holds column 16 s + 4 h + (j & 3) + 8 (j >> 2): the 8 positions (s, h, .) are what lane-half h
// feeds to the MFMA of K-step s, and the columns are the ones that lane-half holds - both for pixels loaded as
// x[px][8q + 4h .. +3] (q = 2s, 2s+1) and for accumulator registers 8s' .. 8s'+7 of a 32-row tile (s = 2 tile + s').
// Pitch = 16 B mod 128 B: ds_read_b128 down the rows is conflict free.
constexpr int WPB = 3 * 128 + 16;    // bytes per row, 64 columns
constexpr int WPB2 = 3 * 256 + 16;   // 128 columns
constexpr int CP_WAVES = 8;
constexpr int GRAM_WGS = 64;  // workgroups per image at least (x 8 waves x 32-pixel tiles): independent of the batch, so results do not depend on it; 8 left a single image on 8 CUs
constexpr int GRAM_RUN_TILES = 32;  // a wave accumulates at most this many tiles (1024 pixels) in fp32 before the fp64 combine
constexpr int PX6[6] = {2, 1, 0, 1, 0, 0};  // six products, least significant first: plane of the first operand ...
constexpr int PY6[6] = {0, 1, 2, 0, 1, 0};  // ... and of the second

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int v = 0; v < 16; ++v) z[v] = 0.f;
  return z;
}

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void split3(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
  p1 = pk_bf16(r0, r1);
  r0 -= __uint_as_float(p1 << 16);
  r1 -= __uint_as_float(p1 & 0xffff0000u);
  p2 = pk_bf16(r0, r1);
}

// 8 consecutive K-slots of a lane -> one MFMA operand per plane
struct Op3 {
  u32x4 p[3];
};
__device__ __forceinline__ Op3 split8(const f32x4 lo, const f32x4 hi) {
  Op3 o;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    uint32_t a, b, c;
    split3(lo[2 * e], lo[2 * e + 1], a, b, c);
    o.p[0][e] = a; o.p[1][e] = b; o.p[2][e] = c;
    split3(hi[2 * e], hi[2 * e + 1], a, b, c);
    o.p[0][2 + e] = a; o.p[1][2 + e] = b; o.p[2][2 + e] = c;
  }
  return o;
}
__device__ __forceinline__ Op3 split8(const f32x16 t, int s) {  // accumulator registers 8s .. 8s+7
  return split8(f32x4{t[8 * s], t[8 * s + 1], t[8 * s + 2], t[8 * s + 3]},
                f32x4{t[8 * s + 4], t[8 * s + 5], t[8 * s + 6], t[8 * s + 7]});
}
__device__ __forceinline__ bf16x8 op(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// six-product fp32-class accumulate: acc += A B with A's planes `a` (first MFMA operand: rows) and B's planes `b`
__device__ __forceinline__ f32x16 mma6(const u32x4* a, const u32x4* b, f32x16 acc) {
#pragma unroll
  for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(a[PX6[t]]), op(b[PY6[t]]), acc, 0, 0, 0);
  return acc;
}

// 64 rows x 64 columns [col0, col0 + 64) of a row-major fp32 matrix -> split LDS image at positions [pos0, pos0 + 64)
__device__ __forceinline__ void stage_split64(const float* __restrict__ w, int ldw, int col0, unsigned char* dst, int pitch,
                                              int pos0, int plane_bytes, int tid, int nthreads) {
  for (int u = tid; u < 64 * 32; u += nthreads) {
    const int row = u >> 5, pp = 2 * (u & 31);  // positions pp, pp + 1 = adjacent columns
    const int s = pp >> 4, hh = (pp >> 3) & 1, j = pp & 7;
    const int col = 16 * s + 4 * hh + (j & 3) + 8 * (j >> 2);
    const f32x2 v = *reinterpret_cast<const f32x2*>(w + (long long)row * ldw + col0 + col);
    uint32_t a, b, c;
    split3(v[0], v[1], a, b, c);
    unsigned char* d = dst + row * pitch + (pos0 + pp) * 2;
    *reinterpret_cast<uint32_t*>(d) = a;
    *reinterpret_cast<uint32_t*>(d + plane_bytes) = b;
    *reinterpret_cast<uint32_t*>(d + 2 * plane_bytes) = c;
  }
}

// the three planes of the 8 positions (ks, h, .) of weight row `row`
__device__ __forceinline__ void wfrag(const unsigned char* base, int row, int pitch, int plane_bytes, int ks, int h, u32x4* out) {
  const unsigned char* a = base + row * pitch + (ks * 16 + 8 * h) * 2;
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = *reinterpret_cast<const u32x4*>(a + k * plane_bytes);
}

// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void crosspath_gram_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                             const float* __restrict__ bias, double* __restrict__ partial,
                                                             long long N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* Ws = smem_raw;                                   // [64][WPB]
  double* Red = reinterpret_cast<double*>(smem_raw + 64 * WPB);   // [3][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const float* __restrict__ xb = x + (long long)b * N * ldx;
  stage_split64(w, 64, 0, Ws, WPB, 0, 128, tid, 512);
  __syncthreads();
  const float bias0 = bias ? bias[r] : 0.f, bias1 = bias ? bias[32 + r] : 0.f;

  // The wave's Gram sums stay in the MFMA accumulators (fp32) over its whole run of tiles - at most GRAM_RUN_TILES of them
  // (segmif_crosspath_gram_blocks sizes the grid for that): every term y_i y_j is non-negative (y = ReLU(.)), so a run's
  // sum has no cancellation and carries ~sqrt(pixels) x 2^-24 of relative error; the runs are then combined in fp64
  // (below, and across workgroups in the fold kernel).  Round 2 converted to fp64 after every 32-pixel tile: 96 of a
  // lane's 256 registers and ~700 vector-ALU cycles per tile, which left no room for a second tile of loads in flight.
  f32x16 g[3] = {zero16(), zero16(), zero16()};

  const long long ntiles = (N + 31) / 32;
  const long long stride = (long long)gridDim.x * CP_WAVES;
  auto load = [&](long long tt, f32x4* dst) {  // x[px][8q + 4h .. +3]; unconditional (clamped row): rows past N are masked below
    long long px = tt * 32 + r;
    px = px < N ? px : N - 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = *reinterpret_cast<const f32x4*>(xb + px * ldx + 8 * q + 4 * h);
  };
  // one 32-pixel tile: `cur` holds its rows; the tile TWO steps ahead is requested into `pre` before the arithmetic starts
  // (three register sets in rotation: two tiles of loads in flight per wave)
  auto tile = [&](long long t, const f32x4* cur, f32x4* pre, long long tpre) {
    int zo = 0;  // opaque zero in every LDS address below: the weight fragments are loop invariant, and hoisted out of
    asm volatile("" : "+v"(zo));  // the tile loop they would occupy 100+ registers for the whole kernel
    if (tpre < ntiles) load(tpre, pre);
    // stage 1: Y[px][n] = relu(sum_k x[px][k] W[n][k] + c[n]); lane = column n, register v = pixel (v&3)+8(v>>2)+4h
    f32x16 y[2] = {zero16(), zero16()};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const Op3 xs = split8(cur[2 * s], cur[2 * s + 1]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        u32x4 wf[3];
        wfrag(Ws + zo, nt * 32 + r, WPB, 128, s, h, wf);
        y[nt] = mma6(xs.p, wf, y[nt]);
      }
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      y[0][v] = fmaxf(y[0][v] + bias0, 0.f);
      y[1][v] = fmaxf(y[1][v] + bias1, 0.f);
    }
    if (t * 32 + 32 > N) {  // the image's last, partial tile: pixels past the end contribute nothing
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const bool pv = t * 32 + (v & 3) + 8 * (v >> 2) + 4 * h < N;
        y[0][v] = pv ? y[0][v] : 0.f;
        y[1][v] = pv ? y[1][v] : 0.f;
      }
    }
    // stage 2: G[i][j] += sum_px Y[px][i] Y[px][j]: registers 8s .. 8s+7 of stage 1 are the 8 K-slots (pixels) of step s
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const Op3 y0 = split8(y[0], s), y1 = split8(y[1], s);
      g[0] = mma6(y0.p, y0.p, g[0]);
      g[1] = mma6(y0.p, y1.p, g[1]);
      g[2] = mma6(y1.p, y1.p, g[2]);
    }
  };
  f32x4 xa[8], xb2[8], xc[8];
  long long t = (long long)blockIdx.x * CP_WAVES + wave;
  if (t < ntiles) load(t, xa);
  if (t + stride < ntiles) load(t + stride, xb2);
  for (; t < ntiles; t += 3 * stride) {
    tile(t, xa, xc, t + 2 * stride);
    if (t + stride < ntiles) tile(t + stride, xb2, xa, t + 3 * stride);
    if (t + 2 * stride < ntiles) tile(t + 2 * stride, xc, xb2, t + 4 * stride);
  }
  // deterministic reduction over the 8 waves, then one partial per workgroup
  for (int wv = 0; wv < CP_WAVES; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          double* p = Red + (a * 16 + v) * 64 + lane;
          *p = wv == 0 ? (double)g[a][v] : *p + (double)g[a][v];
        }
    }
    __syncthreads();
  }
  // canonical layout: tile a in {(0,0), (0,1), (1,1)}, element (i, j) at a*1024 + i*32 + j
  double* dst = partial + ((long long)b * gridDim.x + blockIdx.x) * 3072;
  for (int u = tid; u < 3072; u += 512) {
    const int a = u >> 10, i = (u >> 5) & 31, j = u & 31;
    const int hh = (i >> 2) & 1, v = (i & 3) + 4 * (i >> 3);
    dst[u] = Red[(a * 16 + v) * 64 + hh * 32 + j];
  }
}

// G = sum of the partial Gram matrices (fp64), ctx_h = softmax_{dim -2}((Wk_h G Wv_h^T) scale), folded into end_proj:
//   Weff[b][n][kofs + h*8 + i] = sum_j ctx[b][h][i][j] * Wend[n][wofs + h*8 + j]      (as segmif_linattn_fold_f32)
__global__ __launch_bounds__(1024) void crosspath_fold_kernel(const double* __restrict__ partial, int nblk,
                                                              const float* __restrict__ wkv, const float* __restrict__ wend,
                                                              float* __restrict__ weff, int Nout, int ldw, int wofs, int ldweff,
                                                              int kofs, float scale, uint32_t* __restrict__ cond) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* G = dsm;            // [64][64]
  double* T1 = dsm + 4096;    // [64][64]  Wk G
  double* ctx = T1 + 4096;    // [8][8][8]
  const int tid = threadIdx.x, b = blockIdx.x;
  const double* p = partial + (long long)b * nblk * 3072;
  for (int u = tid; u < 4096; u += 1024) {
    const int i = u >> 6, j = u & 63;
    const int ti = i >> 5, tj = j >> 5;
    const int a = ti == 0 ? tj : 2;                                            // tile (0,0), (0,1), (1,1); (1,0) mirrors (0,1)
    const int idx = (ti == 1 && tj == 0) ? 1024 + (j & 31) * 32 + (i & 31) : a * 1024 + (i & 31) * 32 + (j & 31);
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += p[(long long)k * 3072 + idx];          // fixed order: deterministic
    G[u] = s;
  }
  __syncthreads();
  double* D1 = ctx + 512;     // [64][64]  Wk (G o S): the Gram matrix under the probe pattern S (cond only)
  double* dctx = D1 + 4096;   // [8][8][8]  d logit / d eps under that pattern (cond only)
  for (int u = tid; u < 4096; u += 1024) {  // T1[c][bq] = sum_a Wk[c][a] G[a][bq]
    const int c = u >> 6, bq = u & 63;
    double s = 0.0, sd = 0.0;
    for (int a = 0; a < 64; ++a) {
      const double wg = (double)wkv[c * 64 + a] * G[a * 64 + bq];
      s += wg;
      // S[a][bq] = +-1, symmetric (the Gram matrix and the errors of its entries are), fixed: a multiplicative hash of (min, max)
      const unsigned lo = a < bq ? a : bq, hi = a < bq ? bq : a;
      sd += (((lo * 64u + hi + 1u) * 2654435761u) >> 16) & 1u ? wg : -wg;
    }
    T1[u] = s;
    if (cond) D1[u] = sd;
  }
  __syncthreads();
  if (tid < 512) {  // (K^T V)[h][i][j] = sum_b T1[h*8+i][b] Wv[h*8+j][b]
    const int hh = tid >> 6, i = (tid >> 3) & 7, j = tid & 7;
    double s = 0.0, sd = 0.0;
    for (int q = 0; q < 64; ++q) {
      const double w = (double)wkv[(64 + hh * 8 + j) * 64 + q];
      s += T1[(hh * 8 + i) * 64 + q] * w;
      if (cond) sd += D1[(hh * 8 + i) * 64 + q] * w;
    }
    ctx[tid] = s * (double)scale;
    if (cond) dctx[tid] = sd * (double)scale;
  }
  __syncthreads();
  if (tid < 64) {  // one (h, j) column per thread: softmax over i (dim = -2)
    const int hh = tid >> 3, j = tid & 7;
    double mx = -1e300;
    for (int i = 0; i < 8; ++i) mx = fmax(mx, ctx[hh * 64 + i * 8 + j]);
    double ev[8], sum = 0.0;
    for (int i = 0; i < 8; ++i) {
      ev[i] = exp(ctx[hh * 64 + i * 8 + j] - mx);
      sum += ev[i];
    }
    double mean_d = 0.0;
    for (int i = 0; i < 8; ++i) {
      const double pi = ev[i] / sum;
      ctx[hh * 64 + i * 8 + j] = pi;
      if (cond) mean_d += pi * dctx[hh * 64 + i * 8 + j];
    }
    if (cond) {
      // Conditioning probe of this softmax column.  What the producers' arithmetic leaves on the Gram entries is a RELATIVE
      // error; the probe asks how far the context moves per unit of it under ONE fixed sign pattern S: G -> G o (1 + eps S) moves
      // the logits by eps dL (dL formed above beside the logits themselves) and the softmax, to first order, by
      // d p_i = p_i (dL_i - sum_k p_k dL_k) eps.  kappa = max_i |p_i (dL_i - sum_k p_k dL_k)| over the launch's 64 columns:
      // ~1 for a well-conditioned context, hundreds where two large, cancelling logits compete (tools/cond_probe.py: 300 - 400
      // on the over-exposed pair whose f16x3 error was 6e-3, below 10 on its neighbours).  A worst-case bound (absolute values
      // instead of S) sits at 1e4 for nearly every input and separates nothing - profiles/r05_cond_calibration_abs.txt.
      // A decided (one-hot) column reports ~0 whatever its magnitude; NaN logits report NaN (top of the integer order).
      double kap = 0.0;
      for (int i = 0; i < 8; ++i) kap = fmax(kap, fabs(ctx[hh * 64 + i * 8 + j] * (dctx[hh * 64 + i * 8 + j] - mean_d)));
      const float kappa = (float)kap;
      uint32_t bits = __float_as_uint(kappa);
      if (kappa != kappa) bits = 0x7fc00000u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)bits, o);
        bits = other > bits ? other : bits;
      }
      if (tid == 0 && bits > cond[b]) atomicMax(cond + b, bits);
    }
  }
  __syncthreads();
  for (int o = tid; o < Nout * 64; o += 1024) {
    const int n = o >> 6, c = o & 63, hh = c >> 3, i = c & 7;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf((float)ctx[hh * 64 + i * 8 + j], wend[(long long)n * ldw + wofs + hh * 8 + j], acc);
    weff[((long long)b * Nout + n) * ldweff + kofs + c] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------------------
struct TailK {
  const float* x3; const float* xi;
  const float* w3; const float* b3;   // 64 x 64 rows of channel_proj3 (the y half), its bias
  const float* wi; const float* bi;   // 64 x 64 rows of channel_proj_i (the u half), its bias
  const float* weff;                  // [B][64][128] per-image end_proj with both contexts folded in
  const float* bend; const float* gamma; const float* beta;
  float* out;
  unsigned char* planes;              // optional split-bf16 copy of out (conv3x3_planes.hip format) or null
  int pl_f16;                         // the copy is an f16x3 one (half pairs, 64 bytes per pixel; planes16.h)
  uint32_t* pl_amax;                  // f16x3: range slot(s) for max |out| or null
  int pl_amax_images;                 // > 1: slot index = image (blockIdx.y)
  long long N;
  int ld3, ldi, ldo;
  int W, Hp, Wp, chunks;              // planes geometry (image width, padded dims, chunk images per batch element)
  float eps;
};

// F16: the planes copy is an f16x3 one (its own instantiation: the bf16 kernel sits at the register limit).
template <bool F16>
__global__ __launch_bounds__(512) void crosspath_tail_kernel(const TailK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* W3s = smem_raw;             // [64][WPB]
  unsigned char* Wis = W3s + 64 * WPB;       // [64][WPB]
  unsigned char* Wes = Wis + 64 * WPB;       // [64][WPB2]
  float* Cst = reinterpret_cast<float*>(Wes + 64 * WPB2);  // b3[64] bi[64] bend[64] gamma[64] beta[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  stage_split64(p.w3, 64, 0, W3s, WPB, 0, 128, tid, 512);
  stage_split64(p.wi, 64, 0, Wis, WPB, 0, 128, tid, 512);
  {
    const float* we = p.weff + (long long)b * 64 * 128;
    stage_split64(we, 128, 0, Wes, WPB2, 0, 256, tid, 512);
    stage_split64(we, 128, 64, Wes, WPB2, 64, 256, tid, 512);
    if (tid < 320) {
      const int a = tid >> 6, c = tid & 63;
      const float* src = a == 0 ? p.b3 : a == 1 ? p.bi : a == 2 ? p.bend : a == 3 ? p.gamma : p.beta;
      Cst[tid] = src ? src[c] : (a == 3 ? 1.f : 0.f);
    }
  }
  __syncthreads();
  const float* __restrict__ x3b = p.x3 + (long long)b * p.N * p.ld3;
  const float* __restrict__ xib = p.xi + (long long)b * p.N * p.ldi;
  float* __restrict__ outb = p.out + (long long)b * p.N * p.ldo;

  const long long ntiles = (p.N + 31) / 32;
  const long long stride = (long long)gridDim.x * CP_WAVES;
  uint32_t pl_amx = 0u;  // f16x3 planes copy: largest |out| this lane wrote (p16::absmax_pk patterns)
  auto load = [&](long long tt, const float* __restrict__ base, int ld, f32x4* dst) {  // a pixel's channels 8q + 4h .. +3
    const long long px = tt * 32 + r;
    const bool ok = tt < ntiles && px < p.N;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      dst[q] = ok ? *reinterpret_cast<const f32x4*>(base + px * ld + 8 * q + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // accumulator tile initialised with a per-row constant: register v = row (v&3) + 8(v>>2) + 4h of the 32-row tile
  auto rows16 = [&](const float* c) {
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 cb = *reinterpret_cast<const f32x4*>(c + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * g + e] = cb[e];
    }
    return a;
  };
  // one 32-pixel tile; c3 / ci hold its rows of x_3 / x_i.  The next tile's x_i rows are requested into ni while this one is
  // computed (two register sets used alternately: ci is live until the residual add of the epilogue); the next tile's x_3 rows go
  // back into c3 as soon as stage 1 has split them ((r4) three 32-register sets instead of four: the fourth made hipcc spill, and
  // every spill reload is followed by s_waitcnt vmcnt(0) - which drains the prefetch it sits beside)
  auto tile = [&](long long t, f32x4* c3, const f32x4* ci, f32x4* ni) {
    const long long px = t * 32 + r;
    const bool ok = px < p.N;
    int zo = 0;  // opaque zero in every LDS address below (see the Gram kernel): keeps ~200 registers of loop-invariant
    asm volatile("" : "+v"(zo));  // weight fragments from being hoisted out of the tile loop
    f32x16 z[2];  // end_proj accumulators, bias as the initial value
    z[0] = rows16(Cst + zo + 128);
    z[1] = rows16(Cst + zo + 160);
    // two passes: source 0 = x_3 -> y_3 half (W3, columns 0..63 of Weff), source 1 = x_i -> u_i half (Wi, columns 64..127)
#pragma unroll
    for (int src = 0; src < 2; ++src) {
      if (src == 1) load(t + stride, xib, p.ldi, ni);
      const f32x4* xc = src == 0 ? c3 : ci;
      // stage 1 (transposed): T[c][px] = relu(sum_k W[c][k] x[px][k] + bias[c]); lane = pixel, register v = channel
      // (v&3) + 8 (v>>2) + 4h of the 32-row tile
      const unsigned char* Wsrc = (src == 0 ? W3s : Wis) + zo;
      f32x16 tt[2];
      tt[0] = rows16(Cst + zo + src * 64);
      tt[1] = rows16(Cst + zo + src * 64 + 32);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const Op3 xs = split8(xc[2 * s], xc[2 * s + 1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 wf[3];
          wfrag(Wsrc, nt * 32 + r, WPB, 128, s, h, wf);
          tt[nt] = mma6(wf, xs.p, tt[nt]);
        }
      }
      if (src == 0) load(t + stride, x3b, p.ld3, c3);  // (its rows have been split: the registers take the next tile's)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int v = 0; v < 16; ++v) tt[nt][v] = fmaxf(tt[nt][v], 0.f);
      }
      // stage 2: Z[m][px] += sum_c Weff[m][c] T[c][px]: registers 8s' .. 8s'+7 of tile nt are the K-slots of step 2nt + s'
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const Op3 tk = split8(tt[nt], sp);
          const int ks = src * 4 + nt * 2 + sp;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            u32x4 wf[3];
            wfrag(Wes + zo, mt * 32 + r, WPB2, 256, ks, h, wf);
            z[mt] = mma6(wf, tk.p, z[mt]);
          }
        }
    }
    // epilogue: + residual x_i (same channel layout: channel 32 mt + 8 g + 4 h + e = ci[4 mt + g][e]), LayerNorm over the
    // pixel's 64 channels (32 in this lane, 32 in lane ^ 32)
    float o[32];
    float s1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = z[mt][4 * g + e] + ci[4 * mt + g][e];
          o[mt * 16 + 4 * g + e] = v;
          s1 += v;
        }
    s1 += __shfl_xor(s1, 32);
    const float mean = s1 * (1.0f / 64.0f);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      o[k] -= mean;
      s2 = fmaf(o[k], o[k], s2);
    }
    s2 += __shfl_xor(s2, 32);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 64.0f) + p.eps);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(Cst + zo + 192 + mt * 32 + 8 * g + 4 * h);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(Cst + zo + 256 + mt * 32 + 8 * g + 4 * h);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = o[mt * 16 + 4 * g + e] * rstd * ga[e] + bt[e];
          o[mt * 16 + 4 * g + e] = v[e];
        }
        if (ok && p.out) *reinterpret_cast<f32x4*>(outb + px * p.ldo + mt * 32 + 8 * g + 4 * h) = v;
      }
    if constexpr (F16) {
      if (ok) {
        const int yy = (int)(px / p.W), xx = (int)(px - (long long)yy * p.W);
        unsigned char* dst = p.planes + ((((long long)b * p.chunks) * p.Hp + yy + 2) * p.Wp + xx + 2) * p16::PIXEL_BYTES + h * 16;
        const long long cstride = (long long)p.Hp * p.Wp * p16::PIXEL_BYTES;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          u32x4 hi, lo;
          p16::split8(o + 8 * c, hi, lo);
          *reinterpret_cast<u32x4*>(dst + c * cstride) = hi;
          *reinterpret_cast<u32x4*>(dst + c * cstride + 32) = lo;
          pl_amx = p16::absmax_pk4(pl_amx, hi);
        }
      }
    } else if (p.planes && ok) {  // (F16 implies a planes buffer) positions 8h .. 8h+7 of chunk c = channels 16c + {4h..4h+3, 8+4h..8+4h+3}: this lane's o[8c .. 8c+7]
      const int yy = (int)(px / p.W), xx = (int)(px - (long long)yy * p.W);
      unsigned char* dst = p.planes + ((((long long)b * p.chunks) * p.Hp + yy + 2) * p.Wp + xx + 2) * 96 + h * 16;
      const long long cstride = (long long)p.Hp * p.Wp * 96;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const Op3 pl = split8(f32x4{o[8 * c], o[8 * c + 1], o[8 * c + 2], o[8 * c + 3]},
                              f32x4{o[8 * c + 4], o[8 * c + 5], o[8 * c + 6], o[8 * c + 7]});
        *reinterpret_cast<u32x4*>(dst + c * cstride) = pl.p[0];
        *reinterpret_cast<u32x4*>(dst + c * cstride + 32) = pl.p[1];
        *reinterpret_cast<u32x4*>(dst + c * cstride + 64) = pl.p[2];
      }
    }
  };
  f32x4 a3[8], ai[8], bi[8];
  long long t = (long long)blockIdx.x * CP_WAVES + wave;
  load(t, x3b, p.ld3, a3);
  load(t, xib, p.ldi, ai);
  for (; t < ntiles; t += 2 * stride) {
    tile(t, a3, ai, bi);
    if (t + stride < ntiles) tile(t + stride, a3, bi, ai);
  }
  if constexpr (F16) {
    if (p.pl_amax) p16::fold_pat(p.pl_amax, p.pl_amax_images > 1 ? b : 0, p.pl_amax_images > 1 ? b : 0, pl_amx);
  }
}

}  // namespace

extern "C" int segmif_crosspath_gram_blocks(int64_t N) {
  const long long ntiles = (N + 31) / 32;
  const long long want = (ntiles + CP_WAVES - 1) / CP_WAVES;
  if (want < GRAM_WGS) return (int)(want < 1 ? 1 : want);
  const long long bounded = (ntiles + (long long)CP_WAVES * GRAM_RUN_TILES - 1) / ((long long)CP_WAVES * GRAM_RUN_TILES);
  return (int)(bounded > GRAM_WGS ? bounded : GRAM_WGS);
}

extern "C" int segmif_crosspath_gram_f32(const float* x, int ldx, const float* w, const float* bias, double* partial, int B,
                                         int64_t N, void* stream) {
  if (!x || !w || !partial || B <= 0 || N <= 0 || ldx < 64 || (ldx & 3)) return SEGMIF_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)w) & 15) || ((uintptr_t)partial & 7)) return SEGMIF_EINVAL;
  const int nblk = segmif_crosspath_gram_blocks(N);
  constexpr size_t smem = (size_t)64 * WPB + 3 * 16 * 64 * sizeof(double);
  hipLaunchKernelGGL(crosspath_gram_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(512), smem, (hipStream_t)stream, x, ldx, w,
                     bias, partial, (long long)N);
  return (int)hipGetLastError();
}

extern "C" int segmif_crosspath_fold_f32(const double* partial, int nblk, const float* wkv, const float* wend, float* weff, int B,
                                         int Nout, int ldw, int wofs, int ldweff, int kofs, float scale, uint32_t* cond,
                                         void* stream) {
  if (!partial || !wkv || !wend || !weff || B <= 0 || nblk <= 0 || Nout <= 0) return SEGMIF_EINVAL;
  constexpr size_t smem = (size_t)(3 * 4096 + 2 * 512) * sizeof(double);
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)crosspath_fold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipLaunchKernelGGL(crosspath_fold_kernel, dim3((unsigned)B), dim3(1024), smem, (hipStream_t)stream, partial, nblk, wkv, wend,
                     weff, Nout, ldw, wofs, ldweff, kofs, scale, cond);
  return (int)hipGetLastError();
}

extern "C" int segmif_crosspath_tail_f32(const SegmifCrossTail* d, void* stream) {
  if (!d || !d->x3 || !d->xi || !d->w3 || !d->wi || !d->weff || (!d->out && !d->planes_out) || d->B <= 0 || d->N <= 0) return SEGMIF_EINVAL;  // (out may be NULL when the planes copy is the only consumer)
  if (d->ld3 < 64 || d->ldi < 64 || d->ldo < 64 || ((d->ld3 | d->ldi | d->ldo) & 3)) return SEGMIF_EINVAL;
  if (((uintptr_t)d->x3 | (uintptr_t)d->xi | (uintptr_t)d->w3 | (uintptr_t)d->wi | (uintptr_t)d->weff | (uintptr_t)d->out) & 15)
    return SEGMIF_EINVAL;
  TailK k;
  k.x3 = d->x3; k.xi = d->xi; k.w3 = d->w3; k.b3 = d->b3; k.wi = d->wi; k.bi = d->bi; k.weff = d->weff;
  k.bend = d->bend; k.gamma = d->ln_gamma; k.beta = d->ln_beta; k.out = d->out;
  k.planes = (unsigned char*)d->planes_out;
  k.pl_f16 = k.planes ? d->planes_f16 : 0;
  k.pl_amax = k.pl_f16 ? d->planes_amax : nullptr;
  k.pl_amax_images = d->planes_amax_images;
  if (k.pl_amax && k.pl_amax_images > 1 && k.pl_amax_images != d->B) return SEGMIF_EINVAL;
  k.N = d->N; k.ld3 = d->ld3; k.ldi = d->ldi; k.ldo = d->ldo;
  k.W = 0; k.Hp = 0; k.Wp = 0; k.chunks = 0;
  if (k.planes) {
    if (d->H <= 0 || d->W <= 0 || (int64_t)d->H * d->W != d->N || d->planes_chunks < 4) return SEGMIF_EINVAL;
    int hp, wp;
    if (segmif_planes_dims(d->H, d->W, &hp, &wp) != 0) return SEGMIF_EINVAL;
    k.W = d->W; k.Hp = hp; k.Wp = wp; k.chunks = d->planes_chunks;
  }
  k.eps = d->ln_eps;
  const long long ntiles = (d->N + 31) / 32;
  long long wgs = (ntiles + CP_WAVES - 1) / CP_WAVES;
  const long long per_image = (2 * 256 + d->B - 1) / d->B;  // 100 KB of LDS: one workgroup per CU, two rounds of them
  if (wgs > per_image) wgs = per_image;
  constexpr size_t smem = (size_t)2 * 64 * WPB + 64 * WPB2 + 512 * sizeof(float);
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)crosspath_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)crosspath_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const dim3 grid((unsigned)wgs, (unsigned)d->B);
  hipStream_t st = (hipStream_t)stream;
  if (k.pl_f16) hipLaunchKernelGGL(crosspath_tail_kernel<true>, grid, dim3(512), smem, st, k);
  else hipLaunchKernelGGL(crosspath_tail_kernel<false>, grid, dim3(512), smem, st, k);
  return (int)hipGetLastError();
}
