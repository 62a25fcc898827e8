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

// ---- (r6) f16x3 arithmetic for the tail (planes16.h): weights as three half planes W0 | W - W0 | 2^-11 W0 of the row scaled by a
// power of two, activations as half pairs split in registers; three products per MAC, least significant first: lo W0s, hi Wl, hi W0
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct Op2 {
  u32x4 hi, lo;
};
__device__ __forceinline__ Op2 split8h(const f32x4 a, const f32x4 b) {
  const float y[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  Op2 o;
  p16::split8(y, o.hi, o.lo);
  return o;
}
__device__ __forceinline__ Op2 split8h(const f32x16 t, int s) {  // accumulator registers 8s .. 8s+7
  return split8h(f32x4{t[8 * s], t[8 * s + 1], t[8 * s + 2], t[8 * s + 3]},
                 f32x4{t[8 * s + 4], t[8 * s + 5], t[8 * s + 6], t[8 * s + 7]});
}
__device__ __forceinline__ f16x8 oph(const u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// acc += W X with the weight planes `w` as the MFMA's first operand (rows) and the half pair `x` as its second
__device__ __forceinline__ f32x16 mma3(const u32x4* w, const Op2& x, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oph(w[2]), oph(x.lo), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oph(w[1]), oph(x.hi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oph(w[0]), oph(x.hi), acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void split3h(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 w0 = {(_Float16)x0, (_Float16)x1};
  const h2 wl = {(_Float16)(x0 - (float)w0[0]), (_Float16)(x1 - (float)w0[1])};
  const h2 ws = {(_Float16)((float)w0[0] * (1.f / p16::LSCALE)), (_Float16)((float)w0[1] * (1.f / p16::LSCALE))};
  p0 = __builtin_bit_cast(uint32_t, w0);
  p1 = __builtin_bit_cast(uint32_t, wl);
  p2 = __builtin_bit_cast(uint32_t, ws);
}
__device__ __forceinline__ float pow2_scale(float mx) {  // brings mx into [2^14, 2^15); 1 for zero / non-finite
  int e = 0;
  if (mx >= 1e-30f && mx <= 3e38f) e = 14 - (int)((__float_as_uint(mx) >> 23) - 127);
  return ldexpf(1.f, e);
}
// f16x3 LDS image of 64 rows x NCOL columns (64 | 128) of a row-major fp32 matrix, same positions as stage_split64; the row
// scale 2^-e(n) goes to inv[n].  One row per LPR lanes (a half-wave for 64 columns, a wave for 128): its maximum is a
// shuffle reduction.  Must be called by all `nthreads` threads (a multiple of 64).
template <int NCOL>
__device__ __forceinline__ void stage_split_h(const float* __restrict__ w, int ldw, unsigned char* dst, int pitch, float* inv,
                                              int tid, int nthreads) {
  constexpr int LPR = NCOL / 2;  // lanes (pairs of columns) per row
  for (int u = tid; u < 64 * LPR; u += nthreads) {
    const int row = u / LPR, pp = 2 * (u % LPR);
    const int s = pp >> 4, hh = (pp >> 3) & 1, j = pp & 7;
    const int col = 16 * s + 4 * hh + (j & 3) + 8 * (j >> 2);
    const f32x2 v = *reinterpret_cast<const f32x2*>(w + (long long)row * ldw + col);
    float mx = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float sc = pow2_scale(mx);
    uint32_t a, b, c;
    split3h(v[0] * sc, v[1] * sc, a, b, c);
    unsigned char* d = dst + row * pitch + pp * 2;
    *reinterpret_cast<uint32_t*>(d) = a;
    *reinterpret_cast<uint32_t*>(d + NCOL * 2) = b;
    *reinterpret_cast<uint32_t*>(d + NCOL * 4) = c;
    if (u % LPR == 0) inv[row] = 1.f / sc;
  }
}

// deterministic reduction of the waves' Gram accumulators, then one fp64 partial per workgroup (shared by the two Gram kernels)
__device__ __forceinline__ void gram_reduce_store(const f32x16* g, double* Red, double* __restrict__ partial, int b, int tid, int lane,
                                                  int wave) {
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

// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void crosspath_gram_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                             const float* __restrict__ bias, double* __restrict__ partial,
                                                             long long N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* Ws = smem_raw;                                   // [64][WPB]
  double* Red = reinterpret_cast<double*>(smem_raw + 64 * WPB);   // [3][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  int never = 0;
  asm volatile("" : "+s"(never));
  auto tile = [&](long long t, const f32x4* cur, f32x4* pre, long long tpre) {
    int zo = 0;  // opaque zero in every LDS address below: the weight fragments are loop invariant, and hoisted out of
    asm volatile("" : "+v"(zo));  // the tile loop they would occupy 100+ registers for the whole kernel
    load(tpre, pre);  // ((r5) unconditional - rows past the image clamp to its last one: a branch around the request hides from
                      // the compiler how many are in flight, and the first tile of every three then waited for its own prefetch)
    if (never) asm volatile("s_nop 0");  // (a basic-block end: the scheduler otherwise sinks the request below the tile's MFMAs)
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
  load(t, xa);
  load(t + stride, xb2);
  for (; t < ntiles; t += 3 * stride) {
    tile(t, xa, xc, t + 2 * stride);
    if (t + stride < ntiles) tile(t + stride, xb2, xa, t + 3 * stride);
    if (t + 2 * stride < ntiles) tile(t + 2 * stride, xc, xb2, t + 4 * stride);
  }
  gram_reduce_store(g, Red, partial, b, tid, lane, wave);
}

// LAZY Gram (r5): G_b = sum_px relu(u_px) relu(u_px)^T where u = bilinear(s_low -> H x W) and s_low is the
// (B, ih x iw, lds) LOW-resolution map of channel_proj3's u half already applied (W x + c, no ReLU): the resize is a convex
// combination per channel, so it commutes with the Linear, and the full-resolution segmentation feature is never formed (see
// crosspath_tail_kernel<., LAZY>).  Stage 1 of crosspath_gram_kernel (48 MFMAs, the 3-way split of x, 256 B of HBM per pixel)
// becomes: source values from an L2-resident map, three multiply-adds, a ReLU - produced directly in the layout stage 2 wants
// (lane = channel, register = pixel).
//  * The pixel -> source arithmetic (bilinear_kernel's, csrc/rowops.hip) is the same for the 32 lanes of a half, so it is done once
//    per pixel with lane = pixel and handed over through a per-wave LDS table instead of ~40 vector instructions per pixel per lane.
//  * The four pixels of an aligned group of four share their source rows (W % 4 == 0) and - the resize being an enlargement by
//    three or more (3 iw <= W) - touch at most three adjacent source columns c0 .. c0 + 2: six loads serve a group's 16 taps; a
//    pixel selects (c0, c0 + 1) or (c0 + 1, c0 + 2).  The first version loaded every tap (128 dword loads per 32-pixel tile) and
//    ran at the texture path's instruction rate: 1.69 ms against 1.57 ms for the kernel that reads the full tensor from HBM.
//  * Two 24-register buffers (one K step = 8 pixels x 2 channel tiles each) are requested one tile ahead; GFENCE: see FENCE in
//    crosspath_tail_kernel.
__global__ __launch_bounds__(512, 4) void crosspath_gram_lazy_kernel(const float* __restrict__ s, int lds, int ih, int iw, int W, float sy,
                                                                  float sx, double* __restrict__ partial, long long N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Red = reinterpret_cast<double*>(smem_raw);                                  // [3][16][64]
  // [wave][slot][32 pixels]; TapO (read for group leaders only): offsets of (y0, c0), (y1, c0), and of columns c0 + 1, c0 + 2
  // relative to c0 (clamped at the right edge); TapW: (ly, lx, pixel uses columns c0 + 1 / c0 + 2 ? 1 : 0, -)
  u32x4* TapO = reinterpret_cast<u32x4*>(smem_raw + 3 * 16 * 64 * sizeof(double));
  f32x4* TapW = reinterpret_cast<f32x4*>(smem_raw + 3 * 16 * 64 * sizeof(double) + CP_WAVES * 2 * 32 * sizeof(u32x4));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const float* __restrict__ sb = s + (long long)b * ih * iw * lds;
  f32x16 g[3] = {zero16(), zero16(), zero16()};
  const long long ntiles = (N + 31) / 32;
  const long long stride = (long long)gridDim.x * CP_WAVES;

  // lane = pixel r of tile tt -> table slot `slot` of this wave
  auto table = [&](long long tt, int slot) {
    long long px = tt * 32 + r;
    px = px < N ? px : N - 1;  // (clamped into the image: the loads are unconditional, pixels past N are masked in `consume`)
    const unsigned yy = (unsigned)px / (unsigned)W, xx = (unsigned)px - yy * (unsigned)W;
    const float fy = fmaxf(sy * ((float)yy + 0.5f) - 0.5f, 0.f);
    const float fx = fmaxf(sx * ((float)xx + 0.5f) - 0.5f, 0.f);
    const float fl = fmaxf(sx * ((float)(xx & ~3u) + 0.5f) - 0.5f, 0.f);  // the group's first pixel
    const int y0 = (int)fy, x0 = (int)fx, c0 = (int)fl;
    const int y1 = min(y0 + 1, ih - 1);
    u32x4 o;
    o[0] = (unsigned)((y0 * iw + c0) * lds);
    o[1] = (unsigned)((y1 * iw + c0) * lds);
    o[2] = (unsigned)((min(c0 + 1, iw - 1) - c0) * lds);
    o[3] = (unsigned)((min(c0 + 2, iw - 1) - c0) * lds);
    TapO[(wave * 2 + slot) * 32 + r] = o;  // (both lane halves write the same values)
    TapW[(wave * 2 + slot) * 32 + r] = f32x4{fy - (float)y0, fx - (float)x0, x0 > c0 ? 1.f : 0.f, 0.f};
  };
  // register j of K step ks is pixel 16 ks + 8 (j >> 2) + (j & 3) + 4 h of the tile (v = 8 ks + j: (v & 3) + 8 (v >> 2) + 4 h);
  // group gq = j >> 2 starts at pixel 16 ks + 8 gq + 4 h.   buf[((gq * 2 + nt) * 2 + row) * 3 + col]
  auto issue = [&](int slot, int ks, float* buf) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const u32x4 o = TapO[(wave * 2 + slot) * 32 + 16 * ks + 8 * gq + 4 * h];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const unsigned base = o[row] + (unsigned)(32 * nt + r);
          float* q = buf + ((gq * 2 + nt) * 2 + row) * 3;
          q[0] = sb[base];
          q[1] = sb[base + o[2]];
          q[2] = sb[base + o[3]];
        }
    }
  };
  auto consume = [&](long long t, int slot, int ks, const float* buf) {
    f32x4 y[2][2];  // [nt][low / high four registers of the K step]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gq = j >> 2, l = 16 * ks + 8 * gq + (j & 3);
      const f32x4 w = TapW[(wave * 2 + slot) * 32 + l + 4 * h];
      const float ly = w[0], lx = w[1], hy = 1.f - ly, hx = 1.f - lx;
      const bool right = w[2] != 0.f;
      const bool pv = t * 32 + l + 4 * h < N;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float* q = buf + ((gq * 2 + nt) * 2) * 3;  // row 0: q[0..2], row 1: q[3..5]
        const float a0 = right ? q[1] : q[0], b0 = right ? q[2] : q[1];
        const float a1 = right ? q[4] : q[3], b1 = right ? q[5] : q[4];
        const float v = fmaxf(hy * (hx * a0 + lx * b0) + ly * (hx * a1 + lx * b1), 0.f);
        y[nt][gq][j & 3] = pv ? v : 0.f;
      }
    }
    const Op3 y0 = split8(y[0][0], y[0][1]), y1 = split8(y[1][0], y[1][1]);
    g[0] = mma6(y0.p, y0.p, g[0]);
    g[1] = mma6(y0.p, y1.p, g[1]);
    g[2] = mma6(y1.p, y1.p, g[2]);
  };
  float bufA[24], bufB[24];
  long long t = (long long)blockIdx.x * CP_WAVES + wave;
  int slot = 0;
  int never = 0;
  asm volatile("" : "+s"(never));
#define GFENCE() do { if (never) asm volatile("s_nop 0"); } while (0)
  if (t < ntiles) {
    table(t, 0);
    GFENCE();
    issue(0, 0, bufA);
    GFENCE();
    issue(0, 1, bufB);
    GFENCE();
  }
  for (; t < ntiles; t += stride, slot ^= 1) {
    // (unconditional requests - the last iteration re-requests its own tile: with a branch around them the compiler cannot count
    // what is in flight and the second K step's wait drains the requests just made)
    table(t + stride < ntiles ? t + stride : t, slot ^ 1);
    GFENCE();
    consume(t, slot, 0, bufA);
    GFENCE();
    issue(slot ^ 1, 0, bufA);
    GFENCE();
    consume(t, slot, 1, bufB);
    GFENCE();
    issue(slot ^ 1, 1, bufB);
    GFENCE();
  }
#undef GFENCE
  __syncthreads();
  gram_reduce_store(g, Red, partial, b, tid, lane, wave);
}

// (r6) total[b][u] = sum_k partial[b][k][u] in a fixed order (eight interleaved running sums, then a fixed tree): the fold below runs
// ONE workgroup per image, and pulling an image's 64 - 128 partials (1.5 - 3 MB, 24 KB apart per element) through one CU was
// 165 - 300 us of its launch - 8 launches per pair forward, 8.6 % of a 4-pair step, 4.5 % of the 1024 x 1024 one
// (profiles/r06_config1_kernel_stats.txt, r06_config4_kernel_stats.txt).  Here 12 workgroups per image share the read.
__global__ __launch_bounds__(256) void crosspath_gram_sum_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ total) {
  const int u = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const double* p = partial + (long long)b * nblk * 3072 + u;
  double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int k = 0;
  for (; k + 8 <= nblk; k += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] += p[(long long)(k + j) * 3072];
  }
  for (; k < nblk; ++k) s8[k & 7] += p[(long long)k * 3072];
  total[(long long)b * 3072 + u] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
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
    // fixed order: deterministic.  (r6) Eight independent running sums - partial k goes to sum k & 7 - so that eight loads are
    // in flight per thread instead of one: the single dependent chain over >= 64 partials (24 KB apart) was most of this kernel's
    // 165 us, which is 8 launches x one workgroup per image per pair forward - 9 % of a 4-pair step (profiles/r06_config1_kernel_stats.txt)
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int k = 0;
    for (; k + 8 <= nblk; k += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s8[j] += p[(long long)(k + j) * 3072 + idx];
    }
    for (; k < nblk; ++k) s8[k & 7] += p[(long long)k * 3072 + idx];
    G[u] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
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
  int ih, iw;                         // LAZY: x3 is the (B, ih x iw, ld3) low-resolution map of PROJECTED rows (see below); W = image width
  float sy, sx;                       // LAZY: ih / H, iw / W (the resize's source step, as segmif_bilinear_nhwc_f32 forms it)
  uint32_t* ar_amax;                  // A16: range slot(s) for the operands the kernel splits (x_i, the interpolated y_3) or null
  int ar_amax_images;
};

// F16: the planes copy is an f16x3 one (its own instantiation: the bf16 kernel sits at the register limit).
// LAZY (r5): x_3 = bilinear(seg_low -> H x W) is never materialised.  A bilinear resize is a per-channel convex
// combination of four source pixels, so it commutes with channel_proj3's Linear: W3 x_3 + b3 = bilinear(W3 seg_low + b3).  The host
// runs that Linear at LOW resolution (1/16 or 1/64 of the pixels) and hands the result as x3; here a lane interpolates its pixel's
// row of it (the arithmetic of bilinear_kernel, csrc/rowops.hip), applies the ReLU and has - in the registers the MFMA wants - what
// stage 1 of source 0 produced before: 48 of the tile's 192 MFMAs, the 3-way split of x_3 and 256 of the 768 bytes per pixel of
// HBM traffic go away (the low-resolution map is L2 / MALL resident: 4.9 MB per image at 120 x 160).  The four source rows of one
// K step (16 channels: 8 loads of 16 bytes) travel in the 32 registers that held the prefetched x_3 rows; the four K steps of
// source 0 are spread between the phases of source 1 so that each step's rows are in flight during 24-48 MFMAs of other work.
// LAZY tiles issue a FIXED number of vector-memory operations: rows are read from addresses clamped into the image and lanes past
// the image's last pixel recompute and re-store that pixel (identical bytes to the same address) instead of being predicated.  With
// branches around loads or stores the compiler cannot count what is in flight and every wait for one K step's rows becomes
// s_waitcnt vmcnt(0) - draining the x_i prefetch (HBM latency) four times per tile and the previous tile's stores at its top; that
// made the first LAZY version 11 % faster instead of the third its traffic promised.  OUT = false: no fp32 output exists in the
// instantiation at all (the planes copy is the only consumer - the forward's hot case); OUT = true: p.out is checked at run time.
// A16 (r6; LAZY, planes-only instantiation): the kernel's own contractions on f16x3 operands - Wi and the per-image Weff staged as
// power-of-two-scaled half planes (row scales in LDS, found while staging), x_i / the interpolated y_3 / relu(channel_proj_i) split
// into half pairs in registers: 72 MFMAs and the two-way splits per 32-pixel tile instead of 144 and the three-way ones.  Round 4
// built this for the kernel that still read x_3 from HBM and measured nothing (it moved 768 B per pixel: HBM-bound); with the
// segmentation feature read at its own resolution the tail is bound by exactly this work.  Inside a guarded scope
// only: max |x_i| and max |y_3| go to their own range slot (ar_amax), an overflow of the un-tracked relu(channel_proj_i) turns
// the output into inf / NaN, which the planes copy's slot reports.
template <bool F16, bool LAZY, bool OUT, bool A16 = false>
__global__ __launch_bounds__(512) void crosspath_tail_kernel(const TailK p) {
  static_assert(!A16 || (F16 && LAZY && !OUT), "f16x3 arithmetic: the forward's hot instantiation only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* W3s = smem_raw;             // [64][WPB]
  unsigned char* Wis = W3s + 64 * WPB;       // [64][WPB]
  unsigned char* Wes = Wis + 64 * WPB;       // [64][WPB2]
  float* Cst = reinterpret_cast<float*>(Wes + 64 * WPB2);  // b3[64] bi[64] bend[64] gamma[64] beta[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the tile loop's branches are then scalar too)
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  if constexpr (!LAZY) stage_split64(p.w3, 64, 0, W3s, WPB, 0, 128, tid, 512);
  if constexpr (A16) {  // (Cst + 384: 2^-e of Wi's rows, Cst + 448: of this image's Weff rows)
    stage_split_h<64>(p.wi, 64, Wis, WPB, Cst + 384, tid, 512);
    stage_split_h<128>(p.weff + (long long)b * 64 * 128, 128, Wes, WPB2, Cst + 448, tid, 512);
  } else {
    stage_split64(p.wi, 64, 0, Wis, WPB, 0, 128, tid, 512);
  }
  {
    const float* we = p.weff + (long long)b * 64 * 128;
    if constexpr (!A16) {
      stage_split64(we, 128, 0, Wes, WPB2, 0, 256, tid, 512);
      stage_split64(we, 128, 64, Wes, WPB2, 64, 256, tid, 512);
    }
    if (tid < 320) {
      const int a = tid >> 6, c = tid & 63;
      const float* src = a == 0 ? p.b3 : a == 1 ? p.bi : a == 2 ? p.bend : a == 3 ? p.gamma : p.beta;
      Cst[tid] = src ? src[c] : (a == 3 ? 1.f : 0.f);
    }
  }
  __syncthreads();
  if constexpr (A16) {  // biases of the two f16x3 contractions divided by their rows' scales 2^-e (exact: powers of two)
    if (tid < 64) Cst[64 + tid] /= Cst[384 + tid];
    else if (tid < 128) Cst[128 + tid - 64] /= Cst[448 + tid - 64];
    __syncthreads();
  }
  const float* __restrict__ x3b = p.x3 + (long long)b * (LAZY ? (long long)p.ih * p.iw : p.N) * p.ld3;
  const float* __restrict__ xib = p.xi + (long long)b * p.N * p.ldi;
  float* __restrict__ outb = p.out + (long long)b * p.N * p.ldo;

  const long long ntiles = (p.N + 31) / 32;
  const long long stride = (long long)gridDim.x * CP_WAVES;
  uint32_t pl_amx = 0u;  // f16x3 planes copy: largest |out| this lane wrote (p16::absmax_pk patterns)
  // A16: largest |x_i| this lane split, as an fp32 bit pattern (integer maximum: a NaN stays on top).  The other two operands the kernel
  // splits are not tracked - y_3 is a convex combination of the low-resolution map's rows, relu(channel_proj_i) is bounded by
  // |Wi| |x_i| + |bi| -: an overflow of either turns the output into inf / NaN, which the planes copy's own slot reports, and a
  // half pair's ABSOLUTE error is at most 2^-36 whatever the magnitude, i.e. below fp32's resolution of the residual x_i they are
  // added to as long as x_i itself is in range (which this slot checks).  (Tracking them on the split halves made hipcc spill 79
  // registers in this kernel; every spill reload drains the prefetch it sits beside.)
  uint32_t ar_amx = 0u;
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
  // a tile's epilogue: + residual x_i, LayerNorm, stores (shared by the two tile bodies below)
  auto finish = [&](const f32x16* z, const f32x4* ci, long long px, bool ok, int zo) {
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
          if constexpr (A16) ar_amx = max(ar_amx, __float_as_uint(ci[4 * mt + g][e]) & 0x7fffffffu);  // (x_i: the operand whose range matters, below)
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
        if constexpr (OUT) {
          if (ok && p.out) *reinterpret_cast<f32x4*>(outb + px * p.ldo + mt * 32 + 8 * g + 4 * h) = v;
        }
      }
    if constexpr (F16) {
      if (ok) {
        // (r5) 32-bit pixel arithmetic and byte offsets from a wave-uniform image base: the launcher bounds N and the image's
        // planes bytes below 2^31 (a 64-bit division and 64-bit addresses per store held registers the LAZY body needs)
        const unsigned upx = (unsigned)px;
        const unsigned yy = upx / (unsigned)p.W, xx = upx - yy * (unsigned)p.W;
        unsigned char* __restrict__ pb = p.planes + (long long)b * p.chunks * p.Hp * p.Wp * p16::PIXEL_BYTES;
        const unsigned off = ((yy + 2u) * (unsigned)p.Wp + xx + 2u) * (unsigned)p16::PIXEL_BYTES + (unsigned)h * 16u;
        const unsigned cstride = (unsigned)p.Hp * (unsigned)p.Wp * (unsigned)p16::PIXEL_BYTES;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          u32x4 hi, lo;
          p16::split8(o + 8 * c, hi, lo);
          *reinterpret_cast<u32x4*>(pb + (size_t)(off + c * cstride)) = hi;
          *reinterpret_cast<u32x4*>(pb + (size_t)(off + c * cstride + 32u)) = lo;
          pl_amx = p16::absmax_pk4(pl_amx, hi, lo);
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
    finish(z, ci, px, ok, zo);
  };
  // ---- LAZY: x_3's projected rows interpolated from the low-resolution map ----
  struct Taps {
    int o00, o01, o10, o11;  // element offsets of the four source rows (channel 4h included)
    float ly, lx;
  };
  auto taps = [&](long long tt) {  // of this lane's pixel of tile tt (clamped into the image: the loads are unconditional)
    long long px = tt * 32 + r;
    px = px < p.N ? px : p.N - 1;
    const int yy = (int)((unsigned)px / (unsigned)p.W), xx = (int)px - yy * p.W;
    const float fy = fmaxf(p.sy * ((float)yy + 0.5f) - 0.5f, 0.f);  // (bilinear_kernel's arithmetic, csrc/rowops.hip)
    const float fx = fmaxf(p.sx * ((float)xx + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, p.ih - 1), x1 = min(x0 + 1, p.iw - 1);
    Taps tp;
    tp.ly = fy - (float)y0;
    tp.lx = fx - (float)x0;
    tp.o00 = (y0 * p.iw + x0) * p.ld3 + 4 * h;
    tp.o01 = (y0 * p.iw + x1) * p.ld3 + 4 * h;
    tp.o10 = (y1 * p.iw + x0) * p.ld3 + 4 * h;
    tp.o11 = (y1 * p.iw + x1) * p.ld3 + 4 * h;
    return tp;
  };
  // the four source rows' pieces q = 2 ks, 2 ks + 1 (16 channels = one K step of stage 2): buf[2 tap + j]
  auto issue = [&](const Taps& tp, int ks, f32x4* buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 8 * (2 * ks + j);
      buf[0 + j] = *reinterpret_cast<const f32x4*>(x3b + tp.o00 + c);
      buf[2 + j] = *reinterpret_cast<const f32x4*>(x3b + tp.o01 + c);
      buf[4 + j] = *reinterpret_cast<const f32x4*>(x3b + tp.o10 + c);
      buf[6 + j] = *reinterpret_cast<const f32x4*>(x3b + tp.o11 + c);
    }
  };
  // K step ks of source 0: T = relu(interpolated rows) is already in stage 2's operand layout (registers 8 sp .. 8 sp + 7 of tile nt,
  // ks = 2 nt + sp, are pieces 2 ks and 2 ks + 1)
  auto step0 = [&](const Taps& tp, int ks, const f32x4* buf, f32x16* z, int zo) {
    const float hy1 = 1.f - tp.ly, hx1 = 1.f - tp.lx;
    const f32x2 hy = {hy1, hy1}, hx = {hx1, hx1}, ly = {tp.ly, tp.ly}, lx = {tp.lx, tp.lx};
    f32x4 pc[2];
    // packed arithmetic on the register pairs AS LOADED (elements 0-1 and 2-3 of each 16-byte piece): written per element, the
    // compiler paired values of different source rows for v_pk_mul_f32 and copied them together right behind the loads - a wait
    // on requests issued a few instructions earlier
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 v00 = {buf[0 + j][e], buf[0 + j][e + 1]}, v01 = {buf[2 + j][e], buf[2 + j][e + 1]};
        const f32x2 v10 = {buf[4 + j][e], buf[4 + j][e + 1]}, v11 = {buf[6 + j][e], buf[6 + j][e + 1]};
        const f32x2 v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        pc[j][e] = fmaxf(v[0], 0.f);
        pc[j][e + 1] = fmaxf(v[1], 0.f);
      }
    if constexpr (A16) {
      const Op2 th = split8h(pc[0], pc[1]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        u32x4 wf[3];
        wfrag(Wes + zo, mt * 32 + r, WPB2, 256, ks, h, wf);
        z[mt] = mma3(wf, th, z[mt]);
      }
    } else {
      const Op3 tk = split8(pc[0], pc[1]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        u32x4 wf[3];
        wfrag(Wes + zo, mt * 32 + r, WPB2, 256, ks, h, wf);
        z[mt] = mma6(wf, tk.p, z[mt]);
      }
    }
  };
  // one tile: c3 holds the source rows of its K step 0 (requested by the previous tile); ci / ni as above
  auto load_clamped = [&](long long tt, f32x4* dst) {  // x_i rows, unconditional
    long long px = tt * 32 + r;
    px = px < p.N ? px : p.N - 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = *reinterpret_cast<const f32x4*>(xib + px * p.ldi + 8 * q + 4 * h);
  };
  // Scheduling fences of the LAZY tile: a branch the compiler cannot fold (never taken) ends a basic block, and the instruction
  // scheduler works inside blocks.  Without them it either sinks a K step's requests down to their first use or lifts that use
  // (the interpolation) up to just behind the requests to free their 32 registers - both wait on rows requested a few instructions
  // earlier.  (sched_barrier masks and fake register dependencies each stopped one of the two motions and provoked the other.)
  // The fenced blocks hold no vector-memory operation of their own, so the in-flight counts the waits are built from stay exact.
  int never = 0;
  asm volatile("" : "+s"(never));
#define FENCE() do { if (never) asm volatile("s_nop 0"); } while (0)
  auto tile_lazy = [&](long long t, f32x4* c3, const f32x4* ci, f32x4* ni) {
    long long px = t * 32 + r;
    px = px < p.N ? px : p.N - 1;  // (lanes past the end redo the last pixel: see the template's header)
    constexpr bool ok = true;
    int zo = 0;
    asm volatile("" : "+v"(zo));
    const Taps tp = taps(t);
    f32x16 z[2];
    z[0] = rows16(Cst + zo + 128);  // (A16: the bias divided by the row's scale 2^-e - exact -, the scale applied at the end)
    z[1] = rows16(Cst + zo + 160);
    step0(tp, 0, c3, z, zo);
    issue(tp, 1, c3);
    FENCE();  // (the x_i rows AFTER this K step's rows: the wait below may then leave them in flight)
    load_clamped(t + stride, ni);
    FENCE();
    // source 1 = x_i -> u_i half, stage 1 (as in tile())
    f32x16 tt[2];
    if constexpr (A16) {
      tt[0] = rows16(Cst + zo + 64);  // bias / 2^-e(c): relu(2^-e (W' x) + b) = 2^-e relu(W' x + b 2^e)
      tt[1] = rows16(Cst + zo + 96);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const Op2 xh = split8h(ci[2 * s], ci[2 * s + 1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 wf[3];
          wfrag(Wis + zo, nt * 32 + r, WPB, 128, s, h, wf);
          tt[nt] = mma3(wf, xh, tt[nt]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)  // ReLU, row scale 2^-e(c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(Cst + zo + 384 + nt * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) tt[nt][4 * g + e] = fmaxf(tt[nt][4 * g + e], 0.f) * sc[e];
        }
    } else {
      tt[0] = rows16(Cst + zo + 64);
      tt[1] = rows16(Cst + zo + 96);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const Op3 xs = split8(ci[2 * s], ci[2 * s + 1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 wf[3];
          wfrag(Wis + zo, nt * 32 + r, WPB, 128, s, h, wf);
          tt[nt] = mma6(wf, xs.p, tt[nt]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int v = 0; v < 16; ++v) tt[nt][v] = fmaxf(tt[nt][v], 0.f);
      }
    }
    FENCE();
    step0(tp, 1, c3, z, zo);
    issue(tp, 2, c3);
    FENCE();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const int ks = 4 + nt * 2 + sp;
        if constexpr (A16) {
          const Op2 th = split8h(tt[nt], sp);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            u32x4 wf[3];
            wfrag(Wes + zo, mt * 32 + r, WPB2, 256, ks, h, wf);
            z[mt] = mma3(wf, th, z[mt]);
          }
        } else {
          const Op3 tk = split8(tt[nt], sp);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            u32x4 wf[3];
            wfrag(Wes + zo, mt * 32 + r, WPB2, 256, ks, h, wf);
            z[mt] = mma6(wf, tk.p, z[mt]);
          }
        }
      }
      if (nt == 0) {
        FENCE();
        step0(tp, 2, c3, z, zo);
        issue(tp, 3, c3);
        FENCE();
      }
    }
    FENCE();
    step0(tp, 3, c3, z, zo);
    issue(taps(t + stride), 0, c3);  // the next tile's first K step
    FENCE();
    if constexpr (A16) {  // end_proj: row scale 2^-e(m) (the bias went in divided by it)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(Cst + zo + 448 + mt * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) z[mt][4 * g + e] *= sc[e];
        }
    }
    finish(z, ci, px, ok, zo);
  };
  f32x4 a3[8], ai[8], bi[8];
  long long t = (long long)blockIdx.x * CP_WAVES + wave;
  if constexpr (LAZY) {
    issue(taps(t), 0, a3);
    FENCE();  // (same order and counts as a tile leaves behind: eight operations after the K step's rows)
    load_clamped(t, ai);
    FENCE();
  } else {
    load(t, x3b, p.ld3, a3);
    load(t, xib, p.ldi, ai);
  }
  for (; t < ntiles; t += 2 * stride) {
    if constexpr (LAZY) {
      tile_lazy(t, a3, ai, bi);
      if (t + stride < ntiles) tile_lazy(t + stride, a3, bi, ai);
    } else {
      tile(t, a3, ai, bi);
      if (t + stride < ntiles) tile(t + stride, a3, bi, ai);
    }
  }
#undef FENCE
  if constexpr (F16) {
    if (p.pl_amax) p16::fold_pat(p.pl_amax, p.pl_amax_images > 1 ? b : 0, p.pl_amax_images > 1 ? b : 0, pl_amx);
  }
  if constexpr (A16) {
    if (p.ar_amax) p16::fold_bits(p.ar_amax, p.ar_amax_images > 1 ? b : 0, p.ar_amax_images > 1 ? b : 0, ar_amx);
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

extern "C" int segmif_crosspath_gram_lazy_f32(const float* s_low, int lds, int ih, int iw, int H, int W, double* partial, int B,
                                              void* stream) {
  if (!s_low || !partial || B <= 0 || ih <= 0 || iw <= 0 || H <= 0 || W <= 0 || lds < 64) return SEGMIF_EINVAL;
  if (((uintptr_t)s_low & 3) || ((uintptr_t)partial & 7)) return SEGMIF_EINVAL;
  const long long N = (long long)H * W;
  if (N >= (1ll << 31) || (long long)ih * iw * lds >= (1ll << 31)) return SEGMIF_EINVAL;  // 32-bit offsets inside an image
  if ((W & 3) || 3ll * iw > W) return SEGMIF_EINVAL;  // aligned groups of four pixels share <= 3 source columns (kernel header)
  const int nblk = segmif_crosspath_gram_blocks(N);
  constexpr size_t smem = 3 * 16 * 64 * sizeof(double) + CP_WAVES * 2 * 32 * (sizeof(u32x4) + sizeof(f32x4));
  hipLaunchKernelGGL(crosspath_gram_lazy_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(512), smem, (hipStream_t)stream, s_low, lds,
                     ih, iw, W, (float)ih / (float)H, (float)iw / (float)W, partial, N);
  return (int)hipGetLastError();
}

extern "C" int segmif_crosspath_gram_sum_f64(const double* partial, int nblk, double* total, int B, void* stream) {
  if (!partial || !total || nblk <= 0 || B <= 0 || (((uintptr_t)partial | (uintptr_t)total) & 7)) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(crosspath_gram_sum_kernel, dim3(12u, (unsigned)B), dim3(256), 0, (hipStream_t)stream, partial, nblk, total);
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
  if (!d || !d->x3 || !d->xi || (!d->w3 && d->x3_ih <= 0) || !d->wi || !d->weff || (!d->out && !d->planes_out) || d->B <= 0 || d->N <= 0) return SEGMIF_EINVAL;  // (out may be NULL when the planes copy is the only consumer)
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
    if (d->N >= (1ll << 31) || (long long)d->planes_chunks * hp * wp * 96 >= (1ll << 31)) return SEGMIF_EINVAL;  // 32-bit offsets inside an image
  }
  k.eps = d->ln_eps;
  k.ih = k.iw = 0; k.sy = k.sx = 0.f;
  const bool lazy = d->x3_ih > 0 || d->x3_iw > 0;
  if (lazy) {  // x3 = the low-resolution map of projected rows (header): H x W is the resize's target
    if (d->x3_ih <= 0 || d->x3_iw <= 0 || d->H <= 0 || d->W <= 0 || (int64_t)d->H * d->W != d->N || d->N >= (1ll << 31)) return SEGMIF_EINVAL;
    if ((int64_t)d->x3_ih * d->x3_iw * d->ld3 >= (1ll << 31)) return SEGMIF_EINVAL;  // 32-bit element offsets inside an image
    k.ih = d->x3_ih; k.iw = d->x3_iw; k.W = d->W;
    k.sy = (float)d->x3_ih / (float)d->H; k.sx = (float)d->x3_iw / (float)d->W;
  }
  // (r6) f16x3 arithmetic: the lazy, planes-only, f16-planes launch inside a guarded scope (arith_amax given)
  const bool a16 = d->arith_f16 != 0;
  k.ar_amax = nullptr; k.ar_amax_images = 1;
  if (a16) {
    if (!lazy || !k.pl_f16 || k.out || !d->arith_amax || (d->arith_amax_images > 1 && d->arith_amax_images != d->B)) return SEGMIF_EINVAL;
    k.ar_amax = d->arith_amax; k.ar_amax_images = d->arith_amax_images;
  }
  const long long ntiles = (d->N + 31) / 32;
  long long wgs = (ntiles + CP_WAVES - 1) / CP_WAVES;
  const long long per_image = (2 * 256 + d->B - 1) / d->B;  // 100 KB of LDS: one workgroup per CU, two rounds of them
  if (wgs > per_image) wgs = per_image;
  constexpr size_t smem = (size_t)2 * 64 * WPB + 64 * WPB2 + 512 * sizeof(float);
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipSuccess;
    for (const void* fn : {(const void*)crosspath_tail_kernel<false, false, true>, (const void*)crosspath_tail_kernel<true, false, true>,
                           (const void*)crosspath_tail_kernel<false, true, true>, (const void*)crosspath_tail_kernel<true, true, true>,
                           (const void*)crosspath_tail_kernel<true, true, false>, (const void*)crosspath_tail_kernel<true, true, false, true>})
      if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const dim3 grid((unsigned)wgs, (unsigned)d->B);
  hipStream_t st = (hipStream_t)stream;
  if (lazy) {
    if (a16) hipLaunchKernelGGL((crosspath_tail_kernel<true, true, false, true>), grid, dim3(512), smem, st, k);
    else if (k.pl_f16 && !k.out) hipLaunchKernelGGL((crosspath_tail_kernel<true, true, false>), grid, dim3(512), smem, st, k);
    else if (k.pl_f16) hipLaunchKernelGGL((crosspath_tail_kernel<true, true, true>), grid, dim3(512), smem, st, k);
    else hipLaunchKernelGGL((crosspath_tail_kernel<false, true, true>), grid, dim3(512), smem, st, k);
  } else {
    if (k.pl_f16) hipLaunchKernelGGL((crosspath_tail_kernel<true, false, true>), grid, dim3(512), smem, st, k);
    else hipLaunchKernelGGL((crosspath_tail_kernel<false, false, true>), grid, dim3(512), smem, st, k);
  }
  return (int)hipGetLastError();
}
