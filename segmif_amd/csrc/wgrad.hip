// Weight gradients of the convolutions / linears on the SegMiF hot path, on the exact-fp32 matrix pipe.
//
//   dW[n][k] = sum_m dY[m][n] * A(m, k)        A = the same on-the-fly im2col gather as igemm.hip
//
// The contraction index is the pixel/token index m (millions of rows), the output is small
// (N x K).  Both operands sit in memory with m as their ROW index, so LDS tiles are kept [m][n] and
// [m][k] exactly as loaded and the MFMA fragments are read "down the columns": for the k-pair
// {2j, 2j+1} of v_mfma_f32_32x32x2_f32, lane (r, h) reads dY[m0 + 2j + h][n0 + r] and
// X[m0 + 2j + h][k0 + r] — consecutive lanes hit consecutive banks, the two halves are separate
// lane groups, so the ds_read_b32 stream is conflict free without padding.
//
// Parallelisation: grid = (n-tiles * k-tiles, 1, m-chunks).  A block owns a (32*NT) x (32*KT)
// tile of dW (NT*KT = 4 waves, one 32x32 sub-tile each) and one contiguous chunk of rows; the
// per-chunk partials are summed by a second, deterministic pass (fp64 accumulation) straight into
// the parameter's OIHW / (N, K) gradient layout — no atomics, bitwise reproducible.
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "igemm_common.h"
#include "planes16.h"
#include "segmif_hip.h"

#ifndef WG3_DBG
#define WG3_DBG 0  // tuning aid (profiles/r03_wgrad3x3_phases.txt): 1 compiles the phase-skipping switches of wgrad3x3_split in
#endif

namespace segmif {
namespace {

constexpr int MR = 32;  // rows per LDS tile

struct WgradK {
  const float* dy;
  const float* in;
  float* partial;  // [chunks][N][Kp]
  long long M;
  int N, K, Kp;
  int ldy, lda;
  int H, W, Cin, KH, KW, stride, pad, dil, OH, OW;
  int conv;  // 0 dense, 1 conv (Cin % 4 == 0), 2 generic scalar gather
  int yvec;  // dY rows are 16-byte loadable
  long long rows_per_chunk;
  int nkt, nnt, chunks;
  long long in_zs, dy_zs;  // batch strides (gridDim.z = batches * chunks)
  int nz2;                 // second batch level (attention heads): batch z = z1 * nz2 + z2, offsets z1 * zs + z2 * zs2
  long long in_zs2, dy_zs2;
  float* bias_partial;  // optional [batch][chunk][N]: column sums of dY (bias gradient), k-tile 0 blocks only
};

// FAST: dense problem (nn.Linear / 1x1 conv), 16-byte loadable dY rows, N % BN == 0 and K % BK == 0 - every prefetch load
// is unconditional (rows past the chunk's end are clamped to its last row and zeroed when the tile goes to LDS).  Loads
// under divergent branches made the compiler wait for ALL outstanding loads at each join, i.e. before the MFMA loop the
// prefetch is meant to hide under.
template <int NT, int KT, bool FAST>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradK p) {
  constexpr int BN = 32 * NT, BK = 32 * KT;
  constexpr int YU = MR * BN / 4 / 256 > 0 ? MR * BN / 4 / 256 : 1;  // float4 units per thread (dY tile)
  constexpr int XU = MR * BK / 4 / 256;  // float4 units per thread (X tile)
  static_assert(NT * KT == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float Ys[2][MR * BN];
  __shared__ __attribute__((aligned(16))) float Xs[2][MR * BK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave % NT, wk = wave / NT;
  const int nt = blockIdx.x % p.nnt, kt = blockIdx.x / p.nnt;
  const int n0 = nt * BN, k0 = kt * BK;
  const int zb = blockIdx.z / p.chunks, chunk = blockIdx.z - zb * p.chunks;
  const int zb1 = zb / p.nz2, zb2 = zb - zb1 * p.nz2;
  const float* __restrict__ dyp = p.dy + (long long)zb1 * p.dy_zs + (long long)zb2 * p.dy_zs2;
  const float* __restrict__ inp = p.in + (long long)zb1 * p.in_zs + (long long)zb2 * p.in_zs2;
  const long long m_begin = (long long)chunk * p.rows_per_chunk;
  const long long m_end = (m_begin + p.rows_per_chunk < p.M) ? m_begin + p.rows_per_chunk : p.M;

  // loaders: dY tile MR x BN, X tile MR x BK, 16 B per thread per unit
  constexpr int YUPR = BN / 4, XUPR = BK / 4;
  f32x4 ry[YU], rx[XU];
  f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};  // this thread's column quad of sum_m dY (every tile is loaded once)
  const bool want_bias = p.bias_partial != nullptr && kt == 0;
  unsigned okbits = 0;  // FAST: bit j = ry[j]'s row exists, bit 16 + j = rx[j]'s row exists
  auto gload = [&](long long m0) {
    if (FAST) {
      okbits = 0;
#pragma unroll
      for (int j = 0; j < YU; ++j) {
        const int u = min(tid + 256 * j, MR * YUPR - 1);
        const int row = u / YUPR, q = (u % YUPR) * 4;
        const long long m = m0 + row;
        ry[j] = *reinterpret_cast<const f32x4*>(dyp + (m < m_end ? m : m_end - 1) * p.ldy + n0 + q);
        okbits |= (unsigned)(m < m_end) << j;
      }
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const int u = tid + 256 * j;
        const int row = u / XUPR, q = (u % XUPR) * 4;
        const long long m = m0 + row;
        rx[j] = *reinterpret_cast<const f32x4*>(inp + (m < m_end ? m : m_end - 1) * p.lda + k0 + q);
        okbits |= (unsigned)(m < m_end) << (16 + j);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < YU; ++j) {
      const int u = tid + 256 * j;
      const int row = u / YUPR, q = (u % YUPR) * 4;
      ry[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const long long m = m0 + row;
      if (u < MR * YUPR && m < m_end) {
        const float* src = dyp + m * p.ldy + n0 + q;
        if (p.yvec && n0 + q + 3 < p.N) ry[j] = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + q + e < p.N) ry[j][e] = src[e];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      const int u = tid + 256 * j;
      const int row = u / XUPR, q = (u % XUPR) * 4;
      rx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const long long m = m0 + row;
      const int k = k0 + q;
      if (m < m_end && k < p.K) {
        if (p.conv == 0) {
          rx[j] = *reinterpret_cast<const f32x4*>(inp + m * p.lda + k);  // K % 4 == 0 for dense
        } else {
          const long long ohw = (long long)p.OH * p.OW;
          const long long b = m / ohw;
          const int rem = (int)(m - b * ohw);
          const int oy = rem / p.OW, ox = rem - oy * p.OW;
          if (p.conv == 1) {  // 4 consecutive k share a tap (Cin % 4 == 0)
            const int tap = k / p.Cin, c = k - tap * p.Cin;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
              rx[j] = *reinterpret_cast<const f32x4*>(inp + ((b * p.H + iy) * p.W + ix) * p.lda + c);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int kk = k + e;
              if (kk < p.K) {
                const int tap = kk / p.Cin, c = kk - tap * p.Cin;
                const int ky = tap / p.KW, kx = tap - ky * p.KW;
                const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                  rx[j][e] = inp[((b * p.H + iy) * p.W + ix) * p.lda + c];
              }
            }
          }
        }
      }
    }
  };
  auto sstore = [&](int buf) {
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < YU; ++j) {
      const int u = tid + 256 * j;
      const f32x4 v = (!FAST || ((okbits >> j) & 1)) ? ry[j] : zero4;
      if (u < MR * YUPR) {
        *reinterpret_cast<f32x4*>(&Ys[buf][u * 4]) = v;
        // (the bias sum is taken HERE, where the tile is consumed anyway: summing in gload made every prefetch wait for
        // its own data before the MFMA loop it was meant to hide under)
        if (want_bias) bsum += v;
      }
    }
#pragma unroll
    for (int j = 0; j < XU; ++j)
      *reinterpret_cast<f32x4*>(&Xs[buf][(tid + 256 * j) * 4]) = (!FAST || ((okbits >> (16 + j)) & 1)) ? rx[j] : zero4;
  };

  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;

  const long long ntiles = (m_end - m_begin + MR - 1) / MR;
  if (ntiles > 0) {
    gload(m_begin);
    sstore(0);
  }
  __syncthreads();
  for (long long t = 0; t < ntiles; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < ntiles) gload(m_begin + (t + 1) * MR);
    const float* ya = &Ys[cur][h * BN + wn * 32 + r];
    const float* xa = &Xs[cur][h * BK + wk * 32 + r];
    float fa[MR / 2], fb[MR / 2];  // the tile's fragments first, then the MFMAs: one LDS round trip per tile, not one per pair
#pragma unroll
    for (int j = 0; j < MR / 2; ++j) {
      fa[j] = ya[2 * j * BN];
      fb[j] = xa[2 * j * BK];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < MR / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    if (t + 1 < ntiles) sstore(cur ^ 1);
    __syncthreads();
  }
  if (want_bias) {  // combine the 256 / YUPR threads that share a column quad (Ys is free now)
    f32x4* red = reinterpret_cast<f32x4*>(&Ys[0][0]);
    red[tid] = bsum;
    __syncthreads();
    if (tid < YUPR) {
      f32x4 sacc = red[tid];
      for (int i = 1; i < 256 / YUPR; ++i) sacc += red[tid + i * YUPR];
      float* bo = p.bias_partial + (long long)blockIdx.z * p.N;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + tid * 4 + e < p.N) bo[n0 + tid * 4 + e] = sacc[e];
    }
  }
  // D[i = n][j = k]: lane holds column k = r, rows n = (v&3) + 8*(v>>2) + 4*h
  float* out = p.partial + (long long)blockIdx.z * p.N * p.Kp;  // [batch][chunk][N][Kp]
  const int k = k0 + wk * 32 + r;
  if (k < p.Kp) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int n = n0 + wn * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
      if (n < p.N) out[(long long)n * p.Kp + k] = acc[v];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Halo-tiled weight gradient of a 3x3 stride-1 convolution (dilation 1 or 2): the DRDB convs and the
// other 3x3 convs of the fusion net.  The generic kernel above re-gathers the shifted input once per
// tap and — with only 32 output channels — is bound by that gather (32 % of the MFMA peak measured).
// Here a block owns a 32-channel slice of the input and walks a strip of 8x32 pixel tiles: per tile it
// stages dY (256 px x 32 n) and the input halo ((8+2d) x (32+2d) px x 32 c) in LDS once; wave w takes
// tile rows 2w, 2w+1 and accumulates ALL nine taps (nine 32x32 accumulators): A = dY^T read once per
// pixel pair, B = the halo at nine shifted addresses.  After the strip the four waves' accumulators are
// combined through LDS and written as one partial [N][9*Cin] slab (summed by wgrad_reduce_kernel).
// ---------------------------------------------------------------------------------------------------
struct Wg3K {
  const float* dy;
  const float* in;
  float* partial;  // [strips][N][Kp]
  float* bias_partial;  // [strips][N] or null (only channel-chunk 0 blocks write it)
  int B, H, W, Cin, N, Kp, ldy, lda;
  int tiles_x, tiles_y, tiles_total, tiles_per_strip, nchunks, yvec;
  int dbg;  // diagnosis only, builds with -DWG3_DBG=1 (SEGMIF_WG3_DBG): 1 skip the MFMA loop, 2 skip the split + LDS stores, 4 skip the global loads
  // f16x3 form of the two-team kernel: range slots (bit patterns of max |.|) of dY and of the input's channel blocks
  const uint32_t* dy_amax;
  const uint32_t* in_amax;
  int dy_amax_n, in_amax_n;
};

template <int DIL, bool VEC>  // VEC: dY rows are 16-byte loadable and the block's 32 output channels all exist
__global__ __launch_bounds__(256) void wgrad3x3_halo_kernel(const Wg3K p) {
  constexpr int TH = 8, TW = 32, HH = TH + 2 * DIL, HWD = TW + 2 * DIL, HP = HH * HWD;
  constexpr int YJ = 8;  // dY float4 units per thread: 256 px * 8 / 256
  constexpr int XJ = (HP * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ys = smem;  // [256][32]
  float* Xs = smem + 256 * 32;  // [HP][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int chunk = blockIdx.x % p.nchunks, strip = blockIdx.x / p.nchunks;
  const int ntile = blockIdx.y;  // 32 output channels
  const int c0 = chunk * 32, n0 = ntile * 32;
  const int t_begin = strip * p.tiles_per_strip;
  const int t_end = min(t_begin + p.tiles_per_strip, p.tiles_total);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  f32x4 bsum{0.f, 0.f, 0.f, 0.f};
  const bool want_bias = p.bias_partial != nullptr && chunk == 0;

  f32x4 ry[YJ], rx[XJ];
  unsigned okbits = 0;  // VEC: bit j = ry[j] inside the image, bit 8 + j = rx[j] inside the image
  auto gload = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    const int b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const long long img = (long long)b * p.H * p.W;
    if (VEC) {
      // unconditional loads from clamped addresses, zeroed at the LDS store: a load under a divergent branch makes the
      // compiler wait for every outstanding load at the join, i.e. BEFORE the MFMA loop this prefetch should hide under
      okbits = 0;
#pragma unroll
      for (int j = 0; j < YJ; ++j) {
        const int u = tid + 256 * j;
        const int px = u >> 3, q = (u & 7) * 4;
        const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
        ry[j] = *reinterpret_cast<const f32x4*>(p.dy + (img + (long long)min(gy, p.H - 1) * p.W + min(gx, p.W - 1)) * p.ldy + n0 + q);
        okbits |= (unsigned)(gy < p.H && gx < p.W) << j;
      }
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const int u = min(tid + 256 * j, HP * 8 - 1);
        const int pp = u >> 3, q = (u & 7) * 4;
        const int hy = pp / HWD, hx = pp - hy * HWD;
        const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
        rx[j] = *reinterpret_cast<const f32x4*>(p.in + (img + (long long)min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1)) * p.lda +
                                                c0 + q);
        okbits |= (unsigned)((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) << (8 + j);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      const int u = tid + 256 * j;
      const int px = u >> 3, q = (u & 7) * 4;
      const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
      ry[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gy < p.H && gx < p.W) {
        const float* src = p.dy + (img + (long long)gy * p.W + gx) * p.ldy + n0 + q;
        if (p.yvec && n0 + q + 3 < p.N) ry[j] = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + q + e < p.N) ry[j][e] = src[e];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int u = tid + 256 * j;
      rx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < HP * 8) {
        const int pp = u >> 3, q = (u & 7) * 4;
        const int hy = pp / HWD, hx = pp - hy * HWD;
        const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
          rx[j] = *reinterpret_cast<const f32x4*>(p.in + (img + (long long)gy * p.W + gx) * p.lda + c0 + q);
      }
    }
  };
  auto sstore = [&]() {
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      const f32x4 v = (!VEC || ((okbits >> j) & 1)) ? ry[j] : zero4;
      *reinterpret_cast<f32x4*>(Ys + (tid + 256 * j) * 4) = v;
      if (want_bias) bsum += v;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int u = tid + 256 * j;
      if (u < HP * 8) *reinterpret_cast<f32x4*>(Xs + u * 4) = (!VEC || ((okbits >> (8 + j)) & 1)) ? rx[j] : zero4;
    }
  };

  if (t_begin < t_end) gload(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    sstore();
    __syncthreads();
    if (tile + 1 < t_end) gload(tile + 1);
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      const int ry_ = 2 * wave + row;
      const float* ya = Ys + (ry_ * 32 + h) * 32 + r;  // pixel (ry_, 2j + h), channel n = r
      const float* xa = Xs + (ry_ * HWD + h) * 32 + r;  // halo pixel (ry_ + ky*DIL, 2j + h + kx*DIL), channel c = r
#pragma unroll 4
      for (int j = 0; j < 16; ++j) {
        const float a = ya[2 * j * 32];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                a, xa[((ky * DIL) * HWD + 2 * j + kx * DIL) * 32], acc[ky * 3 + kx], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // combine the four waves (each holds 9 x [32 n][32 c]) tap by tap through LDS, write the partial slab
  float* red = smem;  // [4][32][33]
  float* out = p.partial + (long long)strip * p.N * p.Kp;
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int n = (v & 3) + 8 * (v >> 2) + 4 * h;
      red[(wave * 32 + n) * 33 + r] = acc[t][v];
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
      const int n = i >> 5, c = i & 31;
      const float s4 = (red[(0 * 32 + n) * 33 + c] + red[(1 * 32 + n) * 33 + c]) +
                       (red[(2 * 32 + n) * 33 + c] + red[(3 * 32 + n) * 33 + c]);
      if (n0 + n < p.N) out[(long long)(n0 + n) * p.Kp + t * p.Cin + c0 + c] = s4;
    }
  }
  if (want_bias) {
    __syncthreads();
    f32x4* rb = reinterpret_cast<f32x4*>(smem);
    rb[tid] = bsum;
    __syncthreads();
    if (tid < 8) {
      f32x4 sacc = rb[tid];
      for (int i = 1; i < 32; ++i) sacc += rb[tid + 8 * i];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + tid * 4 + e < p.N) p.bias_partial[(long long)strip * p.N + n0 + tid * 4 + e] = sacc[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe with 3-way split operands ("bf16x6", fp32-class: see
// conv3x3_planes.hip), for the dilation-2 convs of the DRDBs - 15 % of the round-2 fusion training step sat in the
// fp32 kernel above at 57 % of the fp32 pipe.  The contraction index is the PIXEL, and v_mfma_f32_32x32x16_bf16
// wants 8 consecutive contraction slots per lane, while both operands arrive pixel-major / channel-minor: the
// staging pass therefore stores PIXEL PAIRS - one dword = (px 2i, px 2i+1) of one channel, bf16 x 3 planes - as
// [plane][pair][32 channels].  A lane (channel r, half h) then reads its 8 slots as 4 dwords 128 B apart
// (conflict free: the 32 lanes of a half read 32 consecutive dwords), and with dilation 2 the three horizontal taps
// of a halo row are the dword windows [0..3], [1..4], [2..5] of ONE 6-dword read (a shift of two pixels is one
// dword).  Staging: a thread loads two adjacent pixels x 4 channels (16-byte loads, 8 lanes cover a pixel's 128
// bytes), splits them, and writes one 16-byte unit per plane.  Same tiling, strip partials, bias sums and wave
// combine as the fp32 kernel; the partial slabs go through the same deterministic reduce.
// ---------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t wg_pk_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}

// (x0, x1) -> three dwords, each (bf16 of x0 in the low half, of x1 in the high half); x = p0 + p1 + p2 to 24 bits
__device__ __forceinline__ void wg_split3(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = wg_pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
  p1 = wg_pk_bf16(r0, r1);
  r0 -= __uint_as_float(p1 << 16);
  r1 -= __uint_as_float(p1 & 0xffff0000u);
  p2 = wg_pk_bf16(r0, r1);
}

__global__ __launch_bounds__(256) void wgrad3x3_split_kernel(const Wg3K p) {
  constexpr int DIL = 2, TH = 8, TW = 32, HH = TH + 2 * DIL, HWD = TW + 2 * DIL;
  constexpr int YPAIRS = TH * TW / 2;    // 128
  constexpr int XROWP = HWD / 2;         // 18 pairs per halo row
  constexpr int XPAIRS = HH * XROWP;     // 216
  constexpr int YJ = YPAIRS * 8 / 256;   // 4 units (pair x channel quad) per thread
  constexpr int XJ = (XPAIRS * 8 + 255) / 256;  // 7
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0};  // six products, least significant first: plane of dY ...
  constexpr int PB[6] = {0, 1, 2, 0, 1, 0};  // ... and of the input
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t* Ys = reinterpret_cast<uint32_t*>(smem);  // [3][YPAIRS][32]
  uint32_t* Xs = Ys + 3 * YPAIRS * 32;               // [3][XPAIRS][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int chunk = blockIdx.x % p.nchunks, strip = blockIdx.x / p.nchunks;
  const int ntile = blockIdx.y;
  const int c0 = chunk * 32, n0 = ntile * 32;
  const int t_begin = strip * p.tiles_per_strip;
  const int t_end = min(t_begin + p.tiles_per_strip, p.tiles_total);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  f32x4 bsum{0.f, 0.f, 0.f, 0.f};
  const bool want_bias = p.bias_partial != nullptr && chunk == 0;
  const f32x4 zero4{0.f, 0.f, 0.f, 0.f};

  // Prefetch registers + one validity bit per loaded pixel.  Every load is UNCONDITIONAL from a clamped (in-image) address and
  // out-of-image pixels are zeroed when the tile is stored to LDS: a load under a divergent branch made the compiler wait
  // for all outstanding loads at the join (s_waitcnt vmcnt(0) before the MFMA loop the prefetch is meant to hide under -
  // measured: loads 1.12 ms + MFMA 0.85 ms = 2.03 ms per call, no overlap at all).
  f32x4 ry[2 * YJ], rx[2 * XJ];
  unsigned okbits = 0;  // bit 2j+e: ry[2j+e] valid; bit 16 + 2j+e: rx[2j+e] valid
  auto gload = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    const int b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const long long img = (long long)b * p.H * p.W;
    okbits = 0;
    // (launched only for 16-byte loadable dY rows and N % 32 == 0: no element-wise path here)
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      const int u = tid + 256 * j;
      const int pair = u >> 3, q = (u & 7) * 4;
      const int gy = y0 + (pair >> 4), gx = x0 + 2 * (pair & 15);
      const int cy = min(gy, p.H - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int cx = min(gx + e, p.W - 1);
        ry[2 * j + e] = *reinterpret_cast<const f32x4*>(p.dy + (img + (long long)cy * p.W + cx) * p.ldy + n0 + q);
        okbits |= (unsigned)(gy < p.H && gx + e < p.W) << (2 * j + e);
      }
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int u = min(tid + 256 * j, XPAIRS * 8 - 1);  // (the last round's surplus threads re-load the last unit and drop it)
      const int pair = u >> 3, q = (u & 7) * 4;
      const int hy = pair / XROWP, hx = 2 * (pair - hy * XROWP);
      const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
      const int cy = min(max(gy, 0), p.H - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int cx = min(max(gx + e, 0), p.W - 1);
        rx[2 * j + e] = *reinterpret_cast<const f32x4*>(p.in + (img + (long long)cy * p.W + cx) * p.lda + c0 + q);
        okbits |= (unsigned)((unsigned)gy < (unsigned)p.H && (unsigned)(gx + e) < (unsigned)p.W) << (16 + 2 * j + e);
      }
    }
  };
  auto put = [&](uint32_t* base, int npairs, int unit, const f32x4 v0, const f32x4 v1) {
    u32x4 w0, w1, w2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t a, b, d;
      wg_split3(v0[c], v1[c], a, b, d);
      w0[c] = a; w1[c] = b; w2[c] = d;
    }
    uint32_t* dst = base + unit * 4;  // (pair, quad) -> dword pair * 32 + quad * 4
    *reinterpret_cast<u32x4*>(dst) = w0;
    *reinterpret_cast<u32x4*>(dst + npairs * 32) = w1;
    *reinterpret_cast<u32x4*>(dst + 2 * npairs * 32) = w2;
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      const f32x4 v0 = (okbits >> (2 * j)) & 1 ? ry[2 * j] : zero4, v1 = (okbits >> (2 * j + 1)) & 1 ? ry[2 * j + 1] : zero4;
      put(Ys, YPAIRS, tid + 256 * j, v0, v1);
      if (want_bias) bsum += v0 + v1;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int u = tid + 256 * j;
      const f32x4 v0 = (okbits >> (16 + 2 * j)) & 1 ? rx[2 * j] : zero4, v1 = (okbits >> (17 + 2 * j)) & 1 ? rx[2 * j + 1] : zero4;
      if (u < XPAIRS * 8) put(Xs, XPAIRS, u, v0, v1);
    }
  };

  if (t_begin < t_end) gload(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    if (!(WG3_DBG && (p.dbg & 2))) sstore();
    __syncthreads();
    if (tile + 1 < t_end && !(WG3_DBG && (p.dbg & 4))) gload(tile + 1);
    if (WG3_DBG && (p.dbg & 1)) { __syncthreads(); continue; }
    // 12 steps per tile: (row of the wave's two, 16-pixel k-step, vertical tap); a step = 18 MFMAs on one 6-dword halo
    // window per plane (+ the dY fragment, re-read when the pixel run changes).  The LDS reads of step i + 1 are issued
    // BEFORE the MFMAs of step i (one wave per SIMD: nothing else hides their latency).
    auto read_a = [&](int it, u32x4* a) {
      const int row = it / 6, ks = (it / 3) & 1;
      // dY: pixels (ry_, 16 ks + 8 h + 0..7) of channel n = r  ->  pairs ry_*16 + 8 ks + 4 h + 0..3
      const uint32_t* ya = Ys + ((2 * wave + row) * 16 + 8 * ks + 4 * h) * 32 + r;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[pl][i] = ya[(pl * YPAIRS + i) * 32];
    };
    auto read_b = [&](int it, uint32_t (*b6)[6]) {
      const int row = it / 6, ks = (it / 3) & 1, ky = it % 3;
      // halo row ry_ + ky*DIL, halo pixels 16 ks + 8 h + kx*DIL + 0..7  ->  pairs 8 ks + 4 h + kx + 0..3
      const uint32_t* xa = Xs + ((2 * wave + row + ky * DIL) * XROWP + 8 * ks + 4 * h) * 32 + r;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 6; ++i) b6[pl][i] = xa[(pl * XPAIRS + i) * 32];
    };
    u32x4 a[2][3];
    uint32_t b6[2][3][6];
    read_a(0, a[0]);
    read_b(0, b6[0]);
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int cb = it & 1, ca = (it / 3) & 1, ky = it % 3;
      if (it + 1 < 12) {
        if ((it + 1) % 3 == 0) read_a(it + 1, a[((it + 1) / 3) & 1]);
        read_b(it + 1, b6[cb ^ 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // product-major, tap-minor: consecutive MFMAs go to three different accumulators (no back-to-back dependent issue)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int pl = PB[t];
          const u32x4 bw = u32x4{b6[cb][pl][kx], b6[cb][pl][kx + 1], b6[cb][pl][kx + 2], b6[cb][pl][kx + 3]};
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ca][PA[t]]),
                                                                     __builtin_bit_cast(bf16x8, bw), acc[ky * 3 + kx], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // combine the four waves (each holds 9 x [32 n][32 c]) tap by tap through LDS, write the partial slab
  float* red = smem;  // [4][32][33]
  float* out = p.partial + (long long)strip * p.N * p.Kp;
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int n = (v & 3) + 8 * (v >> 2) + 4 * h;
      red[(wave * 32 + n) * 33 + r] = acc[t][v];
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
      const int n = i >> 5, c = i & 31;
      const float s4 = (red[(0 * 32 + n) * 33 + c] + red[(1 * 32 + n) * 33 + c]) +
                       (red[(2 * 32 + n) * 33 + c] + red[(3 * 32 + n) * 33 + c]);
      if (n0 + n < p.N) out[(long long)(n0 + n) * p.Kp + t * p.Cin + c0 + c] = s4;
    }
  }
  if (want_bias) {
    __syncthreads();
    f32x4* rb = reinterpret_cast<f32x4*>(smem);
    rb[tid] = bsum;
    __syncthreads();
    if (tid < 8) {
      f32x4 sacc = rb[tid];
      for (int i = 1; i < 32; ++i) sacc += rb[tid + 8 * i];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + tid * 4 + e < p.N) p.bias_partial[(long long)strip * p.N + n0 + tid * 4 + e] = sacc[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad3x3_split_kernel with the staging taken off the multiplying waves: 8 waves = a STAGING team (global loads, 3-way
// split, pixel-pair packing, LDS stores - and the bias sums) and an MFMA team (fragment reads + the 108 MFMAs of a tile),
// two LDS tile buffers in ping-pong, ONE workgroup barrier per tile:
//     staging:  store(0) | store(1) | store(2) | ...
//     MFMA:              | mult(0)  | mult(1)  | ...
// In the one-team kernel the split phase (0.4 ms of a 1.52 ms call) and the tail of the loads were serial to the MFMA phase
// (0.83 ms).  Tiles are 4 x 32 pixels here (two 80 KB buffers fill the 160 KB of LDS): a wave of the MFMA team owns one tile
// row, the halo is 8 x 36 pixels.  Everything else - pair-packed planes, dword windows for the three horizontal taps, strips,
// partial slabs, reduce - is the one-team kernel's.
// ---------------------------------------------------------------------------------------------------
// F16 (round 4, default on the training path): both operands as half pairs x = hi + lo of the value scaled by the power of two
// that puts the tensor's maximum (device range slots, planes16.h range_scale) in [2^13, 2^14) - lo is kept UNscaled here
// (with the maximum pinned at 2^13 it stays a normal half down to 2^-3, i.e. 2^-16 of the maximum; below that its absolute
// error is 2^-25, 2^-38 of the maximum) so that the three products hi.hi, hi.lo, lo.hi share one accumulator; the partial
// slabs are written back unscaled (exact).  Half the matrix-pipe work and two thirds of the LDS traffic of the bf16x6 form.
template <int DIL, bool F16>  // DIL 2: DRDB convs (tap shift = one dword); 1: conv2 / conv21 (tap shift = half a dword: the middle tap is a funnel shift)
__global__ __launch_bounds__(512) void wgrad3x3_split2_kernel(const Wg3K p) {
  constexpr int NPL = F16 ? 2 : 3, NPROD = F16 ? 3 : 6;
  constexpr int TH = 4, TW = 32, HH = TH + 2 * DIL, HWD = TW + 2 * DIL;
  constexpr int NB = DIL == 2 ? 6 : 5;   // dwords of a halo row a lane reads per (k-step, vertical tap)
  constexpr int YPAIRS = TH * TW / 2;    // 64
  constexpr int XROWP = HWD / 2;         // 18 pairs per halo row
  constexpr int XPAIRS = HH * XROWP;     // 144
  constexpr int YJ = YPAIRS * 8 / 256;   // 2 units per staging thread
  constexpr int XJ = (XPAIRS * 8 + 255) / 256;  // 5
  constexpr int BUF = NPL * YPAIRS * 32 + NPL * XPAIRS * 32;  // dwords per tile buffer (bf16x6: 79 872 bytes)
  // products (dY plane, input plane), least significant first: six for bf16 triples, lo.hi, hi.lo, hi.hi for half pairs
  constexpr int PA[6] = {F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0, 0};
  constexpr int PB[6] = {0, 1, F16 ? 0 : 2, 0, 1, 0};
  float s_y = 1.f, s_x = 1.f;
  if (F16) {
    s_y = p16::range_scale(p.dy_amax, p.dy_amax_n);
    s_x = p16::range_scale(p.in_amax, p.in_amax_n);
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t* bufs = reinterpret_cast<uint32_t*>(smem);
  const int tid = threadIdx.x, team = tid >> 8, t256 = tid & 255, lane = tid & 63, wave = (tid >> 6) & 3;
  const int r = lane & 31, h = lane >> 5;
  const int chunk = blockIdx.x % p.nchunks, strip = blockIdx.x / p.nchunks;
  const int ntile = blockIdx.y;
  const int c0 = chunk * 32, n0 = ntile * 32;
  const int t_begin = strip * p.tiles_per_strip;
  const int t_end = min(t_begin + p.tiles_per_strip, p.tiles_total);
  const bool want_bias = p.bias_partial != nullptr && chunk == 0;
  const f32x4 zero4{0.f, 0.f, 0.f, 0.f};

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  f32x4 bsum = zero4;

  if (team == 1) {
    // ---------------- staging team ----------------
    f32x4 ry[2 * YJ], rx[2 * XJ];
    unsigned okbits = 0;
    auto gload = [&](int tile) {  // unconditional loads from clamped addresses (see the one-team kernel)
      const int tx = tile % p.tiles_x;
      const int ty = (tile / p.tiles_x) % p.tiles_y;
      const int b = tile / (p.tiles_x * p.tiles_y);
      const int x0 = tx * TW, y0 = ty * TH;
      const long long img = (long long)b * p.H * p.W;
      okbits = 0;
#pragma unroll
      for (int j = 0; j < YJ; ++j) {
        const int u = t256 + 256 * j;
        const int pair = u >> 3, q = (u & 7) * 4;
        const int gy = y0 + (pair >> 4), gx = x0 + 2 * (pair & 15);
        const int cy = min(gy, p.H - 1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cx = min(gx + e, p.W - 1);
          ry[2 * j + e] = *reinterpret_cast<const f32x4*>(p.dy + (img + (long long)cy * p.W + cx) * p.ldy + n0 + q);
          okbits |= (unsigned)(gy < p.H && gx + e < p.W) << (2 * j + e);
        }
      }
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const int u = min(t256 + 256 * j, XPAIRS * 8 - 1);
        const int pair = u >> 3, q = (u & 7) * 4;
        const int hy = pair / XROWP, hx = 2 * (pair - hy * XROWP);
        const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
        const int cy = min(max(gy, 0), p.H - 1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cx = min(max(gx + e, 0), p.W - 1);
          rx[2 * j + e] = *reinterpret_cast<const f32x4*>(p.in + (img + (long long)cy * p.W + cx) * p.lda + c0 + q);
          okbits |= (unsigned)((unsigned)gy < (unsigned)p.H && (unsigned)(gx + e) < (unsigned)p.W) << (16 + 2 * j + e);
        }
      }
    };
    auto put = [&](uint32_t* base, int npairs, int unit, const f32x4 v0, const f32x4 v1, float sc) {
      uint32_t* dst = base + unit * 4;
      if constexpr (F16) {
        u32x4 w0, w1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const p16::f2 v = {v0[c] * sc, v1[c] * sc};
          const p16::h2 hi = __builtin_convertvector(v, p16::h2);  // round to nearest even
          const p16::f2 back = __builtin_convertvector(hi, p16::f2);
          const p16::f2 res = {v[0] - back[0], v[1] - back[1]};     // exact in fp32
          w0[c] = __builtin_bit_cast(uint32_t, hi);
          w1[c] = __builtin_bit_cast(uint32_t, __builtin_convertvector(res, p16::h2));
        }
        *reinterpret_cast<u32x4*>(dst) = w0;
        *reinterpret_cast<u32x4*>(dst + npairs * 32) = w1;
      } else {
        u32x4 w0, w1, w2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          uint32_t a, b, d;
          wg_split3(v0[c], v1[c], a, b, d);
          w0[c] = a; w1[c] = b; w2[c] = d;
        }
        *reinterpret_cast<u32x4*>(dst) = w0;
        *reinterpret_cast<u32x4*>(dst + npairs * 32) = w1;
        *reinterpret_cast<u32x4*>(dst + (NPL - 1) * npairs * 32) = w2;
      }
    };
    auto sstore = [&](uint32_t* Ys) {
      uint32_t* Xs = Ys + NPL * YPAIRS * 32;
#pragma unroll
      for (int j = 0; j < YJ; ++j) {
        const f32x4 v0 = (okbits >> (2 * j)) & 1 ? ry[2 * j] : zero4, v1 = (okbits >> (2 * j + 1)) & 1 ? ry[2 * j + 1] : zero4;
        put(Ys, YPAIRS, t256 + 256 * j, v0, v1, s_y);
        if (want_bias) bsum += v0 + v1;
      }
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const int u = t256 + 256 * j;
        const f32x4 v0 = (okbits >> (16 + 2 * j)) & 1 ? rx[2 * j] : zero4, v1 = (okbits >> (17 + 2 * j)) & 1 ? rx[2 * j + 1] : zero4;
        if (u < XPAIRS * 8) put(Xs, XPAIRS, u, v0, v1, s_x);
      }
    };
    if (t_begin < t_end) gload(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
      sstore(bufs + ((tile - t_begin) & 1) * BUF);
      if (tile + 1 < t_end) gload(tile + 1);
      __syncthreads();  // tile's buffer is complete; the MFMA team has left the other buffer
    }
  } else {
    // ---------------- MFMA team: wave w owns tile row w ----------------
    for (int tile = t_begin; tile < t_end; ++tile) {
      __syncthreads();
      const uint32_t* Ys = bufs + ((tile - t_begin) & 1) * BUF;
      const uint32_t* Xs = Ys + NPL * YPAIRS * 32;
      auto read_a = [&](int it, u32x4* a) {  // it = 3 ks + ky
        const int ks = it / 3;
        const uint32_t* ya = Ys + (wave * 16 + 8 * ks + 4 * h) * 32 + r;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int i = 0; i < 4; ++i) a[pl][i] = ya[(pl * YPAIRS + i) * 32];
      };
      auto read_b = [&](int it, uint32_t (*b6)[NB]) {
        const int ks = it / 3, ky = it % 3;
        const uint32_t* xa = Xs + ((wave + ky * DIL) * XROWP + 8 * ks + 4 * h) * 32 + r;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int i = 0; i < NB; ++i) b6[pl][i] = xa[(pl * XPAIRS + i) * 32];
      };
      u32x4 a[2][NPL];
      uint32_t b6[2][NPL][NB];
      read_a(0, a[0]);
      read_b(0, b6[0]);
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int cb = it & 1, ca = (it / 3) & 1, ky = it % 3;
        if (it + 1 < 6) {
          if ((it + 1) % 3 == 0) read_a(it + 1, a[((it + 1) / 3) & 1]);
          read_b(it + 1, b6[cb ^ 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int pl = PB[t];
            u32x4 bw;
            if (DIL == 2) {  // halo pixels 2 kx + 0..7: dwords kx .. kx + 3
              bw = u32x4{b6[cb][pl][kx], b6[cb][pl][kx + 1], b6[cb][pl][kx + 2], b6[cb][pl][kx + 3]};
            } else if (kx != 1) {  // halo pixels kx + 0..7, kx even: dwords kx / 2 .. + 3
              bw = u32x4{b6[cb][pl][kx / 2], b6[cb][pl][kx / 2 + 1], b6[cb][pl][kx / 2 + 2], b6[cb][pl][kx / 2 + 3]};
            } else {  // halo pixels 1..8: (pixel 2i + 1, pixel 2i + 2) = high half of dword i, low half of dword i + 1
#pragma unroll
              for (int i = 0; i < 4; ++i) bw[i] = __builtin_amdgcn_alignbit(b6[cb][pl][i + 1], b6[cb][pl][i], 16);
            }
            if constexpr (F16)
              acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, a[ca][PA[t]]),
                                                                        __builtin_bit_cast(wg_f16x8, bw), acc[ky * 3 + kx], 0, 0, 0);
            else
              acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ca][PA[t]]),
                                                                         __builtin_bit_cast(bf16x8, bw), acc[ky * 3 + kx], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // combine the MFMA team's four waves tap by tap through LDS (all 512 threads take the barriers and share the summation)
  float* red = smem;  // [4][32][33]
  float* out = p.partial + (long long)strip * p.N * p.Kp;
  const float desc = F16 ? (1.f / s_y) * (1.f / s_x) : 1.f;  // the two range scales taken out again (powers of two: exact)
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = (v & 3) + 8 * (v >> 2) + 4 * h;
        red[(wave * 32 + n) * 33 + r] = acc[t][v];
      }
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += 512) {
      const int n = i >> 5, c = i & 31;
      const float s4 = (red[(0 * 32 + n) * 33 + c] + red[(1 * 32 + n) * 33 + c]) +
                       (red[(2 * 32 + n) * 33 + c] + red[(3 * 32 + n) * 33 + c]);
      if (n0 + n < p.N) out[(long long)(n0 + n) * p.Kp + t * p.Cin + c0 + c] = F16 ? s4 * desc : s4;
    }
  }
  if (want_bias) {
    __syncthreads();
    f32x4* rb = reinterpret_cast<f32x4*>(smem);
    if (team == 1) rb[t256] = bsum;
    __syncthreads();
    if (tid < 8) {
      f32x4 sacc = rb[tid];
      for (int i = 1; i < 32; ++i) sacc += rb[tid + 8 * i];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + tid * 4 + e < p.N) p.bias_partial[(long long)strip * p.N + n0 + tid * 4 + e] = sacc[e];
    }
  }
}

// sum over chunks (fp64) and scatter from the packed [N][Kp] (tap-major, channel-minor) order into the
// parameter's own layout: OIHW for convs; element (n, k) -> dw[n*sn + k*sk] for dense problems.
// accumulate != 0: grad += value.
// A block covers T = 256 / S consecutive outputs; S slices of the chunk range are summed side by side (fp64) and combined
// in LDS in slice order (fixed: deterministic).  S = 1 when there are enough outputs to fill the chip on their own; small
// outputs with many chunks (conv1: 576 weights x 2048 chunks; a bias: 32 values x 1000 strips) take S = 16 - one thread
// walking every chunk took 0.6 - 0.9 ms per call in round 2.
template <int S>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int chunks, int N, int K, int Kp, int Cin, int KH, int KW,
                                                           int dense, long long sn, long long sk, long long dw_zs,
                                                           int nz2, long long dw_zs2, int accumulate) {
  constexpr int T = 256 / S;
  __shared__ double red[S > 1 ? 256 : 1];
  const int t = threadIdx.x % T, sl = threadIdx.x / T;
  const long long i = (long long)blockIdx.x * T + t;
  const bool live = i < (long long)N * K;
  const int zb = blockIdx.y;
  const int n = live ? (int)(i / K) : 0, k = live ? (int)(i - (long long)n * K) : 0;
  const float* p = partial + (long long)zb * chunks * N * Kp;
  double s = 0.0;
  if (live)
    for (int c = sl; c < chunks; c += S) s += (double)p[((long long)c * N + n) * Kp + k];
  if (S > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl != 0) return;
    s = red[t];
#pragma unroll
    for (int q = 1; q < S; ++q) s += red[q * T + t];
  }
  if (!live) return;
  long long dst;
  if (dense) dst = n * sn + k * sk;
  else {
    const int tap = k / Cin, ch = k - tap * Cin;
    const int ky = tap / KW, kx = tap - ky * KW;
    dst = (((long long)n * Cin + ch) * KH + ky) * KW + kx;
  }
  dst += (zb / nz2) * dw_zs + (zb % nz2) * dw_zs2;
  dw[dst] = accumulate ? dw[dst] + (float)s : (float)s;
}

void launch_wgrad_reduce(hipStream_t s, const float* partial, float* dw, int chunks, int N, int K, int Kp, int Cin, int KH, int KW,
                         int dense, long long sn, long long sk, long long dw_zs, int accumulate, int nz, int nz2 = 1,
                         long long dw_zs2 = 0) {
  const long long total = (long long)N * K;
  if (total >= 64 * 256 || chunks < 16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((total + 255) / 256), (unsigned)nz), dim3(256), 0, s, partial, dw,
                       chunks, N, K, Kp, Cin, KH, KW, dense, sn, sk, dw_zs, nz2, dw_zs2, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((total + 15) / 16), (unsigned)nz), dim3(256), 0, s, partial, dw,
                       chunks, N, K, Kp, Cin, KH, KW, dense, sn, sk, dw_zs, nz2, dw_zs2, accumulate);
}

// db[n] = sum over (batch slice, chunk) of the per-block column sums: 16 columns x 16 slices per block, slices combined in order
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* __restrict__ partial, float* __restrict__ db,
                                                                int chunks, int nz, int N, int accumulate) {
  __shared__ double red[256];
  const int t = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + t;
  double s = 0.0;
  if (n < N)
    for (int c = sl; c < chunks * nz; c += 16) s += (double)partial[(long long)c * N + n];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl != 0 || n >= N) return;
  s = red[t];
#pragma unroll
  for (int q = 1; q < 16; ++q) s += red[q * 16 + t];
  db[n] = accumulate ? db[n] + (float)s : (float)s;
}

// column sums of a rows x N matrix (bias gradients, LN / dwconv / BN parameter partials):
// stage 1: a block owns cs_rows(rows) rows.  Vector path (N % 4 == 0, 16-byte aligned rows): QPR lanes cover the
// column quads of a row (QPR = 8..64), 256/QPR row slots per block, four rows in flight per thread; scalar
// path otherwise.  Slots are combined in LDS and written as one fp64 partial row; stage 2 sums the partial rows.
// rows per block: a fixed function of the row count (results stay deterministic) that keeps a few hundred blocks in
// flight - with a constant 1024 the parameter-gradient partials of a LayerNorm / dwconv backward (<= 1024 rows of
// 2C / 10C floats) were summed by ONE workgroup (90 us per call, 13 % of the round-2 segmentation step)
__host__ __device__ inline int cs_rows(long long rows) {
  long long r = (rows + 511) / 512;
  r = (r + 15) / 16 * 16;
  return (int)(r < 16 ? 16 : (r > 1024 ? 1024 : r));
}
template <int QPR>
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                                 long long rows, int N, int ldx) {
  constexpr int SLOTS = 256 / QPR;
  __shared__ f32x4 red[256];
  const int q = threadIdx.x % QPR, slot = threadIdx.x / QPR;
  const int rpb = cs_rows(rows);
  const long long r0 = (long long)blockIdx.x * rpb;
  const long long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  const int nq = N >> 2;
  for (int q0 = 0; q0 < nq; q0 += QPR) {
    const int qq = q0 + q;
    f32x4 a0{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (qq < nq) {
      const float* base = x + 4 * qq;
      long long r = r0 + slot;
      for (; r + 3 * SLOTS < r1; r += 4 * SLOTS) {
        a0 += *reinterpret_cast<const f32x4*>(base + r * ldx);
        a1 += *reinterpret_cast<const f32x4*>(base + (r + SLOTS) * ldx);
        a2 += *reinterpret_cast<const f32x4*>(base + (r + 2 * SLOTS) * ldx);
        a3 += *reinterpret_cast<const f32x4*>(base + (r + 3 * SLOTS) * ldx);
      }
      for (; r < r1; r += SLOTS) a0 += *reinterpret_cast<const f32x4*>(base + r * ldx);
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (slot == 0 && qq < nq) {
      double s[4] = {0.0, 0.0, 0.0, 0.0};
      for (int sl = 0; sl < SLOTS; ++sl) {
        const f32x4 v = red[sl * QPR + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) partial[(long long)blockIdx.x * N + 4 * qq + e] = s[e];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                             long long rows, int N, int ldx) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int rpb = cs_rows(rows);
  const long long r0 = (long long)blockIdx.x * rpb;
  const long long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int n = n0 + c;
    float s = 0.f;
    if (n < N)
      for (long long r = r0 + slot; r < r1; r += 4) s += x[r * ldx + n];
    red[slot][c] = s;
    __syncthreads();
    if (slot == 0 && n < N)
      partial[(long long)blockIdx.x * N + n] = ((double)red[0][c] + (double)red[1][c]) + ((double)red[2][c] + (double)red[3][c]);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ partial, float* __restrict__ out,
                                                           int nblk, int N, int accumulate) {
  // 64 columns per block, 4 row-slices combined in LDS (fixed order: deterministic)
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  double s = 0.0;
  if (n < N)
    for (int b = slot; b < nblk; b += 4) s += partial[(long long)b * N + n];
  red[slot][c] = s;
  __syncthreads();
  if (slot == 0 && n < N) {
    const double t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    out[n] = accumulate ? out[n] + (float)t : (float)t;
  }
}

// dx = dy * f'(.) for the fused-epilogue activations.  `ref` is the activation OUTPUT for ReLU /
// PReLU (sign test; PReLU needs slope > 0 for sign(out) == sign(pre)) and the PRE-activation for GELU.
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref,
                                                      float* __restrict__ dx, long long rows, int C, int ldy, int ldr,
                                                      int ldx, int act, const float* __restrict__ slope_p,
                                                      const float* __restrict__ ref2, int ldr2) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const long long row = i / C;
  const int c = (int)(i - row * C);
  const float g = dy[row * ldy + c];
  float v = ref[row * ldr + c];
  if (ref2) v -= ref2[row * ldr2 + c];  // the activation output is ref - ref2 (a residual was added after it)
  float o;
  if (act == SEGMIF_ACT_RELU) o = v > 0.f ? g : 0.f;
  else if (act == SEGMIF_ACT_PRELU) o = v >= 0.f ? g : g * *slope_p;
  else if (act == SEGMIF_ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    o = g * (cdf + v * pdf);
  } else o = g;
  dx[row * ldx + c] = o;
}

}  // namespace
}  // namespace segmif

using namespace segmif;

static long long pick_chunks(long long M, int N, int K) {
  const int Kp = (K + 15) / 16 * 16;
  const bool narrow = N <= 32;
  const int BN = narrow ? 32 : 64, BK = narrow ? 128 : 64;
  const long long tiles = (long long)((N + BN - 1) / BN) * ((Kp + BK - 1) / BK);
  long long chunks = (2048 + tiles - 1) / tiles;  // aim for >= 2048 blocks
  const long long max_chunks = (M + 1023) / 1024;  // at least 1024 rows per chunk
  if (chunks > max_chunks) chunks = max_chunks;
  return chunks < 1 ? 1 : chunks;
}

extern "C" int64_t segmif_wgrad_workspace_size(int64_t M, int N, int K) {
  const int Kp = (K + 15) / 16 * 16;
  return pick_chunks(M, N, K) * ((int64_t)N * Kp + N);  // weight partials + bias partials
}

static int wgrad_impl(const SegmifIgemm* d, const float* dy, int ldy, int64_t dy_zstride, int64_t dy_zstride2, float* dw,
                      int64_t dw_sn, int64_t dw_sk, float* dbias, float* workspace, int accumulate, void* stream);

extern "C" int segmif_wgrad_f32(const SegmifIgemm* d, const float* dy, int ldy, int64_t dy_zstride, float* dw,
                                int64_t dw_sn, int64_t dw_sk, float* dbias, float* workspace, int accumulate,
                                void* stream) {
  if (d && d->nz2 > 1) return SEGMIF_EINVAL;  // two batch levels: segmif_wgrad_batched2_f32
  return wgrad_impl(d, dy, ldy, dy_zstride, 0, dw, dw_sn, dw_sk, dbias, workspace, accumulate, stream);
}

extern "C" int segmif_wgrad_batched2_f32(const SegmifIgemm* d, const float* dy, int ldy, int64_t dy_zstride, int64_t dy_zstride2,
                                         float* dw, int64_t dw_sn, int64_t dw_sk, float* workspace, int accumulate,
                                         void* stream) {
  if (!d || d->nz < 1 || d->nz2 < 1 || !(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0)) return SEGMIF_EINVAL;
  return wgrad_impl(d, dy, ldy, dy_zstride, dy_zstride2, dw, dw_sn, dw_sk, nullptr, workspace, accumulate, stream);
}

static int wgrad_impl(const SegmifIgemm* d, const float* dy, int ldy, int64_t dy_zstride, int64_t dy_zstride2, float* dw,
                      int64_t dw_sn, int64_t dw_sk, float* dbias, float* workspace, int accumulate, void* stream) {
  if (!d || !d->in || !dy || !dw || !workspace || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->in2) return SEGMIF_EINVAL;
  WgradK k;
  k.dy = dy; k.in = d->in; k.partial = workspace; k.M = d->M; k.N = d->N; k.K = d->K; k.Kp = (d->K + 15) / 16 * 16;
  k.ldy = ldy; k.lda = d->lda;
  k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad;
  k.dil = d->dil; k.OH = d->OH; k.OW = d->OW;
  const int nz2 = d->nz2 > 1 ? d->nz2 : 1;
  const int nz = (d->nz > 0 ? d->nz : 1) * nz2;  // total batch slices; z = z1 * nz2 + z2
  k.in_zs = d->in_zstride; k.dy_zs = dy_zstride;
  k.nz2 = nz2; k.in_zs2 = nz2 > 1 ? d->in_zstride2 : 0; k.dy_zs2 = nz2 > 1 ? dy_zstride2 : 0;
  const bool is_conv = !(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0);
  if (!is_conv) {
    if ((d->K % 4) || (d->lda % 4) || ((uintptr_t)d->in & 15) || (d->in_zstride % 4) || (k.in_zs2 % 4)) return SEGMIF_EINVAL;
    k.conv = 0;
    k.Cin = d->K; k.KH = k.KW = 1;
  } else {
    k.conv = (d->Cin % 4 == 0 && d->lda % 4 == 0 && !((uintptr_t)d->in & 15) && d->in_zstride % 4 == 0) ? 1 : 2;
  }
  k.yvec = (ldy % 4 == 0) && !((uintptr_t)dy & 15) && (dy_zstride % 4 == 0) && (k.dy_zs2 % 4 == 0);
  {  // halo-tiled path for 3x3 stride-1 convs (DRDB and friends)
    const bool halo = is_conv && d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) &&
                      d->pad == d->dil && d->OH == d->H && d->OW == d->W && d->Cin % 32 == 0 && d->lda % 4 == 0 &&
                      !((uintptr_t)d->in & 15) && nz == 1 && d->N <= 64 && d->M % ((long long)d->H * d->W) == 0;
    if (halo) {
      Wg3K w;
      w.dy = dy; w.in = d->in; w.B = (int)(d->M / ((long long)d->H * d->W)); w.H = d->H; w.W = d->W; w.Cin = d->Cin;
      w.N = d->N; w.Kp = k.Kp; w.ldy = ldy; w.lda = d->lda; w.yvec = k.yvec;
      w.tiles_x = (d->W + 31) / 32; w.tiles_y = (d->H + 7) / 8; w.tiles_total = w.B * w.tiles_x * w.tiles_y;
      w.nchunks = d->Cin / 32;
      const int ntiles_n = (d->N + 31) / 32;
      // strips: as many as the workspace sized by segmif_wgrad_workspace_size allows, aiming at ~1024 blocks
      long long strips = pick_chunks(d->M, d->N, d->K);
      const long long want = (1024 + (long long)w.nchunks * ntiles_n - 1) / ((long long)w.nchunks * ntiles_n);
      if (strips > want) strips = want;
      if (strips > w.tiles_total) strips = w.tiles_total;
      if (strips < 1) strips = 1;
      w.tiles_per_strip = (int)((w.tiles_total + strips - 1) / strips);
      strips = (w.tiles_total + w.tiles_per_strip - 1) / w.tiles_per_strip;
      w.partial = workspace;
      w.bias_partial = dbias ? workspace + strips * d->N * k.Kp : nullptr;
      hipStream_t s = (hipStream_t)stream;
      dim3 grid((unsigned)(strips * w.nchunks), (unsigned)ntiles_n);
      const int HPd = (8 + 2 * d->dil) * (32 + 2 * d->dil);
      const size_t smem = (size_t)(256 * 32 + HPd * 32) * sizeof(float);
      // SEGMIF_WGRAD3X3 (read once per process; a training step is ~3000 launches): "fp32" = the exact-fp32 kernel for
      // dilation 2 as well, "split1" = the one-team bf16x6 kernel (8 x 32 tiles, dilation 2)
      static const int mode = [] { const char* e = getenv("SEGMIF_WGRAD3X3"); return !e ? 0 : !strcmp(e, "fp32") ? 1 : !strcmp(e, "split1") ? 2 : 0; }();
      const bool fp32_only = mode == 1, one_team = mode == 2;
#if WG3_DBG
      static const int dbg = [] { const char* e = getenv("SEGMIF_WG3_DBG"); return e ? atoi(e) : 0; }();
      w.dbg = dbg;
#else
      w.dbg = 0;
#endif
      if (!fp32_only && !(one_team && d->dil == 2) && k.yvec && d->N % 32 == 0) {
        // two teams, 4 x 32 pixel tiles: re-derive the tile grid and the strips for that tile height
        w.tiles_y = (d->H + 3) / 4;
        w.tiles_total = w.B * w.tiles_x * w.tiles_y;
        long long strips2 = strips;
        if (strips2 > w.tiles_total) strips2 = w.tiles_total;
        w.tiles_per_strip = (int)((w.tiles_total + strips2 - 1) / strips2);
        strips = (w.tiles_total + w.tiles_per_strip - 1) / w.tiles_per_strip;  // never more than the workspace was sized for
        w.bias_partial = dbias ? workspace + strips * d->N * k.Kp : nullptr;
        grid = dim3((unsigned)(strips * w.nchunks), (unsigned)ntiles_n);
        const int xpairs = (4 + 2 * d->dil) * (32 + 2 * d->dil) / 2;
        const bool f16 = d->split_f16 != 0;
        if (f16 && (!d->split_in_amax || !d->wgrad_dy_amax || d->split_in_amax_n < 1 || d->split_in_amax_n > 64 ||
                    d->wgrad_dy_amax_n < 1 || d->wgrad_dy_amax_n > 64))
          return SEGMIF_EINVAL;
        w.dy_amax = d->wgrad_dy_amax; w.dy_amax_n = d->wgrad_dy_amax_n; w.in_amax = d->split_in_amax; w.in_amax_n = d->split_in_amax_n;
        const int npl = f16 ? 2 : 3;
        const size_t smem2 = (size_t)2 * (npl * 64 * 32 + npl * xpairs * 32) * sizeof(uint32_t);
        auto fn2 = d->dil == 2 ? (f16 ? wgrad3x3_split2_kernel<2, true> : wgrad3x3_split2_kernel<2, false>)
                               : (f16 ? wgrad3x3_split2_kernel<1, true> : wgrad3x3_split2_kernel<1, false>);
        static segmif::PerDeviceFlag raised_flag4[4];
        bool& raised4 = raised_flag4[(d->dil == 2) * 2 + f16].here();
        if (!raised4) {
          hipError_t e4 = hipFuncSetAttribute((const void*)fn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
          if (e4 != hipSuccess) return (int)e4;
          raised4 = true;
        }
        hipLaunchKernelGGL(fn2, grid, dim3(512), smem2, s, w);
      } else if (d->dil == 2 && !fp32_only && k.yvec && d->N % 32 == 0) {
        constexpr size_t smem_split = (size_t)(3 * 128 * 32 + 3 * 216 * 32) * sizeof(uint32_t);
        static segmif::PerDeviceFlag raised_flag3;
        bool& raised3 = raised_flag3.here();
        if (!raised3) { hipFuncSetAttribute((const void*)wgrad3x3_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_split); raised3 = true; }
        hipLaunchKernelGGL(wgrad3x3_split_kernel, grid, dim3(256), smem_split, s, w);
      } else {
        const bool vec = k.yvec && d->N % 32 == 0;
        auto fn = d->dil == 1 ? (vec ? wgrad3x3_halo_kernel<1, true> : wgrad3x3_halo_kernel<1, false>)
                              : (vec ? wgrad3x3_halo_kernel<2, true> : wgrad3x3_halo_kernel<2, false>);
        static segmif::PerDeviceFlag raised_halo[4];
        bool& raised = raised_halo[(d->dil == 2) * 2 + vec].here();
        if (!raised) {
          hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
          if (e != hipSuccess) return (int)e;
          raised = true;
        }
        hipLaunchKernelGGL(fn, grid, dim3(256), smem, s, w);
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
      launch_wgrad_reduce(s, workspace, dw, (int)strips, d->N, d->K, k.Kp, k.Cin, k.KH, k.KW, 0, (long long)dw_sn, (long long)dw_sk,
                          0LL, accumulate, 1);
      if (dbias)
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((d->N + 15) / 16)), dim3(256), 0, s,
                           w.bias_partial, dbias, (int)strips, 1, d->N, accumulate);
      return (int)hipGetLastError();
    }
  }
  long long chunks = pick_chunks(d->M, d->N, d->K);
  k.rows_per_chunk = ((d->M + chunks - 1) / chunks + MR - 1) / MR * MR;
  chunks = (d->M + k.rows_per_chunk - 1) / k.rows_per_chunk;
  k.chunks = (int)chunks;
  k.bias_partial = dbias ? workspace + (long long)chunks * nz * d->N * k.Kp : nullptr;
  const bool narrow = d->N <= 32;
  const int BN = narrow ? 32 : 64, BK = narrow ? 128 : 64;
  k.nnt = (d->N + BN - 1) / BN;
  k.nkt = (k.Kp + BK - 1) / BK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(k.nnt * k.nkt), 1, (unsigned)(chunks * nz));
  const int BNl = narrow ? 32 : 64, BKl = narrow ? 128 : 64;
  const bool fast = k.conv == 0 && k.yvec && d->N % BNl == 0 && d->K % BKl == 0;
  if (narrow) {
    if (fast) hipLaunchKernelGGL((wgrad_kernel<1, 4, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((wgrad_kernel<1, 4, false>), grid, dim3(256), 0, s, k);
  } else {
    // (a bf16x6 version of this kernel - row pairs packed like wgrad3x3_split_kernel - measured 2 ms per step SLOWER than
    // the FAST fp32 path on both training steps: these problems are bound by their loads and partial slabs, not by the
    // fp32 matrix pipe; profiles/r03_wgrad_dense_bf16x6_ab.txt)
    if (fast) hipLaunchKernelGGL((wgrad_kernel<2, 2, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((wgrad_kernel<2, 2, false>), grid, dim3(256), 0, s, k);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  launch_wgrad_reduce(s, workspace, dw, (int)chunks, d->N, d->K, k.Kp, k.Cin, k.KH, k.KW, is_conv ? 0 : 1, (long long)dw_sn,
                      (long long)dw_sk, (long long)d->out_zstride, accumulate, nz, nz2, nz2 > 1 ? (long long)d->out_zstride2 : 0LL);
  if (dbias)  // bias gradient: sum over every chunk of every batch slice
    hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((d->N + 15) / 16)), dim3(256), 0, s,
                       k.bias_partial, dbias, (int)chunks, nz, d->N, accumulate);
  return (int)hipGetLastError();
}

extern "C" int segmif_colsum_blocks(int64_t rows) { return (int)((rows + cs_rows(rows) - 1) / cs_rows(rows)); }

extern "C" int segmif_colsum_f32(const float* x, float* out, double* workspace, int64_t rows, int N, int ldx,
                                 int accumulate, void* stream) {
  if (!x || !out || !workspace || rows <= 0 || N <= 0 || ldx < N) return SEGMIF_EINVAL;
  const int nblk = segmif_colsum_blocks(rows);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = !(N & 3) && !(ldx & 3) && !((uintptr_t)x & 15);
  const int nq = N >> 2;
  if (!vec) hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, s, x, workspace, (long long)rows, N, ldx);
  else if (nq <= 8) hipLaunchKernelGGL(colsum_partial_vec_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, s, x, workspace, (long long)rows, N, ldx);
  else if (nq <= 16) hipLaunchKernelGGL(colsum_partial_vec_kernel<16>, dim3((unsigned)nblk), dim3(256), 0, s, x, workspace, (long long)rows, N, ldx);
  else if (nq <= 32) hipLaunchKernelGGL(colsum_partial_vec_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, s, x, workspace, (long long)rows, N, ldx);
  else hipLaunchKernelGGL(colsum_partial_vec_kernel<64>, dim3((unsigned)nblk), dim3(256), 0, s, x, workspace, (long long)rows, N, ldx);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, workspace, out, nblk, N,
                     accumulate);
  return (int)hipGetLastError();
}

// 16-byte form of act_bwd_kernel: a block owns `rb` rows and walks their (row, channel quad) units, 32-bit index math only
__global__ __launch_bounds__(256) void act_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ ref,
                                                          float* __restrict__ dx, long long rows, int c4n, int rb, int ldy, int ldr,
                                                          int ldx, int act, const float* __restrict__ slope_p,
                                                          const float* __restrict__ ref2, int ldr2) {
  const long long row0 = (long long)blockIdx.x * rb;
  const int nrows = (int)(rows - row0 < rb ? rows - row0 : rb);
  const float a = act == SEGMIF_ACT_PRELU ? *slope_p : 0.f;
  for (int u = threadIdx.x; u < nrows * c4n; u += 256) {
    const int r = u / c4n, cq = u - r * c4n;
    const long long row = row0 + r;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + row * ldy + 4 * cq);
    f32x4 v = *reinterpret_cast<const f32x4*>(ref + row * ldr + 4 * cq);
    if (ref2) {
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(ref2 + row * ldr2 + 4 * cq);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] -= v2[e];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (act == SEGMIF_ACT_RELU) o[e] = v[e] > 0.f ? g[e] : 0.f;
      else if (act == SEGMIF_ACT_PRELU) o[e] = v[e] >= 0.f ? g[e] : g[e] * a;
      else if (act == SEGMIF_ACT_GELU) {
        const float cdf = 0.5f * (1.0f + erff(v[e] * 0.70710678118654752440f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v[e] * v[e]);
        o[e] = g[e] * (cdf + v[e] * pdf);
      } else o[e] = g[e];
    }
    *reinterpret_cast<f32x4*>(dx + row * ldx + 4 * cq) = o;
  }
}

static int act_bwd_dispatch(const float* dy, const float* ref, const float* ref2, int ldr2, float* dx, int64_t rows, int C, int ldy,
                            int ldr, int ldx, int act, const float* slope, void* stream) {
  if (!dy || !ref || !dx || rows <= 0 || C <= 0) return SEGMIF_EINVAL;
  if (act == SEGMIF_ACT_PRELU && !slope) return SEGMIF_EINVAL;
  const long long total = (long long)rows * C;
  if (!((C | ldy | ldr | ldx | (ref2 ? ldr2 : 0)) & 3) &&
      !(((uintptr_t)dy | (uintptr_t)ref | (uintptr_t)dx | (uintptr_t)ref2) & 15) && rows < (1ll << 33)) {
    const int c4n = C >> 2;
    const int rb = c4n >= 1024 ? 1 : 1024 / c4n;  // ~four 16-byte units per thread
    hipLaunchKernelGGL(act_bwd_vec_kernel, dim3((unsigned)((rows + rb - 1) / rb)), dim3(256), 0, (hipStream_t)stream, dy, ref, dx,
                       (long long)rows, c4n, rb, ldy, ldr, ldx, act, slope, ref2, ldr2);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, ref,
                     dx, (long long)rows, C, ldy, ldr, ldx, act, slope, ref2, ldr2);
  return (int)hipGetLastError();
}

extern "C" int segmif_act_bwd_f32(const float* dy, const float* ref, float* dx, int64_t rows, int C, int ldy, int ldr,
                                  int ldx, int act, const float* slope, void* stream) {
  return act_bwd_dispatch(dy, ref, nullptr, 0, dx, rows, C, ldy, ldr, ldx, act, slope, stream);
}

extern "C" int segmif_act_bwd2_f32(const float* dy, const float* ref, const float* ref2, float* dx, int64_t rows, int C, int ldy,
                                   int ldr, int ldr2, int ldx, int act, const float* slope, void* stream) {
  if (!ref2) return SEGMIF_EINVAL;
  return act_bwd_dispatch(dy, ref, ref2, ldr2, dx, rows, C, ldy, ldr, ldx, act, slope, stream);
}
