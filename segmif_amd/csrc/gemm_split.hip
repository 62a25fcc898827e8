// Dense GEMM on the bf16 matrix pipe with 3-way split operands ("bf16x6", numerics as in conv3x3_split.hip): the MiT
// encoder's nn.Linear layers (core/mix_transformer.py:46-53 fc1 / fc2, :94-115 q / kv / proj) and the SegFormer head's,
//     out = res + act(A W^T + bias),   A: (M, K) fp32 rows with pitch lda,  W: (N, K).
// Round 1 ran these on the exact-fp32 matrix pipe at 60-85 % of its 157 TFLOP/s; six bf16 products per fp32-equivalent
// MAC cost 6 x 32 cycles per 32x32x16 block against 8 x 64, i.e. the same fp32-class result at 2.7x the matrix rate.
//
// Tile 128 x 128 x 32, 4 waves of 64 x 64.  A stays fp32 in HBM: a thread loads 4 x 16 bytes of the next K step while
// the current one multiplies, splits them in registers (x = x0 + x1 + x2, bf16 each, the two residual subtractions are
// exact) and writes the three planes to LDS.  W is split once per parameter version (segmif_gemm_split_pack) into the
// kernel's LDS image, one contiguous 26 KB block per (column tile, K step) that goes HBM/L2 -> LDS by LDS-DMA.
// LDS rows are [K half][plane][16 bf16] = 192 bytes + 16 of padding: 13 sixteen-byte slots per row, so the 16 lanes of
// a ds_read_b128 service group land on 16 distinct slots.  53 KB per workgroup: three workgroups share a CU and cover
// each other's staging phases.
//
// F16 = true ("f16x3", segmif_gemm_split16_*; the format and its range guard are described in conv3x3_planes.hip /
// planes16.h): A is split into half pairs (144-byte LDS rows: 2 K halves x 2 planes x 32 B + 16), the weight image keeps
// its 208-byte rows with the planes W0 | W - W0 | 2^-11 W0 of the row scaled by 2^e(n), three products per MAC, the
// epilogue multiplies by 2^-e(n); max |A| of what a workgroup staged is folded into the guard slot.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_once.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "igemm_common.h"
#include "planes16.h"
#include "segmif_hip.h"

namespace segmif {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#ifndef GEMM_DBG
#define GEMM_DBG 0  // tuning aid: 1 = s_memtime timeline probe (tools/gemm_timeline.py)
#endif
#if GEMM_DBG
__device__ unsigned long long gemm_timeline[1024][16][8];
#define GEMM_TL(step, slot)                                                                        \
  do {                                                                                            \
    if (tid == 0 && blockIdx.x < 1024 && (step) < 16) gemm_timeline[blockIdx.x][step][slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define GEMM_TL(step, slot)
#endif

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int GPITCH = 208;                   // bytes per LDS row
constexpr int GTILE = GBM * GPITCH;           // 26 624 bytes = 26 DMA instructions of 1 KB
constexpr int GPITCH_H = 144;                 // f16x3: bytes per LDS row of A (9 sixteen-byte slots: odd, conflict-free)

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

struct GemmSplitK {
  const float* a;
  const unsigned char* w;  // packed [n-tile][k-step][128][208]
  const float* bias;
  const float* res;
  const float* prelu;
  float* out;
  long long M;
  int N, K, lda, ldo, ldr, act;
  int ntm, ntn;
  int epi;  // 1: the output leaves through LDS (rows written 256 contiguous bytes at a time), 0: straight from the accumulators
  const float* wscale;  // f16x3: 2^-e(n) per (padded) output column
  uint32_t* amax;       // f16x3: range slots for max |A| (or null): slot row / amax_rows
  long long amax_rows;  // rows per image (M when every row reports to amax[0])
  // patch mode (patch_k > 0): a k x k convolution with stride patch_st and zero padding patch_pad over a dense NHWC image
  // (pixel pitch lda = C, C % GBK == 0): A row (b, oy, ox) is the patch at (st oy - pad, st ox - pad), k = (ky, kx, c); a K step
  // of GBK floats lies inside one tap (ky, kx), whose pixel is either inside the image or contributes zeros
  int patch_k, patch_st, patch_pad, patch_H, patch_W, patch_OH, patch_OW;
};

template <bool F16>
__device__ __forceinline__ f32x16 gemm_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__global__ __launch_bounds__(256) void gemm_split_kernel(const GemmSplitK p) {
  constexpr int APITCH = F16 ? GPITCH_H : GPITCH;   // LDS row of A
  constexpr int AHALF = F16 ? 64 : 96;              // bytes per K half of an A row
  constexpr int NPA = F16 ? 2 : 3, NPROD = F16 ? 3 : 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
  unsigned char* As = smem_g;
  unsigned char* Bs = smem_g + GBM * APITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  {  // XCD-aware remap: an XCD owns a contiguous run of tiles; column tiles of one row block are neighbours (A from L2)
    const int nwg = gridDim.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int mt = bid / p.ntn, nt = bid - mt * p.ntn;
  const long long m0 = (long long)mt * GBM;
  const int n0 = nt * GBN;
  const int nks = p.K / GBK;

  // A staging: unit u = tid + 256 j -> row u >> 3, K quarter kq = u & 7 (floats 4 kq .. 4 kq + 3 of the step)
  const int kq = tid & 7;
  const float* a_ptr[4];
  bool a_ok[4];
  int a_dst[4];
  int a_iy[4] = {0, 0, 0, 0}, a_ix[4] = {0, 0, 0, 0};  // patch mode: image coordinates of the patch's top-left pixel
  bool ra_ok[4] = {true, true, true, true};           // patch mode: the tap of the step held in ra[] lies inside the image
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (tid >> 3) + 32 * j;
    a_ok[j] = m0 + row < p.M;
    const long long m = a_ok[j] ? (m0 + row) : 0;
    if (p.patch_k) {
      const long long ohw = (long long)p.patch_OH * p.patch_OW;
      const long long b = m / ohw;
      const int rem = (int)(m - b * ohw);
      const int oy = rem / p.patch_OW, ox = rem - oy * p.patch_OW;
      a_iy[j] = oy * p.patch_st - p.patch_pad;
      a_ix[j] = ox * p.patch_st - p.patch_pad;
      // (pointer of the patch's top-left pixel: may lie outside the image - only dereferenced for taps that are inside)
      a_ptr[j] = p.a + ((b * p.patch_H + a_iy[j]) * p.patch_W + a_ix[j]) * p.lda + kq * 4;
    } else {
      a_ptr[j] = p.a + m * (long long)p.lda + kq * 4;
    }
    a_dst[j] = row * APITCH + (kq >> 2) * AHALF + (kq & 3) * 8;
  }
  f32x4 ra[4];
  uint32_t amx = 0u;  // f16x3: largest |A| this lane has split (p16::absmax_pk patterns)
  auto gload = [&](int ks) {
    if (p.patch_k) {  // (uniform) K step ks = tap (ky, kx), channels c0 .. c0 + GBK - 1
      const int tap = (ks * GBK) / p.lda, c0 = ks * GBK - tap * p.lda;
      const int ky = tap / p.patch_k, kx = tap - ky * p.patch_k;
      const long long ko = ((long long)ky * p.patch_W + kx) * p.lda + c0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // unconditional load from a clamped address + a mask applied at the LDS store: a load under a divergent branch makes
        // hipcc wait vmcnt(0) at the join, i.e. before the MFMAs it was issued to hide under (profiles/r03_wgrad3x3_phases.txt)
        ra_ok[j] = (unsigned)(a_iy[j] + ky) < (unsigned)p.patch_H && (unsigned)(a_ix[j] + kx) < (unsigned)p.patch_W;
        const float* src = ra_ok[j] ? a_ptr[j] + ko : p.a + kq * 4;
        ra[j] = *reinterpret_cast<const f32x4*>(src);
      }
      return;
    }
    const long long ko = (long long)ks * GBK;
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4*>(a_ptr[j] + ko);
  };
  auto a_store = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 x = (a_ok[j] && ra_ok[j]) ? ra[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (F16) {
        uint32_t ha, la, hb, lb;
        p16::split2(x[0], x[1], ha, la);
        p16::split2(x[2], x[3], hb, lb);
        *reinterpret_cast<u32x2*>(As + a_dst[j]) = u32x2{ha, hb};
        *reinterpret_cast<u32x2*>(As + a_dst[j] + 32) = u32x2{la, lb};
        amx = p16::absmax_pk(p16::absmax_pk(amx, ha, la), hb, lb);
      } else {
        uint32_t p0a, p1a, p2a, p0b, p1b, p2b;
        split3(x[0], x[1], p0a, p1a, p2a);
        split3(x[2], x[3], p0b, p1b, p2b);
        *reinterpret_cast<u32x2*>(As + a_dst[j]) = u32x2{p0a, p0b};
        *reinterpret_cast<u32x2*>(As + a_dst[j] + 32) = u32x2{p1a, p1b};
        *reinterpret_cast<u32x2*>(As + a_dst[j] + 64) = u32x2{p2a, p2b};
      }
    }
  };
  const unsigned char* __restrict__ wt = p.w + (long long)nt * nks * GTILE + lane * 16;
  auto b_dma = [&](int ks) {
    const unsigned char* src = wt + (long long)ks * GTILE;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int i = j * 4 + wave;
      if (i < GTILE / 1024)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(Bs + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const unsigned char* a_lane = As + (wm * 64 + r) * APITCH + h * 16;
  const unsigned char* b_lane = Bs + (wn * 64 + r) * GPITCH + h * 16;
  // products, least significant first (A plane, W plane): bf16x6 a2 w0, a1 w1, a0 w2, a1 w0, a0 w1, a0 w0; f16x3 l W0s, a0 Wl, a0 W0
  constexpr int PA[6] = {F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0, 0};
  constexpr int PW[6] = {F16 ? 2 : 0, 1, F16 ? 0 : 2, 0, 1, 0};

  // Single-buffered on purpose: three workgroups per CU cover each other's staging phases.  (An 8-wave, LDS
  // double-buffered variant with one barrier per step at one workgroup per CU measured 10 % slower over the encoder's
  // shapes: profiles/r02_gemm_split.txt.)
  gload(0);
  a_store();  // before the DMA: beside an LDS-DMA in flight hipcc waits vmcnt(0) at the first use of a plain load's result
  b_dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks = 0; ks < nks; ++ks) {
    const bool more = ks + 1 < nks;
    GEMM_TL(ks, 0);
    if (more) gload(ks + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[2][NPA], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          if (pl < NPA) fa[i][pl] = *reinterpret_cast<const u32x4*>(a_lane + i * 32 * APITCH + s * AHALF + pl * 32);
          fb[i][pl] = *reinterpret_cast<const u32x4*>(b_lane + i * 32 * GPITCH + s * 96 + pl * 32);
        }
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = gemm_mfma<F16>(fb[j][PW[t]], fa[i][PA[t]], acc[i][j]);
    }
    GEMM_TL(ks, 1);
    __syncthreads();  // every wave is done reading this step's tiles
    GEMM_TL(ks, 2);
    if (more) {
      a_store();
      GEMM_TL(ks, 3);
      b_dma(ks + 1);
      GEMM_TL(ks, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GEMM_TL(ks, 5);
    __syncthreads();
    GEMM_TL(ks, 6);
  }
  GEMM_TL(15, 7);
  if constexpr (F16) {
    if (p.amax) {  // the tile's rows m0 .. m0 + 127 may straddle images: it reports to each (conservative)
      const long long m1 = m0 + GBM - 1 < p.M ? m0 + GBM - 1 : p.M - 1;
      p16::fold_pat(p.amax, (int)(m0 / p.amax_rows), (int)(m1 / p.amax_rows), amx);
    }
  }

  // ---- epilogue: bias + activation (+ residual).  The products run transposed (weights as the MFMA's row operand), so a
  // lane owns one output row and its registers 4g .. 4g+3 are four consecutive columns: 16-byte stores, a quarter of the
  // vector-memory instructions of a row-per-register layout.  (tools/gemm_timeline.py: the epilogue is 20 000+ cycles of a
  // K = 320 tile's 95 000 - every workgroup of a round stores at the same time and the burst drains at HBM write rate.)
  // (r6) one uniform branch on "the activation is a GELU" in front of the epilogue: inside the per-element switch every accumulator
  // element carried its own copy of the erf GELU (see igemm.hip) - tens of KB of ISA for an activation these kernels' callers never ask for
  auto epilogue_r6 = [&](auto gelu_c) {
    constexpr bool GELU_EPI = decltype(gelu_c)::value;
    const float slope = (p.act == SEGMIF_ACT_PRELU) ? *p.prelu : 0.f;
    if (p.epi) {
      // Through LDS: a lane owns an output ROW of the accumulators, so a direct store instruction touches 32 rows x 32 bytes;
      // staged in a wave-private [32][68] tile (the K loop's buffers are free after its last barrier) the same data leaves as
      // 4 rows x 256 contiguous bytes per instruction, and the residual is read the same way.
      float* T = reinterpret_cast<float*>(smem_g) + wave * (32 * 68);
  #pragma unroll
      for (int i = 0; i < 2; ++i) {
  #pragma unroll
        for (int j = 0; j < 2; ++j)
  #pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cl = j * 32 + 8 * g + 4 * h;
            const int n = n0 + wn * 64 + cl;
            f32x4 y = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            if constexpr (F16) y *= *reinterpret_cast<const f32x4*>(p.wscale + n);  // (the scale array covers the padded columns)
            if (p.bias && n < p.N) y += *reinterpret_cast<const f32x4*>(p.bias + n);
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (GELU_EPI) { y[e] = gelu_exact(y[e]); } else if (p.act == SEGMIF_ACT_RELU) y[e] = fmaxf(y[e], 0.f);
              else if (p.act == SEGMIF_ACT_PRELU) y[e] = y[e] >= 0.f ? y[e] : slope * y[e];
            }
            *reinterpret_cast<f32x4*>(T + r * 68 + cl) = y;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: the wave's own LDS writes have landed
  #pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int row = t * 4 + (lane >> 4), cl = (lane & 15) * 4;
          const long long m = m0 + wm * 64 + i * 32 + row;
          const int n = n0 + wn * 64 + cl;
          f32x4 y = *reinterpret_cast<const f32x4*>(T + row * 68 + cl);
          if (m < p.M && n < p.N) {
            if (p.res) y += *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + n);
            *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = y;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next half overwrites the tile
      }
      return;
    }
  #pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long m = m0 + wm * 64 + i * 32 + r;
      if (m >= p.M) continue;
  #pragma unroll
      for (int j = 0; j < 2; ++j)
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * h;
          if (n >= p.N) continue;  // N % 4 == 0: a group of four is inside or outside as a whole
          f32x4 y = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if constexpr (F16) y *= *reinterpret_cast<const f32x4*>(p.wscale + n);
          if (p.bias) y += *reinterpret_cast<const f32x4*>(p.bias + n);
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (GELU_EPI) { y[e] = gelu_exact(y[e]); } else if (p.act == SEGMIF_ACT_RELU) y[e] = fmaxf(y[e], 0.f);
            else if (p.act == SEGMIF_ACT_PRELU) y[e] = y[e] >= 0.f ? y[e] : slope * y[e];
          }
          if (p.res) y += *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + n);
          *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = y;
        }
    }
    GEMM_TL(14, 7);
  };
  if (p.act == SEGMIF_ACT_GELU) epilogue_r6(std::true_type{});
  else epilogue_r6(std::false_type{});
}

// fp32 [N][ldw] -> [n-tile][k-step][128 rows][K half][plane][16] bf16 with 208-byte rows, zero filled past N / K
__global__ void gemm_split_pack_kernel(const float* __restrict__ w, int N, int K, int ldw, int nks, long long total,
                                       uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kk = (int)(idx & 31);
  long long t = idx >> 5;
  const int row = (int)(t & 127); t >>= 7;
  const int ks = (int)(t % nks);
  const int nt = (int)(t / nks);
  const int n = nt * GBN + row, k = ks * GBK + kk;
  const float x = (n < N && k < K) ? w[(long long)n * ldw + k] : 0.f;
  uint32_t p0, p1, p2;
  split3(x, 0.f, p0, p1, p2);
  uint16_t* dst = out + (((long long)nt * nks + ks) * GBM + row) * (GPITCH / 2) + (kk >> 4) * 48 + (kk & 15);
  dst[0] = (uint16_t)(p0 & 0xffffu);
  dst[16] = (uint16_t)(p1 & 0xffffu);
  dst[32] = (uint16_t)(p2 & 0xffffu);
  if (kk < 8) out[(((long long)nt * nks + ks) * GBM + row) * (GPITCH / 2) + 96 + kk] = 0;  // the 16 padding bytes
}

// f16x3 weights: row scale 2^-e(n) (2^14 <= 2^e max|w[n][.]| < 2^15; 1 for vanishing rows and for the padding rows)
__global__ void gemm_split16_scale_kernel(const float* __restrict__ w, int N, int K, int ldw, float* __restrict__ inv_scale) {
  const int n = blockIdx.x;  // grid = padded N
  float mx = 0.f;
  if (n < N)
    for (int k = threadIdx.x; k < K; k += 64) mx = fmaxf(mx, fabsf(w[(long long)n * ldw + k]));
  mx = p16::wave_max(mx);
  if (threadIdx.x == 0) {
    int e = 0;
    if (mx >= 1e-30f && mx <= 3e38f) e = 14 - (int)((__float_as_uint(mx) >> 23) - 127);
    inv_scale[n] = ldexpf(1.f, -e);
  }
}

// same image as gemm_split_pack_kernel with the planes W0 | W - W0 | 2^-11 W0 of the scaled row (halves)
__global__ void gemm_split16_pack_kernel(const float* __restrict__ w, int N, int K, int ldw, int nks, long long total,
                                         const float* __restrict__ inv_scale, uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kk = (int)(idx & 31);
  long long t = idx >> 5;
  const int row = (int)(t & 127); t >>= 7;
  const int ks = (int)(t % nks);
  const int nt = (int)(t / nks);
  const int n = nt * GBN + row, k = ks * GBK + kk;
  const float x = (n < N && k < K) ? w[(long long)n * ldw + k] * (1.f / inv_scale[n]) : 0.f;  // exact: power of two
  const _Float16 w0 = (_Float16)x;
  const _Float16 wl = (_Float16)(x - (float)w0);
  const _Float16 ws = (_Float16)((float)w0 * (1.f / p16::LSCALE));
  uint16_t* dst = out + (((long long)nt * nks + ks) * GBM + row) * (GPITCH / 2) + (kk >> 4) * 48 + (kk & 15);
  dst[0] = __builtin_bit_cast(uint16_t, w0);
  dst[16] = __builtin_bit_cast(uint16_t, wl);
  dst[32] = __builtin_bit_cast(uint16_t, ws);
  if (kk < 8) out[(((long long)nt * nks + ks) * GBM + row) * (GPITCH / 2) + 96 + kk] = 0;  // the 16 padding bytes
}

}  // namespace
}  // namespace segmif

using namespace segmif;

#if GEMM_DBG
extern "C" int segmif_debug_gemm_timeline(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gemm_timeline), bytes < sizeof(gemm_timeline) ? bytes : sizeof(gemm_timeline));
}
#endif

extern "C" int64_t segmif_gemm_split_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % GBK) return 0;
  return (int64_t)((N + GBN - 1) / GBN) * (K / GBK) * GTILE;
}

extern "C" int segmif_gemm_split_pack(const float* w, int N, int K, int ldw, void* out, void* stream) {
  if (!w || !out || segmif_gemm_split_weight_bytes(N, K) == 0 || ldw < K) return SEGMIF_EINVAL;
  const int nks = K / GBK;
  const long long total = (long long)((N + GBN - 1) / GBN) * nks * GBM * GBK;
  hipLaunchKernelGGL(gemm_split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K,
                     ldw, nks, total, (uint16_t*)out);
  return (int)hipGetLastError();
}

extern "C" int64_t segmif_gemm_split16_weight_bytes(int N, int K) {
  const int64_t image = segmif_gemm_split_weight_bytes(N, K);  // + one float per padded output column: 2^-e(n)
  return image ? image + (int64_t)((N + GBN - 1) / GBN) * GBN * 4 : 0;
}

extern "C" int segmif_gemm_split16_pack(const float* w, int N, int K, int ldw, void* out, void* stream) {
  if (!w || !out || segmif_gemm_split_weight_bytes(N, K) == 0 || ldw < K) return SEGMIF_EINVAL;
  const int nks = K / GBK, npad = (N + GBN - 1) / GBN * GBN;
  const long long total = (long long)(npad / GBN) * nks * GBM * GBK;
  float* inv_scale = reinterpret_cast<float*>((unsigned char*)out + segmif_gemm_split_weight_bytes(N, K));
  hipLaunchKernelGGL(gemm_split16_scale_kernel, dim3((unsigned)npad), dim3(64), 0, (hipStream_t)stream, w, N, K, ldw, inv_scale);
  hipLaunchKernelGGL(gemm_split16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K,
                     ldw, nks, total, inv_scale, (uint16_t*)out);
  return (int)hipGetLastError();
}

static int gemm_split_impl(const SegmifGemmSplit* d, bool f16, uint32_t* amax, int amax_images, void* stream);

extern "C" int segmif_gemm_split_f32(const SegmifGemmSplit* d, void* stream) { return gemm_split_impl(d, false, nullptr, 1, stream); }

extern "C" int segmif_gemm_split16_f32(const SegmifGemmSplit* d, uint32_t* amax, int amax_images, void* stream) {
  return gemm_split_impl(d, true, amax, amax_images, stream);
}

static int gemm_split_impl(const SegmifGemmSplit* d, bool f16, uint32_t* amax, int amax_images, void* stream) {
  if (amax && (amax_images < 1 || !d || d->M % amax_images)) return SEGMIF_EINVAL;  // whole images: M = images x rows per image
  if (!d || !d->a || !d->w || !d->out || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->K % GBK) return SEGMIF_EINVAL;
  const bool patch = d->patch_k > 0;
  if ((!patch && d->lda < d->K) || (d->lda & 3) || ((uintptr_t)d->a & 15) || ((uintptr_t)d->w & 15)) return SEGMIF_EINVAL;
  int poh = 0, pow_ = 0;
  if (patch) {  // K = k * k * C, C = lda a multiple of GBK (a K step never straddles two taps)
    const int kk = d->patch_k, st = d->patch_st, pad = d->patch_pad;
    if (st < 1 || pad < 0 || pad >= kk || d->patch_H + 2 * pad < kk || d->patch_W + 2 * pad < kk || d->K != kk * kk * d->lda || d->lda % GBK)
      return SEGMIF_EINVAL;
    poh = (d->patch_H + 2 * pad - kk) / st + 1;
    pow_ = (d->patch_W + 2 * pad - kk) / st + 1;
    if (d->M % ((long long)poh * pow_)) return SEGMIF_EINVAL;
  }
  if (d->act == SEGMIF_ACT_PRELU && !d->prelu) return SEGMIF_EINVAL;
  if (d->res && (d->ldr <= 0 || (d->ldr & 3) || ((uintptr_t)d->res & 15))) return SEGMIF_EINVAL;
  if ((d->N & 3) || (d->ldo & 3) || ((uintptr_t)d->out & 15) || ((uintptr_t)d->bias & 15)) return SEGMIF_EINVAL;  // float4 epilogue
  GemmSplitK k;
  k.a = d->a; k.w = (const unsigned char*)d->w; k.bias = d->bias; k.res = d->res; k.prelu = d->prelu; k.out = d->out;
  k.M = d->M; k.N = d->N; k.K = d->K; k.lda = d->lda; k.ldo = d->ldo; k.ldr = d->ldr; k.act = d->act;
  static const bool epi_direct = [] { const char* e = getenv("SEGMIF_GEMM_EPI"); return e && !strcmp(e, "direct"); }();
  k.epi = epi_direct ? 0 : 1;  // "direct" (read once per process): stores straight from the accumulators (round 2)
  k.ntm = (int)((d->M + GBM - 1) / GBM);
  k.ntn = (d->N + GBN - 1) / GBN;
  k.patch_k = patch ? d->patch_k : 0; k.patch_st = d->patch_st; k.patch_pad = d->patch_pad;
  k.patch_H = d->patch_H; k.patch_W = d->patch_W; k.patch_OH = poh; k.patch_OW = pow_;
  k.amax = f16 ? amax : nullptr;
  k.amax_rows = d->M / (amax && amax_images > 1 ? amax_images : 1);
  k.wscale = reinterpret_cast<const float*>(k.w + segmif_gemm_split_weight_bytes(d->N, d->K));  // (f16x3 images only)
  constexpr size_t smem = 2 * (size_t)GTILE, smem16 = (size_t)GBM * GPITCH_H + GTILE;
  static_assert(smem16 >= 4 * 32 * 68 * sizeof(float), "the LDS epilogue tile must fit");
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const dim3 grid((unsigned)((long long)k.ntm * k.ntn));
  if (f16) hipLaunchKernelGGL(gemm_split_kernel<true>, grid, dim3(256), smem16, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(gemm_split_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, k);
  return (int)hipGetLastError();
}
