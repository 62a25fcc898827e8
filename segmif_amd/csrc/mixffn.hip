// Mix-FFN of a MiT block in ONE kernel for the wide-image stages (C = 64 | 128: stages 1-2 of mit_b1 .. b5):
//     out = x + fc2( GELU( dwconv3x3( fc1( LayerNorm(x) ) ) ) )        core/mix_transformer.py:46-53 (Mlp), :376-387 (DWConv),
//                                                                       :152-155 (Block: norm2, mlp, residual)
// Round 3 ran this as LayerNorm -> GEMM -> dwconv+GELU -> GEMM: the 4C-wide hidden tensor crossed HBM four times (315 MB
// in, 1.26 GB out, 1.26 GB in / out, 1.26 GB in per 64 images at stage 1) and the three kernels sat at 3.2 - 4.4 TB/s:
// 1.65 ms per block.  Here a workgroup owns a 12 x 16 pixel tile and the hidden tensor never leaves the CU:
//
//   A      the tile's 14 x 18 halo of tokens (252 -> 8 row blocks of 32) is loaded ONCE, normalised (LayerNorm in registers:
//          a token's row is split over the two half-waves, one cross-half shuffle per statistic) and split into half pairs
//          straight into MFMA operand registers (f16x3 format, planes16.h) - no LDS for A;
//   per chunk of 32 hidden channels (4C / 32 chunks):
//   P1     fc1 on the matrix pipe (the chunk's weights + constants arrive as ONE LDS-DMA image, double-buffered and issued a
//          whole chunk ahead; 3 products per MAC), + bias, zero outside the image (the dwconv's
//          zero padding applies to the HIDDEN image) -> Hs[256 tokens][32] fp32 in LDS
//   P2     depthwise 3x3 + bias + GELU (erf form, A&S 7.1.26: fp32-level) on the vector ALU (a thread slides a 3 x 3 window of float4s down a column),
//          result split into half pairs -> Gs in the B-operand layout of fc2 (its own LDS at C = 64; aliasing Hs at C = 128)
//   P3     fc2 partial product over these 32 hidden channels into the output accumulators (registers)
//   end    out = x + acc * 2^-e(n) + b2.
// HBM traffic: x once (1.31x with the halo, mostly L2 hits), out once.  One workgroup of 8 waves per CU (72 KB of LDS at
// C = 64, 111 KB at C = 128; ~200 registers per lane: the halo's operand fragments, the output accumulators and the
// depthwise window live side by side).
//
// f16x3 only (half pairs, range slots for the two tensors the kernel splits: LN(x) and the GELU output); outside a guarded
// scope, or for a pair the guard sends back, the host runs the round-3 chain on bf16x6.
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>

#include "igemm_common.h"
#include "planes16.h"
#include "segmif_hip.h"

#ifndef MF_DBG
#define MF_DBG 0  // tuning aid: 1 = s_memtime timeline probe (tools/mixffn_timeline.py; build with HIPCC_EXTRA=-DMF_DBG=1)
#endif

namespace segmif {
namespace {

#if MF_DBG
__device__ unsigned long long mixffn_timeline[256][16][8];
#define MF_TL(chunk, slot)                                                                                       \
  do {                                                                                                           \
    if (tid == MF_DBG_TID && blockIdx.x < 256 && (chunk) < 16) mixffn_timeline[blockIdx.x][chunk][slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#ifndef MF_DBG_TID
#define MF_DBG_TID 0
#endif
#else
#define MF_TL(chunk, slot)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int MF_TH = 12, MF_TW = 16;          // output tile (pixels)
constexpr int MF_HH = MF_TH + 2, MF_HW = MF_TW + 2;
constexpr int MF_NTOK = MF_HH * MF_HW;         // 252 halo tokens -> 8 row blocks
constexpr int MF_NPX = MF_TH * MF_TW;          // 192 output pixels -> 6 row blocks
constexpr int MF_HS_BYTES = 256 * 128;         // Hs: [256 tokens][32 ch] fp32, 16-byte slots swizzled by (token & 7)
constexpr int MF_GP = 144;                     // Gs: bytes per pixel (32 hi halves | 32 lo halves | 16 pad); 9 slots: odd

template <int C>
struct MfGeom {
  static constexpr int HID = 4 * C, NCH = HID / 32, KS = C / 16, CB = C / 32;
  static constexpr int P1 = 6 * C + 16;        // fc1 row pitch: 3 planes x C halves + 16 (an odd number of 16-byte slots)
  static constexpr int W1B = 32 * P1;          // one chunk of fc1 weights: 32 hidden rows
  static constexpr int P2 = 208;               // fc2 row pitch: 3 planes x 32 halves + 16
  static constexpr int W2B = C * P2;           // one chunk of fc2 weights: C output rows x 32 hidden columns
  // a chunk's image = everything the kernel needs for 32 hidden channels, one contiguous DMA: fc1 rows | fc2 rows |
  // floats: depthwise taps [9][32], depthwise bias [32], fc1 bias [32], fc1 row scale 2^-e [32]
  static constexpr int O_F = W1B + W2B, F_DWB = 9 * 32, F_B1 = 10 * 32, F_S1 = 11 * 32;
  static constexpr int CHB = O_F + 12 * 32 * 4;
  static_assert(CHB % 1024 == 0, "a chunk image is a whole number of 1 KB DMA instructions");
  static constexpr bool GS_SEPARATE = (C == 64);  // the fc2 operand tile has LDS of its own (one barrier fewer per chunk)
  static constexpr int GS_BYTES = MF_NPX * MF_GP;
  // constants (floats): gamma[C] beta[C] b2[C] s2[C]
  static constexpr int O_GAMMA = 0, O_BETA = C, O_B2 = 2 * C, O_S2 = 3 * C, NCONST = 4 * C;
  static constexpr int SMEM = MF_HS_BYTES + (GS_SEPARATE ? GS_BYTES : 0) + 2 * CHB + NCONST * 4;
};

struct MixFfnK {
  const float* x;             // (B, H*W, C) tokens, dense
  float* out;                 // (B, H*W, C), must not alias x (halo tokens are read by neighbouring workgroups)
  const unsigned char* wimg;  // [NCH][CHB] chunk images | s2[C]   (segmif_mixffn_pack)
  const float* gamma; const float* beta; float eps;
  const float* b2;
  int B, H, W, tiles_x, tiles_y;
  uint32_t* amax_a;           // range slots of LN(x) and of the GELU output (or null); index = image when amax_images > 1
  uint32_t* amax_g;
  int amax_images;
};

// GELU(x) = x Phi(x) on ~16 vector-ALU operations (the libm erff of the stand-alone dwconv kernel costs ~60; this kernel is bound
// by its vector ALU phase).  Phi through erfc(z), z = |x| / sqrt 2, in the rational-times-Gaussian form of Abramowitz & Stegun
// 7.1.26 (|error of erf| <= 1.5e-7, i.e. <= 7.5e-8 |x| on GELU: about one fp32 ulp of x):
//     erfc(z) ~ (a1 t + .. + a5 t^5) exp(-z^2),  t = 1 / (1 + p z);   Phi = x >= 0 ? 1 - erfc / 2 : erfc / 2.
__device__ __forceinline__ float gelu_fast(float x) {
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

__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (r6) SADDR form (csrc/conv3x3_planes.hip, dma16s): wave-uniform base in SGPRs + 32-bit lane offset
__device__ __forceinline__ void mf_dma16s(const unsigned char* sbase, uint32_t voff, unsigned char* lds_wave_base) {
  const uint32_t m = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ void mf_dma16(const unsigned char* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int C, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void mixffn_kernel(const MixFfnK p) {
  using G = MfGeom<C>;
  constexpr int T = 64 * NW;
  constexpr int RBW = 8 / NW;                 // halo row blocks per wave in fc1 (1 at 8 waves)
  constexpr int RPT = MF_TH / (T / 128);      // output rows per thread in the depthwise phase (3 at 8 waves)
  constexpr int HALF = C / 2;                 // channels of a token held by one half-wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];
  unsigned char* Hs = smem_m;
  unsigned char* Gs = G::GS_SEPARATE ? smem_m + MF_HS_BYTES : smem_m;  // C = 128: aliases Hs (LDS budget)
  unsigned char* Wb = smem_m + MF_HS_BYTES + (G::GS_SEPARATE ? G::GS_BYTES : 0);  // [2][CHB] chunk images
  float* Cst = reinterpret_cast<float*>(Wb + 2 * G::CHB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  {  // XCD-aware remap: an XCD owns a contiguous run of tiles (neighbouring tiles share their halo rows in one L2)
    const int nwg = gridDim.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int tx = bid % p.tiles_x, ty = (bid / p.tiles_x) % p.tiles_y, b = bid / (p.tiles_x * p.tiles_y);
  const int x0 = tx * MF_TW, y0 = ty * MF_TH;
  const long long img = (long long)b * p.H * p.W;

  // chunk j's image -> Wb[j & 1] by LDS-DMA, issued a whole chunk ahead (right after the barrier that opens chunk j - 1: by
  // then every wave is done with chunk j - 2, the buffer's previous tenant)
  auto dma = [&](int j) {
    const unsigned char* src = p.wimg + (long long)j * G::CHB;  // (uniform: SADDR form, the lane offset is a constant)
    unsigned char* dst = Wb + (j & 1) * G::CHB;
    for (int i = wave; i < G::CHB / 1024; i += NW) mf_dma16s(src + i * 1024, (uint32_t)(lane * 16), dst + i * 1024);
  };
  dma(0);

  // ---- constants -> LDS ------------------------------------------------------------------------------------------------
  {
    const float* s2g = reinterpret_cast<const float*>(p.wimg + (long long)G::NCH * G::CHB);
    for (int i = tid; i < G::NCONST; i += T) {
      float v;
      if (i < G::O_BETA) v = p.gamma[i];
      else if (i < G::O_B2) v = p.beta[i - G::O_BETA];
      else if (i < G::O_S2) v = p.b2[i - G::O_B2];
      else v = s2g[i - G::O_S2];
      Cst[i] = v;
    }
  }
  __syncthreads();

  // ---- A: this wave's halo row blocks, LayerNorm, split -> operand registers ------------------------------------------
  // MFMA contraction slot (step s, half h, element e) <-> channel HALF h + 8 s + e (the fc1 weight image is packed the same
  // way), so a lane reads one contiguous run of HALF floats of its token.
  u32x4 a_hi[RBW][G::KS], a_lo[RBW][G::KS];
  bool tok_in[RBW];
  int tok_id[RBW];
  uint32_t amx_a = 0u, amx_g = 0u;
#pragma unroll
  for (int i = 0; i < RBW; ++i) {
    const int tok = (wave + i * NW) * 32 + r;
    tok_id[i] = tok;
    const int hy = tok / MF_HW, hx = tok - hy * MF_HW;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool in = tok < MF_NTOK && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    tok_in[i] = in;
    const float* src = p.x + (img + (long long)(in ? gy : 0) * p.W + (in ? gx : 0)) * C + HALF * h;
    f32x4 v[HALF / 4];
#pragma unroll
    for (int k = 0; k < HALF / 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + 4 * k);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < HALF / 4; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < HALF / 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[k][e] -= mean;
        q += v[k][e] * v[k][e];
      }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / C) + p.eps);
#pragma unroll
    for (int s8 = 0; s8 < G::KS; ++s8) {
      float y[8];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(Cst + G::O_GAMMA + HALF * h + 8 * s8 + 4 * k);
        const f32x4 be = *reinterpret_cast<const f32x4*>(Cst + G::O_BETA + HALF * h + 8 * s8 + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[4 * k + e] = in ? v[2 * s8 + k][e] * rstd * ga[e] + be[e] : 0.f;
      }
      p16::split8(y, a_hi[i][s8], a_lo[i][s8]);
      amx_a = p16::absmax_pk4(amx_a, a_hi[i][s8], a_lo[i][s8]);
    }
  }

  // output jobs of this wave in fc2: column block cb (32 output channels), row blocks rb0 + i RSTEP (i < NJ, those < 6)
  constexpr int NJ = (6 * G::CB + NW - 1) / NW, RSTEP = NW / G::CB;
  const int cb = wave % G::CB, rb0 = wave / G::CB;
  f32x16 acc2[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc2[i][v] = 0.f;

  // depthwise phase geometry of this thread: channel quad q4, column xc, rows yb .. yb + RPT - 1 of the output tile
  const int q4 = tid & 7, xc = (tid >> 3) & 15, yb = (tid >> 7) * RPT;

  for (int j = 0; j < G::NCH; ++j) {
    const unsigned char* Wc = Wb + (j & 1) * G::CHB;
    const float* Fc = reinterpret_cast<const float*>(Wc + G::O_F);
    MF_TL(j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk's image has landed (issued a whole chunk ago)
    MF_TL(j, 1);
    __syncthreads();                                     // ... for every wave; and every wave is done with the previous chunk
    MF_TL(j, 2);
    if (j + 1 < G::NCH) dma(j + 1);

    // ---- P1: fc1 for 32 hidden channels ---------------------------------------------------------------------------------
    {
      const unsigned char* w1_lane = Wc + r * G::P1 + h * 16;
      f32x16 acc1[RBW];
#pragma unroll
      for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1[i][v] = 0.f;
      u32x4 w[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[0][pl] = *reinterpret_cast<const u32x4*>(w1_lane + pl * (2 * C));
#pragma unroll
      for (int s = 0; s < G::KS; ++s) {
        if (s + 1 < G::KS) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) w[(s + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(w1_lane + pl * (2 * C) + (s + 1) * 32);
        }
#pragma unroll
        for (int i = 0; i < RBW; ++i) {  // products least significant first: l W0s, x0 Wl, x0 W0
          acc1[i] = mfma16(w[s & 1][2], a_lo[i][s], acc1[i]);
          acc1[i] = mfma16(w[s & 1][1], a_hi[i][s], acc1[i]);
          acc1[i] = mfma16(w[s & 1][0], a_hi[i][s], acc1[i]);
        }
      }
      // + bias (x row scale), zero outside the image, -> Hs
      f32x4 bb[4], ss[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bb[g] = *reinterpret_cast<const f32x4*>(Fc + G::F_B1 + 8 * g + 4 * h);
        ss[g] = *reinterpret_cast<const f32x4*>(Fc + G::F_S1 + 8 * g + 4 * h);
      }
#pragma unroll
      for (int i = 0; i < RBW; ++i) {
        const int tok = tok_id[i];
        unsigned char* row = Hs + tok * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = tok_in[i] ? fmaf(acc1[i][4 * g + e], ss[g][e], bb[g][e]) : 0.f;
          *reinterpret_cast<f32x4*>(row + (((2 * g + h) ^ (tok & 7)) * 16)) = y;
        }
      }
    }
    MF_TL(j, 3);
    __syncthreads();  // Hs complete (and, Gs separate: every wave is past the previous chunk's P3)
    MF_TL(j, 4);

    // ---- P2: depthwise 3x3 + bias + GELU over the output tile, 4 channels x RPT rows per thread -----------------------------
    uint32_t g_hi[RPT][2], g_lo[RPT][2];
    {
      f32x4 wt[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(Fc + t * 32 + 4 * q4);
      const f32x4 bias = *reinterpret_cast<const f32x4*>(Fc + G::F_DWB + 4 * q4);
      f32x4 win[3][3];
      auto ld_row = [&](int hy, f32x4* dst) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int tok = hy * MF_HW + xc + dx;
          dst[dx] = *reinterpret_cast<const f32x4*>(Hs + tok * 128 + ((q4 ^ (tok & 7)) * 16));
        }
      };
      ld_row(yb, win[0]);
      ld_row(yb + 1, win[1]);
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        ld_row(yb + i + 2, win[2]);
        f32x4 a = bias;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaf(win[dy][dx][e], wt[dy * 3 + dx][e], a[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = gelu_fast(a[e]);
        p16::split2(a[0], a[1], g_hi[i][0], g_lo[i][0]);
        p16::split2(a[2], a[3], g_hi[i][1], g_lo[i][1]);
        const bool ok = y0 + yb + i < p.H && x0 + xc < p.W;
        const uint32_t m = p16::absmax_pk(p16::absmax_pk(amx_g, g_hi[i][0], g_lo[i][0]), g_hi[i][1], g_lo[i][1]);
        amx_g = ok ? m : amx_g;
        if constexpr (G::GS_SEPARATE) {
          unsigned char* px = Gs + ((yb + i) * MF_TW + xc) * MF_GP + q4 * 8;
          *reinterpret_cast<u32x2*>(px) = u32x2{g_hi[i][0], g_hi[i][1]};
          *reinterpret_cast<u32x2*>(px + 64) = u32x2{g_lo[i][0], g_lo[i][1]};
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          win[0][dx] = win[1][dx];
          win[1][dx] = win[2][dx];
        }
      }
    }
    MF_TL(j, 5);
    __syncthreads();  // Gs complete (separate) / every thread has read its Hs window (aliased)
    MF_TL(j, 6);
    if constexpr (!G::GS_SEPARATE) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        unsigned char* px = Gs + ((yb + i) * MF_TW + xc) * MF_GP + q4 * 8;
        *reinterpret_cast<u32x2*>(px) = u32x2{g_hi[i][0], g_hi[i][1]};
        *reinterpret_cast<u32x2*>(px + 64) = u32x2{g_lo[i][0], g_lo[i][1]};
      }
      __syncthreads();  // Gs complete
    }

    // ---- P3: fc2 partial product over these 32 hidden channels ----------------------------------------------------------
    {
      const unsigned char* w2_lane = Wc + G::W1B + (32 * cb + r) * G::P2 + h * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 w[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const u32x4*>(w2_lane + pl * 64 + s * 32);
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
          if (rb0 + RSTEP * i < 6) {  // (wave-uniform)
            const unsigned char* px = Gs + ((rb0 + RSTEP * i) * 32 + r) * MF_GP + (2 * s + h) * 16;
            const u32x4 ghi = *reinterpret_cast<const u32x4*>(px);
            const u32x4 glo = *reinterpret_cast<const u32x4*>(px + 64);
            acc2[i] = mfma16(w[2], glo, acc2[i]);
            acc2[i] = mfma16(w[1], ghi, acc2[i]);
            acc2[i] = mfma16(w[0], ghi, acc2[i]);
          }
        }
      }
    }
  }

  MF_TL(G::NCH - 1, 7);
  // ---- epilogue: out = x + acc * 2^-e(n) + b2 ---------------------------------------------------------------------------
  {
    f32x4 bb[4], ss[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bb[g] = *reinterpret_cast<const f32x4*>(Cst + G::O_B2 + 32 * cb + 8 * g + 4 * h);
      ss[g] = *reinterpret_cast<const f32x4*>(Cst + G::O_S2 + 32 * cb + 8 * g + 4 * h);
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int px = (rb0 + RSTEP * i) * 32 + r;
      const int gy = y0 + px / MF_TW, gx = x0 + (px % MF_TW);
      if (rb0 + RSTEP * i < 6 && gy < p.H && gx < p.W) {
        const long long m = (img + (long long)gy * p.W + gx) * C + 32 * cb + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 xr = *reinterpret_cast<const f32x4*>(p.x + m + 8 * g);
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = xr[e] + fmaf(acc2[i][4 * g + e], ss[g][e], bb[g][e]);
          *reinterpret_cast<f32x4*>(p.out + m + 8 * g) = y;
        }
      }
    }
  }
  const int slot = p.amax_images > 1 ? b : 0;
  if (p.amax_a) p16::fold_pat(p.amax_a, slot, slot, amx_a);
  if (p.amax_g) p16::fold_pat(p.amax_g, slot, slot, amx_g);
}

// row scale 2^-e(n) of an (N, K) weight: 2^14 <= 2^e max |w[n][.]| < 2^15 (1 for a vanishing row) -> dst[n * stride_hi + ...]
__global__ void mixffn_scale_kernel(const float* __restrict__ w, int K, float* __restrict__ dst, int group, long long group_stride) {
  const int n = blockIdx.x;  // written to dst[(n / group) * group_stride + n % group]
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 64) mx = fmaxf(mx, fabsf(w[(long long)n * K + k]));
  mx = p16::wave_max(mx);
  if (threadIdx.x == 0) {
    int e = 0;
    if (mx >= 1e-30f && mx <= 3e38f) e = 14 - (int)((__float_as_uint(mx) >> 23) - 127);
    dst[(long long)(n / group) * group_stride + n % group] = ldexpf(1.f, -e);
  }
}

__device__ __forceinline__ void mf_put3(unsigned char* row, int plane_bytes, int pos, float x) {
  const _Float16 w0 = (_Float16)x;
  const _Float16 wl = (_Float16)(x - (float)w0);
  const _Float16 ws = (_Float16)((float)w0 * (1.f / p16::LSCALE));
  reinterpret_cast<_Float16*>(row)[pos] = w0;
  reinterpret_cast<_Float16*>(row + plane_bytes)[pos] = wl;
  reinterpret_cast<_Float16*>(row + 2 * plane_bytes)[pos] = ws;
}

// fc1 (HID, C) -> chunk (n >> 5), row n & 31: planes W0 | Wl | W0s of the scaled row, position (s, h, e) <- channel C/2 h + 8 s + e
template <int C>
__global__ void mixffn_pack1_kernel(const float* __restrict__ w1, unsigned char* __restrict__ out) {
  using G = MfGeom<C>;
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one thread per (n, position)
  if (idx >= G::HID * C) return;
  const int n = idx / C, pos = idx - n * C;
  const int s = pos >> 4, hh = (pos >> 3) & 1, e = pos & 7;
  const int c = (C / 2) * hh + 8 * s + e;
  unsigned char* chunk = out + (long long)(n >> 5) * G::CHB;
  const float inv = reinterpret_cast<const float*>(chunk + G::O_F)[G::F_S1 + (n & 31)];
  const float x = w1[(long long)n * C + c] * (1.f / inv);  // exact: power of two
  unsigned char* row = chunk + (n & 31) * G::P1;
  mf_put3(row, 2 * C, pos, x);
  if (pos < 8) reinterpret_cast<uint16_t*>(row + 6 * C)[pos] = 0;  // the 16 padding bytes
}

// fc2 (C, HID) -> chunk j = k >> 5, row n: position (s, h, e) <- hidden channel 32 j + 16 s + 8 h + e
template <int C>
__global__ void mixffn_pack2_kernel(const float* __restrict__ w2, const float* __restrict__ s2, unsigned char* __restrict__ out) {
  using G = MfGeom<C>;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= C * G::HID) return;
  const int n = idx / G::HID, k = idx - n * G::HID;
  const int j = k >> 5, pos = k & 31;
  const float x = w2[(long long)n * G::HID + k] * (1.f / s2[n]);
  unsigned char* row = out + (long long)j * G::CHB + G::W1B + n * G::P2;
  mf_put3(row, 64, pos, x);
  if (pos < 8) reinterpret_cast<uint16_t*>(row + 192)[pos] = 0;
}

// per-chunk floats: depthwise taps [9][32] (from the [9][HID] packing), depthwise bias, fc1 bias
template <int C>
__global__ void mixffn_pack_consts_kernel(const float* __restrict__ dw9, const float* __restrict__ dwb, const float* __restrict__ b1,
                                          unsigned char* __restrict__ out) {
  using G = MfGeom<C>;
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (chunk, slot 0..10, lane 0..31)
  if (idx >= G::NCH * 11 * 32) return;
  const int c = idx & 31, slot = (idx >> 5) % 11, j = idx / (11 * 32);
  const int n = 32 * j + c;
  const float v = slot < 9 ? dw9[(long long)slot * G::HID + n] : slot == 9 ? dwb[n] : b1[n];
  reinterpret_cast<float*>(out + (long long)j * G::CHB + G::O_F)[slot * 32 + c] = v;
}

template <int C>
int64_t image_bytes() {
  using G = MfGeom<C>;
  return (int64_t)G::NCH * G::CHB + (int64_t)C * 4;
}

template <int C>
int pack(const float* w1, const float* b1, const float* dw9, const float* dwb, const float* w2, void* out, hipStream_t s) {
  using G = MfGeom<C>;
  unsigned char* o = (unsigned char*)out;
  float* s2 = reinterpret_cast<float*>(o + (long long)G::NCH * G::CHB);
  float* s1_first = reinterpret_cast<float*>(o + G::O_F) + G::F_S1;  // fc1 row scales live inside the chunks
  hipLaunchKernelGGL(mixffn_scale_kernel, dim3(G::HID), dim3(64), 0, s, w1, C, s1_first, 32, (long long)(G::CHB / 4));
  hipLaunchKernelGGL(mixffn_scale_kernel, dim3(C), dim3(64), 0, s, w2, G::HID, s2, C, 0ll);
  hipLaunchKernelGGL(mixffn_pack1_kernel<C>, dim3((G::HID * C + 255) / 256), dim3(256), 0, s, w1, o);
  hipLaunchKernelGGL(mixffn_pack2_kernel<C>, dim3((C * G::HID + 255) / 256), dim3(256), 0, s, w2, s2, o);
  hipLaunchKernelGGL(mixffn_pack_consts_kernel<C>, dim3((G::NCH * 11 * 32 + 255) / 256), dim3(256), 0, s, dw9, dwb, b1, o);
  return (int)hipGetLastError();
}

template <int C, int NW>
int launch(const MixFfnK& k, hipStream_t s) {
  using G = MfGeom<C>;
  static_assert(G::SMEM <= 160 * 1024, "LDS budget");
  static_assert(MF_NPX * MF_GP <= MF_HS_BYTES, "Gs must fit the Hs region when it aliases it");
  auto fn = mixffn_kernel<C, NW>;
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const long long tiles = (long long)k.B * k.tiles_x * k.tiles_y;
  hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(64 * NW), G::SMEM, s, k);
  return (int)hipGetLastError();
}

}  // namespace
}  // namespace segmif

using namespace segmif;

#if MF_DBG
extern "C" int segmif_debug_mixffn_timeline(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mixffn_timeline), bytes < sizeof(mixffn_timeline) ? bytes : sizeof(mixffn_timeline));
}
#endif

extern "C" int64_t segmif_mixffn_weight_bytes(int C) {
  return C == 64 ? image_bytes<64>() : C == 128 ? image_bytes<128>() : 0;
}

extern "C" int segmif_mixffn_pack(const float* w1, const float* b1, const float* dw9, const float* dwb, const float* w2, int C,
                                  void* out, void* stream) {
  if (!w1 || !b1 || !dw9 || !dwb || !w2 || !out || ((uintptr_t)out & 15)) return SEGMIF_EINVAL;
  if (C == 64) return pack<64>(w1, b1, dw9, dwb, w2, out, (hipStream_t)stream);
  if (C == 128) return pack<128>(w1, b1, dw9, dwb, w2, out, (hipStream_t)stream);
  return SEGMIF_EINVAL;
}

extern "C" int segmif_mixffn_f16x3(const SegmifMixFfn* d, void* stream) {
  if (!d || !d->x || !d->out || !d->wimg || !d->ln_gamma || !d->ln_beta || !d->b2)
    return SEGMIF_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->x == d->out) return SEGMIF_EINVAL;
  if (((uintptr_t)d->x | (uintptr_t)d->out | (uintptr_t)d->wimg) & 15) return SEGMIF_EINVAL;
  if ((d->amax_a || d->amax_g) && d->amax_images != 1 && d->amax_images != d->B) return SEGMIF_EINVAL;
  MixFfnK k;
  k.x = d->x; k.out = d->out; k.wimg = (const unsigned char*)d->wimg;
  k.gamma = d->ln_gamma; k.beta = d->ln_beta; k.eps = d->ln_eps;
  k.b2 = d->b2;
  k.B = d->B; k.H = d->H; k.W = d->W;
  k.tiles_x = (d->W + MF_TW - 1) / MF_TW; k.tiles_y = (d->H + MF_TH - 1) / MF_TH;
  k.amax_a = d->amax_a; k.amax_g = d->amax_g; k.amax_images = d->amax_images;
  if ((long long)k.B * k.tiles_x * k.tiles_y >= (1ll << 31)) return SEGMIF_EINVAL;
  if (d->C == 64) return launch<64, 8>(k, (hipStream_t)stream);
  if (d->C == 128) return launch<128, 8>(k, (hipStream_t)stream);
  return SEGMIF_EINVAL;
}
