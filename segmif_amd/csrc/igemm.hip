// Implicit-GEMM convolution / linear for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: 64 cycles per SIMD per instruction, 157 TFLOP/s chip peak).
//
// One kernel family serves every dense contraction on SegMiF's hot path: the dilated 3x3 convs
// of the DRDB blocks (core/model_fusion.py:121-157 — 59 % of a pair's FLOPs), the plain 3x3 / 1x1
// convs of the fusion net, the overlap-patch / spatial-reduction convs and all nn.Linear layers
// of the MiT encoder and SegFormer head.  Activations are NHWC (== the reference's token layout),
// so the im2col matrix is never built: each thread gathers 16-byte channel runs for its rows
// straight from the image, zero-filling the padding halo, and stages them through LDS.
//
// Tiling (256 threads = 4 waves, one per SIMD; wave64):
//   block tile BM x BN, K step BK; wave tile WM x WN made of 32x32 MFMA sub-tiles.
//   LDS holds A[BM][BK+4] and B[BN][BK+4], double buffered; the +4 float row pad makes the
//   ds_read_b128 fragment reads conflict free (row stride 20 or 36 dwords: 16 lanes of a read
//   group land on 16 distinct 4-bank slots).
//   Fragment trick: lane (r = lane&31, h = lane>>5) reads ONE float4 = k-offsets 4h..4h+3 of row r
//   and issues 4 MFMAs from it; MFMA s consumes the k-pair {s, 4+s}.  The k order inside a sum is
//   free as long as A and B agree, so one 16-byte LDS read feeds four 64-cycle matrix ops.
//   Global -> register -> LDS staging is software pipelined: tile k+1's loads are issued before
//   tile k's MFMAs and written to the other LDS buffer after them (one barrier per K step).
//   Block ids are remapped so that each XCD (private L2) owns a contiguous range of M tiles:
//   neighbouring tiles share their 3x3 halo rows.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_once.h"
#include <stdint.h>

#include "segmif_hip.h"
#include "igemm_common.h"
#include "planes16.h"


using namespace segmif;

namespace {

typedef __bf16 ig_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ig_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t ig_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ig_pk_bf16(float a, float b) {
  ig_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ig_bf16x2));
}
// x = p0 + p1 + p2, three bf16 each (round to nearest, exact residuals): the planes format's split (conv3x3_planes.hip)
__device__ __forceinline__ void ig_split3(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = ig_pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
  p1 = ig_pk_bf16(r0, r1);
  r0 -= __uint_as_float(p1 << 16);
  r1 -= __uint_as_float(p1 & 0xffff0000u);
  p2 = ig_pk_bf16(r0, r1);
}

enum { MODE_DENSE = 0, MODE_CONV = 1, MODE_GENERIC = 2, MODE_DENSE2 = 3 };

template <int BM, int BN, int WM, int WN, int BK, int MODE, int PF = 1>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmK p) {
  constexpr int BKP = BK + 4;
  constexpr int UPR = BK / 4;  // float4 units per tile row
  constexpr int RPP = 256 / UPR;  // rows covered by one pass of the 256 threads
  constexpr int AU = BM / RPP;  // A units per thread
  constexpr int BU = (BN + RPP - 1) / RPP;  // B units per thread (last may be partial)
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "block must hold exactly 4 waves");
  static_assert(BM % RPP == 0, "A tile must be a whole number of passes");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;  // [2][BM][BKP]
  float* Bs = smem + 2 * BM * BKP;  // [2][BN][BKP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware (bijective) tile remap: dispatcher puts block b on XCD b % 8.
  int bid = blockIdx.x;
  {
    const int nwg = p.ntm * p.ntn;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = bid / p.ntn, nt = bid - mt * p.ntn;
  const long long m0 = (long long)mt * BM;
  const int n0 = nt * BN;
  const int zsplit = p.splitk > 1 ? (int)blockIdx.z : 0;  // split-K is only used with nz == 1
  const long long zb = p.splitk > 1 ? 0 : blockIdx.z / p.nz2, z2 = p.splitk > 1 ? 0 : blockIdx.z - zb * p.nz2;
  const float* __restrict__ in = p.in + zb * p.in_zs + z2 * p.in_zs2;
  const float* __restrict__ in2 = (MODE == MODE_DENSE2) ? p.in2 + zb * p.in2_zs : nullptr;
  const float* __restrict__ wt = p.wt + zb * p.wt_zs + z2 * p.wt_zs2;

  const int kq = tid % UPR;
  const int row_t = tid / UPR;

  // ---- per-thread row state for the A gather -------------------------------------------------
  bool a_ok[AU];
  long long a_off[AU];  // dense: row*lda ; conv: image base in pixels
  int a_iy0[AU], a_ix0[AU];
#pragma unroll
  for (int j = 0; j < AU; ++j) {
    const long long m = m0 + row_t + j * RPP;
    a_ok[j] = m < p.M;
    if (MODE == MODE_DENSE || MODE == MODE_DENSE2) {
      a_off[j] = m;
      a_iy0[j] = a_ix0[j] = 0;
    } else {
      const long long ohw = (long long)p.OH * p.OW;
      const long long b = m / ohw;
      const int rem = (int)(m - b * ohw);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_off[j] = b * p.H * p.W;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad;
    }
  }
  bool b_ok[BU];
  const float* b_ptr[BU];
#pragma unroll
  for (int j = 0; j < BU; ++j) {
    const int nrow = row_t + j * RPP;
    b_ok[j] = (nrow < BN) && (n0 + nrow < p.N);
    b_ptr[j] = wt + (long long)(n0 + nrow) * p.ldw + kq * 4;
  }

  f32x4 ra0[AU], rb0[BU], ra1[PF == 2 ? AU : 1], rb1[PF == 2 ? BU : 1];
  int tap_ky = 0, tap_kx = 0, tap_c0 = 0;  // CONV mode: wave-uniform position of the next K tile

  auto gload = [&](int kc, f32x4* ra, f32x4* rb) {
    const int k0 = kc * BK;
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      rb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b_ok[j]) rb[j] = *reinterpret_cast<const f32x4*>(b_ptr[j] + k0);
    }
    if (MODE == MODE_DENSE) {
#pragma unroll
      for (int j = 0; j < AU; ++j) {
        ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_ok[j]) ra[j] = *reinterpret_cast<const f32x4*>(in + a_off[j] * p.lda + k0 + kq * 4);
      }
    } else if (MODE == MODE_DENSE2) {
      const bool second = k0 >= p.K1;
#pragma unroll
      for (int j = 0; j < AU; ++j) {
        ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_ok[j]) {
          const float* src = second ? in2 + a_off[j] * p.lda2 + (k0 - p.K1) + kq * 4
                                    : in + a_off[j] * p.lda + k0 + kq * 4;
          ra[j] = *reinterpret_cast<const f32x4*>(src);
        }
      }
    } else if (MODE == MODE_CONV) {
      const int dy = tap_ky * p.dil, dx = tap_kx * p.dil;
#pragma unroll
      for (int j = 0; j < AU; ++j) {
        const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
        ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          const long long pix = a_off[j] + (long long)iy * p.W + ix;
          ra[j] = *reinterpret_cast<const f32x4*>(in + pix * p.lda + tap_c0 + kq * 4);
        }
      }
      tap_c0 += BK;
      if (tap_c0 >= p.Cin) {
        tap_c0 = 0;
        if (++tap_kx == p.KW) {
          tap_kx = 0;
          ++tap_ky;
        }
      }
    } else {  // MODE_GENERIC: any Cin / K, scalar gathers (patch_embed1: Cin = 3; conv1_*: Cin = 1)
#pragma unroll
      for (int j = 0; j < AU; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + kq * 4 + e;
          v[e] = 0.f;
          if (a_ok[j] && k < p.K) {
            const int tap = k / p.Cin, c = k - tap * p.Cin;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int iy = a_iy0[j] + ky * p.dil, ix = a_ix0[j] + kx * p.dil;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
              v[e] = in[(a_off[j] + (long long)iy * p.W + ix) * p.lda + c];
          }
        }
        ra[j] = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  };

  auto sstore = [&](int buf, const f32x4* ra, const f32x4* rb) {
    float* a_dst = As + buf * (BM * BKP) + row_t * BKP + kq * 4;
#pragma unroll
    for (int j = 0; j < AU; ++j) *reinterpret_cast<f32x4*>(a_dst + j * RPP * BKP) = ra[j];
    float* b_dst = Bs + buf * (BN * BKP) + row_t * BKP + kq * 4;
#pragma unroll
    for (int j = 0; j < BU; ++j)
      if (row_t + j * RPP < BN) *reinterpret_cast<f32x4*>(b_dst + j * RPP * BKP) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  int kc_begin = 0, nk = p.Kp / BK;
  if (p.splitk > 1) {
    kc_begin = zsplit * p.ksteps_per_split;
    const int kc_end = kc_begin + p.ksteps_per_split;
    nk = kc_end < nk ? kc_end : nk;
    if (MODE == MODE_CONV) {  // position of this split's first K tile
      const int k0 = kc_begin * BK;
      const int tap = k0 / p.Cin;
      tap_c0 = k0 - tap * p.Cin;
      tap_ky = tap / p.KW;
      tap_kx = tap - tap_ky * p.KW;
    }
  }
  const int frag_off = (lane & 31) * BKP + (lane >> 5) * 4;
  auto compute = [&](int cur) {
    const float* a_base = As + cur * (BM * BKP) + (wm * WM) * BKP + frag_off;
    const float* b_base = Bs + cur * (BN * BKP) + (wn * WN) * BKP + frag_off;
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BKP + t * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BKP + t * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][s], a[i][s], acc[i][j], 0, 0, 0);  // D[n][m]: a lane owns output row m
    }
  };

  if constexpr (PF == 1) {
    gload(kc_begin, ra0, rb0);
    sstore(0, ra0, rb0);
    __syncthreads();
    for (int kc = kc_begin; kc < nk; ++kc) {
      const int cur = (kc - kc_begin) & 1;
      if (kc + 1 < nk) gload(kc + 1, ra0, rb0);
      compute(cur);
      if (kc + 1 < nk) sstore(cur ^ 1, ra0, rb0);
      __syncthreads();
    }
  } else {
    // prefetch distance 2: a tile's global loads are issued two K steps before its LDS store, so ~2 x the MFMA
    // work of a step covers the L2 / HBM latency (latency-bound grids of a few blocks per CU)
    gload(kc_begin, ra0, rb0);
    sstore(0, ra0, rb0);
    if (kc_begin + 1 < nk) gload(kc_begin + 1, ra1, rb1);
    if (kc_begin + 2 < nk) gload(kc_begin + 2, ra0, rb0);
    __syncthreads();
    for (int kc = kc_begin; kc < nk; kc += 2) {
      compute(0);  // tile kc
      if (kc + 1 < nk) sstore(1, ra1, rb1);
      if (kc + 3 < nk) gload(kc + 3, ra1, rb1);
      __syncthreads();
      if (kc + 1 < nk) {
        compute(1);  // tile kc + 1
        if (kc + 2 < nk) sstore(0, ra0, rb0);
        if (kc + 4 < nk) gload(kc + 4, ra0, rb0);
        __syncthreads();
      }
    }
  }

  // ---- epilogues.  The products run transposed (weights as the MFMA's row operand; same products, same summation order,
  // so the values are bit-identical to the row-per-register form): lane (r, h) owns output row 32 i + r of the wave tile
  // and its registers 4g .. 4g+3 of sub-tile j are columns 32 j + 8 g + 4 h .. + 3.  16-byte accesses when the operands
  // allow (p.vec4), all loads ahead of the first store: on gfx9 vmcnt counts stores as well, so a load issued after a
  // store waits out the store's full round trip.
  const int er = lane & 31, eh = lane >> 5;
  auto col_of = [&](int j, int g) { return n0 + wn * WN + j * 32 + 8 * g + 4 * eh; };
  auto row_of = [&](int i) { return m0 + wm * WM + i * 32 + er; };
  if (p.splitk > 1) {  // raw partial sums; bias / activation / residual are applied by splitk_reduce_kernel
    float* __restrict__ wsp = p.ws + (long long)zsplit * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long long m = row_of(i);
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = col_of(j, g);
          if (p.vec4) {
            if (n < p.N)
              *reinterpret_cast<f32x4*>(wsp + m * p.N + n) = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) wsp[m * p.N + n + e] = acc[i][j][4 * g + e];
          }
        }
    }
    return;
  }
  float* __restrict__ out = p.out + zb * p.out_zs + z2 * p.out_zs2;
  const float* __restrict__ res = p.res ? p.res + zb * p.res_zs + z2 * p.res_zs2 : nullptr;
  const float* __restrict__ mask = p.mask ? p.mask + zb * p.mask_zs : nullptr;  // (16-byte epilogue only: checked at launch)
  // per-column constants of this block's BN columns -> LDS (the K loop's tiles are dead): bias, and for the fused
  // LayerNorm gamma and beta; read back as float4 (lgkmcnt, not in the way of the vector-memory queue)
  float* Cs = As;  // [3][BN]
  __syncthreads();
  for (int u = tid; u < BN; u += 256) {
    const int n = n0 + u;
    Cs[u] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    if (p.ln_gamma) {
      Cs[BN + u] = n < p.N ? p.ln_gamma[n] : 0.f;
      Cs[2 * BN + u] = n < p.N ? p.ln_beta[n] : 0.f;
    }
  }
  __syncthreads();
  const float* cl = Cs + wn * WN + 4 * eh;  // + 32 j + 8 g
  if constexpr (WN == 64 && BN == 64) {
    if (p.ln_gamma) {  // out = LayerNorm_64(res + acc + bias) (vec4 guaranteed by the host): a row = 32 values here + 32 in lane ^ 32
      f32x4 rr[TM][2][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m = row_of(i);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            rr[i][j][g] = (res && m < p.M) ? *reinterpret_cast<const f32x4*>(res + m * p.ldr + col_of(j, g)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m = row_of(i);
        float y[2][16];
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(cl + 32 * j + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              y[j][4 * g + e] = acc[i][j][4 * g + e] + bb[e] + rr[i][j][g][e];
              s1 += y[j][4 * g + e];
            }
          }
        s1 += __shfl_xor(s1, 32);
        const float mean = s1 * (1.0f / 64.0f);
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            y[j][v] -= mean;
            s2 = fmaf(y[j][v], y[j][v], s2);
          }
        s2 += __shfl_xor(s2, 32);
        const float rstd = 1.0f / sqrtf(s2 * (1.0f / 64.0f) + p.ln_eps);
        if (m < p.M) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 ga = *reinterpret_cast<const f32x4*>(cl + BN + 32 * j + 8 * g);
              const f32x4 be = *reinterpret_cast<const f32x4*>(cl + 2 * BN + 32 * j + 8 * g);
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = y[j][4 * g + e] * rstd * ga[e] + be[e];
              *reinterpret_cast<f32x4*>(out + m * p.ldo + col_of(j, g)) = o;
            }
        }
      }
      return;
    }
  }
  // (r6) ONE uniform branch on "is the activation a GELU" in front of the whole epilogue.  Written with the erf GELU inside the
  // per-element activation switch, every accumulator element of a lane carried its own copy of it: 9 000 - 23 000 lines of ISA per
  // instantiation (54 KB and more - the 64 KB instruction cache two CUs share does not hold one kernel), for an activation no GEMM
  // of the measured paths applies (GELU follows the depthwise conv).  The 40-us Linears of a training step paid for it in
  // instruction fetch.  (Same finding as gemm_pairs.hip's epilogue.)
  auto epilogue = [&](auto gelu_c) {
    constexpr bool GELU = decltype(gelu_c)::value;
    const float slope = (p.act == SEGMIF_ACT_PRELU) ? *p.prelu : 0.f;
    auto activate = [&](float y) {
      if constexpr (GELU) return gelu_exact(y);
      if (p.act == SEGMIF_ACT_RELU) y = fmaxf(y, 0.f);
      else if (p.act == SEGMIF_ACT_PRELU) y = y >= 0.f ? y : slope * y;
      return y;
    };
    if (p.vec4) {
      // the planes copy exists in the generic-gather instantiations only (a conv that asks for it is routed there: its one
      // user is conv1 with Cin = 1); the dense and 16-channel-chunk conv tiles stay free of its registers (20-30 VGPRs)
      constexpr bool PLANES = MODE == MODE_GENERIC;
      const int pl_pitch = PLANES && p.pl_f16 ? p16::PIXEL_BYTES : 96;
      uint32_t pl_amx = 0u;  // f16x3 planes: largest |output| this lane wrote (p16::absmax_pk patterns)
      f32x4 rr[TM][TN][4];
      if (res) {
  #pragma unroll
        for (int i = 0; i < TM; ++i) {
          const long long m = row_of(i);
  #pragma unroll
          for (int j = 0; j < TN; ++j)
  #pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = col_of(j, g);
              rr[i][j][g] = (m < p.M && n < p.N) ? *reinterpret_cast<const f32x4*>(res + m * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
      }
  #pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m = row_of(i);
        if (m >= p.M) continue;
        unsigned char* pl_px = nullptr;
        if (PLANES && p.planes) {  // output pixel m = (b, oy, ox) -> its 96-byte (f16x3: 64-byte) slot in chunk image 0 of batch element b
          const long long ohw = (long long)p.OH * p.OW;
          const long long b = m / ohw;
          const int rem = (int)(m - b * ohw);
          const int oy = rem / p.OW, ox = rem - oy * p.OW;
          pl_px = p.planes + ((((long long)b * p.pl_chunks + p.pl_chunk0) * p.pl_Hp + oy + 2) * p.pl_Wp + ox + 2) * pl_pitch + eh * 16;
        }
  #pragma unroll
        for (int j = 0; j < TN; ++j)
  #pragma unroll
          for (int q = 0; q < 2; ++q) {  // columns 32 j + 16 q .. + 15 = planes chunk (n0 + wn WN + 32 j) / 16 + q
            float yy[8];
  #pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              const int g = 2 * q + gg;
              const int n = col_of(j, g);
              if (n >= p.N) continue;  // N % 4 == 0: a group of four is inside or outside as a whole
              const f32x4 bb = *reinterpret_cast<const f32x4*>(cl + 32 * j + 8 * g);
              f32x4 y;
  #pragma unroll
              for (int e = 0; e < 4; ++e) {
                y[e] = activate(acc[i][j][4 * g + e] + bb[e]);
                if (res) y[e] += rr[i][j][g][e];
                yy[4 * gg + e] = y[e];
              }
              if (mask) {  // through a ReLU whose OUTPUT the mask tensor is (the consumer's forward activation)
                const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + m * p.ldm + n);
  #pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = mk[e] > 0.f ? y[e] : 0.f;
              }
              if (p.out) *reinterpret_cast<f32x4*>(out + m * p.ldo + n) = y;
            }
            if (PLANES && p.planes && col_of(j, 2 * q) < p.N) {
              // this lane's 8 of the chunk's 16 channels are positions 8 h .. 8 h + 7 of the chunk (sigma order): one
              // 16-byte store per plane
              const int chunk = (n0 + wn * WN + 32 * j) / 16 + q;
              unsigned char* dst = pl_px + (long long)chunk * p.pl_Hp * p.pl_Wp * pl_pitch;
              if (p.pl_f16) {
                ig_u32x4 hi, lo;
                p16::split8(yy, hi, lo);
                *reinterpret_cast<ig_u32x4*>(dst) = hi;
                *reinterpret_cast<ig_u32x4*>(dst + 32) = lo;
                pl_amx = p16::absmax_pk4(pl_amx, hi, lo);
              } else {
                ig_u32x4 pp[3];
  #pragma unroll
                for (int e = 0; e < 4; ++e) {
                  uint32_t a, b, c;
                  ig_split3(yy[2 * e], yy[2 * e + 1], a, b, c);
                  pp[0][e] = a; pp[1][e] = b; pp[2][e] = c;
                }
  #pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<ig_u32x4*>(dst + k * 32) = pp[k];
              }
            }
          }
      }
      if (PLANES && p.pl_amax) {  // (kernel argument: uniform over the grid) this wave's rows -> their images' range slots
        const long long ohw = (long long)p.OH * p.OW;
        const long long ma = m0 + wm * WM, mb = ma + TM * 32 - 1 < p.M ? ma + TM * 32 - 1 : p.M - 1;
        if (ma < p.M) p16::fold_pat(p.pl_amax, p.pl_amax_images > 1 ? (int)(ma / ohw) : 0, p.pl_amax_images > 1 ? (int)(mb / ohw) : 0, pl_amx);
      }
    } else {  // ragged N or unaligned views: element by element
  #pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m = row_of(i);
        if (m >= p.M) continue;
  #pragma unroll
        for (int j = 0; j < TN; ++j)
  #pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = col_of(j, g);
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n + e >= p.N) continue;
              float y = activate(acc[i][j][4 * g + e] + cl[32 * j + 8 * g + e]);
              if (res) y += res[m * p.ldr + n + e];
              out[m * p.ldo + n + e] = y;
            }
          }
      }
    }
  };
  if (p.act == SEGMIF_ACT_GELU) epilogue(std::true_type{});
  else epilogue(std::false_type{});
}

struct TileCfg {
  int BM, BN, BK;
  const char* name;
};
constexpr int kNumTiles = 15;  // 0-8, 11-13: implicit-GEMM tiles; 9, 10: halo-tiled 3x3 (conv3x3.hip); 14: split-bf16 halo 3x3
const TileCfg kTiles[kNumTiles] = {
    {256, 32, 16, "256x32x16"}, {256, 32, 32, "256x32x32"}, {128, 64, 16, "128x64x16"},
    {128, 64, 32, "128x64x32"}, {128, 128, 16, "128x128x16"}, {128, 128, 32, "128x128x32"},
    {64, 64, 16, "64x64x16"},   {256, 64, 16, "256x64x16"},   {256, 64, 32, "256x64x32"},
    {256, 64, 16, "halo8x32c16"}, {256, 64, 8, "halo8x32c8"}, {64, 64, 32, "64x64x32"},
    {64, 64, 16, "64x64x16p2"}, {128, 64, 16, "128x64x16p2"}, {512, 64, 16, "halo16x32c16_bf16x6"},
};
constexpr int kHaloTile0 = 9;
constexpr int kSplitTile = 14;  // desc.wt = segmif_conv3x3_split_pack() image, not the fp32 packing

template <int BM, int BN, int WM, int WN, int BK, int MODE, int PF = 1>
int launch(const IgemmK& k, int nz, hipStream_t stream) {
  constexpr size_t smem = 2ull * (BM + BN) * (BK + 4) * sizeof(float);
  auto fn = igemm_kernel<BM, BN, WM, WN, BK, MODE, PF>;
  if (smem > 64 * 1024) {
    static segmif::PerDeviceFlag raised_flag;  // idempotent attribute; benign race
  bool& raised = raised_flag.here();
    if (!raised) {
      hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      raised = true;
    }
  }
  dim3 grid((unsigned)(k.ntm * k.ntn), 1, (unsigned)nz);
  hipLaunchKernelGGL(fn, grid, dim3(256), smem, stream, k);
  return (int)hipGetLastError();
}

template <int MODE>
int dispatch_tile(int tile, const IgemmK& k, int nz, hipStream_t s) {
  switch (tile) {
    case 0: return launch<256, 32, 64, 32, 16, MODE>(k, nz, s);
    case 1: return launch<256, 32, 64, 32, 32, MODE>(k, nz, s);
    case 2: return launch<128, 64, 64, 32, 16, MODE>(k, nz, s);
    case 3: return launch<128, 64, 64, 32, 32, MODE>(k, nz, s);
    case 4: return launch<128, 128, 64, 64, 16, MODE>(k, nz, s);
    case 5: return launch<128, 128, 64, 64, 32, MODE>(k, nz, s);
    case 6: return launch<64, 64, 32, 32, 16, MODE>(k, nz, s);
    case 7: return launch<256, 64, 64, 64, 16, MODE>(k, nz, s);
    case 8: return launch<256, 64, 64, 64, 32, MODE>(k, nz, s);
    case 11: return launch<64, 64, 32, 32, 32, MODE>(k, nz, s);
    case 12: return launch<64, 64, 32, 32, 16, MODE, 2>(k, nz, s);
    case 13: return launch<128, 64, 64, 32, 16, MODE, 2>(k, nz, s);
  }
  return SEGMIF_EINVAL;
}

// GENERIC (scalar gather) is only ever used for the two tiny-Cin stem convs: keep two tiles.
int dispatch_generic(int tile, const IgemmK& k, int nz, hipStream_t s) {
  switch (tile) {
    case 0: return launch<256, 32, 64, 32, 16, MODE_GENERIC>(k, nz, s);
    case 2: return launch<128, 64, 64, 32, 16, MODE_GENERIC>(k, nz, s);
    case 6: return launch<64, 64, 32, 32, 16, MODE_GENERIC>(k, nz, s);
    case 7: return launch<256, 64, 64, 64, 16, MODE_GENERIC>(k, nz, s);  // the fused-LayerNorm tile (stage-1 patch embed, Cin = 3)
  }
  return SEGMIF_EINVAL;
}

// Tile choice, from the MI355X sweep in profiles/r01_kernel_bench.txt: BK = 32 never pays (its LDS
// footprint costs a block per CU), 64x64 wins whenever the grid is not enormous, the big tiles only
// pay for long-K / wide-N problems with tens of thousands of 64x64 tiles.
int pick_tile(long long M, int N, int K, int nz, bool generic) {
  auto blocks = [&](int t) {
    return ((M + kTiles[t].BM - 1) / kTiles[t].BM) * ((N + kTiles[t].BN - 1) / kTiles[t].BN) * nz;
  };
  if (N <= 32) return 0;
  if (generic) return blocks(6) >= 16384 ? 2 : 6;
  if (N >= 128 && K >= 512 && blocks(4) >= 1024) return 4;
  if (N <= 64 && K >= 128 && blocks(7) >= 2048) return 7;
  if (blocks(6) >= 16384) return 2;
  return 6;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmK p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M * p.N) return;
  const long long m = i / p.N;
  const int n = (int)(i - m * p.N);
  float y = 0.f;
  for (int s = 0; s < p.splitk; ++s) y += p.ws[(long long)s * p.M * p.N + i];  // fixed order: deterministic
  if (p.bias) y += p.bias[n];
  if (p.act == SEGMIF_ACT_RELU) y = fmaxf(y, 0.f);
  else if (p.act == SEGMIF_ACT_PRELU) y = y >= 0.f ? y : *p.prelu * y;
  else if (p.act == SEGMIF_ACT_GELU) y = gelu_exact(y);
  if (p.res) y += p.res[m * p.ldr + n];
  p.out[m * p.ldo + n] = y;
}

// split-K plan: only for single-slice dense / conv problems whose tile grid leaves most CUs idle
int plan_splitk(long long M, int N, int Kp, int tile) {
  const long long blocks = ((M + kTiles[tile].BM - 1) / kTiles[tile].BM) * ((N + kTiles[tile].BN - 1) / kTiles[tile].BN);
  const int nk = Kp / kTiles[tile].BK;
  if (blocks >= 512 || nk < 8) return 1;
  long long s = (1024 + blocks - 1) / blocks;
  if (s > nk / 4) s = nk / 4;
  if (s > 16) s = 16;
  return s < 2 ? 1 : (int)s;
}

__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Cin, int KH,
                                   int KW, int Kp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * Kp) return;
  const int n = (int)(i / Kp), k = (int)(i - (long long)n * Kp);
  float v = 0.f;
  if (k < KH * KW * Cin) {
    const int tap = k / Cin, c = k - tap * Cin;
    const int ky = tap / KW, kx = tap - ky * KW;
    v = src[(((long long)n * Cin + c) * KH + ky) * KW + kx];
  }
  dst[i] = v;
}

}  // namespace

extern "C" int segmif_igemm_num_tiles(void) { return kNumTiles; }
extern "C" const char* segmif_igemm_tile_name(int t) { return (t >= 0 && t < kNumTiles) ? kTiles[t].name : "?"; }

extern "C" int segmif_pack_conv_weight(const float* src, float* dst, int N, int Cin, int KH, int KW, void* stream) {
  if (!src || !dst || N <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return SEGMIF_EINVAL;
  const int Kp = (KH * KW * Cin + 15) / 16 * 16;
  const long long total = (long long)N * Kp;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, N, Cin, KH, KW, Kp);
  return (int)hipGetLastError();
}

static int igemm_resolve(const SegmifIgemm* d, IgemmK& k, int& mode_out, int& tile_out, int& nz_out, bool& halo_out);

extern "C" int64_t segmif_igemm_workspace_floats(const SegmifIgemm* d) {
  IgemmK k;
  int mode, tile, nz;
  bool halo;
  if (igemm_resolve(d, k, mode, tile, nz, halo) != 0 || halo) return 0;
  return k.splitk > 1 ? (int64_t)k.splitk * k.M * k.N : 0;
}

extern "C" int segmif_igemm_f32(const SegmifIgemm* d, void* stream) {
  IgemmK k;
  int mode, tile, nz;
  bool halo;
  const int rc = igemm_resolve(d, k, mode, tile, nz, halo);
  if (rc != 0) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((d->split_f16 || d->split_out_amax) && !(halo && tile == kSplitTile)) return SEGMIF_EINVAL;  // split 3x3 kernel only
  if (d->relu_mask && !(halo && tile == kSplitTile)) {
    // elsewhere the mask lives in the 16-byte epilogue of the implicit-GEMM tiles: no halo kernel, no fused LayerNorm, no planes
    // copy, no split-K (igemm_resolve keeps a masked problem un-split)
    if (halo || !k.vec4 || k.ln_gamma || k.planes || k.splitk > 1 || (k.ldm & 3) || ((uintptr_t)k.mask & 15) || (k.mask_zs & 3) || k.nz2 > 1)
      return SEGMIF_EINVAL;
  }
  if (halo) return tile == kSplitTile ? conv3x3_split_launch(k, s) : conv3x3_halo_launch(k, tile - kHaloTile0, s);
  if (k.splitk > 1) {
    if (!d->workspace || d->workspace_floats < (int64_t)k.splitk * k.M * k.N) k.splitk = 1;  // no room: plain launch
    else k.ws = d->workspace;
    if ((uintptr_t)k.ws & 15) k.vec4 = 0;
  }
  int r;
  const int gz = k.splitk > 1 ? k.splitk : nz;
  switch (mode) {
    case MODE_DENSE: r = dispatch_tile<MODE_DENSE>(tile, k, gz, s); break;
    case MODE_CONV: r = dispatch_tile<MODE_CONV>(tile, k, gz, s); break;
    case MODE_DENSE2: r = dispatch_tile<MODE_DENSE2>(tile, k, gz, s); break;
    default: r = dispatch_generic(tile, k, gz, s);
  }
  if (r != 0 || k.splitk <= 1) return r;
  const long long total = k.M * k.N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k);
  return (int)hipGetLastError();
}

static int igemm_resolve(const SegmifIgemm* d, IgemmK& k, int& mode_out, int& tile_out, int& nz_out, bool& halo_out) {
  if (!d || !d->in || !d->wt || (!d->out && !d->planes_out) || d->M <= 0 || d->N <= 0 || d->K <= 0) return SEGMIF_EINVAL;  // (out may be NULL when the planes copy is the only consumer)
  halo_out = false;
  k.in = d->in; k.in2 = d->in2; k.wt = d->wt; k.bias = d->bias; k.res = d->res; k.prelu = d->prelu; k.out = d->out;
  k.M = d->M; k.N = d->N; k.K = d->K; k.Kp = (d->K + 15) / 16 * 16;
  k.lda = d->lda; k.lda2 = d->lda2; k.K1 = d->K1; k.ldo = d->ldo; k.ldr = d->ldr;
  k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad;
  k.dil = d->dil; k.OH = d->OH; k.OW = d->OW; k.act = d->act;
  k.in_zs = d->in_zstride; k.in2_zs = d->in2_zstride; k.wt_zs = d->wt_zstride; k.out_zs = d->out_zstride;
  k.res_zs = d->res_zstride;
  k.nz2 = d->nz2 > 0 ? d->nz2 : 1;
  k.in_zs2 = d->in_zstride2; k.wt_zs2 = d->wt_zstride2; k.out_zs2 = d->out_zstride2; k.res_zs2 = d->res_zstride2;
  k.ln_gamma = d->ln_gamma; k.ln_beta = d->ln_beta; k.ln_eps = d->ln_eps;
  k.mask = d->relu_mask; k.ldm = d->ld_mask; k.mask_zs = d->mask_zstride;
  k.split_f16 = d->split_f16; k.in_amax = d->split_in_amax; k.in_amax_n = d->split_in_amax_n; k.out_amax = d->split_out_amax;
  k.out_amax_n = d->split_out_amax_n > 0 ? d->split_out_amax_n : 1;
  if (k.mask && k.ldm < d->N) return SEGMIF_EINVAL;
  if (k.ln_gamma && (!k.ln_beta || d->N != 64 || d->act != SEGMIF_ACT_NONE)) return SEGMIF_EINVAL;
  k.ldw = d->ldw > 0 ? d->ldw : k.Kp;
  if (k.ldw % 4) return SEGMIF_EINVAL;
  const int nz = (d->nz > 0 ? d->nz : 1) * k.nz2;
  if (d->act == SEGMIF_ACT_PRELU && !d->prelu) return SEGMIF_EINVAL;
  if (d->res && d->ldr <= 0) return SEGMIF_EINVAL;
  k.vec4 = !(d->N & 3) && !(d->ldo & 3) && !((uintptr_t)d->out & 15) && !(k.out_zs & 3) && !(k.out_zs2 & 3) &&
           (!d->res || (!(d->ldr & 3) && !((uintptr_t)d->res & 15) && !(k.res_zs & 3) && !(k.res_zs2 & 3)));
  if (k.ln_gamma && !k.vec4) return SEGMIF_EINVAL;  // the fused LayerNorm epilogue only exists in its 16-byte form
  k.planes = (unsigned char*)d->planes_out;
  k.pl_Hp = k.pl_Wp = k.pl_chunks = k.pl_chunk0 = 0;
  k.pl_f16 = k.planes ? d->planes_f16 : 0;
  k.pl_amax = k.pl_f16 ? d->planes_amax : nullptr;
  k.pl_amax_images = d->planes_amax_images > 1 ? d->planes_amax_images : 1;
  if (k.pl_amax && k.pl_amax_images > 1 && (long long)k.pl_amax_images * d->OH * d->OW != d->M) return SEGMIF_EINVAL;
  if (!d->out && (!k.planes || nz > 1 || d->res)) return SEGMIF_EINVAL;  // planes-only output: the planes epilogue is the one that may skip the fp32 store
  if (k.planes) {
    int hp, wp;
    if (!k.vec4 || (d->N & 15) || nz > 1 || k.ln_gamma || d->planes_chunk0 < 0 || d->planes_chunk0 + d->N / 16 > d->planes_chunks ||
        d->M % ((long long)d->OH * d->OW) || segmif_planes_dims(d->OH, d->OW, &hp, &wp) != 0 || ((uintptr_t)k.planes & 15))
      return SEGMIF_EINVAL;
    k.pl_Hp = hp; k.pl_Wp = wp; k.pl_chunks = d->planes_chunks; k.pl_chunk0 = d->planes_chunk0;
  }

  const bool is_conv = !(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0);
  int mode;
  if (d->in2) {
    if (is_conv || d->K1 % 32 != 0 || (d->K - d->K1) % 16 != 0 || d->lda % 4 || d->lda2 % 4) return SEGMIF_EINVAL;
    mode = MODE_DENSE2;
  } else if (!is_conv) {
    mode = (d->K % 16 == 0 && d->lda % 4 == 0) ? MODE_DENSE : MODE_GENERIC;
  } else {
    mode = (d->Cin % 16 == 0 && d->lda % 4 == 0) ? MODE_CONV : MODE_GENERIC;
  }
  if (mode != MODE_GENERIC && (((uintptr_t)d->in & 15) || (d->in2 && ((uintptr_t)d->in2 & 15)))) mode = d->in2 ? -1 : MODE_GENERIC;
  if (mode < 0 || ((uintptr_t)d->wt & 15)) return SEGMIF_EINVAL;
  if (k.planes) {  // the planes copy is a conv epilogue, compiled into the generic-gather tiles
    if (mode == MODE_DENSE || mode == MODE_DENSE2) return SEGMIF_EINVAL;
    mode = MODE_GENERIC;
  }
  if (mode == MODE_GENERIC && (d->KH * d->KW * d->Cin != d->K)) return SEGMIF_EINVAL;

  const bool bk32_ok = (k.Kp % 32 == 0) && (mode != MODE_CONV || d->Cin % 32 == 0) &&
                       (mode != MODE_DENSE2 || (d->K1 % 32 == 0));
  int tile = d->tile;
  const bool halo_ok = mode == MODE_CONV && nz == 1 && k.ldw == k.Kp && !k.ln_gamma && !k.planes && conv3x3_halo_eligible(k);
  if (k.ln_gamma) {
    if (tile >= 0 && tile != 7 && (tile != 8 || mode == MODE_GENERIC)) return SEGMIF_EINVAL;
    if (tile < 0) tile = 7;  // the fused LayerNorm needs a wave tile spanning all 64 columns
  }
  if (tile < 0 && halo_ok) tile = kHaloTile0 + 1;  // 8-channel chunks: best on every shape (profiles/r01_kernel_bench_halo.txt)
  k.splitk = 1;
  k.ksteps_per_split = 0;
  k.ws = nullptr;
  mode_out = mode;
  nz_out = nz;
  if ((tile >= kHaloTile0 && tile < kHaloTile0 + 2) || tile == kSplitTile) {
    if (!halo_ok) return SEGMIF_EINVAL;
    halo_out = true;
    tile_out = tile;
    return 0;
  }
  const bool auto_tile = tile < 0;
  if (tile < 0) tile = pick_tile(d->M, d->N, d->K, nz, mode == MODE_GENERIC);
  if (tile >= kNumTiles) return SEGMIF_EINVAL;
  if (kTiles[tile].BK == 32 && !bk32_ok) return SEGMIF_EINVAL;
  if (auto_tile && tile == 6 && d->K >= 128 && mode != MODE_GENERIC) tile = 12;  // prefetch distance 2: +3..9 % (profiles/r01_enc_gemm_tiles.txt)
  k.ntm = (int)((d->M + kTiles[tile].BM - 1) / kTiles[tile].BM);
  k.ntn = (d->N + kTiles[tile].BN - 1) / kTiles[tile].BN;
  if (auto_tile && nz == 1 && (mode == MODE_DENSE || mode == MODE_CONV) && k.ldw == k.Kp && !k.ln_gamma && !k.planes && !k.mask) {
    k.splitk = plan_splitk(d->M, d->N, k.Kp, tile);
    if (k.splitk > 1) {
      const int nk = k.Kp / kTiles[tile].BK;
      k.ksteps_per_split = (nk + k.splitk - 1) / k.splitk;
      k.splitk = (nk + k.ksteps_per_split - 1) / k.ksteps_per_split;
    }
  }
  tile_out = tile;
  return 0;
}
