// Halo-tiled 3x3 stride-1 convolution on the bf16 matrix pipe with a 3-way operand split
// ("bf16x6"): fp32-class results at 2.7x the fp32-MFMA rate.  Same role as conv3x3.hip — the DRDB
// dilated convs of core/model_fusion.py:121-157 — selected with tile id 14.
//
// Numerics.  Every fp32 operand is written as x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1) (round to nearest, the two subtractions are exact in fp32), i.e. 3 x 8 = 24
// significand bits.  A product a*w is evaluated as the six terms of weight >= 2^-16
//     a0 w0 + (a0 w1 + a1 w0) + (a0 w2 + a1 w1 + a2 w0)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped
// terms (a1 w2, a2 w1, a2 w2) are <= 2^-23 |a w| with random sign — the size of one fp32 rounding.
// The result is therefore as accurate as the exact-fp32 MFMA path (tests: both against fp64), at
// 6 x 32 = 192 matrix-pipe cycles per 32x32x16 block instead of 8 x 64 = 512.  Inf inputs turn
// into NaN (inf - inf in the split); finite inputs of any magnitude are safe (bf16 has the fp32
// exponent range).
//
// Layout.  A workgroup owns an STH x 32 patch of output pixels (STH = 8: 4 waves, 79 KB of LDS, two
// workgroups per CU — the NOUT = 32 DRDB case; STH = 16: 8 waves, one workgroup per CU, when 64 output
// channels fill the LDS).  Per 16-channel chunk it stages the (STH+2d) x (32+2d) input halo in LDS as
// three bf16 planes per pixel ([pixel][plane][16 ch], 96 B + 16 B pad = 112 B pitch: conflict-free
// ds_read_b128 across 16 consecutive pixels) — the split happens in registers between the global fp32
// load and the LDS write, HBM still holds plain fp32 NHWC — and the nine taps' pre-split weights
// ([tap][n][plane][16], packed once per parameter by segmif_conv3x3_split_pack).  A wave computes
// two patch rows R0 and R0 + d: tap ky of the second reads the halo row tap ky + 1 of the first reads,
// so 4 x 3 fragment sets (not 2 x 9) cover both.  Fragments of step s + 1 are requested before the
// MFMAs of step s (double-buffered A set, 3-deep weight ring, sched_barrier-pinned), chunk c + 1
// streams into registers under chunk c's MFMAs and is split between its last steps.
//
// Where the time goes (round 1, 128 -> 32 DRDB conv at 8 x 480 x 640: 0.95 ms against 1.49 ms for the
// exact-fp32 halo kernel; tools/split_ablate.sh, tools/split_timeline.py, tools/micro/mfma_chain.hip).
// The matrix pipe alone needs 0.53 ms (16.4 ns per MFMA per SIMD at the ~1.95 GHz the chip holds under
// bf16 MFMA load; one accumulation chain per wave already issues back-to-back).  The s_memtime timeline
// of a wave splits a chunk into: LDS stores 18 % (the VGPR -> LDS store path moves ~1.3 B per cycle per
// wave and all eight waves of the CU use it at once), MFMA steps that also issue the next halo's global
// loads 40 % (78 ticks per MFMA: a vector-memory issue stalls the in-order wave), MFMA steps that also
// split 32 % (41 ticks per MFMA: the pipe is saturated there), barriers 4 %.  What helped: issuing the
// next chunk's loads one slot per step instead of as a burst (and unconditionally — behind a branch the
// compiler drains vmcnt to 0 before every load), weights late / halo early.  Tried and measured no
// better: 16-row patches at one workgroup per CU, a ping-pong workgroup (two halves held in anti-phase
// by a barrier per phase), static wave priority by LDS slot, four accumulation chains, contiguous
// instead of 64-byte halo reads.  Next (round 2): a persistent, LDS-double-buffered schedule.
//
// f16x3 form (round 4, the training path's default; F16 = true below).  Same kernel, half pairs instead of bf16 triples:
// x = hi + 2^-11 lo (planes16.h), weights as W0 | Wl | 2^-11 W0 of the row scaled by a power of two (conv3x3_planes.hip's
// format), THREE products (lo W0s, hi Wl, hi W0) instead of six - the matrix pipe was 56 % of a call.  A half has 5 exponent
// bits, and this kernel also multiplies GRADIENTS (1e-7 and below), so the range is not guarded but MADE: the caller passes
// the device slots holding max |x| of the input's channel blocks (their producers' epilogues, or segmif_amax_f32), the kernel
// scales the staged values by the power of two that puts that maximum in [2^13, 2^14) and takes it out again in the
// epilogue (exact both ways).  Every element then carries 22 significant bits down to 2^-27 of the tensor's maximum -
// below fp32's own resolution of a sum that contains the maximum.  A NaN / inf maximum makes the scale NaN: the output is
// all NaN instead of quietly wrong.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_once.h"
#include <stdint.h>

#include "igemm_common.h"
#include "planes16.h"
#include "segmif_hip.h"

#ifndef SPLIT_INTERLEAVE
#define SPLIT_INTERLEAVE 1
#endif
#ifndef SPLIT_DBG
#define SPLIT_DBG 0  // tuning aid: compile-time ablation mask (tools/split_ablate.sh)
#endif

#if SPLIT_DBG & 32  // timeline probe: s_memtime at the phase boundaries of wave 0 of the first blocks
#define SPLIT_TL_BLOCKS 2048
#define SPLIT_TL_CHUNKS 12
__device__ unsigned long long split_timeline[SPLIT_TL_BLOCKS][SPLIT_TL_CHUNKS + 1][8];
#define TL(slot)                                                                                      \
  do {                                                                                                \
    if (tid == 0 && blockIdx.x < SPLIT_TL_BLOCKS && c < SPLIT_TL_CHUNKS)                              \
      split_timeline[blockIdx.x][c][slot] = __builtin_amdgcn_s_memtime();                             \
  } while (0)
#else
#define TL(slot) do {} while (0)
#endif

namespace segmif {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int STW = 32;
constexpr int ROWB = 112;    // LDS bytes per weight row (3 planes) and per bf16x3 pixel
constexpr int ROWB_H = 80;   // per f16x3 pixel: 2 planes x 32 B + 16 B pad (conflict-free ds_read_b128 over 16 pixels: 20-dword stride)

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: a -> low half
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

inline int split_nout(int N) { return N <= 32 ? 32 : 64; }

// STH = patch height: 8 (4 waves, 79 KB LDS with NOUT = 32 -> two workgroups per CU, one loading while
// the other multiplies) or 16 (8 waves, one workgroup per CU; needed when NOUT = 64 fills the LDS).
template <int NOUT, int DIL, int STH, bool F16>
__global__ __launch_bounds__(STH * 32) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_split_kernel(const IgemmK p, int tiles_x, int tiles_y) {
  constexpr int NT = STH * 32;
  constexpr int NPA = F16 ? 2 : 3;          // activation planes
  constexpr int AROW = F16 ? ROWB_H : ROWB; // LDS bytes per staged pixel
  constexpr int NPROD = F16 ? 3 : 6;
  constexpr int HH = STH + 2 * DIL, HW = STW + 2 * DIL, HP = HH * HW;
  constexpr int A_UNITS = HP * 4;         // float4 units: 16 channels = 4 per pixel
  constexpr int B_UNITS = 9 * NOUT * 6;   // 16-byte units of split weights per chunk
  constexpr int AJ = (A_UNITS + NT - 1) / NT, BJ = (B_UNITS + NT - 1) / NT;
  constexpr int TN = NOUT / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* As = smem_b;              // [HP][AROW]
  unsigned char* Bs = smem_b + HP * AROW;  // [9 * NOUT][ROWB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  {  // XCD-aware remap: each XCD owns a contiguous run of patches (shared halo rows stay in one L2)
    const int nwg = gridDim.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int ntile = blockIdx.y, nbase = ntile * NOUT;
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int b = bid / (tiles_x * tiles_y);
  const int x0 = tx * STW, y0 = ty * STH;
  const float* __restrict__ in = p.in + (long long)b * p.H * p.W * p.lda;
  const int nchunks = p.Cin / 16;
  const u32x4* __restrict__ wsp = reinterpret_cast<const u32x4*>(p.wt) + (long long)ntile * nchunks * B_UNITS;

  // per-thread staging slots, fixed across chunks.  Loads are unconditional (out-of-image and
  // surplus slots read pixel 0 / the last weight unit and are zeroed / dropped later) so that the
  // compiler can count them: s_waitcnt vmcnt(n) on the oldest slot while the rest are in flight.
  long long a_off[AJ];  // float offset inside the image: H * W * lda can pass 2^31 (e.g. 3100 x 3100 x 224)
  int a_dst[AJ];
  bool a_ok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int u = tid + NT * j;
    const int pp = u >> 2, q4 = u & 3;
    const int hy = pp / HW, hx = pp - hy * HW;
    const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
    a_dst[j] = u < A_UNITS ? pp * AROW + q4 * 8 : -1;
    a_ok[j] = u < A_UNITS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    a_off[j] = (a_ok[j] ? ((long long)gy * p.W + gx) * p.lda : 0) + q4 * 4;
  }
  int b_src[BJ], b_dst[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int u = tid + NT * j;
    const int row = u / 6, s6 = u - row * 6;
    b_dst[j] = u < B_UNITS ? row * ROWB + s6 * 16 : -1;
    b_src[j] = u < B_UNITS ? u : B_UNITS - 1;
  }

  // f16x3: the power of two that puts the input's largest magnitude (the callers' range slots) into [2^13, 2^14)
  float s_in = 1.f, s_out = 1.f;
  if (F16) {
    s_in = p16::range_scale(p.in_amax, p.in_amax_n);  // (NaN for an inf / NaN input: fail loudly, all NaN)
    s_out = 1.f / s_in;                               // exact (a power of two)
  }

  f32x4 ra[AJ];
  u32x2 pa[AJ][NPA];  // the same units after the split: [plane] = 4 bf16 / halves
  u32x4 rb[BJ];
  auto gload_a = [&](int j, int c) { ra[j] = *reinterpret_cast<const f32x4*>(in + a_off[j] + c * 16); };
  auto gload_b = [&](int j, int c) { rb[j] = wsp[(long long)c * B_UNITS + b_src[j]]; };
  auto split_unit = [&](int j) {  // registers only: interleaved with the tail of the previous chunk's MFMAs
    const f32x4 x = a_ok[j] ? ra[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (F16) {
      uint32_t ha, la, hb, lb;
      p16::split2(x[0] * s_in, x[1] * s_in, ha, la);
      p16::split2(x[2] * s_in, x[3] * s_in, hb, lb);
      pa[j][0] = u32x2{ha, hb};
      pa[j][1] = u32x2{la, lb};
    } else {
      uint32_t p0a, p1a, p2a, p0b, p1b, p2b;
      split3(x[0], x[1], p0a, p1a, p2a);
      split3(x[2], x[3], p0b, p1b, p2b);
      pa[j][0] = u32x2{p0a, p0b};
      pa[j][1] = u32x2{p1a, p1b};
      pa[j][NPA - 1] = u32x2{p2a, p2b};
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      if (a_dst[j] >= 0) {
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) *reinterpret_cast<u32x2*>(As + a_dst[j] + pl * 32) = pa[j][pl];
      }
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      if (b_dst[j] >= 0) *reinterpret_cast<u32x4*>(Bs + b_dst[j]) = rb[j];
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // The wave's two sub-tiles are patch rows R0 and R0 + DIL: tap ky of the second reads the halo row
  // tap ky + 1 of the first reads, so 4 x 3 fragment sets (not 2 x 9) cover both.  A chunk is 12
  // steps (kx, m): halo row R0 + m * DIL at column offset kx * DIL feeds sub-tile 0 with tap ky = m
  // (m < 3) and sub-tile 1 with tap ky = m - 1 (m > 0).  The fragments of step s + 1 are requested
  // before the MFMAs of step s are issued (double-buffered A set, 3-deep weight ring), so the LDS
  // latency hides under 6-12 MFMAs instead of stalling the wave once per fragment.
  const int R0 = (DIL == 2) ? ((wave >> 1) * 4 + (wave & 1)) : 2 * wave;
  const unsigned char* a_lane = As + (R0 * HW + r) * AROW + h * 16;
  const unsigned char* b_lane = Bs + r * ROWB + h * 16;
  // products (activation plane, weight plane), least significant first: six for bf16 triples,
  // three for half pairs (lo x 2^-11 W0, hi x Wl, hi x W0)
  constexpr int PA[6] = {F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0, 0};
  constexpr int PW[6] = {F16 ? 2 : 0, 1, F16 ? 0 : 2, 0, 1, 0};
  constexpr int SPLIT_FROM = 12 - AJ;

  constexpr bool W_AHEAD = TN == 1;  // two output-channel tiles: the 3-deep weight ring would spill
  bf16x8 F[2][NPA], Wr[W_AHEAD ? 3 : 2][TN][3];  // (16-byte register quads; reinterpreted as halves for the f16 MFMA)
  auto load_f = [&](int st) {
    const int kx = st >> 2, m = st & 3;
#pragma unroll
    for (int pl = 0; pl < NPA; ++pl)
      F[st & 1][pl] = *reinterpret_cast<const bf16x8*>(a_lane + (m * DIL * HW + kx * DIL) * AROW + pl * 32);
  };
  auto mma = [&](const bf16x8& a, const bf16x8& b, f32x16 c) {
    if constexpr (F16)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };
  auto load_w = [&](int st) {
    const int kx = st >> 2, m = st & 3;
    if (m < 3) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          Wr[W_AHEAD ? m : (m & 1)][j][pl] =
              *reinterpret_cast<const bf16x8*>(b_lane + ((m * 3 + kx) * NOUT + j * 32) * ROWB + pl * 32);
    }
  };

#pragma unroll
  for (int j = 0; j < AJ; ++j) gload_a(j, 0);
#pragma unroll
  for (int j = 0; j < BJ; ++j) gload_b(j, 0);
#pragma unroll
  for (int j = 0; j < AJ; ++j) split_unit(j);
  static_assert(AJ <= 7 && BJ <= 12, "staging slots are issued one per MFMA step");
  for (int c = 0; c < nchunks; ++c) {
    TL(0);
    if (!(SPLIT_DBG & 16) || c == 0) sstore();
    TL(1);
    __syncthreads();
    TL(2);
    const int cnext = c + 1 < nchunks ? c + 1 : c;
    load_f(0);
    if (W_AHEAD) load_w(0);
    TL(3);
#pragma unroll
    for (int st = 0; st < 12; ++st) {
      const int m = st & 3;
      const int wc = W_AHEAD ? m : (m & 1), wp = W_AHEAD ? m - 1 : ((m - 1) & 1);
      if (!(SPLIT_DBG & 2)) {
        if (!W_AHEAD) load_w(st);
        if (st + 1 < 12) {
          load_f(st + 1);
          if (W_AHEAD) load_w(st + 1);
        }
      }
      // next chunk's global loads, one slot of each kind per step: a burst at the chunk start queues
      // all eight waves of the CU on the 64 B/clk vector-memory issue path (1700 of 8750 cycles per chunk)
      // (unconditional — the last chunk re-reads itself — so the loop body stays one basic block and the
      // compiler keeps exact vmcnt counts instead of draining the queue before every load)
      if (!(SPLIT_DBG & 8)) {
        if (st < AJ) gload_a(st, cnext);                        // halo slots early: split from step 12 - AJ on
        if (st >= 12 - BJ) gload_b(st - (12 - BJ), cnext);     // weight slots late: only the LDS store needs them
      }
#if !SPLIT_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (SPLIT_DBG & 1) continue;
          if (m < 3) acc[0][j] = mma(F[st & 1][PA[t]], Wr[wc][j][PW[t]], acc[0][j]);
          if (m > 0) acc[1][j] = mma(F[st & 1][PA[t]], Wr[wp][j][PW[t]], acc[1][j]);
        }
      if (st >= SPLIT_FROM && !(SPLIT_DBG & 4)) split_unit(st - SPLIT_FROM);
#if SPLIT_INTERLEAVE
      // one scheduling region per step: every MFMA is followed by one of the next step's fragment reads,
      // a staging load and a few of the split's VALU ops, so that a wave alone on its SIMD (its partner
      // storing to LDS or parked at a barrier) still feeds the matrix pipe back-to-back
#pragma unroll
      for (int t = 0; t < 2 * NPROD * TN; ++t) {  // (steps with one sub-tile leave the tail groups without an MFMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
        if (t < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);             // VALU
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (st == SPLIT_FROM - 1) TL(4);
    }
    TL(5);
    __syncthreads();
    TL(6);
  }
#if SPLIT_DBG & 32
  if (tid == 0 && blockIdx.x < SPLIT_TL_BLOCKS) {
    split_timeline[blockIdx.x][SPLIT_TL_CHUNKS][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    split_timeline[blockIdx.x][SPLIT_TL_CHUNKS][1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    split_timeline[blockIdx.x][SPLIT_TL_CHUNKS][2] = __builtin_amdgcn_s_getreg((31 << 11) | 6);   // LDS_ALLOC
  }
#endif

  // (r6) one uniform branch on "the activation is a GELU" in front of the epilogue: inside the per-element switch every accumulator
  // element carried its own copy of the erf GELU (see igemm.hip) - tens of KB of ISA for an activation these kernels' callers never ask for
  auto epilogue_r6 = [&](auto gelu_c) {
    constexpr bool GELU_EPI = decltype(gelu_c)::value;
    // ---- epilogue (same contract as the fp32 halo kernel) ----------------------------------------
    const float slope = (p.act == SEGMIF_ACT_PRELU) ? *p.prelu : 0.f;
    const long long img = (long long)b * p.H * p.W;
    // f16x3: the rows' own power-of-two scales sit behind the weight image (one float per padded output channel)
    const float* wdesc = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.wt) +
                                                        (long long)gridDim.y * nchunks * B_UNITS * 16);
    uint32_t amx = 0u;  // largest |output| this lane wrote (bit pattern), for the consumer's scale
  #pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = y0 + R0 + i * DIL;
      if (oy >= p.H) continue;
  #pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 32 + r;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        const float desc = F16 ? s_out * wdesc[n] : 1.f;
        // residual and mask values first, all sixteen of each in flight at once (unconditional loads from clamped columns):
        // fetched one by one between the stores they were a chain of exposed latencies at the end of a 25 us workgroup
        float rv[16], mv[16];
        const long long row = img + (long long)oy * p.W;
        if (p.res) {
  #pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int ox = min(x0 + (v & 3) + 8 * (v >> 2) + 4 * h, p.W - 1);
            rv[v] = p.res[(row + ox) * p.ldr + n];
          }
        }
        if (p.mask) {
  #pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int ox = min(x0 + (v & 3) + 8 * (v >> 2) + 4 * h, p.W - 1);
            mv[v] = p.mask[(row + ox) * p.ldm + n];
          }
        }
  #pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int ox = x0 + (v & 3) + 8 * (v >> 2) + 4 * h;
          if (ox >= p.W) continue;
          float y = (F16 ? acc[i][j][v] * desc : acc[i][j][v]) + bv;
          if constexpr (GELU_EPI) { y = gelu_exact(y); } else if (p.act == SEGMIF_ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == SEGMIF_ACT_PRELU) y = y >= 0.f ? y : slope * y;
          if (p.res) y += rv[v];
          if (p.mask && !(mv[v] > 0.f)) y = 0.f;  // DRDB backward: through the receiving block's ReLU
          p.out[(row + ox) * p.ldo + n] = y;
          amx = p16::absmax_bits(amx, y, 0.f);
        }
      }
    }
    if (p.out_amax) {
      // One atomic per WORKGROUP, spread over out_amax_n words by workgroup index (the consumer takes the maximum over all of
      // them).  (Measured against one atomic per wave on a single cold word at 8 x 480 x 640: no difference either way -
      // tools/split_conv_bench.py "1-word report"; kept because it cannot be worse and the slots are per-launch anyway.)
      amx = p16::wave_umax(amx);
      uint32_t* red = reinterpret_cast<uint32_t*>(smem_b);  // (every wave is past the chunk loop's closing barrier: the staging area is free)
      if (lane == 0) red[wave] = amx;
      __syncthreads();
      if (tid == 0) {
        uint32_t m = red[0];
  #pragma unroll
        for (int w = 1; w < NT / 64; ++w) m = red[w] > m ? red[w] : m;
        uint32_t* slot = p.out_amax + (blockIdx.x & (p.out_amax_n - 1));
        if (m > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, m);
      }
    }
  };
  if (p.act == SEGMIF_ACT_GELU) epilogue_r6(std::true_type{});
  else epilogue_r6(std::false_type{});
}

template <int NOUT, int DIL, int STH, bool F16>
int launch(const IgemmK& k, hipStream_t stream) {
  constexpr int HP = (STH + 2 * DIL) * (STW + 2 * DIL);
  constexpr size_t smem = (size_t)HP * (F16 ? ROWB_H : ROWB) + (size_t)9 * NOUT * ROWB + ((SPLIT_DBG & 64) && STH == 8 ? 40960 : 0);  // 64: one workgroup per CU
  auto fn = conv3x3_split_kernel<NOUT, DIL, STH, F16>;
  static segmif::PerDeviceFlag raised_flag;  // idempotent attribute; benign race
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const int tiles_x = (k.W + STW - 1) / STW, tiles_y = (k.H + STH - 1) / STH;
  const long long B = k.M / ((long long)k.H * k.W);
  dim3 grid((unsigned)(B * tiles_x * tiles_y), (unsigned)((k.N + NOUT - 1) / NOUT));
  hipLaunchKernelGGL(fn, grid, dim3(STH * 32), smem, stream, k, tiles_x, tiles_y);
  return (int)hipGetLastError();
}

// packed fp32 [N][ldw] (k = tap * Cin + c)  ->  [n-tile][chunk][tap][n][plane][16] bf16
__global__ void split_pack_kernel(const float* __restrict__ w, int N, int Cin, int ldw, int NOUT, long long total,
                                  uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nchunks = Cin / 16;
  const int c16 = (int)(idx & 15);
  long long t = idx >> 4;
  const int n = (int)(t % NOUT); t /= NOUT;
  const int tap = (int)(t % 9); t /= 9;
  const int chunk = (int)(t % nchunks);
  const int nt = (int)(t / nchunks);
  const int gn = nt * NOUT + n;
  const float x = gn < N ? w[(long long)gn * ldw + tap * Cin + chunk * 16 + c16] : 0.f;
  uint32_t p0, p1, p2;
  split3(x, 0.f, p0, p1, p2);
  const long long row = (((long long)nt * nchunks + chunk) * 9 + tap) * NOUT + n;
  out[row * 48 + c16] = (uint16_t)(p0 & 0xffffu);
  out[row * 48 + 16 + c16] = (uint16_t)(p1 & 0xffffu);
  out[row * 48 + 32 + c16] = (uint16_t)(p2 & 0xffffu);
}

// f16x3 weights: [n-tile][chunk][tap][n][W0 | Wl | 2^-11 W0][16] halves of the row scaled by 2^e(n), then one float per
// padded output channel: 2^-e(n)  (2^14 <= 2^e(n) max|w[n][.]| < 2^15; e = 0 for an all-zero row)
__global__ void split16_scale_kernel(const float* __restrict__ w, int N, int K, int ldw, int Npad, float* __restrict__ inv_scale) {
  const int n = blockIdx.x;
  float mx = 0.f;
  if (n < N)
    for (int k = threadIdx.x; k < K; k += 64) mx = fmaxf(mx, fabsf(w[(long long)n * ldw + k]));
  mx = p16::wave_max(mx);
  if (threadIdx.x == 0 && n < Npad) {
    int e = 0;
    if (mx >= 1e-30f && mx <= 3e38f) e = 14 - (int)((__float_as_uint(mx) >> 23) - 127);
    inv_scale[n] = ldexpf(1.f, -e);
  }
}

__global__ void split16_pack_kernel(const float* __restrict__ w, int N, int Cin, int ldw, int NOUT, long long total,
                                    const float* __restrict__ inv_scale, uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nchunks = Cin / 16;
  const int c16 = (int)(idx & 15);
  long long t = idx >> 4;
  const int n = (int)(t % NOUT); t /= NOUT;
  const int tap = (int)(t % 9); t /= 9;
  const int chunk = (int)(t % nchunks);
  const int nt = (int)(t / nchunks);
  const int gn = nt * NOUT + n;
  const float x = gn < N ? w[(long long)gn * ldw + tap * Cin + chunk * 16 + c16] * (1.f / inv_scale[gn]) : 0.f;  // exact: power of two
  const _Float16 w0 = (_Float16)x;
  const _Float16 wl = (_Float16)(x - (float)w0);
  const _Float16 ws = (_Float16)((float)w0 * (1.f / p16::LSCALE));
  const long long row = (((long long)nt * nchunks + chunk) * 9 + tap) * NOUT + n;
  out[row * 48 + c16] = __builtin_bit_cast(uint16_t, w0);
  out[row * 48 + 16 + c16] = __builtin_bit_cast(uint16_t, wl);
  out[row * 48 + 32 + c16] = __builtin_bit_cast(uint16_t, ws);
}

// max |x| of a rows view (rows x C, pitch ld) folded into one range slot (IEEE bit pattern; a NaN stays on top)
__global__ __launch_bounds__(256) void amax_rows_kernel(const float* __restrict__ x, long long rows, int c4n, int rb, int ld,
                                                        uint32_t* __restrict__ slot, int nslots) {
  __shared__ uint32_t red[4];
  const long long row0 = (long long)blockIdx.x * rb;
  const int nrows = (int)(rows - row0 < rb ? rows - row0 : rb);
  uint32_t m = 0u;
  for (int u = threadIdx.x; u < nrows * c4n; u += 256) {
    const int r = u / c4n, cq = u - r * c4n;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (row0 + r) * ld + 4 * cq);
    m = p16::absmax_bits(p16::absmax_bits(m, v[0], v[1]), v[2], v[3]);
  }
  // one atomic per block, spread over nslots words by block index (see the conv's epilogue)
  m = p16::wave_umax(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = red[0];
    for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
    uint32_t* s = slot + (blockIdx.x & (nslots - 1));
    if (m > __atomic_load_n(s, __ATOMIC_RELAXED)) atomicMax(s, m);
  }
}

}  // namespace

template <bool F16>
static int split_launch(const IgemmK& k, hipStream_t s) {
  const bool wide = split_nout(k.N) == 64;
  if (wide) return k.dil == 2 ? launch<64, 2, 16, F16>(k, s) : launch<64, 1, 16, F16>(k, s);
#if SPLIT_DBG & 128  // 16-row patches for NOUT = 32 too (one workgroup per CU)
  return k.dil == 2 ? launch<32, 2, 16, F16>(k, s) : launch<32, 1, 16, F16>(k, s);
#endif
  return k.dil == 2 ? launch<32, 2, 8, F16>(k, s) : launch<32, 1, 8, F16>(k, s);
}

int conv3x3_split_launch(const IgemmK& k, hipStream_t s) {
  if (k.split_f16) {
    if (!k.in_amax || k.in_amax_n <= 0 || k.in_amax_n > 64) return SEGMIF_EINVAL;  // the f16x3 form needs its input's range slots
  }
  if (k.out_amax && (k.out_amax_n < 1 || k.out_amax_n > 64 || (k.out_amax_n & (k.out_amax_n - 1)))) return SEGMIF_EINVAL;
  return k.split_f16 ? split_launch<true>(k, s) : split_launch<false>(k, s);
}

}  // namespace segmif

extern "C" int64_t segmif_conv3x3_split_weight_bytes(int N, int Cin) {
  if (N <= 0 || Cin <= 0 || Cin % 16) return 0;
  const int nout = segmif::split_nout(N);
  return (int64_t)((N + nout - 1) / nout) * nout * 9 * Cin * 6;
}

extern "C" int segmif_conv3x3_split_pack(const float* packed, int N, int Cin, int ldw, void* out, void* stream) {
  if (!packed || !out || N <= 0 || Cin <= 0 || Cin % 16 || ldw < 9 * Cin) return SEGMIF_EINVAL;
  const int nout = segmif::split_nout(N);
  const long long total = (long long)((N + nout - 1) / nout) * nout * 9 * Cin;
  hipLaunchKernelGGL(segmif::split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     packed, N, Cin, ldw, nout, total, (uint16_t*)out);
  return (int)hipGetLastError();
}

#if SPLIT_DBG & 32
extern "C" int segmif_debug_split_timeline(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(split_timeline), bytes < sizeof(split_timeline) ? bytes : sizeof(split_timeline));
}
#endif

extern "C" int64_t segmif_conv3x3_split16_weight_bytes(int N, int Cin) {
  const int64_t image = segmif_conv3x3_split_weight_bytes(N, Cin);  // same 96-byte rows, then one float per padded output channel
  if (!image) return 0;
  const int nout = segmif::split_nout(N);
  return image + (int64_t)((N + nout - 1) / nout) * nout * 4;
}

extern "C" int segmif_conv3x3_split16_pack(const float* packed, int N, int Cin, int ldw, void* out, void* stream) {
  if (!packed || !out || N <= 0 || Cin <= 0 || Cin % 16 || ldw < 9 * Cin) return SEGMIF_EINVAL;
  const int nout = segmif::split_nout(N);
  const int npad = (N + nout - 1) / nout * nout;
  const long long total = (long long)npad * 9 * Cin;
  float* inv_scale = reinterpret_cast<float*>((unsigned char*)out + total * 6);
  hipLaunchKernelGGL(segmif::split16_scale_kernel, dim3((unsigned)npad), dim3(64), 0, (hipStream_t)stream, packed, N, 9 * Cin, ldw, npad,
                     inv_scale);
  hipLaunchKernelGGL(segmif::split16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     packed, N, Cin, ldw, nout, total, inv_scale, (uint16_t*)out);
  return (int)hipGetLastError();
}

extern "C" int segmif_amax_f32(const float* x, int64_t rows, int C, int ld, uint32_t* slots, int nslots, void* stream) {
  uint32_t* slot = slots;
  if (!x || !slot || rows <= 0 || C <= 0 || (C & 3) || (ld & 3) || ld < C || ((uintptr_t)x & 15) || nslots < 1 || nslots > 64 ||
      (nslots & (nslots - 1)))
    return SEGMIF_EINVAL;
  const int c4n = C >> 2;
  const int rb = c4n >= 2048 ? 1 : 2048 / c4n;  // ~eight 16-byte units per thread
  hipLaunchKernelGGL(segmif::amax_rows_kernel, dim3((unsigned)((rows + rb - 1) / rb)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)rows, c4n, rb, ld, slot, nslots);
  return (int)hipGetLastError();
}
