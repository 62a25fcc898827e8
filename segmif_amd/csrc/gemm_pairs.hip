// Dense GEMM on f16x3 operands with BOTH operands pre-split and brought in by LDS-DMA (round 5): the MiT encoder's nn.Linear
// layers of stages 2-4 (core/mix_transformer.py:46-53 fc1 / fc2, :94-115 q / kv / proj) and Attention's spatial-reduction conv
// (:73-75, :98-101) in patch mode,
//     out = res + act(A W^T + bias),   A: (M, K) activations in PAIRS format,  W: (N, K).
//
// Why a second GEMM beside gemm_split.hip.  There a workgroup's life is: fp32 loads -> split in registers -> LDS stores ->
// barrier -> MFMAs -> barrier, single-buffered; the matrix pipe was 22 % busy (profiles/r04_pmc_sq_counters.txt) and the A tile
// is split again by every column tile that reads it.  Here the PRODUCER of A (LayerNorm, dwconv + GELU, the attention kernel)
// writes half pairs once - same byte count as fp32 - and the GEMM does no vector arithmetic on its operands at all:
//
//   PAIRS format: a row of K values = K / 16 groups of 64 bytes, [16 hi halves | 16 lo halves], x = hi + 2^-11 lo
//                 (planes16.h; range-guarded by the producer like every f16x3 tensor).
//
//   * tile 256 x 128 (8 waves of 64 x 64; WM = 2: 128 x 128, 4 waves, for problems too short to fill the chip), K step 16,
//     a ring of S = 3 stages of 24 KB (A 16 KB + W 8 KB): 72 KB per workgroup, TWO workgroups per CU - one's epilogue and
//     prologue under the other's K loop;
//   * A and W go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, 3 per wave per step), issued two
//     steps ahead; ONE workgroup barrier per step (s_waitcnt vmcnt on the stage about to be read, barrier, issue the stage that
//     was read last, multiply);
//   * LDS rows are 64 bytes = 4 sixteen-byte slots; LDS-DMA forces a lane-linear image, so the bank swizzle is applied to the
//     SOURCE address: slot j of row R lives at j ^ ((R >> 2) & 3), which puts the 16 lanes of every ds_read_b128 service group
//     on 16 distinct slots (tools/lds_bank_check.py --pairs);
//   * the weight image holds TWO planes per row (W0 | W - W0 of the row scaled by 2^e(n)); the third operand of the f16x3
//     product, 2^-11 W0, is made in registers (v_pk_mul_f16 by 2^-11 - exactly the value gemm_split's third plane stores:
//     half denormals are honoured), 8 vector instructions per 12 MFMAs instead of a third of the weight traffic;
//   * products least significant first: lo (2^-11 W0) + hi (W - W0) + hi W0, transposed (weights as the MFMA's row operand) so
//     that a lane owns an output row; the epilogue is gemm_split's (through a wave-private LDS tile, 256 contiguous bytes per
//     row and instruction).
// Per K step a workgroup moves 24 KB into LDS for 96 MFMAs = 768 CU cycles: 31 B/clk per CU, inside what LDS-DMA sustains
// beside a busy matrix pipe (~39 B/clk from eight issuing waves); gemm_split's 128 x 128 x 32 step needs 55.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "device_once.h"
#include "igemm_common.h"
#include "planes16.h"
#include "segmif_hip.h"

namespace segmif {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#ifndef PAIRS_DBG
#define PAIRS_DBG 0  // tuning aid: 1 = s_memtime timeline probe (tools/pairs_timeline.py)
#endif
#if PAIRS_DBG
__device__ unsigned long long pairs_timeline[1024][32][4];
#define PAIRS_TL(step, slot)                                                                                         \
  do {                                                                                                               \
    if (tid == 0 && blockIdx.x < 1024 && (step) < 32) pairs_timeline[blockIdx.x][step][slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PAIRS_TL(step, slot)
#endif

constexpr int PBK = 16;              // K step
constexpr int PNT = 128;             // column tile
constexpr int PROW = 64;             // bytes per LDS row (a 16-group of pairs: 2 planes x 16 halves)
constexpr int PWSTEP = PNT * PROW;   // weight bytes per (column tile, K step): 8 KB

struct GemmPairsK {
  const unsigned char* a;    // pairs rows, row pitch lda BYTES
  const unsigned char* w;    // [n-tile][k-step][128 rows][64 B], slots swizzled like the LDS image
  const float* wscale;       // 2^-e(n) per padded output column
  const float* bias;
  const float* res;
  const float* prelu;
  float* out;
  const unsigned char* zero;  // patch mode: 64 zero bytes (taps outside the image)
  long long M;
  long long lda;              // bytes
  int N, K, ldo, ldr, act;
  int ntm, ntn;
  // patch mode (patch_k > 0): A row (b, oy, ox) is the k x k patch at (st oy - pad, st ox - pad) of a dense NHWC pairs image
  // (B, H, W, C), C % 16 == 0, pixel pitch 4 C bytes; K = k k C in (ky, kx, c) order; a K step lies inside one tap
  int patch_k, patch_st, patch_pad, patch_H, patch_W, patch_OH, patch_OW, patch_C;
  int saddr;                  // (r6) plain rows whose byte offsets from `a` fit 32 bits: LDS-DMA in SADDR form (below)
};

// (r6) global_load_lds_dwordx4 in its SADDR form - a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane offset (see
// csrc/conv3x3_planes.hip, dma16s: per-lane 64-bit address pairs cost two VGPR reads per instruction and, advanced by vector adds, a
// write-after-read interlock on a vector-memory operand; here the lane offsets are constants of the kernel and the base moves by SALU)
__device__ __forceinline__ void dma16s(const unsigned char* sbase, uint32_t voff, unsigned char* lds_wave_base) {
  const uint32_t m = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m) : "memory", "m0");
}

__device__ __forceinline__ u32x4 times_2m11(const u32x4 v) {  // 8 halves x 2^-11 (v_pk_mul_f16; exact up to the half's own rounding)
  const f16x8 s = {(_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f,
                   (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f};
  return __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8, v) * s);
}

__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// The step barrier.  __syncthreads() is a workgroup-scope fence + s_barrier, and hipcc lowers the fence to s_waitcnt vmcnt(0)
// lgkmcnt(0): every LDS-DMA in flight - the stages issued AHEAD - would have to land before each barrier, which is the round trip
// the ring exists to hide.  So: wait for exactly the stage about to be read (all but the N most recent vector-memory
// instructions of this wave), drain the wave's own LDS reads, and the bare barrier (gfx950 backs off a barrier with memory
// operations outstanding; no implicit wait).  The "memory" clobber keeps the compiler from moving LDS accesses across it.
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// WM: waves along M (4: 256-row tile, 8 waves; 2: 128-row tile, 4 waves); S: stages of the LDS ring; PATCH: A rows are patches
// of an NHWC pairs image (its own instantiation: the plain-row K loop carries no tap bookkeeping)
template <int WM, int S, bool PATCH>
__global__ __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(WM == 4 ? 4 : 3))) void gemm_pairs_kernel(const GemmPairsK p) {
  constexpr int NW = WM * 2, MT = WM * 64;
  constexpr int STAGE = MT * PROW + PWSTEP;
  constexpr int APW = (MT / 16) / NW;   // A pieces (1 KB = 16 rows) per wave per step: 2
  constexpr int WPW = 8 / NW;           // W pieces per wave per step: 1 (8 waves) | 2 (4 waves)
  constexpr int P = APW + WPW;          // LDS-DMA instructions per wave per step
  static_assert(APW >= 1 && WPW >= 1 && S >= 3 && S <= 4, "tile / ring geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  {  // XCD-aware remap (as gemm_split): an XCD owns a contiguous run of tiles; column tiles of one row block are neighbours
    const int nwg = gridDim.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int mt = bid / p.ntn, nt = bid - mt * p.ntn;
  const long long m0 = (long long)mt * MT;
  const int n0 = nt * PNT;
  const int nks = p.K / PBK;

  // ---- LDS-DMA sources.  Piece pc of the A tile = rows 16 pc .. 16 pc + 15; lane -> (row 16 pc + (lane >> 2), physical slot
  // lane & 3), which holds logical slot (lane & 3) ^ ((row >> 2) & 3) of that row's 64 bytes.
  const unsigned char* a_src[APW];
  int a_iy[APW], a_ix[APW];
#pragma unroll
  for (int q = 0; q < APW; ++q) {
    const int row = (wave * APW + q) * 16 + (lane >> 2);
    const int j = (lane & 3) ^ ((row >> 2) & 3);
    long long m = m0 + row;
    m = m < p.M ? m : p.M - 1;  // rows past M repeat the last one (their results are never stored)
    if constexpr (PATCH) {
      const long long ohw = (long long)p.patch_OH * p.patch_OW;
      const long long b = m / ohw;
      const int rem = (int)(m - b * ohw);
      const int oy = rem / p.patch_OW, ox = rem - oy * p.patch_OW;
      a_iy[q] = oy * p.patch_st - p.patch_pad;
      a_ix[q] = ox * p.patch_st - p.patch_pad;
      a_src[q] = p.a + (((b * p.patch_H + a_iy[q]) * p.patch_W + a_ix[q]) * (long long)p.patch_C) * 4 + j * 16;
    } else {
      a_iy[q] = a_ix[q] = 0;
      a_src[q] = p.a + m * p.lda + j * 16;
    }
  }
  uint32_t a_off[APW];  // (SADDR form) the same sources as 32-bit offsets from p.a
#pragma unroll
  for (int q = 0; q < APW; ++q) a_off[q] = PATCH ? 0u : (uint32_t)(a_src[q] - p.a);
  const bool sa = !PATCH && p.saddr;  // (uniform)
  const unsigned char* w_src = p.w + (long long)nt * nks * PWSTEP + (wave * WPW) * 1024 + lane * 16;
  const unsigned char* zsrc = p.zero + (lane & 3) * 16;
  int t_ky = 0, t_kx = 0, t_c0 = 0;  // patch mode: tap and channel offset of the NEXT step to be issued (uniform)
  auto issue = [&](int ks, int buf) {
    unsigned char* base = smem_p + buf * STAGE;
    if constexpr (PATCH) {
      const long long off = (((long long)t_ky * p.patch_W + t_kx) * p.patch_C + t_c0) * 4;
#pragma unroll
      for (int q = 0; q < APW; ++q) {
        const bool ok = (unsigned)(a_iy[q] + t_ky) < (unsigned)p.patch_H && (unsigned)(a_ix[q] + t_kx) < (unsigned)p.patch_W;
        const unsigned char* src = ok ? a_src[q] + off : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + (wave * APW + q) * 1024), 16, 0, 0);
      }
      t_c0 += PBK;
      if (t_c0 == p.patch_C) {
        t_c0 = 0;
        if (++t_kx == p.patch_k) {
          t_kx = 0;
          ++t_ky;
        }
      }
    } else if (sa) {
#pragma unroll
      for (int q = 0; q < APW; ++q) dma16s(p.a + (long long)ks * PROW, a_off[q], base + (wave * APW + q) * 1024);
    } else {
#pragma unroll
      for (int q = 0; q < APW; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (long long)ks * PROW),
                                         (__attribute__((address_space(3))) void*)(base + (wave * APW + q) * 1024), 16, 0, 0);
    }
    if (sa) {
      const unsigned char* wb = p.w + (long long)nt * nks * PWSTEP + (wave * WPW) * 1024 + (long long)ks * PWSTEP;  // (uniform)
#pragma unroll
      for (int q = 0; q < WPW; ++q) dma16s(wb + q * 1024, (uint32_t)(lane * 16), base + MT * PROW + (wave * WPW + q) * 1024);
    } else {
#pragma unroll
      for (int q = 0; q < WPW; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + (long long)ks * PWSTEP + q * 1024),
                                         (__attribute__((address_space(3))) void*)(base + MT * PROW + (wave * WPW + q) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // fragment addresses inside a stage: row (wm 64 + i 32 + r) -> swizzle (r >> 2) & 3 (the tile offsets are multiples of 32)
  const int sw = (r >> 2) & 3;
  const int off_hi = ((h ^ sw) << 4), off_lo = (((2 | h) ^ sw) << 4);
  const int a_lane = (wm * 64 + r) * PROW, w_lane = MT * PROW + (wn * 64 + r) * PROW;
  const bool active = n0 + wn * 64 < p.N;  // (uniform) a wave whose 64 columns are all padding multiplies nothing

  PAIRS_TL(29, 0);
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nks) issue(s, s);
  PAIRS_TL(29, 1);
  int buf = 0;
  for (int ks = 0; ks < nks; ++ks) {
    // the stage about to be read has landed: everything this wave issued except the (at most S - 2) later stages
    const int later = nks - 1 - ks;
    PAIRS_TL(ks, 0);
    if (later >= S - 2) wait_vm_barrier<P * (S - 2)>();
    else if (S == 4 && later == 1) wait_vm_barrier<P>();
    else wait_vm_barrier<0>();  // ... and everybody's; all waves are done reading the stage of step ks - 1
    PAIRS_TL(ks, 1);
    if (ks + S - 1 < nks) issue(ks + S - 1, buf == 0 ? S - 1 : buf - 1);
    PAIRS_TL(ks, 2);
    if (active) {
      const unsigned char* sb = smem_p + buf * STAGE;
      u32x4 ah[2], al[2], w0[2], wl[2], ws[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const u32x4*>(sb + a_lane + i * 32 * PROW + off_hi);
        al[i] = *reinterpret_cast<const u32x4*>(sb + a_lane + i * 32 * PROW + off_lo);
        w0[i] = *reinterpret_cast<const u32x4*>(sb + w_lane + i * 32 * PROW + off_hi);
        wl[i] = *reinterpret_cast<const u32x4*>(sb + w_lane + i * 32 * PROW + off_lo);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) ws[j] = times_2m11(w0[j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(ws[j], al[i], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(wl[j], ah[i], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(w0[j], ah[i], acc[i][j]);
    }
    PAIRS_TL(ks, 3);
    buf = buf + 1 == S ? 0 : buf + 1;
  }
  PAIRS_TL(30, 0);
  __syncthreads();  // the ring is free: the epilogue stages its rows there
  PAIRS_TL(30, 1);

  // ---- epilogue.  The products ran transposed (a lane owns an output ROW), so the accumulators go through a wave-private
  // [32][68] LDS tile and leave as 4 rows x 256 contiguous bytes per instruction (gemm_split's scheme).  Two things differ,
  // both about vmcnt counting stores as well as loads on gfx9: a load issued AFTER a store cannot be waited for without
  // draining that store (tools/pairs_timeline.py: the first version's epilogue was 34 % of a workgroup's life, a chain of such
  // drains).  (1) Row scale, bias and activation are applied AFTER the transposition, where a lane's four columns are the
  // same for every row: two 16-byte loads per lane for the whole tile, issued before any store.  (2) The residual rows of BOTH
  // halves are requested before the first store (64 registers: half of the accumulators are dead by then).
  const float slope = (p.act == SEGMIF_ACT_PRELU) ? *p.prelu : 0.f;
  float* T = reinterpret_cast<float*>(smem_p) + wave * (32 * 68);
  const int ecl = (lane & 15) * 4, en = n0 + wn * 64 + ecl;  // this lane's four columns in the store phase
  const bool en_ok = en < p.N;
  const f32x4 esc = *reinterpret_cast<const f32x4*>(p.wscale + en);  // (the scale array covers the padded columns)
  const f32x4 ebi = (p.bias && en_ok) ? *reinterpret_cast<const f32x4*>(p.bias + en) : f32x4{0.f, 0.f, 0.f, 0.f};
  // (r6) The case every Linear of the encoder is - no activation - gets a straight-line epilogue of its own (with / without the
  // residual), chosen by ONE uniform branch.  Written with `p.act` and `p.res` tested where they are used, each of a lane's 64
  // elements carried the whole activation switch - 64 copies of the erf GELU behind ~480 uniform branches, 7 000 lines of ISA
  // (45 KB: most of the 64 KB instruction cache two CUs share) of which such a Linear executed a fraction, one taken branch every
  // few instructions; the epilogue was 25 % of a workgroup's life (profiles/r05_gemm_pairs_timeline.txt).  The general code
  // stays for the activations.
  auto epilogue = [&](auto lean_c, auto res_c) {
    constexpr bool LEANE = decltype(lean_c)::value;  // no activation: the case every Linear of the encoder is
    constexpr bool RESC = decltype(res_c)::value;    // (LEANE only) residual known at compile time
    const bool has_res = LEANE ? RESC : (p.res != nullptr);
    f32x4 rres[LEANE ? 2 : 1][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(T + r * 68 + j * 32 + 8 * g + 4 * h) =
              f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
      if constexpr (LEANE) __builtin_amdgcn_sched_barrier(0);  // (the residual rows only after the first half's accumulators are in LDS, i.e. dead: 64 + 64 registers do not fit)
      if (LEANE && i == 0 && has_res) {  // (uniform) both halves' residual rows, while nothing has been stored yet
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const long long m = m0 + wm * 64 + ii * 32 + t * 4 + (lane >> 4);
            rres[LEANE ? ii : 0][t] = (m < p.M && en_ok) ? *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + en) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: the wave's own LDS writes have landed
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = t * 4 + (lane >> 4);
        const long long m = m0 + wm * 64 + i * 32 + row;
        f32x4 y = *reinterpret_cast<const f32x4*>(T + row * 68 + ecl) * esc + ebi;
        if constexpr (!LEANE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (p.act == SEGMIF_ACT_RELU) y[e] = fmaxf(y[e], 0.f);
            else if (p.act == SEGMIF_ACT_PRELU) y[e] = y[e] >= 0.f ? y[e] : slope * y[e];
            else if (p.act == SEGMIF_ACT_GELU) y[e] = gelu_exact(y[e]);
          }
        }
        if constexpr (LEANE) {
          if (has_res) y += rres[LEANE ? i : 0][t];
        } else if (has_res && m < p.M && en_ok) {  // (an activation AND a residual: no caller on the measured path; loaded where it is used)
          y += *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + en);
        }
        if (m < p.M && en_ok) *reinterpret_cast<f32x4*>(p.out + m * p.ldo + en) = y;
        if constexpr (LEANE) {
          if (t & 1) __builtin_amdgcn_sched_barrier(0);  // (two rows in flight: branch-free, the scheduler would hoist all eight tile reads and spill)
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next half overwrites the tile
    }
  };
  if (p.act == SEGMIF_ACT_NONE) {
    if (p.res) epilogue(std::true_type{}, std::true_type{});
    else epilogue(std::true_type{}, std::false_type{});
  } else {
    epilogue(std::false_type{}, std::false_type{});
  }
#if PAIRS_DBG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PAIRS_TL(31, 0);
#endif
}

// f16x3 row scale 2^-e(n): 2^14 <= 2^e max |w[n][.]| < 2^15 (1 for vanishing rows and for the padding rows)
__global__ void gemm_pairs_scale_kernel(const float* __restrict__ w, int N, int K, int ldw, float* __restrict__ inv_scale) {
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

// fp32 [N][ldw] -> [n-tile][k-step][128 rows][4 slots x 8 halves]: logical slots W0[0..7] | W0[8..15] | Wl[0..7] | Wl[8..15] of
// the scaled row (Wl = W - W0), slot j stored at j ^ ((row >> 2) & 3); zero filled past N / K
__global__ void gemm_pairs_pack_kernel(const float* __restrict__ w, int N, int K, int ldw, int nks, long long total,
                                       const float* __restrict__ inv_scale, uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kk = (int)(idx & 15);
  long long t = idx >> 4;
  const int row = (int)(t & 127); t >>= 7;
  const int ks = (int)(t % nks);
  const int nt = (int)(t / nks);
  const int n = nt * PNT + row, k = ks * PBK + kk;
  const float x = (n < N && k < K) ? w[(long long)n * ldw + k] * (1.f / inv_scale[n]) : 0.f;  // exact: power of two
  const _Float16 w0 = (_Float16)x;
  const _Float16 wl = (_Float16)(x - (float)w0);
  const int sw = (row >> 2) & 3;
  uint16_t* dst = out + (((long long)nt * nks + ks) * PNT + row) * (PROW / 2);
  dst[(((kk >> 3)) ^ sw) * 8 + (kk & 7)] = __builtin_bit_cast(uint16_t, w0);
  dst[((2 | (kk >> 3)) ^ sw) * 8 + (kk & 7)] = __builtin_bit_cast(uint16_t, wl);
}

// fp32 rows -> pairs rows (a producer for tensors whose own kernel has no pairs epilogue; tests).  One thread = 4 values.
__global__ __launch_bounds__(256) void pairs_from_f32_kernel(const float* __restrict__ x, long long ldx, unsigned char* __restrict__ y,
                                                             long long ldy, long long rows, int C, uint32_t* __restrict__ amax,
                                                             long long amax_rows) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = idx / c4n;
  const int c = (int)(idx - row * c4n) * 4;
  uint32_t amx = 0u;
  if (row < rows) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
    uint32_t ha, la, hb, lb;
    p16::split2(v[0], v[1], ha, la);
    p16::split2(v[2], v[3], hb, lb);
    unsigned char* dst = y + row * ldy + (c >> 4) * 64 + (c & 15) * 2;
    *reinterpret_cast<u32x2*>(dst) = u32x2{ha, hb};
    *reinterpret_cast<u32x2*>(dst + 32) = u32x2{la, lb};
    amx = p16::absmax_pk(p16::absmax_pk(amx, ha, la), hb, lb);
  }
  if (amax) {  // a wave's 64 threads cover 256 consecutive values: at most two rows' images when C >= 128 ... report per row range
    const long long r0 = ((long long)blockIdx.x * 256 + (threadIdx.x & ~63)) / c4n;
    long long r1 = ((long long)blockIdx.x * 256 + (threadIdx.x | 63)) / c4n;
    if (r1 >= rows) r1 = rows - 1;
    if (r0 < rows) p16::fold_pat(amax, (int)(r0 / amax_rows), (int)(r1 / amax_rows), amx);
  }
}

__global__ __launch_bounds__(256) void pairs_to_f32_kernel(const unsigned char* __restrict__ x, long long ldx, float* __restrict__ y,
                                                           long long ldy, long long rows, int C) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = idx / c4n;
  const int c = (int)(idx - row * c4n) * 4;
  if (row >= rows) return;
  const unsigned char* src = x + row * ldx + (c >> 4) * 64 + (c & 15) * 2;
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
  const p16::h2 a0 = __builtin_bit_cast(p16::h2, s32[0]), a1 = __builtin_bit_cast(p16::h2, s32[1]);
  const p16::h2 b0 = __builtin_bit_cast(p16::h2, s32[8]), b1 = __builtin_bit_cast(p16::h2, s32[9]);
  constexpr float inv = 1.f / p16::LSCALE;
  // (written out: the two-iteration loop over hi[e] / lo[e] this replaces was compiled to ONE dword load per plane and left
  // values 2, 3 of every quad undefined - found by the round trip of tools/pairs_debug.py)
  const f32x4 o = {fmaf((float)b0[0], inv, (float)a0[0]), fmaf((float)b0[1], inv, (float)a0[1]),
                   fmaf((float)b1[0], inv, (float)a1[0]), fmaf((float)b1[1], inv, (float)a1[1])};
  *reinterpret_cast<f32x4*>(y + row * ldy + c) = o;
}

}  // namespace
}  // namespace segmif

using namespace segmif;

#if PAIRS_DBG
extern "C" int segmif_debug_pairs_timeline(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pairs_timeline), bytes < sizeof(pairs_timeline) ? bytes : sizeof(pairs_timeline));
}
#endif

extern "C" int64_t segmif_gemm_pairs_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % PBK) return 0;
  const int64_t ntn = (N + PNT - 1) / PNT;
  return ntn * (K / PBK) * PWSTEP + ntn * PNT * 4;  // the image + one float 2^-e(n) per padded output column
}

extern "C" int segmif_gemm_pairs_pack(const float* w, int N, int K, int ldw, void* out, void* stream) {
  if (!w || !out || segmif_gemm_pairs_weight_bytes(N, K) == 0 || ldw < K || ((uintptr_t)out & 15)) return SEGMIF_EINVAL;
  const int nks = K / PBK, npad = (N + PNT - 1) / PNT * PNT;
  const long long total = (long long)(npad / PNT) * nks * PNT * PBK;
  float* inv_scale = reinterpret_cast<float*>((unsigned char*)out + (int64_t)(npad / PNT) * nks * PWSTEP);
  hipLaunchKernelGGL(gemm_pairs_scale_kernel, dim3((unsigned)npad), dim3(64), 0, (hipStream_t)stream, w, N, K, ldw, inv_scale);
  hipLaunchKernelGGL(gemm_pairs_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K,
                     ldw, nks, total, inv_scale, (uint16_t*)out);
  return (int)hipGetLastError();
}

extern "C" int segmif_pairs_from_f32(const float* x, int64_t ldx, void* y, int64_t ldy_bytes, int64_t rows, int C, uint32_t* amax,
                                     int amax_images, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 15) || (ldx & 3) || ldx < C || ldy_bytes < 4LL * C || (ldy_bytes & 15) ||
      (((uintptr_t)x | (uintptr_t)y) & 15))
    return SEGMIF_EINVAL;
  if (amax && (amax_images < 1 || rows % amax_images)) return SEGMIF_EINVAL;
  const long long total = rows * (C >> 2);
  hipLaunchKernelGGL(pairs_from_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx,
                     (unsigned char*)y, (long long)ldy_bytes, (long long)rows, C, amax, (long long)(rows / (amax ? amax_images : 1)));
  return (int)hipGetLastError();
}

extern "C" int segmif_pairs_to_f32(const void* x, int64_t ldx_bytes, float* y, int64_t ldy, int64_t rows, int C, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 15) || (ldy & 3) || ldy < C || ldx_bytes < 4LL * C || (ldx_bytes & 15) ||
      (((uintptr_t)x | (uintptr_t)y) & 15))
    return SEGMIF_EINVAL;
  const long long total = rows * (C >> 2);
  hipLaunchKernelGGL(pairs_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)x, (long long)ldx_bytes, y, (long long)ldy, (long long)rows, C);
  return (int)hipGetLastError();
}

__device__ __attribute__((aligned(256))) unsigned char g_pairs_zero_page[256];  // all zero, never written

extern "C" int segmif_gemm_pairs_f32(const SegmifGemmPairs* d, void* stream) {
  if (!d || !d->a || !d->w || !d->out || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->K % PBK) return SEGMIF_EINVAL;
  const bool patch = d->patch_k > 0;
  if (((uintptr_t)d->a & 15) || ((uintptr_t)d->w & 15)) return SEGMIF_EINVAL;
  if (!patch && (d->lda_bytes < 4LL * d->K || (d->lda_bytes & 15))) return SEGMIF_EINVAL;
  int poh = 0, pow_ = 0;
  if (patch) {  // K = k * k * C, C a multiple of the K step (a step never straddles two taps); pixel pitch 4 C bytes
    const int kk = d->patch_k, st = d->patch_st, pad = d->patch_pad, C = d->patch_C;
    if (st < 1 || pad < 0 || pad >= kk || C <= 0 || C % PBK || d->patch_H + 2 * pad < kk || d->patch_W + 2 * pad < kk ||
        d->K != kk * kk * C)
      return SEGMIF_EINVAL;
    poh = (d->patch_H + 2 * pad - kk) / st + 1;
    pow_ = (d->patch_W + 2 * pad - kk) / st + 1;
    if (d->M % ((long long)poh * pow_)) return SEGMIF_EINVAL;
  }
  if (d->act == SEGMIF_ACT_PRELU && !d->prelu) return SEGMIF_EINVAL;
  if (d->res && (d->ldr <= 0 || (d->ldr & 3) || ((uintptr_t)d->res & 15))) return SEGMIF_EINVAL;
  if ((d->N & 3) || (d->ldo & 3) || d->ldo < d->N || ((uintptr_t)d->out & 15) || ((uintptr_t)d->bias & 15)) return SEGMIF_EINVAL;
  // 64 zero bytes on this device (patch mode's out-of-image taps): a zero-initialised __device__ array - (r6) no
  // hipMalloc / null-stream hipMemset inside a launch function (unordered against a non-blocking stream, illegal under capture)
  static segmif::PerDeviceValue<unsigned char*> zero_page;
  unsigned char*& zp = zero_page.here();
  if (!zp) {
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_pairs_zero_page));
    if (e != hipSuccess || !sym) return e != hipSuccess ? (int)e : SEGMIF_EINVAL;
    zp = (unsigned char*)sym;
  }
  GemmPairsK k;
  k.a = (const unsigned char*)d->a; k.w = (const unsigned char*)d->w; k.bias = d->bias; k.res = d->res; k.prelu = d->prelu;
  k.out = d->out; k.zero = zp;
  k.M = d->M; k.N = d->N; k.K = d->K; k.lda = d->lda_bytes; k.ldo = d->ldo; k.ldr = d->ldr; k.act = d->act;
  {  // SADDR form of the LDS-DMA: every A byte the kernel touches within 2^32 of `a` (SEGMIF_GEMM_PAIRS_SADDR=0: the builtin's form, for A/B)
    static const bool on = [] { const char* e = getenv("SEGMIF_GEMM_PAIRS_SADDR"); return !(e && e[0] == '0'); }();
    k.saddr = on && (unsigned long long)d->M * (unsigned long long)d->lda_bytes < (1ull << 32) ? 1 : 0;
  }
  k.ntn = (d->N + PNT - 1) / PNT;
  k.wscale = reinterpret_cast<const float*>(k.w + (int64_t)k.ntn * (d->K / PBK) * PWSTEP);
  k.patch_k = patch ? d->patch_k : 0; k.patch_st = d->patch_st; k.patch_pad = d->patch_pad;
  k.patch_H = d->patch_H; k.patch_W = d->patch_W; k.patch_OH = poh; k.patch_OW = pow_; k.patch_C = d->patch_C;
  // tile height: 256 rows while that still gives every CU its two workgroups; 128 rows for shorter problems (SEGMIF_GEMM_PAIRS_MT
  // = 128 | 256 forces one, read once per process: a tuning aid)
  static const int mt_env = [] { const char* e = getenv("SEGMIF_GEMM_PAIRS_MT"); return e ? atoi(e) : 0; }();
  int tile = d->tile_rows ? d->tile_rows : mt_env;
  if (tile != 128 && tile != 256) tile = ((d->M + 255) / 256) * k.ntn >= 512 ? 256 : 128;
  constexpr size_t smem256 = 3 * (size_t)(256 * PROW + PWSTEP), smem128 = 3 * (size_t)(128 * PROW + PWSTEP);
  static_assert(smem256 >= 8 * 32 * 68 * sizeof(float) && smem128 >= 4 * 32 * 68 * sizeof(float), "the epilogue tiles must fit the ring");
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pairs_kernel<4, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem256);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_pairs_kernel<4, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem256);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_pairs_kernel<2, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_pairs_kernel<2, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (tile == 256) {
    k.ntm = (int)((d->M + 255) / 256);
    const dim3 grid((unsigned)((long long)k.ntm * k.ntn));
    if (patch) hipLaunchKernelGGL((gemm_pairs_kernel<4, 3, true>), grid, dim3(512), smem256, st, k);
    else hipLaunchKernelGGL((gemm_pairs_kernel<4, 3, false>), grid, dim3(512), smem256, st, k);
  } else {
    k.ntm = (int)((d->M + 127) / 128);
    const dim3 grid((unsigned)((long long)k.ntm * k.ntn));
    if (patch) hipLaunchKernelGGL((gemm_pairs_kernel<2, 3, true>), grid, dim3(256), smem128, st, k);
    else hipLaunchKernelGGL((gemm_pairs_kernel<2, 3, false>), grid, dim3(256), smem128, st, k);
  }
  return (int)hipGetLastError();
}
