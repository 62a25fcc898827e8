// DRDB dilated 3x3 convs (core/model_fusion.py:121-157) on the bf16 matrix pipe with 3-way split
// operands ("bf16x6", numerics as in conv3x3_split.hip), reading activations that are ALREADY split.
//
// Why a second kernel.  conv3x3_split.hip keeps fp32 activations in HBM and splits them on their way to
// LDS: a DRDB re-reads channels [0, 64 + 32 i) in conv i, so every element is split ~3 times over,
// times the 1.69x halo overlap, and the split + VGPR -> LDS store phase is what kept the matrix pipe at
// 61 % (profiles/r01_pmc_sq_counters_bf16x6.txt).  Here the producer writes the three bf16 planes once
// ("planes" image, below) and a conv does no vector arithmetic at all on its input: the halo and the
// weights go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write), the
// waves only issue ds_read_b128 + MFMA (schedule: see the kernel).
//
// Planes image (include/segmif_hip.h, segmif_planes_*):  [b][chunk][Hp][Wp][plane 0..2][16] bf16, one
// 16-channel chunk per image, Hp = ceil(H/8)*8 + 4, Wp = ceil(W/32)*32 + 4: a zero border of 2 pixels plus
// the round-up to whole 8 x 32 patches, so a halo row is one contiguous run of (32 + 2d) * 96 bytes
// and needs no bounds logic.  Position j of a chunk holds channel 16*chunk + sigma(j),
// sigma(j) = (j & 3) + 4 (j >> 3) + 8 ((j >> 2) & 1): the order in which the 32x32 MFMA accumulator
// hands a lane its 8 values of a 16-row block, so an epilogue writes 16 contiguous bytes per plane and —
// in the fused tail — re-uses its accumulators as the next MFMA's B operand without any data movement.
//
// LDS image: [pixel][96 B] with NO padding (41.5 + 27 KB per workgroup; the padded 112-byte pitch of
// conv3x3_split.hip cannot be written by a lane-linear DMA).  A 96-byte pitch alone is a 2-way
// ds_read_b128 bank conflict; the image is therefore built with the two 16-byte halves of every plane
// swapped in halo columns with bit 3 set (weights: rows with bit 4 set) — the DMA's per-lane SOURCE
// address does that for free — and a lane reads half h ^ f.  For every 16-lane service group of
// ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) and every tap offset this puts the 16 lanes
// on 16 distinct 16-byte slots of the 256-byte bank row.
//
// Fused tail (FUSE): the DRDB's closing 1x1 conv 224 -> 64 + ReLU + residual (ref :155-157) rides on the
// fifth dilated conv: its input at a pixel is that conv's centre-tap fragment (chunks 0..11, already in
// LDS) plus the 32 channels the conv itself produces (in the accumulators), so 12 + 24 extra MFMAs per
// sub-tile replace a separate HBM-bound pass over the 224-channel buffer.
//
// Second operand format, F16 = true ("f16x3", segmif_planes16_* / segmif_conv3x3_planes_f16x3): an activation is a PAIR of
// halves, x = x0 + 2^-11 l  with  x0 = RN16(x), l = RN16(2^11 (x - x0))  — 11 + 1 + 11 significand bits, within one bit
// of fp32 for 2^-13 <= |x| < 65504 —, a weight row is scaled by a power of two 2^e(n) so that its largest entry lies in
// [2^14, 2^15) and stored as three halves W0 = RN16(W), Wl = RN16(W - W0), W0s = 2^-11 W0; THREE products per fp32
// product, x0 W0 + x0 Wl + l W0s, the epilogue multiplies by 2^-e(n).  Half the matrix work and 2/3 of the activation
// bytes of bf16x6 at the same error level (tests/test_gpu_kernels.py holds both to the exact-fp32 MFMA kernel's error),
// but only inside the half's exponent range: every producer records max |x| of what it wrote (atomic max on the bit
// pattern, `amax`), and the host re-runs a forward whose planes left [2^-13, 65504) on the bf16x6 kernels.
// LDS image: [pixel][64 B] = 4 pieces of 16 bytes (2 * plane + half); piece q of halo column c sits in slot
// q ^ ((c >> 2) & 3), which puts the 16 lanes of a ds_read_b128 service group on 16 distinct slots of a bank row.
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>
#include <stdlib.h>

#include "igemm_common.h"
#include "planes16.h"
#include "segmif_hip.h"

#ifndef PLANES_DBG
#define PLANES_DBG 0  // tuning aid: 32 = s_memtime timeline probe (tools/planes_timeline.py), 64 = no planes stores, 128 = s_setprio 1 around the MFMA block; fused tail's epilogue: 256 = no out1 stores, 512 = no residual, 1024 = no 1x1 MFMAs on the conv's own channels, 2048 = no epilogue at all
#endif
#ifndef PLANES_DEFER
#define PLANES_DEFER 2  // (r6) plain f16x3 four-sub-tile conv: a patch's plane stores leave in the team's NEXT LOAD phase (1: before its DMA burst, 2: after it); 0: at the end of the COMPUTE phase
#endif
#if PLANES_DBG & 32
#define PLANES_TL_ITEMS 64
__device__ unsigned long long planes_timeline[256][2][PLANES_TL_ITEMS][8];
#define TL(slot)                                                                                   \
  do {                                                                                             \
    if (wave == 0 && lane == 0 && blockIdx.x < 256 && i < PLANES_TL_ITEMS)                         \
      planes_timeline[blockIdx.x][team][i][slot] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
// stamps INSIDE the epilogue (slots: 0 entry | 1 constants in registers | 2 sub-tile 0 activated | 3 its 1x1 MFMAs done | 4 its out1
// values ready | 5 its stores issued | 6 sub-tile 1 done); `keep` = registers the stamp must come after
__device__ unsigned long long planes_epi_timeline[256][2][PLANES_TL_ITEMS][8];
#define ETL(slot, ...)                                                                             \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    asm volatile("" ::__VA_ARGS__);                                                                \
    if (wave == 0 && lane == 0 && blockIdx.x < 256 && item < PLANES_TL_ITEMS)                      \
      planes_epi_timeline[blockIdx.x][team][item][slot] = __builtin_amdgcn_s_memtime();           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define TL(slot) do {} while (0)
#define ETL(slot, ...) do {} while (0)
#endif

namespace segmif {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int PB = 2;            // zero border of a planes image (pixels) = the largest dilation served
constexpr int TH = 8, TW = 32;   // output patch of a workgroup
constexpr int PXB = 96;          // bytes per pixel per chunk: 3 planes x 16 bf16 (also a weight row in both formats)
constexpr int PXH = p16::PIXEL_BYTES;  // f16x3: bytes per pixel per chunk, 2 planes x 16 halves
constexpr float LSCALE = p16::LSCALE;  // f16x3: the low half carries 2^11 x the residual
constexpr int W3_BYTES = 9 * 32 * PXB;   // one chunk of 3x3 weights, 32 output channels
constexpr int W1_BYTES = 64 * PXB;       // one chunk of the fused 1x1 weights, 64 output channels

inline int planes_hp(int H) { return (H + TH - 1) / TH * TH + 2 * PB; }
inline int planes_wp(int W) { return (W + TW - 1) / TW * TW + 2 * PB; }

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

// 8 fp32 values (positions 8h .. 8h+7 of a chunk) -> one 16-byte piece per plane
__device__ __forceinline__ void split8(const float* y, u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t a, b, c;
    split3(y[2 * e], y[2 * e + 1], a, b, c);
    p0[e] = a;
    p1[e] = b;
    p2[e] = c;
  }
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma_split(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// x = hi + 2^-11 lo of element E (0 | 1) of a dword of high halves and the matching dword of low halves: ONE v_fma_mix_f32 (the halves
// are read in place - op_sel picks the half, op_sel_hi marks the f16 sources; exact in fp32) instead of two conversions and an fma
template <int E>
__device__ __forceinline__ float pair_value(uint32_t hi2, uint32_t lo2) {
  float d;
  const float c = 1.f / LSCALE;
  if constexpr (E == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(lo2), "v"(c), "v"(hi2));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(lo2), "v"(c), "v"(hi2));
  return d;
}

__host__ __device__ inline int sigma16(int j) { return (j & 3) + 4 * (j >> 3) + 8 * ((j >> 2) & 1); }

struct PlanesConvK {
  const unsigned char* pin;   // planes input  [B][in_total][Hp][Wp][96]
  unsigned char* pout;        // planes output [B][out_total][Hp][Wp][96] (may be the same buffer) or null
  const unsigned char* wt;    // [chunk][tap][32][96] split, half-swapped
  const float* bias;
  const float* prelu;
  float* out;                 // optional fp32 copy of the 32 outputs, pitch ldo
  int ldo;
  int B, H, W, Hp, Wp;
  int nchunks, in_total, out_total, out_chunk0;
  int act;
  const unsigned char* w1;    // fused 1x1: [nchunks + 2][64][96]
  const float* bias1;
  const float* res;
  int res_planes;             // f16x3 fused tail: the residual is the conv's own input chunks 0..3 (hi + 2^-11 lo), not an fp32 tensor
  float* out1;
  int ldr, ldo1, act1;
  int tiles_x, tiles_y;
  const float* wscale;        // f16x3: 2^-e(n) of the 32 conv rows / the 64 rows of the fused 1x1
  const float* w1scale;
  uint32_t* amax;             // f16x3: range slots of max |conv output| (or null), one per image when amax_images == B
  int amax_images;            // 1: every image reports to amax[0]
};

__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#ifndef PLANES_SADDR
#define PLANES_SADDR 1  // (r6) LDS-DMA with a scalar base + a 32-bit lane offset (no address arithmetic per instruction)
#endif
// (r6) The same instruction in its SADDR form: a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane offset.  From the builtin
// hipcc makes, per DMA instruction, a v_lshl_add_u64 into ONE shared 64-bit address register pair and the load from it - the next
// instruction's address write has to wait until the load in front has read that pair (a write-after-read interlock on a
// vector-memory operand: tens to hundreds of cycles beside the other team's MFMA stream), and every address costs two VGPR reads.
// Here the offsets sit in their own registers for the kernel's life and a DMA instruction is an s_mov of M0 and the load.
__device__ __forceinline__ void dma16s(const unsigned char* sbase, uint32_t voff, unsigned char* lds_wave_base) {
  const uint32_t m = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m) : "memory", "m0");
}

// One persistent workgroup per CU, 8 waves = two TEAMS of four (one wave per SIMD each).  A team owns one 8 x 32
// patch at a time and alternates between two roles, the teams in anti-phase, one workgroup barrier per phase:
//   LOAD     issue the LDS-DMA of its next (patch, chunk) item — team 0 also the chunk's weights, which both teams
//            use —, run the epilogue of a patch it has just finished, wait for its DMA;
//   COMPUTE  12 steps of ds_read_b128 + MFMA on the item loaded in its previous phase.
// While one team waits on memory (a wave blocked in vector-memory issue cannot feed the matrix pipe) the other has
// every SIMD to itself.  Two independent workgroups per CU drift through all relative phases instead (measured: mean
// |lag| 0.28 of a period, both in their MFMA segment 68 % of the time, both issuing DMA into the 64 B/clk L1 path at
// once the rest — profiles/r02_planes_timeline.txt); the barrier pins the phase.  Item i of team 0 is loaded in phase
// 2i and multiplied in phase 2i + 1; team 1 runs one phase later and finds the weights of item i still in W[i & 1].
// SUB = sub-tiles (patch rows) per wave: 2 (a team's patch is 8 x 32) or 4 (16 x 32: the plain f16x3 conv on heights that
// are whole 16-row patches - 0.58 fragment reads per MFMA instead of 0.94, halo overhead 1.41x instead of 1.69x).
// LEAN (r6): the instantiations an inference forward runs - f16x3, no fp32 copy of the conv's output; plain conv: planes out, any
// activation; fused tail: ReLU after the conv and after the 1x1, residual from the input planes, no planes out.  Its epilogue
// carries none of the other cases' (uniform, but per-element and per-store) branches - 53 s_cbranch in the general tail - and its
// ReLU is ONE instruction: biases and row scales are parked in LDS HALVED, t' = t / 2 comes out of the same fma exactly, and
// relu(t) = t' + |t'| (v_add_f32 with the |.| source modifier; bit for bit max(t, 0) for finite t, NaN stays NaN; -inf gives NaN
// where max gives 0 - both are overflow signals the range guard repeats the pair for) instead of v_cmp + 2 wait states + v_cndmask.
template <int DIL, bool FUSE, bool F16, int SUB = 2, bool LEAN = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_planes_kernel(const PlanesConvK p) {
  static_assert(!LEAN || F16, "the lean epilogue is the f16x3 kernels'");
  static_assert(SUB == 2 || (SUB == 4 && !FUSE), "four sub-tiles per wave: plain conv only (the fused tail's accumulators do not fit)");
  constexpr int THS = 4 * SUB;                    // patch rows of a team
  constexpr int HH = THS + 2 * DIL, HW = TW + 2 * DIL;
  constexpr int PXA = F16 ? PXH : PXB;            // bytes per activation pixel per chunk
  constexpr int UPP = PXA / 16;                   // 16-byte pieces per pixel
  constexpr int NPA = F16 ? 2 : 3;                // activation planes
  constexpr int NPROD = F16 ? 3 : 6;              // MFMA products per fp32 product
  constexpr int UR = HW * UPP;                    // 16-byte units per halo row
  constexpr int A_UNITS = HH * UR;
  constexpr int A_INSTR = (A_UNITS + 63) / 64;    // wave-level DMA instructions (64 units each)
  constexpr int AJ = (A_INSTR + 3) / 4;
  constexpr int A_BYTES = A_INSTR * 1024;
  constexpr int W_INSTR = W3_BYTES / 1024;        // 27
  constexpr int WJ = (W_INSTR + 3) / 4;
  constexpr int W1_INSTR = W1_BYTES / 1024;       // 6
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave8 >> 2, wave = wave8 & 3;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* As = smem_b + team * A_BYTES;                    // this team's halo
  unsigned char* Wsb = smem_b + 2 * A_BYTES;                      // [2][W3_BYTES], shared by the teams
  unsigned char* W1sb = Wsb + 2 * W3_BYTES;                       // [2][W1_BYTES] (FUSE)
  unsigned char* Cst = W1sb + (FUSE ? 2 * W1_BYTES : 0);          // 512 B: bias[32], bias1[64], act slopes (+ 512 B f16x3: row scales)
  unsigned char* W1res = Cst + (F16 ? 1024 : 512);                // (r6) f16x3 fused tail: the 1x1 weights of the conv's own 32 channels (chunks nchunks, nchunks + 1), resident

  // patch pairs of this workgroup: XCD x (= blockIdx % 8 under round-robin dispatch) owns a contiguous range of
  // pairs and its workgroups stride through it together, so vertically adjacent patches meet in one L2
  const int npatch = p.B * p.tiles_x * p.tiles_y, npairs = (npatch + 1) >> 1;
  const int G = gridDim.x, nx = G < 8 ? G : 8;
  const int xcd = blockIdx.x % nx, jx = blockIdx.x / nx, GX = (G - xcd + nx - 1) / nx;
  const int q_lo = (int)((long long)npairs * xcd / nx), q_hi = (int)((long long)npairs * (xcd + 1) / nx);
  const int n_my = q_lo + jx < q_hi ? (q_hi - q_lo - jx + GX - 1) / GX : 0;
  const int n_items = n_my * p.nchunks;
  const long long chunk_bytes = (long long)p.Hp * p.Wp * PXA;

  // DMA slots of this lane relative to the patch origin.  LDS unit U (16 bytes, lane-linear) = halo pixel U / UPP,
  // sub-unit U % UPP = 2 * plane + half; it receives the source's half ^ f(halo column) (f16x3: piece sub ^ g(halo
  // column)).  Kept in registers: a vector instruction of the loading team waits ~16 cycles for an issue slot beside the
  // other team's MFMA stream.
  uint32_t a_rel[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int u = (j * 4 + wave) * 64 + lane;
    const int hy = u / UR, rem = u - hy * UR;
    const int hx = rem / UPP, sub = rem - hx * UPP;
    if constexpr (F16) {
      a_rel[j] = (uint32_t)((hy * p.Wp + hx) * PXA + ((sub ^ ((hx >> 2) & 3)) * 16));
    } else {
      const int f = (hx >> 3) & 1;
      a_rel[j] = (uint32_t)((hy * p.Wp + hx) * PXA + (sub >> 1) * 32 + (((sub & 1) ^ f) * 16));
    }
  }

  struct Patch {
    int b, x0, y0;
    bool valid;
  };
  auto patch_of = [&](int k) {  // k-th pair of this workgroup -> this team's patch
    const int pt = 2 * (q_lo + jx + k * GX) + team;
    Patch t;
    t.valid = pt < npatch;
    const int pc = t.valid ? pt : npatch - 1;
    t.x0 = (pc % p.tiles_x) * TW;
    t.y0 = ((pc / p.tiles_x) % p.tiles_y) * THS;
    t.b = pc / (p.tiles_x * p.tiles_y);
    return t;
  };

  auto stage = [&](const Patch& pt, int c) {
    {
      const unsigned char* __restrict__ ab = p.pin + ((long long)pt.b * p.in_total + c) * chunk_bytes +
                                             ((long long)(pt.y0 + (PB - DIL)) * p.Wp + pt.x0 + (PB - DIL)) * PXA;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int i = j * 4 + wave;
        if (j < AJ - 1 || i * 64 + lane < A_UNITS) {
          if constexpr (PLANES_SADDR && F16) dma16s(ab, a_rel[j], As + i * 1024);
          else dma16(ab + a_rel[j], As + i * 1024);
        }
      }
    }
  };
  // Weights of chunk c into slot: instructions [0, W_SPLIT) by team 0 together with the item's halo, the rest (and the
  // fused 1x1 chunk) by team 1 one item ahead — 55 KB of DMA per team per phase instead of 69 + 41.
  constexpr int W_SPLIT = 14;
  auto stage_w = [&](int c, int slot) {
    const unsigned char* __restrict__ wb = p.wt + (long long)c * W3_BYTES + lane * 16;
    unsigned char* wd = Wsb + slot * W3_BYTES;
    const int lo = team == 0 ? 0 : W_SPLIT, hi = team == 0 ? W_SPLIT : W_INSTR;
#pragma unroll
    for (int j = 0; j < (W_SPLIT + 3) / 4; ++j) {
      const int i = lo + j * 4 + wave;
      if (i < hi) {
        if constexpr (PLANES_SADDR && F16) dma16s(p.wt + (long long)c * W3_BYTES + i * 1024, (uint32_t)(lane * 16), wd + i * 1024);
        else dma16(wb + i * 1024, wd + i * 1024);
      }
    }
    if (FUSE && team == 1) {
      const unsigned char* __restrict__ w1b = p.w1 + (long long)c * W1_BYTES + lane * 16;
      unsigned char* w1d = W1sb + slot * W1_BYTES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = j * 4 + wave;
        if (i < W1_INSTR) {
          if constexpr (PLANES_SADDR && F16) dma16s(p.w1 + (long long)c * W1_BYTES + i * 1024, (uint32_t)(lane * 16), w1d + i * 1024);
          else dma16(w1b + i * 1024, w1d + i * 1024);
        }
      }
    }
  };

  if (tid < 98) {  // epilogue constants -> LDS (first read is after at least one workgroup barrier)
    float v;
    constexpr float HALF = LEAN ? 0.5f : 1.f;  // (LEAN: see the kernel's header - exact, a power of two)
    if (tid < 32) v = HALF * p.bias[tid];
    else if (tid < 96) v = FUSE ? HALF * p.bias1[tid - 32] : 0.f;
    else if (tid == 96) v = p.act == SEGMIF_ACT_RELU ? 0.f : (p.act == SEGMIF_ACT_PRELU ? *p.prelu : 1.f);
    else v = p.act1 == SEGMIF_ACT_RELU ? 0.f : 1.f;
    reinterpret_cast<float*>(Cst)[tid] = v;
  } else if (F16 && tid >= 128 && tid < 224) {  // row scales 2^-e(n): floats [128, 160) conv, [160, 224) fused 1x1
    const int n = tid - 128;
    reinterpret_cast<float*>(Cst)[tid] = (LEAN ? 0.5f : 1.f) * (n < 32 ? p.wscale[n] : (FUSE ? p.w1scale[n - 32] : 1.f));
  }
  if constexpr (FUSE && F16) {  // 12 KB = 12 wave-level DMA instructions: waves 0..3 of each team take 3 / 0 (landed before the first barrier's s_waitcnt vmcnt(0))
    if (team == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) dma16(p.w1 + (long long)p.nchunks * W1_BYTES + (j * 4 + wave) * 1024 + lane * 16, W1res + (j * 4 + wave) * 1024);
    }
  }
  uint32_t amx = 0u;  // f16x3: largest |output| this lane has split for image amx_b (p16::absmax_pk patterns)
  int amx_b = 0;

  f32x16 acc[SUB], acc1[FUSE ? 2 : 1][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < SUB; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
#pragma unroll
    for (int i = 0; i < (FUSE ? 2 : 1); ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1[i][j][v] = 0.f;
  };
  zero_acc();

  // The wave's two sub-tiles are patch rows R0 and R0 + DIL (tap ky of the second reads the halo row tap
  // ky + 1 of the first reads): a chunk is 12 steps (kx, m), halo row R0 + m * DIL at column offset kx * DIL
  // feeds sub-tile 0 with tap ky = m (m < 3) and sub-tile 1 with tap ky = m - 1 (m > 0).
  // (SUB = 4: rows R0 + s * DIL, s = 0..3; halo row R0 + m * DIL feeds sub-tile s with tap ky = m - s)
  const int R0 = (DIL == 2) ? ((wave >> 1) * (2 * SUB) + (wave & 1)) : SUB * wave;
  const unsigned char* a_lane[3][F16 ? 2 : 1];  // bf16: plane pl at + 32 pl; f16x3: a pointer per plane (slot q ^ g)
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int col = kx * DIL + r;
    if constexpr (F16) {
      const int g = (col >> 2) & 3;
      a_lane[kx][0] = As + (R0 * HW + col) * PXA + ((h ^ g) * 16);
      a_lane[kx][1] = As + (R0 * HW + col) * PXA + (((2 + h) ^ g) * 16);
    } else {
      a_lane[kx][0] = As + (R0 * HW + col) * PXA + ((h ^ ((col >> 3) & 1)) * 16);
    }
  }
  const int wsw = (h ^ (r >> 4)) * 16;
  // products, least significant first: (activation plane, weight plane).  bf16x6: x2 w0, x1 w1, x0 w2, x1 w0, x0 w1,
  // x0 w0; f16x3: l W0s, x0 Wl, x0 W0
  constexpr int PA[6] = {F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0, 0};
  constexpr int PW[6] = {F16 ? 2 : 0, 1, F16 ? 0 : 2, 0, 1, 0};

  // A chunk = 9 balanced steps of 12 MFMAs, three per column offset kx, in which the two accumulators alternate:
  //   A: acc0 += W[1][kx] F1, acc1 += W[0][kx] F1      F_m = halo row R0 + m * DIL at column offset kx * DIL
  //   B: acc0 += W[2][kx] F2, acc1 += W[1][kx] F2
  //   C: acc0 += W[0][kx] F0, acc1 += W[2][kx] F3      (the two taps without a partner share a step)
  // Two consecutive MFMAs never target the same accumulator, so the fragment reads issued between them cost ~6 cycles
  // each instead of the ~43-cycle re-issue penalty of a dependent pair (the 12-step order had 36 such MFMAs per chunk).
  // A step's operands are requested while the previous step multiplies.
  auto compute = [&](int slot) {
    const unsigned char* w_lane = Wsb + slot * W3_BYTES + r * PXB + wsw;
    const unsigned char* w1_lane = W1sb + slot * W1_BYTES + r * PXB + wsw;
    u32x4 Fa[2][NPA], Fb[NPA], Fc[2][NPA], W0[2][3], W1[2][3], W2[3];  // [kx & 1] where a fragment outlives its column
    auto ld_f = [&](u32x4* dst, int kx, int m) {
#pragma unroll
      for (int pl = 0; pl < NPA; ++pl) {
        if constexpr (F16) dst[pl] = *reinterpret_cast<const u32x4*>(a_lane[kx][pl] + (m * DIL * HW) * PXA);
        else dst[pl] = *reinterpret_cast<const u32x4*>(a_lane[kx][0] + (m * DIL * HW) * PXA + pl * 32);
      }
    };
    auto ld_w = [&](u32x4* dst, int kx, int ky) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const u32x4*>(w_lane + ((ky * 3 + kx) * 32) * PXB + pl * 32);
    };
    auto mm = [&](const u32x4* wa, const u32x4* fa, const u32x4* wb, const u32x4* fb) {
#pragma unroll
      for (int t = 0; t < NPROD; ++t) {
        acc[0] = mfma_split<F16>(wa[PW[t]], fa[PA[t]], acc[0]);
        acc[1] = mfma_split<F16>(wb[PW[t]], fb[PA[t]], acc[1]);
      }
    };
    auto fence = [&](int n) {  // one scheduling region per step: every MFMA is followed by one (f16x3: up to two) of the next step's reads
#pragma unroll
      for (int t = 0; t < n; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, F16 ? 2 : 1, 0);  // DS read
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (SUB == 4) {
      // Four sub-tiles per wave (r4: measured 355 -> 378 TFLOP/s at Cin 192, B 16; 3.55 -> 3.33 ms per 64-image launch).
      // Per column offset kx the six halo rows F_m (m = 0..5) meet the three taps W_ky in five groups whose MFMAs
      // never repeat an accumulator back to back:
      //   G1  m=1: acc0 += W1 F1, acc1 += W0 F1            G4  m=4: acc2 += W2 F4, acc3 += W1 F4
      //   G2  m=2: acc0 += W2 F2, acc1 += W1 F2, acc2 += W0 F2
      //   G3  m=3: acc1 += W2 F3, acc2 += W1 F3, acc3 += W0 F3   G5  m=0 / m=5: acc0 += W0 F0, acc3 += W2 F5
      // 36 MFMAs (x NPROD / 3) against 12 + 9 fragment reads per kx (two sub-tiles: 18 against 17): the LDS port is the
      // limiter of the two-sub-tile kernel on f16x3 operands.  A group's operands are requested while the
      // previous group multiplies; the weights of the next kx arrive during G5 into the other half of Wk.
      u32x4 F[3][NPA], Wk[2][3][3];  // F: rotating fragment sets; Wk[kx & 1][ky][plane]
      auto g2 = [&](int sa, const u32x4* wa, const u32x4* fa, int sb, const u32x4* wb, const u32x4* fb) {
#pragma unroll
        for (int t = 0; t < NPROD; ++t) {
          acc[sa] = mfma_split<F16>(wa[PW[t]], fa[PA[t]], acc[sa]);
          acc[sb] = mfma_split<F16>(wb[PW[t]], fb[PA[t]], acc[sb]);
        }
      };
      auto g3 = [&](int sa, const u32x4* wa, int sb, const u32x4* wb, int sc, const u32x4* wc, const u32x4* f) {
#pragma unroll
        for (int t = 0; t < NPROD; ++t) {
          acc[sa] = mfma_split<F16>(wa[PW[t]], f[PA[t]], acc[sa]);
          acc[sb] = mfma_split<F16>(wb[PW[t]], f[PA[t]], acc[sb]);
          acc[sc] = mfma_split<F16>(wc[PW[t]], f[PA[t]], acc[sc]);
        }
      };
      ld_f(F[0], 0, 1);
      ld_w(Wk[0][1], 0, 1);
      ld_w(Wk[0][0], 0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int c = kx & 1, n = c ^ 1;
        // G1 (F[0] = F_1, W0, W1 on hand); request G2's
        ld_f(F[1], kx, 2);
        ld_w(Wk[c][2], kx, 2);
        g2(0, Wk[c][1], F[0], 1, Wk[c][0], F[0]);
        fence(2 * NPROD);
        // G2 (F[1] = F_2); request G3's
        ld_f(F[2], kx, 3);
        g3(0, Wk[c][2], 1, Wk[c][1], 2, Wk[c][0], F[1]);
        fence(3 * NPROD);
        // G3 (F[2] = F_3); request G4's
        ld_f(F[0], kx, 4);
        g3(1, Wk[c][2], 2, Wk[c][1], 3, Wk[c][0], F[2]);
        fence(3 * NPROD);
        // G4 (F[0] = F_4); request G5's
        ld_f(F[1], kx, 0);
        ld_f(F[2], kx, 5);
        g2(2, Wk[c][2], F[0], 3, Wk[c][1], F[0]);
        fence(2 * NPROD);
        // G5 (F[1] = F_0, F[2] = F_5); request the next column's G1
        if (kx < 2) {
          ld_f(F[0], kx + 1, 1);
          ld_w(Wk[n][1], kx + 1, 1);
          ld_w(Wk[n][0], kx + 1, 0);
        }
        g2(0, Wk[c][0], F[1], 3, Wk[c][2], F[2]);
        fence(2 * NPROD);
      }
      return;
    }
    ld_f(Fa[0], 0, 1);
    ld_w(W1[0], 0, 1);
    ld_w(W0[0], 0, 0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int c = kx & 1, n = c ^ 1;
      // step A (operands loaded during the previous step); request step B's
      ld_f(Fb, kx, 2);
      ld_w(W2, kx, 2);
      mm(W1[c], Fa[c], W0[c], Fa[c]);
      if (FUSE && kx == 1) {  // centre tap of sub-tile 0: the 1x1 conv's input at this pixel
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 W1f[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) W1f[pl] = *reinterpret_cast<const u32x4*>(w1_lane + (nt * 32) * PXB + pl * 32);
#pragma unroll
          for (int t = 0; t < NPROD; ++t) acc1[0][nt] = mfma_split<F16>(W1f[PW[t]], Fa[c][PA[t]], acc1[0][nt]);
        }
      }
      fence(FUSE && kx == 1 ? 4 * NPROD : 2 * NPROD);
      // step B; request step C's
      ld_f(Fc[0], kx, 0);
      ld_f(Fc[1], kx, 3);
      mm(W2, Fb, W1[c], Fb);
      if (FUSE && kx == 1) {  // centre tap of sub-tile 1
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 W1f[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) W1f[pl] = *reinterpret_cast<const u32x4*>(w1_lane + (nt * 32) * PXB + pl * 32);
#pragma unroll
          for (int t = 0; t < NPROD; ++t) acc1[1][nt] = mfma_split<F16>(W1f[PW[t]], Fb[PA[t]], acc1[1][nt]);
        }
      }
      fence(FUSE && kx == 1 ? 4 * NPROD : 2 * NPROD);
      // step C; request the next column's step A
      if (kx < 2) {
        ld_f(Fa[n], kx + 1, 1);
        ld_w(W1[n], kx + 1, 1);
        ld_w(W0[n], kx + 1, 0);
      }
      mm(W0[c], Fc[0], W2, Fc[1]);
      fence(2 * NPROD);
    }
  };

  // accumulator register v of lane (r, h): output channel (v & 3) + 8 (v >> 2) + 4 h of pixel x0 + r
  auto epilogue = [&](const Patch& pt, int item) {
    int z = 0;  // opaque zero added to the weight pointer below: keeps loop-invariant loads of the epilogue from being
    asm volatile("" : "+s"(z));  // hoisted out of the phase loop
    ETL(0, "s"(z));
    // branch-free: act(t) = t >= 0 ? t : nslope * t with nslope = 0 (ReLU), the PReLU slope, or 1 (none).  Biases and
    // slope were parked in LDS at kernel start: a global load here costs a loaded-memory-system round trip (2 000+
    // cycles measured) in a phase the other team is waiting on, and per-element "pointer ? load : 0" code cost 5 000.
    const float* cst = reinterpret_cast<const float*>(Cst);
    const float nslope = cst[96];
    if constexpr (F16) {  // per-image range slots: hand in the running maximum when the patch sequence moves to another image
      const int pb = p.amax_images > 1 ? pt.b : 0;  // (wave-uniform)
      if (pb != amx_b) {
        if (p.amax) p16::fold_pat(p.amax, amx_b, amx_b, amx);
        amx = 0u;
        amx_b = pb;
      }
    }
    float bv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(cst + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[4 * g + e] = t[e];
    }
    float sv[F16 ? 16 : 1];  // f16x3: 2^-e(n) of this lane's 16 output channels
    if constexpr (F16) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(cst + 128 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[4 * g + e] = t[e];
      }
    }
    ETL(1, "v"(bv[0]), "v"(bv[15]), "v"(sv[0]), "v"(sv[F16 ? 15 : 0]));
    const int ox = pt.x0 + r;
    // (r6) f16x3 fused tail: nothing in this epilogue waits for global memory where it is used.  Before, the 1x1 weights of the
    // conv's own 32 channels and the residual pieces were loaded at their point of use - 32 dependent round trips per patch (load,
    // s_waitcnt vmcnt(0) - which also waits for the previous store -, use, store), ~700 ticks each under the other team's DMA
    // burst: the 21 900-tick epilogue of profiles/r06_planes16_timeline_before.txt, 38 % of the kernel's time with no MFMA running
    // on the CU.  Now those weights (12 KB) are resident in LDS for the kernel's life (W1res), a sub-tile's residual - this lane's
    // 16-byte pieces of chunks 0..3, hi and lo: 8 loads - is requested in one batch a sub-tile ahead, and the 8 stores of a
    // sub-tile leave together after its arithmetic.
    constexpr bool PRE = FUSE && F16;
    u32x4 rp[2][2][2];                            // [nt][chunk 2 nt + c2][plane]: one sub-tile's residual pieces
    const bool res_pl = PRE && (LEAN || (!p.res && p.res_planes)) && !(PLANES_DBG & 512);  // (uniform)
    auto request_residual = [&](int i) {  // (the padded planes image holds every pixel of a rounded-up or clamped patch: no bounds needed)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            rp[nt][c2][k] = *reinterpret_cast<const u32x4*>(
                p.pin + ((long long)pt.b * p.in_total + 2 * nt + c2) * chunk_bytes +
                ((long long)(pt.y0 + R0 + i * DIL + PB) * p.Wp + ox + PB) * PXA + h * 16 + k * 32);
    };
    if constexpr (PRE) {
      if (res_pl) request_residual(0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < SUB; ++i) {
      const int oy = pt.y0 + R0 + i * DIL;
      const bool ok = pt.valid && oy < p.H && ox < p.W;
      float y[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float t = F16 ? fmaf(acc[i][v], sv[F16 ? v : 0], bv[v]) : acc[i][v] + bv[v];
        y[v] = LEAN ? t + __builtin_fabsf(t) : (t >= 0.f ? t : nslope * t);
      }
      if (i == 0) ETL(2, "v"(y[0]), "v"(y[7]), "v"(y[8]), "v"(y[15]));
#if PLANES_DBG & 32
      if (i == 0 && wave == 0 && lane == 0 && blockIdx.x < 256 && item < PLANES_TL_ITEMS) {
        asm volatile("" ::"v"(y[0]), "v"(y[15]));
        planes_timeline[blockIdx.x][team][item][7] = __builtin_amdgcn_s_memtime();
      }
#endif
      const long long m = ((long long)pt.b * p.H + oy) * p.W + ox;
      if (!LEAN && p.out && ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(p.out + m * p.ldo + 8 * g + 4 * h) = f32x4{y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        u32x4 pl[3];
        if constexpr (F16) {
          p16::split8(y + 8 * q, pl[0], pl[1]);
          pl[2] = pl[1];
          const uint32_t mx = p16::absmax_pk4(amx, pl[0], pl[1]);
          amx = ok ? mx : amx;
        } else {
          split8(y + 8 * q, pl[0], pl[1], pl[2]);
        }
        if (PLANES_DBG & 64) asm volatile("" ::"v"(pl[0]), "v"(pl[1]), "v"(pl[2]));
        if (!(LEAN && FUSE) && p.pout && ok && !(PLANES_DBG & 64)) {
          unsigned char* dst = p.pout + ((long long)pt.b * p.out_total + p.out_chunk0 + q) * chunk_bytes +
                               ((long long)(oy + PB) * p.Wp + ox + PB) * PXA + h * 16;
#pragma unroll
          for (int k = 0; k < NPA; ++k) *reinterpret_cast<u32x4*>(dst + k * 32) = pl[k];
        }
        if (FUSE) {  // 1x1 weights of the 16 channels just produced: chunk nchunks + q (re-read per sub-tile: L2 hits, keeps registers free)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            u32x4 W1g[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              if constexpr (PRE) W1g[k] = *reinterpret_cast<const u32x4*>(W1res + (q * 64 + nt * 32 + r) * PXB + k * 32 + wsw);
              else W1g[k] = *reinterpret_cast<const u32x4*>(p.w1 + z + ((long long)(p.nchunks + q) * 64 + nt * 32 + r) * PXB + k * 32 + wsw);
            }
            if (PLANES_DBG & 1024) asm volatile("" ::"v"(W1g[0]), "v"(W1g[1]), "v"(W1g[2]), "v"(pl[0]), "v"(pl[1]));
            else
#pragma unroll
              for (int t = 0; t < NPROD; ++t) acc1[i][nt] = mfma_split<F16>(W1g[PW[t]], pl[PA[t]], acc1[i][nt]);
          }
        }
        if constexpr (LEAN) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (FUSE) {
        if (i == 0) ETL(3, "v"(acc1[0][0][0]), "v"(acc1[0][1][0]), "v"(acc1[0][0][15]), "v"(acc1[0][1][15]));
        // out1 = act1(1x1) + residual, 32 channels of the pixel at a time: arithmetic first (in registers), then their four stores;
        // nothing between the residual's arrival and the stores waits on memory
        const float nslope1 = cst[97];
        // (r6) the 1x1's biases and row scales in batches of 8 LDS reads (one per half of this lane's 32 channels; read where
        // they were used, each of the 16 reads was its own round trip of a few hundred ticks under the other team's LDS-DMA writes)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x4 cb1[4], cs1[4], o[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            cb1[g] = *reinterpret_cast<const f32x4*>(cst + z + 32 + nt * 32 + 8 * g + 4 * h);
            if constexpr (F16) cs1[g] = *reinterpret_cast<const f32x4*>(cst + z + 160 + nt * 32 + 8 * g + 4 * h);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = nt * 32 + 8 * g + 4 * h;
            const f32x4 b1 = cb1[g];
            f32x4 s1 = {1.f, 1.f, 1.f, 1.f};
            if constexpr (F16) s1 = cs1[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = F16 ? fmaf(acc1[i][nt][4 * g + e], s1[e], b1[e]) : acc1[i][nt][4 * g + e] + b1[e];
              o[g][e] = LEAN ? t + __builtin_fabsf(t) : (t >= 0.f ? t : nslope1 * t);
            }
            if (!LEAN && p.res) {
              if (ok) o[g] += *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + n0);
            } else if (F16 && (LEAN || p.res_planes) && !(PLANES_DBG & 512)) {
              // (r4) residual = the DRDB's own input, read back from its planes: channel n0 + e is element 4 (g & 1) + e of this
              // lane's piece of chunk 2 nt + (g >> 1) - the order the accumulators hand out; x = hi + 2^-11 lo (23 bits)
              // (r6: from the pieces requested at the top of the epilogue / during the previous sub-tile)
              const uint32_t hw[2] = {rp[nt][g >> 1][0][2 * (g & 1)], rp[nt][g >> 1][0][2 * (g & 1) + 1]};
              const uint32_t lw[2] = {rp[nt][g >> 1][1][2 * (g & 1)], rp[nt][g >> 1][1][2 * (g & 1) + 1]};
#pragma unroll
              for (int e2 = 0; e2 < 2; ++e2) {
                o[g][2 * e2] += pair_value<0>(hw[e2], lw[e2]);
                o[g][2 * e2 + 1] += pair_value<1>(hw[e2], lw[e2]);
              }
            }
          }
          if (i == 0 && nt == 1) ETL(4, "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
          if constexpr (PRE) {
            if (nt == 1 && res_pl && i + 1 < SUB) {  // (this sub-tile's pieces are consumed: the next one's ride ahead of the stores)
              __builtin_amdgcn_sched_barrier(0);
              request_residual(i + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (PLANES_DBG & 256) {
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(o[g]));
          } else if (ok) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(p.out1 + m * p.ldo1 + nt * 32 + 8 * g + 4 * h) = o[g];
          }
          if constexpr (LEAN) __builtin_amdgcn_sched_barrier(0);
        }
        if (i == 0) ETL(5, "s"(z));
        if (i == SUB - 1) ETL(6, "s"(z));
        if constexpr (LEAN) __builtin_amdgcn_sched_barrier(0);  // (branch-free, the scheduler would hoist the next sub-tile's arithmetic above these stores and spill)
      }
    }
  };

  // ---- (r6) deferred stores ---------------------------------------------------------------------------
  // profiles/r06_planes16_timeline_before.txt: the epilogue of a 16 x 32 patch at the end of its last COMPUTE phase took 13 000
  // ticks - as long as three COMPUTE phases, with no MFMA running on the CU meanwhile, twice per patch pair; 64 values of vector
  // arithmetic per lane do not take that long: its 16 store instructions per wave queue in the CU's in-order vector-memory path.
  // DEFER: the COMPUTE phase ends with the arithmetic only (bias, activation, split into half pairs - in registers that are free
  // once the phase's fragments are dead), the workgroup barrier follows at once, and the 16 stores leave at the start of the
  // team's next LOAD phase, beside the other team's MFMA stream, where this wave would only be waiting for vector-memory issue.
  constexpr bool DEFER = PLANES_DEFER && F16 && !FUSE && SUB == 4;
  u32x4 dpk[DEFER ? SUB : 1][4];  // [sub-tile][2 * q + plane]
  Patch dpt = {0, 0, 0, false};
  bool dpending = false;          // (wave-uniform)
  auto defer_math = [&](const Patch& pt) {
    const float* cst = reinterpret_cast<const float*>(Cst);
    const float nslope = cst[96];
    {
      const int pb = p.amax_images > 1 ? pt.b : 0;  // (wave-uniform)
      if (pb != amx_b) {
        if (p.amax) p16::fold_pat(p.amax, amx_b, amx_b, amx);
        amx = 0u;
        amx_b = pb;
      }
    }
    f32x4 bv[4], sv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bv[g] = *reinterpret_cast<const f32x4*>(cst + 8 * g + 4 * h);
      sv[g] = *reinterpret_cast<const f32x4*>(cst + 128 + 8 * g + 4 * h);
    }
    const int ox = pt.x0 + r;
#pragma unroll
    for (int i = 0; i < SUB; ++i) {
      const int oy = pt.y0 + R0 + i * DIL;
      const bool ok = pt.valid && oy < p.H && ox < p.W;
      const long long m = ((long long)pt.b * p.H + oy) * p.W + ox;
      float y[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float t = fmaf(acc[i][v], sv[v >> 2][v & 3], bv[v >> 2][v & 3]);
        y[v] = LEAN ? t + __builtin_fabsf(t) : (t >= 0.f ? t : nslope * t);
      }
      if constexpr (LEAN) {
        // PReLU / no activation (uniform; the dilation-1 convs): y = max(t, 0) + slope min(t, 0), both parts from the halved t
        // without a compare - t' + |t'| and t' - |t'| are exact, the product rounds once as slope * t did
        if (nslope != 0.f) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float th = fmaf(acc[i][v], sv[v >> 2][v & 3], bv[v >> 2][v & 3]);
            y[v] = fmaf(nslope, th - __builtin_fabsf(th), y[v]);
          }
        }
      }
      if (!LEAN && p.out && ok) {  // (the optional fp32 copy is not deferred: no caller of the hot path asks for it)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(p.out + m * p.ldo + 8 * g + 4 * h) = f32x4{y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        p16::split8(y + 8 * q, dpk[i][2 * q], dpk[i][2 * q + 1]);
        const uint32_t mx = p16::absmax_pk4(amx, dpk[i][2 * q], dpk[i][2 * q + 1]);
        amx = ok ? mx : amx;
      }
    }
    dpt = pt;
    dpending = true;
  };
  auto defer_stores = [&]() {
    if ((LEAN || p.pout) && !(PLANES_DBG & 64)) {
      const int ox = dpt.x0 + r;
#pragma unroll
      for (int i = 0; i < SUB; ++i) {
        const int oy = dpt.y0 + R0 + i * DIL;
        if (dpt.valid && oy < p.H && ox < p.W) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            unsigned char* dst = p.pout + ((long long)dpt.b * p.out_total + p.out_chunk0 + q) * chunk_bytes +
                                 ((long long)(oy + PB) * p.Wp + ox + PB) * PXA + h * 16;
            *reinterpret_cast<u32x4*>(dst) = dpk[i][2 * q];
            *reinterpret_cast<u32x4*>(dst + 32) = dpk[i][2 * q + 1];
          }
        }
      }
    }
    dpending = false;
  };

  // ---- phase loop -----------------------------------------------------------------------------------
  // Straight-line per item (LOAD, barrier, COMPUTE, barrier) with team 1 skewed by one barrier: no branch around the
  // MFMA block and the accumulators are cleared by a select, so they stay in one set of registers (a conditional
  // compute / clear made hipcc keep two copies and spill in the fused kernel).
  if (n_items > 0) {
    if (team == 1) {  // skewed by one phase; uses it to fetch its share of the first chunk's weights
      stage_w(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    Patch cur = patch_of(0);
    int k = 0, c = 0;
    for (int i = 0; i < n_items; ++i) {
      // LOAD phase: this item's DMA and nothing else (a vector instruction issued here competes with the other team's
      // MFMA stream for the SIMD's issue port and gets a slot every ~16 cycles: the epilogue took 5 500 cycles here)
      TL(0);
      if (DEFER && PLANES_DEFER == 1 && dpending) {
        defer_stores();
        __builtin_amdgcn_sched_barrier(0);
      }
      stage(cur, c);
      if (team == 0) stage_w(c, i & 1);
      else if (i + 1 < n_items) stage_w(c + 1 == p.nchunks ? 0 : c + 1, (i + 1) & 1);
      if (DEFER && PLANES_DEFER == 2 && dpending) {
        __builtin_amdgcn_sched_barrier(0);
        defer_stores();
      }
      TL(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TL(3);
      __syncthreads();
      TL(4);
      // COMPUTE phase (an invalid tail patch multiplies the clamped patch's data; its stores are masked).  A patch's
      // epilogue closes the phase of its last chunk: by then the other team's DMA burst has been issued and mostly
      // landed, so the stores do not queue behind it in the CU's vector-memory path, and the SIMDs are otherwise idle.
      if (PLANES_DBG & 128) __builtin_amdgcn_s_setprio(1);
      compute(i & 1);
      if (PLANES_DBG & 128) __builtin_amdgcn_s_setprio(0);
      TL(1);
      if (c + 1 == p.nchunks) {
        if constexpr (DEFER) defer_math(cur);
        else if (!(FUSE && (PLANES_DBG & 2048))) epilogue(cur, i);
        zero_acc();
      }
      TL(5);
      __syncthreads();
      TL(6);
      if (++c == p.nchunks) {
        c = 0;
        if (++k < n_my) cur = patch_of(k);
      }
    }
    if (team == 0) __syncthreads();
    if (DEFER && dpending) defer_stores();  // the workgroup's last patch
  }
  if constexpr (F16) {
    if (p.amax) p16::fold_pat(p.amax, amx_b, amx_b, amx);
  }
}

template <int DIL, bool FUSE, bool F16, int SUB = 2, bool LEAN = false>
int launch(PlanesConvK k, hipStream_t stream) {
  constexpr int A_UNITS = (4 * SUB + 2 * DIL) * (TW + 2 * DIL) * (F16 ? PXH : PXB) / 16;
  constexpr size_t smem = 2 * ((size_t)((A_UNITS + 63) / 64) * 1024 + W3_BYTES + (FUSE ? W1_BYTES : 0)) + (F16 ? 1024 : 512) +
                          (FUSE && F16 ? 2 * W1_BYTES : 0);
  static_assert(smem <= 160 * 1024, "LDS budget");
  k.tiles_y = (k.H + 4 * SUB - 1) / (4 * SUB);
  auto fn = conv3x3_planes_kernel<DIL, FUSE, F16, SUB, LEAN>;
  static segmif::PerDeviceFlag raised_flag;  // idempotent attribute; benign race
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  static segmif::PerDeviceValue<int> ncu_dev;
  int& ncu = ncu_dev.here();
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SEGMIF_EINVAL;
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long long npairs = ((long long)k.B * k.tiles_x * k.tiles_y + 1) / 2;
  dim3 grid((unsigned)(npairs < ncu ? npairs : ncu));  // one persistent workgroup (two teams) per CU
  hipLaunchKernelGGL(fn, grid, dim3(512), smem, stream, k);
  return (int)hipGetLastError();
}

__device__ float planes_zero_bias[64];  // stands in for a NULL bias (static storage: zero-initialised)

// ---- producers of the planes format ---------------------------------------------------------------

// fp32 rows (pixel pitch ldx) -> planes chunks [chunk0, chunk0 + nconv).  One thread = (pixel, chunk, half).
template <bool F16>
__global__ void planes_from_f32_kernel(const float* __restrict__ x, int ldx, unsigned char* __restrict__ planes, int B,
                                       int H, int W, int Hp, int Wp, int total, int chunk0, int nconv, uint32_t* amax,
                                       int amax_images) {
  constexpr int PXA = F16 ? PXH : PXB;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long npix = (long long)B * H * W;
  const bool live = idx < npix * nconv * 2;
  if (!F16 && !live) return;
  if (!live) idx = npix * nconv * 2 - 1;  // f16x3: dead lanes of the last block still take part in the wave maximum below (as the last pixel)
  const int hf = (int)(idx & 1);
  const int ch = (int)((idx >> 1) % nconv);
  const long long pix = (idx >> 1) / nconv;
  const int xx = (int)(pix % W);
  const int yy = (int)((pix / W) % H);
  const int b = (int)(pix / ((long long)W * H));
  const float* src = x + pix * ldx + ch * 16 + 4 * hf;  // positions 8 hf + e <-> channels 4 hf + (e & 3) + 8 (e >> 2)
  const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 8);
  const float y[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  unsigned char* dst = planes + (((long long)b * total + chunk0 + ch) * Hp + yy + PB) * (long long)Wp * PXA + (long long)(xx + PB) * PXA + hf * 16;
  if constexpr (F16) {
    u32x4 p0, p1;
    p16::split8(y, p0, p1);
    if (live) {
      *reinterpret_cast<u32x4*>(dst) = p0;
      *reinterpret_cast<u32x4*>(dst + 32) = p1;
    }
    if (amax) {  // a wave covers consecutive pixels: images of its first and last lane (conservative when it straddles two)
      const int b_lo = __shfl(b, 0, 64), b_hi = __shfl(b, 63, 64);
      p16::fold_pat(amax, amax_images > 1 ? b_lo : 0, amax_images > 1 ? b_hi : 0, live ? p16::absmax_pk4(0u, p0, p1) : 0u);
    }
  } else {
    u32x4 p0, p1, p2;
    split8(y, p0, p1, p2);
    *reinterpret_cast<u32x4*>(dst) = p0;
    *reinterpret_cast<u32x4*>(dst + 32) = p1;
    *reinterpret_cast<u32x4*>(dst + 64) = p2;
  }
}

// (r6) conv1_ir / conv1_vis of the fusion net (core/model_fusion.py:1029-1030, :1051-1053, :1055-1057): a 3 x 3 'same' conv from ONE
// channel to 64, + bias + the shared PReLU, written as the first DRDB's input planes (half pairs, chunks chunk0 .. chunk0 + 3) and /
// or as fp32 rows.  As an implicit GEMM this was K = 9 gathered scalar by scalar (igemm<128,64,..,generic>: 2.4 ms per 64-image
// launch, matrix pipe 11 % busy); it is 18 flops per output value against 4 output bytes - a store-bound stencil.  One wave = one
// 16-channel chunk x 32 consecutive pixels x the two halves of the chunk: a lane keeps its 8 channels' 72 taps + 8 biases in
// registers for the workgroup's whole run of pixels (all inside ONE image: a single range report per workgroup), reads its
// pixel's 3 x 3 window (nine loads shared by the lane pair, neighbours one float apart), and stores one 16-byte piece per plane:
// a wave's two store instructions together fill 2 KB of contiguous planes memory.  Store traffic 256 B per pixel = 5.0 GB per
// 64 images: ~0.8 ms at the achievable 6.3 TB/s.
constexpr int C1_PIX = 32;  // pixels per workgroup step (4 waves = the 4 chunks of the 64 output channels)
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ prelu, int act,
                                                         unsigned char* __restrict__ planes, int total, int chunk0,
                                                         float* __restrict__ out, int ldo, int H, int W, int Hp, int Wp, int steps,
                                                         uint32_t* __restrict__ amax, int amax_images) {
  const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6, hf = lane & 1, pl = lane >> 1;
  const int b = blockIdx.y;
  const long long npix = (long long)H * W;
  float wt[8][9], bs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int n = 16 * c + 4 * hf + (e & 3) + 8 * (e >> 2);  // position 8 hf + e of chunk c (sigma order)
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[e][t] = w[n * 9 + t];
    bs[e] = bias ? bias[n] : 0.f;
  }
  const float slope = act == SEGMIF_ACT_PRELU ? prelu[0] : 0.f;
  const float* xb = x + (long long)b * npix;
  unsigned char* pb = planes ? planes + ((long long)b * total + chunk0 + c) * Hp * (long long)Wp * PXH : nullptr;
  float* ob = out ? out + (long long)b * npix * ldo : nullptr;
  uint32_t amx = 0u;
  for (int it = 0; it < steps; ++it) {
    const long long pix = ((long long)blockIdx.x * steps + it) * C1_PIX + pl;
    const bool live = pix < npix;
    const long long pc = live ? pix : npix - 1;
    const int yy = (int)(pc / W), xx = (int)(pc - (long long)yy * W);
    float win[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int sy = yy + dy - 1, sx = xx + dx - 1;
        const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
        const float v = xb[(long long)(in ? sy : yy) * W + (in ? sx : xx)];  // (unconditional load from a valid address)
        win[dy * 3 + dx] = in ? v : 0.f;
      }
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = bs[e];
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(win[t], wt[e][t], a);
      if (act == SEGMIF_ACT_RELU) a = fmaxf(a, 0.f);
      else if (act == SEGMIF_ACT_PRELU) a = a > 0.f ? a : a * slope;
      y[e] = a;
    }
    if (ob && live) {
      float* o = ob + pix * ldo + 16 * c + 4 * hf;
      *reinterpret_cast<f32x4*>(o) = f32x4{y[0], y[1], y[2], y[3]};
      *reinterpret_cast<f32x4*>(o + 8) = f32x4{y[4], y[5], y[6], y[7]};
    }
    if (pb) {
      u32x4 p0, p1;
      p16::split8(y, p0, p1);
      if (live) {
        unsigned char* dst = pb + ((long long)(yy + PB) * Wp + xx + PB) * PXH + hf * 16;
        *reinterpret_cast<u32x4*>(dst) = p0;
        *reinterpret_cast<u32x4*>(dst + 32) = p1;
        amx = p16::absmax_pk4(amx, p0, p1);
      }
    }
  }
  if (pb && amax) p16::fold_pat(amax, amax_images > 1 ? b : 0, amax_images > 1 ? b : 0, amx);
}

// zero the border / round-up region of every chunk image: one block per padded row
__global__ void planes_zero_border_kernel(unsigned char* __restrict__ planes, int H, int W, int Hp, int Wp, int upp) {
  const int row = blockIdx.x;  // upp = 16-byte pieces per pixel (6 | 4)
  u32x4* line = reinterpret_cast<u32x4*>(planes + ((long long)blockIdx.y * Hp + row) * (long long)Wp * upp * 16);
  const u32x4 z = {0u, 0u, 0u, 0u};
  if (row < PB || row >= PB + H) {
    for (int u = threadIdx.x; u < Wp * upp; u += blockDim.x) line[u] = z;
  } else {
    const int right = Wp - (PB + W);
    for (int u = threadIdx.x; u < (PB + right) * upp; u += blockDim.x) line[u < PB * upp ? u : (PB + W) * upp + (u - PB * upp)] = z;
  }
}

// fp32 [N][ldw] (k = tap * Cin + c, taps = 9 | 1) -> [chunk][tap][n][96 B]: plane-major, half-swapped rows
__global__ void planes_pack_weight_kernel(const float* __restrict__ w, int N, int Cin, int taps, int ldw, long long total,
                                          uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx & 15);
  long long t = idx >> 4;
  const int n = (int)(t % N); t /= N;
  const int tap = (int)(t % taps);
  const int chunk = (int)(t / taps);
  const float x = w[(long long)n * ldw + tap * Cin + chunk * 16 + sigma16(j)];
  uint32_t p0, p1, p2;
  split3(x, 0.f, p0, p1, p2);
  const int f = (n >> 4) & 1;
  const long long row = ((long long)chunk * taps + tap) * N + n;
  uint16_t* dst = out + row * 48 + (((j >> 3) ^ f) * 8) + (j & 7);
  dst[0] = (uint16_t)(p0 & 0xffffu);
  dst[16] = (uint16_t)(p1 & 0xffffu);
  dst[32] = (uint16_t)(p2 & 0xffffu);
}

// f16x3 weights.  Row scale: one wave per output row n finds max |w[n][.]| and stores 2^-e(n) with
// 2^14 <= 2^e(n) max < 2^15 (e = 0 for an all-zero or vanishing row).
__global__ void planes16_weight_scale_kernel(const float* __restrict__ w, int K, int ldw, float* __restrict__ inv_scale) {
  const int n = blockIdx.x;
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 64) mx = fmaxf(mx, fabsf(w[(long long)n * ldw + k]));
  mx = p16::wave_max(mx);
  if (threadIdx.x == 0) {
    int e = 0;
    if (mx >= 1e-30f && mx <= 3e38f) e = 14 - (int)((__float_as_uint(mx) >> 23) - 127);
    inv_scale[n] = ldexpf(1.f, -e);
  }
}

// fp32 [N][ldw] (k = tap * Cin + c) -> [chunk][tap][n][96 B]: planes W0 | Wl | W0s of the scaled row, half-swapped rows
__global__ void planes16_pack_weight_kernel(const float* __restrict__ w, int N, int Cin, int taps, int ldw, long long total,
                                            const float* __restrict__ inv_scale, uint16_t* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx & 15);
  long long t = idx >> 4;
  const int n = (int)(t % N); t /= N;
  const int tap = (int)(t % taps);
  const int chunk = (int)(t / taps);
  const float x = w[(long long)n * ldw + tap * Cin + chunk * 16 + sigma16(j)] * (1.f / inv_scale[n]);  // exact: power of two
  const _Float16 w0 = (_Float16)x;
  const _Float16 wl = (_Float16)(x - (float)w0);
  const _Float16 ws = (_Float16)((float)w0 * (1.f / LSCALE));
  const int f = (n >> 4) & 1;
  const long long row = ((long long)chunk * taps + tap) * N + n;
  uint16_t* dst = out + row * 48 + (((j >> 3) ^ f) * 8) + (j & 7);
  dst[0] = __builtin_bit_cast(uint16_t, w0);
  dst[16] = __builtin_bit_cast(uint16_t, wl);
  dst[32] = __builtin_bit_cast(uint16_t, ws);
}

}  // namespace
}  // namespace segmif

using namespace segmif;

#if PLANES_DBG & 32
extern "C" int segmif_debug_planes_epi_timeline(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(planes_epi_timeline), bytes < sizeof(planes_epi_timeline) ? bytes : sizeof(planes_epi_timeline));
}
extern "C" int segmif_debug_planes_timeline(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(planes_timeline), bytes < sizeof(planes_timeline) ? bytes : sizeof(planes_timeline));
}
#endif

extern "C" int segmif_planes_dims(int H, int W, int* Hp, int* Wp) {
  if (H <= 0 || W <= 0 || !Hp || !Wp) return SEGMIF_EINVAL;
  *Hp = planes_hp(H);
  *Wp = planes_wp(W);
  return 0;
}

static int64_t planes_bytes_impl(int B, int H, int W, int chunks, int pxa) {
  if (B <= 0 || H <= 0 || W <= 0 || chunks <= 0) return 0;
  return (int64_t)B * chunks * planes_hp(H) * planes_wp(W) * pxa;
}

static int planes_zero_border_impl(void* planes, int B, int H, int W, int chunks, int pxa, void* stream) {
  if (!planes || B <= 0 || H <= 0 || W <= 0 || chunks <= 0) return SEGMIF_EINVAL;
  const int Hp = planes_hp(H), Wp = planes_wp(W);
  hipLaunchKernelGGL(planes_zero_border_kernel, dim3((unsigned)Hp, (unsigned)(B * chunks)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned char*)planes, H, W, Hp, Wp, pxa / 16);
  return (int)hipGetLastError();
}

template <bool F16>
static int planes_from_f32_impl(const float* x, int ldx, void* planes, int B, int H, int W, int chunks, int chunk0, int nconv,
                                uint32_t* amax, int amax_images, void* stream) {
  if (!x || !planes || B <= 0 || H <= 0 || W <= 0 || nconv <= 0 || chunk0 < 0 || chunk0 + nconv > chunks || ldx % 4 ||
      ((uintptr_t)x & 15) || ldx < 16 * nconv || (amax && amax_images != 1 && amax_images != B))
    return SEGMIF_EINVAL;
  const long long n = (long long)B * H * W * nconv * 2;
  hipLaunchKernelGGL(planes_from_f32_kernel<F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (unsigned char*)planes, B, H, W, planes_hp(H), planes_wp(W), chunks, chunk0, nconv, amax, amax_images);
  return (int)hipGetLastError();
}

extern "C" int64_t segmif_planes_bytes(int B, int H, int W, int chunks) { return planes_bytes_impl(B, H, W, chunks, PXB); }
extern "C" int64_t segmif_planes16_bytes(int B, int H, int W, int chunks) { return planes_bytes_impl(B, H, W, chunks, PXH); }

extern "C" int segmif_planes_zero_border(void* planes, int B, int H, int W, int chunks, void* stream) {
  return planes_zero_border_impl(planes, B, H, W, chunks, PXB, stream);
}
extern "C" int segmif_planes16_zero_border(void* planes, int B, int H, int W, int chunks, void* stream) {
  return planes_zero_border_impl(planes, B, H, W, chunks, PXH, stream);
}

extern "C" int segmif_planes_from_f32(const float* x, int ldx, void* planes, int B, int H, int W, int chunks, int chunk0,
                                      int nconv, void* stream) {
  return planes_from_f32_impl<false>(x, ldx, planes, B, H, W, chunks, chunk0, nconv, nullptr, 1, stream);
}
extern "C" int segmif_planes16_from_f32(const float* x, int ldx, void* planes, int B, int H, int W, int chunks, int chunk0,
                                        int nconv, uint32_t* amax, int amax_images, void* stream) {
  return planes_from_f32_impl<true>(x, ldx, planes, B, H, W, chunks, chunk0, nconv, amax, amax_images, stream);
}

extern "C" int segmif_conv3x3_c1_f16x3(const float* x, const float* w, const float* bias, const float* prelu, int act, void* planes,
                                       int chunks, int chunk0, float* out, int ldo, int B, int H, int W, uint32_t* amax,
                                       int amax_images, void* stream) {
  if (!x || !w || (!planes && !out) || B <= 0 || H <= 0 || W <= 0 || act < 0 || act > SEGMIF_ACT_PRELU) return SEGMIF_EINVAL;
  if (act == SEGMIF_ACT_PRELU && !prelu) return SEGMIF_EINVAL;
  if (planes && (chunk0 < 0 || chunk0 + 4 > chunks || ((uintptr_t)planes & 15) || (amax && amax_images != 1 && amax_images != B)))
    return SEGMIF_EINVAL;
  if (out && (ldo < 64 || (ldo & 3) || ((uintptr_t)out & 15))) return SEGMIF_EINVAL;
  const long long npix = (long long)H * W;
  const long long groups = (npix + C1_PIX - 1) / C1_PIX;
  // workgroups per image: about four per CU over the batch, each a run of `steps` 32-pixel groups inside one image
  long long per_image = (1024 + B - 1) / B;
  if (per_image > groups) per_image = groups;
  if (per_image < 1) per_image = 1;
  const int steps = (int)((groups + per_image - 1) / per_image);
  const unsigned gx = (unsigned)((groups + steps - 1) / steps);
  hipLaunchKernelGGL(conv3x3_c1_kernel, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, w, bias, prelu, act,
                     (unsigned char*)planes, chunks, chunk0, out, ldo, H, W, planes_hp(H), planes_wp(W), steps, amax, amax_images);
  return (int)hipGetLastError();
}

extern "C" int64_t segmif_planes_weight_bytes(int N, int Cin, int taps) {
  if (N <= 0 || N % 32 || Cin <= 0 || Cin % 16 || (taps != 9 && taps != 1)) return 0;
  return (int64_t)N * Cin * taps * 6;
}

extern "C" int segmif_planes_pack_weight(const float* packed, int N, int Cin, int taps, int ldw, void* out, void* stream) {
  if (!packed || !out || segmif_planes_weight_bytes(N, Cin, taps) == 0 || ldw < taps * Cin) return SEGMIF_EINVAL;
  const long long total = (long long)N * Cin * taps;
  hipLaunchKernelGGL(planes_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     packed, N, Cin, taps, ldw, total, (uint16_t*)out);
  return (int)hipGetLastError();
}

extern "C" int64_t segmif_planes16_weight_bytes(int N, int Cin, int taps) {
  const int64_t image = segmif_planes_weight_bytes(N, Cin, taps);  // same 96-byte rows, then N floats: 2^-e(n)
  return image ? image + (int64_t)N * 4 : 0;
}

extern "C" int segmif_planes16_pack_weight(const float* packed, int N, int Cin, int taps, int ldw, void* out, void* stream) {
  if (!packed || !out || segmif_planes_weight_bytes(N, Cin, taps) == 0 || ldw < taps * Cin) return SEGMIF_EINVAL;
  const long long total = (long long)N * Cin * taps;
  float* inv_scale = reinterpret_cast<float*>((unsigned char*)out + total * 6);
  hipLaunchKernelGGL(planes16_weight_scale_kernel, dim3((unsigned)N), dim3(64), 0, (hipStream_t)stream, packed, taps * Cin, ldw,
                     inv_scale);
  hipLaunchKernelGGL(planes16_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     packed, N, Cin, taps, ldw, total, inv_scale, (uint16_t*)out);
  return (int)hipGetLastError();
}

static int conv3x3_planes_impl(const SegmifConvPlanes* d, bool f16, uint32_t* amax, int amax_images, void* stream);

extern "C" int segmif_conv3x3_planes_bf16x6(const SegmifConvPlanes* d, void* stream) {
  return conv3x3_planes_impl(d, false, nullptr, 1, stream);
}

extern "C" int segmif_conv3x3_planes_f16x3(const SegmifConvPlanes* d, uint32_t* amax, int amax_images, void* stream) {
  return conv3x3_planes_impl(d, true, amax, amax_images, stream);
}

static int conv3x3_planes_impl(const SegmifConvPlanes* d, bool f16, uint32_t* amax, int amax_images, void* stream) {
  if (amax && amax_images != 1 && (!d || amax_images != d->B)) return SEGMIF_EINVAL;
  if (!d || !d->planes_in || !d->wt || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cin <= 0 || d->cin % 16) return SEGMIF_EINVAL;
  if (d->dil != 1 && d->dil != 2) return SEGMIF_EINVAL;
  if (d->act != SEGMIF_ACT_NONE && d->act != SEGMIF_ACT_RELU && d->act != SEGMIF_ACT_PRELU) return SEGMIF_EINVAL;
  if (d->act == SEGMIF_ACT_PRELU && !d->prelu) return SEGMIF_EINVAL;
  const bool fuse = d->w1 != nullptr;
  if (!fuse && !d->out && !d->planes_out) return SEGMIF_EINVAL;
  if (d->out && (d->ldo % 4 || ((uintptr_t)d->out & 15))) return SEGMIF_EINVAL;
  if (d->bias && ((uintptr_t)d->bias & 15)) return SEGMIF_EINVAL;
  PlanesConvK k;
  k.pin = (const unsigned char*)d->planes_in;
  k.pout = (unsigned char*)d->planes_out;
  k.wt = (const unsigned char*)d->wt;
  static segmif::PerDeviceValue<const float*> zero_bias_dev;  // a __device__ symbol has one address per device
  const float*& zero_bias = zero_bias_dev.here();
  if (!zero_bias && hipGetSymbolAddress((void**)&zero_bias, HIP_SYMBOL(planes_zero_bias)) != hipSuccess) return SEGMIF_EINVAL;
  k.bias = d->bias ? d->bias : zero_bias;
  k.prelu = d->prelu;
  k.out = d->out;
  k.ldo = d->ldo;
  k.B = d->B; k.H = d->H; k.W = d->W;
  k.Hp = planes_hp(d->H); k.Wp = planes_wp(d->W);
  k.nchunks = d->cin / 16;
  k.in_total = d->in_chunks;
  k.out_total = d->out_chunks;
  k.out_chunk0 = d->out_chunk0;
  k.act = d->act;
  if (k.nchunks > k.in_total) return SEGMIF_EINVAL;
  if (k.pout && (k.out_chunk0 < 0 || k.out_chunk0 + 2 > k.out_total)) return SEGMIF_EINVAL;
  if (k.pout && k.pout == k.pin && k.out_chunk0 < k.nchunks) return SEGMIF_EINVAL;  // would overwrite its own input
  if ((long long)k.Hp * k.Wp * (f16 ? PXH : PXB) >= (1ll << 32)) return SEGMIF_EINVAL;
  k.amax = amax;
  k.amax_images = amax_images;
  k.wscale = reinterpret_cast<const float*>(k.wt + (long long)32 * d->cin * 9 * 6);  // f16x3 images end with the row scales
  k.w1scale = d->w1 ? reinterpret_cast<const float*>((const unsigned char*)d->w1 + (long long)64 * (d->cin + 32) * 6) : nullptr;
  k.w1 = (const unsigned char*)d->w1;
  k.bias1 = d->bias1 ? d->bias1 : zero_bias;
  k.res = d->res;
  k.res_planes = (f16 && fuse && !d->res) ? d->res_from_planes : 0;
  k.out1 = d->out1;
  k.ldr = d->ldr; k.ldo1 = d->ldo1; k.act1 = d->act1;
  if (fuse) {
    if (!d->out1 || d->ldo1 % 4 || ((uintptr_t)d->out1 & 15) || (d->res && (d->ldr % 4 || ((uintptr_t)d->res & 15))) ||
        (d->bias1 && ((uintptr_t)d->bias1 & 15)) || (d->act1 != SEGMIF_ACT_NONE && d->act1 != SEGMIF_ACT_RELU))
      return SEGMIF_EINVAL;
  }
  k.tiles_x = (d->W + TW - 1) / TW;
  k.tiles_y = (d->H + TH - 1) / TH;
  hipStream_t s = (hipStream_t)stream;
  if (f16) {
    // plain conv on heights that are whole 16-row patches: four sub-tiles per wave (16 x 32 patches; +5..6 % over two,
    // profiles/r04_planes_sub4_ab.txt).  SEGMIF_PLANES_SUB=2 (read once per process) keeps the two-sub-tile kernel for A/B runs.
    static const bool sub4 = [] { const char* e = getenv("SEGMIF_PLANES_SUB"); return !(e && e[0] == '2'); }();
    // (r6) the DRDB shapes of an inference forward take the LEAN instantiations (see the kernel); SEGMIF_PLANES_LEAN=0 keeps the general ones
    static const bool lean_on = [] { const char* e = getenv("SEGMIF_PLANES_LEAN"); return !(e && e[0] == '0'); }();
    const bool relu = d->act == SEGMIF_ACT_RELU && !d->out;
    if (lean_on && !d->out) {
      if (!fuse && sub4 && d->H % 16 == 0 && k.pout) return d->dil == 2 ? launch<2, false, true, 4, true>(k, s) : launch<1, false, true, 4, true>(k, s);
      if (fuse && relu && d->dil == 2 && d->act1 == SEGMIF_ACT_RELU && !d->res && k.res_planes && !k.pout) return launch<2, true, true, 2, true>(k, s);
    }
    if (sub4 && !fuse && d->H % 16 == 0) return d->dil == 2 ? launch<2, false, true, 4>(k, s) : launch<1, false, true, 4>(k, s);
    if (d->dil == 2) return fuse ? launch<2, true, true>(k, s) : launch<2, false, true>(k, s);
    return fuse ? launch<1, true, true>(k, s) : launch<1, false, true>(k, s);
  }
  if (d->dil == 2) return fuse ? launch<2, true, false>(k, s) : launch<2, false, false>(k, s);
  return fuse ? launch<1, true, false>(k, s) : launch<1, false, false>(k, s);
}
