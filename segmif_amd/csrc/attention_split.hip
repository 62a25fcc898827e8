// Spatial-reduction attention of the MiT encoder (core/mix_transformer.py:107-111) on the bf16 matrix pipe:
//   O = softmax(Q K^T * scale) V   per (batch, head), head_dim 64, Nk = a few hundred reduced keys.
//
// Same transposed data flow as csrc/attention.hip (a query is a lane; S^T = K Q^T, O^T += V^T P^T; softmax per lane
// plus one cross-half shuffle; the N x Nk score matrix never exists), but every product runs as six
// v_mfma_f32_32x32x16_bf16 over three-way bf16 splits of both operands (x = x0 + x1 + x2, round to nearest at each
// step: fp32-class, see csrc/conv3x3_planes.hip) instead of v_mfma_f32_32x32x2_f32: 48 matrix instructions of 32
// cycles per 32-key tile instead of 64 of 64 cycles.  The fp32 kernel sat at 60 % of the fp32 matrix pipe
// (profiles/r02_pmc_sq_counters_planes.txt); vector work next to a busy matrix pipe is what costs here, so it is kept
// off the inner loop:
//   * K and V are split ONCE per call by a small pack kernel into ready-made LDS images, one per 32-key tile, which
//     the attention workgroups pull in by LDS-DMA (no registers, no ds_write, no arithmetic);
//   * Q is split once per wave (its 32 queries stay in registers for the whole key loop);
//   * per tile a lane splits only its 16 probabilities, and rescales O only when some lane's running maximum moved.
// Operand order.  The MFMA takes, per lane-half h and K-step s, 8 consecutive K-slots.  For S^T the slots are head
// dimensions: position 16 s + 8 h + j <-> dimension 16 s + 4 h + (j & 3) + 8 (j >> 2), which is both what a lane gets
// from two float4 loads of its query row and the order the K image is written in.  For O^T the slots are keys, and
// the accumulator registers 8 s .. 8 s + 7 of S^T (rows (v&3) + 8 (v>>2) + 4 h) ARE the 8 slots of step s - key
// 16 s + 4 h + (j & 3) + 8 (j >> 2) again - so P feeds the second product without leaving its registers and the V
// image is stored transposed in that key order.
//
// F16 = true ("f16x3", segmif_sr_attention_split16_f32; the format: csrc/conv3x3_planes.hip / planes16.h): THREE
// v_mfma_f32_32x32x16_f16 per product instead of six.  K and V^T play the weights' role (three half planes W0 | W - W0 |
// 2^-11 W0 of the values scaled by a power of two), Q and P the activations' (half pairs x = hi + 2^-11 lo, split in
// registers).  The scale is per (key tile, head, image), found by the pack workgroup that writes the tile's image: the largest
// |K| (|V|) of the tile lands in [2^14, 2^15).  2^-eK multiplies the tile's scores inside the exponent's fma (softmax needs
// the scores' ABSOLUTE accuracy: a tile-wide scale gives the small ones at least the precision of the largest); 2^-eV is the
// unit of the running output sums, which are rescaled - by an exact power of two - when a tile's unit differs from the
// running one, together with the online softmax's own rescale.  Range slots: max |scaled Q| per image (P lies in [0, 1]; K
// and V are in range by construction).
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>

#include "planes16.h"
#include "segmif_hip.h"

namespace p16 = segmif::p16;

#ifndef ATTN_ABL
#define ATTN_ABL 0  // tuning aid (tools/attn_ablate.sh): 1 = no tile DMA after the first, 2 = no softmax arithmetic, 4 = no Q K^T MFMAs, 8 = no P V MFMAs, 16 = no per-tile barrier (with 1), 32 = no output stores
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KT = 32;                      // keys per tile
// Tile image, per operand format.  bf16x6: three bf16 planes.  f16x3 (r6): TWO half planes W0 | W - W0 - the third operand of the
// f16x3 product, 2^-11 W0, is made in registers from W0 (v_pk_mul_f16 by 2^-11: exactly the value the pack kernel used to store;
// gemm_pairs.hip does the same): a third less LDS-DMA traffic and a third fewer fragment reads, and the tile stream is what
// bounds this kernel (profiles/r06_attn_ablation.txt: removing every MFMA and the softmax arithmetic gains 18 %, removing the
// tile DMA alone 17 %).
template <bool F16>
struct Img {
  static constexpr int NPL = F16 ? 2 : 3;             // planes stored per value
  static constexpr int KPITCH = NPL * 128 + 16;       // K image row (one key): [plane][64 positions] + 16 B (conflict-free ds_read_b128)
  static constexpr int VPITCH = NPL * 64 + 16;        // V^T image row (one head dimension): [plane][32 key positions] + 16 B
  static constexpr int K_BYTES = KT * KPITCH;         // 12800 | 8704
  static constexpr int V_BYTES = 64 * VPITCH;         // 13312 | 9216
  static constexpr int BYTES = F16 ? 18432 : 26624;   // K image + V^T image, rounded up to whole 1 KB LDS-DMA wave instructions
  static constexpr int O_TILE_SCALES = K_BYTES + V_BYTES;  // f16x3: floats 2^-eK, 2^-eV of the tile (in the round-up's slack)
  static_assert(O_TILE_SCALES + 8 <= BYTES, "tile scales must fit the image");
};
constexpr int IMG_MAX = Img<false>::BYTES;  // what segmif_sr_attention_split_workspace sizes a tile for (either format)
constexpr int PX6[6] = {2, 1, 0, 1, 0, 0};  // six products, least significant first: plane of the first operand ...
constexpr int PY6[6] = {0, 1, 2, 0, 1, 0};  // ... and of the second

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
__device__ __forceinline__ bf16x8 op(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 mma6(const u32x4* a, const u32x4* b, f32x16 acc) {
#pragma unroll
  for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(a[PX6[t]]), op(b[PY6[t]]), acc, 0, 0, 0);
  return acc;
}
// f16x3: an activation operand is a half pair (hi, lo = 2^11 residual); products least significant first: lo W0s, hi Wl, hi W0
struct Op2 {
  u32x4 hi, lo;
};
__device__ __forceinline__ Op2 split8h(const f32x4 a, const f32x4 b) {
  const float y[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  Op2 o;
  p16::split8(y, o.hi, o.lo);
  return o;
}
__device__ __forceinline__ f16x8 oph(const u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ u32x4 times_2m11(const u32x4 v) {  // 8 halves x 2^-11 (v_pk_mul_f16; exact up to the half's own rounding)
  const f16x8 s = {(_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f,
                   (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f};
  return __builtin_bit_cast(u32x4, oph(v) * s);
}
__device__ __forceinline__ f32x16 mma3(const u32x4* w, const Op2& x, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oph(w[2]), oph(x.lo), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oph(w[1]), oph(x.hi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oph(w[0]), oph(x.hi), acc, 0, 0, 0);
  return acc;
}
// three half planes W0 | W - W0 | 2^-11 W0 of two adjacent scaled values -> one dword per plane
__device__ __forceinline__ void split3h(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 w0 = {(_Float16)x0, (_Float16)x1};
  const h2 wl = {(_Float16)(x0 - (float)w0[0]), (_Float16)(x1 - (float)w0[1])};
  const h2 ws = {(_Float16)((float)w0[0] * (1.f / p16::LSCALE)), (_Float16)((float)w0[1] * (1.f / p16::LSCALE))};
  p0 = __builtin_bit_cast(uint32_t, w0);
  p1 = __builtin_bit_cast(uint32_t, wl);
  p2 = __builtin_bit_cast(uint32_t, ws);
}
// power of two that brings mx into [2^14, 2^15) (1 for zero / non-finite input)
__device__ __forceinline__ float pow2_scale(float mx) {
  int e = 0;
  if (mx >= 1e-30f && mx <= 3e38f) e = 14 - (int)((__float_as_uint(mx) >> 23) - 127);
  return ldexpf(1.f, e);
}

__device__ __forceinline__ int slot_to_index(int pos) {  // position 16 s + 8 h + j -> 16 s + 4 h + (j & 3) + 8 (j >> 2)
  const int s = pos >> 4, hh = (pos >> 3) & 1, j = pos & 7;
  return 16 * s + 4 * hh + (j & 3) + 8 * (j >> 2);
}
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {  // a wave-uniform pointer, into SGPRs
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const unsigned char*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}
// (r6) the SADDR form of the same instruction: wave-uniform base in SGPRs + a 32-bit lane offset (csrc/conv3x3_planes.hip, dma16s: no
// per-instruction address arithmetic, one VGPR read per lane instead of two, no write-after-read interlock on a shared address pair)
__device__ __forceinline__ void dma16s(const unsigned char* sbase, uint32_t voff, unsigned char* lds_wave_base) {
  const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One workgroup per (key tile, head, batch): the tile's K rows and V^T rows, split, in LDS-image order.
template <bool F16>
__global__ __launch_bounds__(256) void sr_attention_pack_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                                unsigned char* __restrict__ img, int Nk, int ldkv, int ntiles) {
  const int kt = blockIdx.x, head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int tid = threadIdx.x;
  const float* kb = k + (long long)b * Nk * ldkv + head * 64;
  const float* vb = v + (long long)b * Nk * ldkv + head * 64;
  using I = Img<F16>;
  unsigned char* dst = img + (((long long)b * heads + head) * ntiles + kt) * I::BYTES;
  // every thread's share stays in registers between the range pass and the split (KT * 32 / 256 = 4 K pairs, 4 V pairs)
  f32x2 kv2[4], vv2[4];
  float mk = 0.f, mv = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // K: row = key, positions pp, pp + 1 = adjacent head dimensions
    const int u = tid + 256 * i;
    const int row = u >> 5, pp = 2 * (u & 31);
    const int key = kt * KT + row, d = slot_to_index(pp);
    kv2[i] = f32x2{0.f, 0.f};
    if (key < Nk) kv2[i] = *reinterpret_cast<const f32x2*>(kb + (long long)key * ldkv + d);
    mk = fmaxf(mk, fmaxf(fabsf(kv2[i][0]), fabsf(kv2[i][1])));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // V^T: row = head dimension, positions pp, pp + 1 = adjacent keys
    const int u = tid + 256 * i;
    const int d = u & 63, pp = 2 * (u >> 6);
    const int key = kt * KT + slot_to_index(pp);
    vv2[i][0] = key < Nk ? vb[(long long)key * ldkv + d] : 0.f;
    vv2[i][1] = key + 1 < Nk ? vb[(long long)(key + 1) * ldkv + d] : 0.f;
    mv = fmaxf(mv, fmaxf(fabsf(vv2[i][0]), fabsf(vv2[i][1])));
  }
  float sk = 1.f, sv = 1.f;
  if constexpr (F16) {  // tile-wide power-of-two scales (a NaN in the tile: fmaxf drops it, the values themselves carry it on)
    __shared__ float red[2][4];
    mk = p16::wave_max(mk);
    mv = p16::wave_max(mv);
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = mk;
      red[1][tid >> 6] = mv;
    }
    __syncthreads();
    sk = pow2_scale(fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
    sv = pow2_scale(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
    if (tid == 0) {
      reinterpret_cast<float*>(dst + I::O_TILE_SCALES)[0] = 1.f / sk;  // exact: powers of two
      reinterpret_cast<float*>(dst + I::O_TILE_SCALES)[1] = 1.f / sv;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i;
    const int row = u >> 5, pp = 2 * (u & 31);
    uint32_t a, bb, c;
    if constexpr (F16) split3h(kv2[i][0] * sk, kv2[i][1] * sk, a, bb, c);
    else split3(kv2[i][0], kv2[i][1], a, bb, c);
    unsigned char* o = dst + row * I::KPITCH + pp * 2;
    *reinterpret_cast<uint32_t*>(o) = a;
    *reinterpret_cast<uint32_t*>(o + 128) = bb;
    if constexpr (!F16) *reinterpret_cast<uint32_t*>(o + 256) = c;  // (f16x3: 2^-11 W0 is made by the reader)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i;
    const int d = u & 63, pp = 2 * (u >> 6);
    uint32_t a, bb, c;
    if constexpr (F16) split3h(vv2[i][0] * sv, vv2[i][1] * sv, a, bb, c);
    else split3(vv2[i][0], vv2[i][1], a, bb, c);
    unsigned char* o = dst + I::K_BYTES + d * I::VPITCH + pp * 2;
    *reinterpret_cast<uint32_t*>(o) = a;
    *reinterpret_cast<uint32_t*>(o + 64) = bb;
    if constexpr (!F16) *reinterpret_cast<uint32_t*>(o + 128) = c;
  }
}

// Workgroup = 4 waves = 128 queries of one (batch, head); key tiles double-buffered in LDS (one barrier per tile).
template <bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void sr_attention_split_kernel(const float* __restrict__ q, const unsigned char* __restrict__ img,
                                                                 float* __restrict__ out, int N, int Nk, int ldq, int ldo,
                                                                 float scale, int ntiles, uint32_t* amax, int amax_images,
                                                                 int out_pairs, uint32_t* out_amax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][IMG]
  using I = Img<F16>;
  constexpr int IMG = I::BYTES, NPL = I::NPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool q_ok = qi < N;
  const unsigned char* src = img + ((long long)b * heads + head) * ntiles * IMG;

  // 26 (f16x3: 18) wave-sized (1 KB) DMA instructions per tile, dealt round-robin to the four waves
  auto stage = [&](int kt, int buf) {
    const unsigned char* s = src + (long long)kt * IMG;
    unsigned char* d = smem + buf * IMG;
    for (int i = wave; i < IMG / 1024; i += 4) dma16s(uniform_ptr(s + i * 1024), (uint32_t)(lane * 16), d + i * 1024);
  };
  stage(0, 0);

  // the query row, split: K-step s = head dimensions 16 s + {4h .. 4h+3, 8 + 4h .. 8 + 4h+3}
  Op3 qp[F16 ? 1 : 4];
  Op2 qh[F16 ? 4 : 1];
  uint32_t amx = 0u;
  {
    const float* qrow = q + ((long long)b * N + (q_ok ? qi : 0)) * ldq + head * 64 + 4 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      if (q_ok) {
        lo = *reinterpret_cast<const f32x4*>(qrow + 16 * s);
        hi = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8);
      }
      // softmax_e(scale q.k) = softmax_2((scale log2(e) q).k): one multiply per query element here, a bare v_exp_f32 per
      // score in the key loop
      if constexpr (F16) {
        qh[s] = split8h(lo * (scale * 1.44269504088896340736f), hi * (scale * 1.44269504088896340736f));
        amx = p16::absmax_pk4(amx, qh[s].hi, qh[s].lo);
      } else {
        qp[s] = split8(lo * (scale * 1.44269504088896340736f), hi * (scale * 1.44269504088896340736f));
      }
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  float vunit = 1.f;  // f16x3: the running output sums are in units of 2^-eV of the last tile taken (o_true = o * vunit)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (!(ATTN_ABL & 1) && kt + 1 < ntiles) stage(kt + 1, cur ^ 1);  // buffer cur^1 was last read in iteration kt-1, before its closing barrier
    const unsigned char* Kt = smem + cur * IMG;
    const unsigned char* Vt = Kt + I::K_BYTES;
    float kinv = 1.f, vinv = 1.f;
    if constexpr (F16) {  // the tile's 2^-eK, 2^-eV (every lane reads the same two words)
      kinv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(Kt + I::O_TILE_SCALES)));
      vinv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(Kt + I::O_TILE_SCALES + 4)));
    }

    // ---- S^T = K (scale log2(e) Q)^T: scores in the base-2 exponent domain -----------------------
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      u32x4 kf[3];
#pragma unroll
      for (int k = 0; k < NPL; ++k) kf[k] = *reinterpret_cast<const u32x4*>(Kt + r * I::KPITCH + k * 128 + (16 * st + 8 * h) * 2);
      if constexpr (F16) kf[2] = times_2m11(kf[0]);
      if (ATTN_ABL & 4) asm volatile("" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]));
      else if constexpr (F16) s = mma3(kf, qh[st], s);
      else s = mma6(kf, qp[st].p, s);
    }
    // ---- per-lane online softmax over this lane's 16 keys (+ partner half) -----------------
    if ((kt + 1) * KT > Nk) {  // last, partial tile: keys past the end get no weight
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * KT + (e & 3) + 8 * (e >> 2) + 4 * h;
        s[e] = key < Nk ? s[e] : -INFINITY;
      }
    }
    float mx = s[0];
#if !(ATTN_ABL & 2)
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if constexpr (F16) mx *= kinv;  // (kinv > 0: the maximum commutes with the tile's scale)
    const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = F16 ? __builtin_amdgcn_exp2f(fmaf(s[e], kinv, -m_new)) : __builtin_amdgcn_exp2f(s[e] - m_new);
      psum += s[e];
    }
    const bool unit_moved = F16 && vinv != vunit;  // (uniform)
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0 || unit_moved) {  // some lane's maximum (or the unit) moved: rescale the running sums
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      const float beta = F16 ? alpha * (vunit / vinv) : alpha;  // (a ratio of two powers of two: exact)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= beta;
      vunit = vinv;
    }
    l_run += psum;
    m_run = m_new;
#endif
    // ---- O^T += V^T P^T: registers 8 sp .. 8 sp + 7 of s are the K-slots (keys) of step sp ------------
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const f32x4 pa = {s[8 * sp], s[8 * sp + 1], s[8 * sp + 2], s[8 * sp + 3]};
      const f32x4 pb = {s[8 * sp + 4], s[8 * sp + 5], s[8 * sp + 6], s[8 * sp + 7]};
      Op3 pk;
      Op2 ph;
      if constexpr (F16) ph = split8h(pa, pb);
      else pk = split8(pa, pb);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x4 vf[3];
#pragma unroll
        for (int k = 0; k < NPL; ++k)
          vf[k] = *reinterpret_cast<const u32x4*>(Vt + (dt * 32 + r) * I::VPITCH + k * 64 + (16 * sp + 8 * h) * 2);
        if constexpr (F16) vf[2] = times_2m11(vf[0]);
        if (ATTN_ABL & 8) asm volatile("" ::"v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(ph.hi), "v"(ph.lo));
        else if constexpr (F16) o[dt] = mma3(vf, ph, o[dt]);
        else o[dt] = mma6(vf, pk.p, o[dt]);
      }
    }
    if (!(ATTN_ABL & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile kt + 1 has landed
      __syncthreads();
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = vunit / l_tot;
  uint32_t oamx = 0u;
  // (r6) The output leaves through LDS.  A lane owns a QUERY: written straight from the accumulators, every store instruction was 64
  // fragments of 8 (pairs) or 16 bytes (fp32) in 32 different rows - 16 / 8 such instructions per wave - and the ablation of
  // tools/attn_ablate.sh put them at 31 % of the stage-3 call and 65 % of the stage-2 call (profiles/r06_attn_ablation.txt).  The
  // key-tile buffers are free after the loop's last barrier: each wave parks its 32 x 256-byte tile there (row pitch 272: 16-byte
  // aligned, 4 banks of skew per row) and stores it as 8 instructions of 4 rows x 256 contiguous bytes.  Both formats fill the same
  // tile: fp32 rows are 64 channels x 4 bytes, PAIRS rows 4 groups of [16 hi | 16 lo] halves.
  constexpr int OPITCH = 272;
  unsigned char* T = smem + wave * (32 * OPITCH);
  if (!(ATTN_ABL & 32)) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
        if (F16 && out_pairs) {
          // PAIRS output (gemm_pairs.hip: the proj Linear's A operand): channel c = 32 dt + 8 g + 4 h .. + 3 of the head lies in
          // 16-group c >> 4 at half position c & 15; same byte count as the fp32 row
          const int c = 32 * dt + 8 * g + 4 * h;
          uint32_t ha, la, hb, lb;
          p16::split2(w[0], w[1], ha, la);
          p16::split2(w[2], w[3], hb, lb);
          unsigned char* d8 = T + r * OPITCH + (c >> 4) * 64 + (c & 15) * 2;
          *reinterpret_cast<u32x2*>(d8) = u32x2{ha, hb};
          *reinterpret_cast<u32x2*>(d8 + 32) = u32x2{la, lb};
          if (q_ok) oamx = p16::absmax_pk(p16::absmax_pk(oamx, ha, la), hb, lb);
        } else {
          *reinterpret_cast<f32x4*>(T + r * OPITCH + (32 * dt + 8 * g + 4 * h) * 4) = w;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private tile: the wave's own LDS writes have landed; no barrier)
    __builtin_amdgcn_wave_barrier();
    const int q0 = blockIdx.x * 128 + wave * 32;
    unsigned char* obase = reinterpret_cast<unsigned char*>(out + ((long long)b * N + q0) * ldo + head * 64) + (lane & 15) * 16;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = 4 * it + (lane >> 4);
      const u32x4 v = *reinterpret_cast<const u32x4*>(T + row * OPITCH + (lane & 15) * 16);
      if (q0 + row < N) *reinterpret_cast<u32x4*>(obase + (long long)row * ldo * 4) = v;
    }
  } else {  // (ablation 32: the accumulators stay live - without this the whole key loop is dead code)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(o[dt][e] * inv));
  }
  if constexpr (F16) {
    if (amax) p16::fold_pat(amax, amax_images > 1 ? b : 0, amax_images > 1 ? b : 0, q_ok ? amx : 0u);
    if (out_pairs && out_amax) p16::fold_pat(out_amax, amax_images > 1 ? b : 0, amax_images > 1 ? b : 0, oamx);
  }
}

}  // namespace

extern "C" int64_t segmif_sr_attention_split_workspace(int B, int heads, int Nk) {
  if (B <= 0 || heads <= 0 || Nk <= 0) return 0;
  return (int64_t)B * heads * ((Nk + KT - 1) / KT) * IMG_MAX;
}

static int sr_attention_split_impl(bool f16, const float* q, const float* k, const float* v, float* out, void* workspace, int B,
                                   int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale, uint32_t* amax,
                                   int amax_images, void* stream, int out_pairs = 0, uint32_t* out_amax = nullptr) {
  if (!q || !k || !v || !out || !workspace || B <= 0 || heads <= 0 || N <= 0 || Nk <= 0 || hd != 64) return SEGMIF_EINVAL;
  if ((ldq | ldo) & 3 || (ldkv & 1)) return SEGMIF_EINVAL;
  if (((uintptr_t)q | (uintptr_t)out | (uintptr_t)workspace) & 15) return SEGMIF_EINVAL;
  if (((uintptr_t)k | (uintptr_t)v) & 7) return SEGMIF_EINVAL;
  if (amax && amax_images != 1 && amax_images != B) return SEGMIF_EINVAL;
  const int ntiles = (Nk + KT - 1) / KT;
  hipStream_t s = (hipStream_t)stream;
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)sr_attention_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Img<false>::BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sr_attention_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Img<true>::BYTES);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const dim3 pgrid((unsigned)ntiles, (unsigned)heads, (unsigned)B), grid((unsigned)((N + 127) / 128), (unsigned)heads, (unsigned)B);
  if (f16) {
    hipLaunchKernelGGL(sr_attention_pack_kernel<true>, pgrid, dim3(256), 0, s, k, v, (unsigned char*)workspace, Nk, ldkv, ntiles);
    hipLaunchKernelGGL(sr_attention_split_kernel<true>, grid, dim3(256), 2 * Img<true>::BYTES, s, q, (const unsigned char*)workspace, out, N, Nk,
                       ldq, ldo, scale, ntiles, amax, amax_images, out_pairs, out_amax);
  } else {
    hipLaunchKernelGGL(sr_attention_pack_kernel<false>, pgrid, dim3(256), 0, s, k, v, (unsigned char*)workspace, Nk, ldkv, ntiles);
    hipLaunchKernelGGL(sr_attention_split_kernel<false>, grid, dim3(256), 2 * Img<false>::BYTES, s, q, (const unsigned char*)workspace, out, N, Nk,
                       ldq, ldo, scale, ntiles, (uint32_t*)nullptr, 1, 0, (uint32_t*)nullptr);
  }
  return (int)hipGetLastError();
}

extern "C" int segmif_sr_attention_split_f32(const float* q, const float* k, const float* v, float* out, void* workspace, int B,
                                             int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale,
                                             void* stream) {
  return sr_attention_split_impl(false, q, k, v, out, workspace, B, heads, N, Nk, hd, ldq, ldkv, ldo, scale, nullptr, 1, stream);
}

extern "C" int segmif_sr_attention_split16_f32(const float* q, const float* k, const float* v, float* out, void* workspace, int B,
                                               int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale,
                                               uint32_t* amax, int amax_images, void* stream) {
  return sr_attention_split_impl(true, q, k, v, out, workspace, B, heads, N, Nk, hd, ldq, ldkv, ldo, scale, amax, amax_images, stream);
}

// (r5) the same with the output written in PAIRS format (gemm_pairs.hip; `out` is then a pairs buffer of ldo * 4 bytes per row)
// and max |out| reported to out_amax (indexed like amax; or NULL): the proj Linear reads it by LDS-DMA with no split of its own
extern "C" int segmif_sr_attention_split16_pairs_f32(const float* q, const float* k, const float* v, void* out, void* workspace, int B,
                                                     int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale,
                                                     uint32_t* amax, uint32_t* out_amax, int amax_images, void* stream) {
  return sr_attention_split_impl(true, q, k, v, reinterpret_cast<float*>(out), workspace, B, heads, N, Nk, hd, ldq, ldkv, ldo, scale, amax,
                                 amax_images, stream, 1, out_amax);
}
