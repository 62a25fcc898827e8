// Private to segmif_amd/csrc: the f16x3 operand format's device helpers (conv3x3_planes.hip holds the description).
// An activation is a pair of halves, x = hi + 2^-11 lo with hi = RN16(x), lo = RN16(2^11 (x - hi)); every producer of
// such planes folds max |x| into a guard slot so that the host can tell whether the tensor stayed inside the half's
// exponent range (ops.Planes16Guard).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace segmif {
namespace p16 {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int PIXEL_BYTES = 64;     // per 16-channel chunk: 2 planes x 16 halves
constexpr float LSCALE = 2048.f;    // the low half carries 2^11 x the residual

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const f2 v = {x0, x1};
  const h2 a = __builtin_convertvector(v, h2);  // v_cvt_pk_f16_f32, round to nearest even
  hi = __builtin_bit_cast(uint32_t, a);
  // (r6) 2^11 (x - hi) as ONE v_fma_mix_f32 per value - hi read in place as a half, times -2^11, plus 2^11 x: the same real number as
  // (x - float(hi)) * 2^11 (every step of either form is exact), without the two v_cvt_f32_f16 and the subtraction
  const f2 xs = v * LSCALE;
  f2 res;
  const float m = -LSCALE;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(res[0]) : "v"(hi), "v"(m), "v"(xs[0]));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(res[1]) : "v"(hi), "v"(m), "v"(xs[1]));
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(res, h2));
}

// 8 values (positions 8h .. 8h+7 of a chunk) -> one 16-byte piece per plane
__device__ __forceinline__ void split8(const float* y, u4& hi, u4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t a, b;
    split2(y[2 * e], y[2 * e + 1], a, b);
    hi[e] = a;
    lo[e] = b;
  }
}

// PAIRS rows (gemm_pairs.hip): a 16-channel group is 64 bytes, [16 hi | 16 lo].  Four consecutive lanes that hold 4 consecutive
// channels each (split into ha, hb / la, lb by split2) own one group; storing 8 bytes per lane and plane would be 32-byte
// partial writes with 32-byte holes - measured 10x slower than the fp32 store on the LayerNorm (profiles/r05_pairs_producers_log.txt).
// Instead the quad trades dwords (DPP quad_perm, no LDS) so that lane q stores the 16-byte piece q of the group: one fully
// coalesced dwordx4 store per lane.  -> the piece of lane (threadIdx.x & 3); all four lanes of the quad must be active.
__device__ __forceinline__ u4 quad_piece(uint32_t ha, uint32_t hb, uint32_t la, uint32_t lb) {
  // source lanes 2 (q & 1) and 2 (q & 1) + 1 of the quad: quad_perm [0, 2, 0, 2] = 0x88 and [1, 3, 1, 3] = 0xdd
  const uint32_t h0a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ha, 0x88, 0xf, 0xf, false);
  const uint32_t h0b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hb, 0x88, 0xf, 0xf, false);
  const uint32_t h1a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ha, 0xdd, 0xf, 0xf, false);
  const uint32_t h1b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hb, 0xdd, 0xf, 0xf, false);
  const uint32_t l0a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)la, 0x88, 0xf, 0xf, false);
  const uint32_t l0b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lb, 0x88, 0xf, 0xf, false);
  const uint32_t l1a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)la, 0xdd, 0xf, 0xf, false);
  const uint32_t l1b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lb, 0xdd, 0xf, 0xf, false);
  const bool low = (threadIdx.x & 2) != 0;  // lanes 2, 3 of the quad store the lo plane's pieces
  return u4{low ? l0a : h0a, low ? l0b : h0b, low ? l1a : h1a, low ? l1b : h1b};
}

// ---- range bookkeeping -----------------------------------------------------------------------------------------------
// A producer tracks the largest magnitude it wrote on the HIGH halves' bit patterns (|half| as a 15-bit unsigned integer is
// monotone in magnitude; +inf = 0x7c00, every NaN lies above it), two halves per operation (v_and_b32 + v_pk_max_u16 - the
// same count as one v_max_f32 per value, and unlike v_max_f32 an integer maximum cannot drop a NaN).  Values beyond the
// half's range have already become inf in the conversion, so the pattern carries exactly what the guard asks about.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t absmax_pk(uint32_t m, uint32_t halves) {
  const us2 a = __builtin_bit_cast(us2, m), b = __builtin_bit_cast(us2, halves & 0x7fff7fffu);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a, b));
}

// (r6) The same with the STICKY bit of the low halves: a value whose high half rounded to zero (|x| < 2^-25) while its low half did
// not is carried by the low half alone - fewer than 23 bits - yet read 0 on the high halves' patterns, like an exact zero, and
// passed the guard (the one hole of the range contract).  Such a value now counts as the smallest subnormal
// (pattern 1 = 2^-24): a tensor whose LARGEST magnitude is that small reports a maximum below the guard's lower bound and is
// repeated on bf16 triples; a healthy tensor's maximum is unaffected (its patterns are far above 1).  v_and + v_pk_min_u16 +
// v_pk_max_u16 per pair of values, in epilogues only.
// (r6, later) min(|lo|, 1) is written as v_pk_min_u16 by hand: from __builtin_elementwise_min against the constant {1, 1} hipcc makes
// "lo != 0 ? 1 : 0" per half - v_bfe, two v_cmp_ne_u16 + s_nop + v_cndmask, v_perm: 12 instructions and 4 wait states per pair of values,
// ~800 per 16 x 32 patch of the dominant conv's epilogue (found in the ISA behind profiles/r06_tail_epilogue_timeline.txt).
__device__ __forceinline__ uint32_t sticky_pk(uint32_t lo_abs) {  // per half: 1 where the (sign-stripped) half is non-zero
  uint32_t s;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(s) : "v"(lo_abs), "v"(0x00010001u));
  return s;
}
__device__ __forceinline__ uint32_t absmax_pk(uint32_t m, uint32_t hi, uint32_t lo) {
  const uint32_t s = sticky_pk(lo & 0x7fff7fffu);
  return absmax_pk(__builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, m), __builtin_bit_cast(us2, s))), hi);
}

// four dwords of each plane at once: the sticky bits of the four low dwords are OR-ed first (a half of the OR is non-zero exactly when
// one of its four halves is: the same maximum as four absmax_pk calls, 14 instructions instead of 20)
__device__ __forceinline__ uint32_t absmax_pk4(uint32_t m, const u4& hi, const u4& lo) {
  const uint32_t s = sticky_pk((lo[0] | lo[1] | lo[2] | lo[3]) & 0x7fff7fffu);
  m = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, m), __builtin_bit_cast(us2, s)));
  return absmax_pk(absmax_pk(absmax_pk(absmax_pk(m, hi[0]), hi[1]), hi[2]), hi[3]);
}

__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
    v = v > w ? v : w;
  }
  return v;
}

__device__ __forceinline__ float wave_max(float v) {  // (weights' row scales: finite by construction)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The look before the atomic: an atomic load (hipcc emits it with the cache-bypassing sc0 sc1 bits, so it sees other CUs'
// atomics; a plain load would keep hitting a stale line in the CU's L1 and every wave would go on to the atomic).  It has a
// price: its LATENCY (several microseconds under load) is what the reporting wave ends on, and atomics on one cache line queue
// (~10 ns each; the slot words of 16 neighbouring images share a line).  Fine for kernels whose waves live for tens of
// microseconds - the convs, attention, the CrossPath tail: one look per wave or per persistent workgroup.  NOT for row
// kernels whose waves live for a microsecond: the first LayerNorm that wrote half pairs spent 330 us instead of 22 this way
// (profiles/r05_pairs_producers_log.txt) - those use fold_pat_block / fold_pat_async below.
__device__ __forceinline__ uint32_t slot_peek(const uint32_t* slot) {
  return __atomic_load_n(slot, __ATOMIC_RELAXED);
}

// Fold this wave's running maximum (absmax_pk patterns) into the range slots of images b_lo .. b_hi (a wave whose rows
// straddle an image boundary reports to both: conservative).  A slot holds the IEEE bit pattern of a non-negative float
// (or NaN: it compares above inf).  The plain load first: a tensor has a few hundred thousand waves and all but a handful
// carry a value below the running maximum - one atomic per wave on a single address serialises in the L2 (measured 10x
// on segmif_planes16_from_f32).
__device__ __forceinline__ void fold_pat(uint32_t* slots, int b_lo, int b_hi, uint32_t pat) {
  uint32_t v = (pat & 0xffffu) > (pat >> 16) ? (pat & 0xffffu) : (pat >> 16);
  v = wave_umax(v);
  if ((threadIdx.x & 63) == 0 && v) {
    const uint32_t bits = __float_as_uint((float)__builtin_bit_cast(_Float16, (unsigned short)v));  // exact; NaN stays NaN
    for (int b = b_lo; b <= b_hi; ++b)
      if (bits > slot_peek(slots + b)) atomicMax(slots + b, bits);
  }
}

// fold_pat without the look: the atomic is issued unconditionally and its result is not used, so the wave does not wait for it
// (a no-return global_atomic_umax leaves like a store).  For kernels whose waves live for a microsecond - the LayerNorm that
// writes half pairs - the look's LATENCY (a cache-bypassing load: several microseconds under load) is what costs: every wave
// ends on it and holds its slot meanwhile (measured: 20 us -> 190 us).  Such a kernel spreads its reports over several rows of
// slots instead, so that the atomics do not queue on one address.
__device__ __forceinline__ void fold_pat_async(uint32_t* slots, int b_lo, int b_hi, uint32_t pat) {
  uint32_t v = (pat & 0xffffu) > (pat >> 16) ? (pat & 0xffffu) : (pat >> 16);
  v = wave_umax(v);
  if ((threadIdx.x & 63) == 0 && v) {
    const uint32_t bits = __float_as_uint((float)__builtin_bit_cast(_Float16, (unsigned short)v));
    for (int b = b_lo; b <= b_hi; ++b) (void)__hip_atomic_fetch_max(slots + b, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same for a whole workgroup of up to 16 waves reporting to ONE image: the waves' maxima meet in LDS and one lane makes the
// slot access (row kernels whose waves are short: see slot_peek).  Must be reached by every wave of the workgroup that is
// still alive, partial waves included (lanes that left a wave early count as zero).
__device__ __forceinline__ void fold_pat_block(uint32_t* slots, int b, uint32_t pat) {
  __shared__ uint32_t red_pat[16];
  uint32_t v = (pat & 0xffffu) > (pat >> 16) ? (pat & 0xffffu) : (pat >> 16);
  v = wave_umax(v);
  const int nw = (int)((blockDim.x * blockDim.y * blockDim.z + 63) >> 6);
  if (threadIdx.x < 16) red_pat[threadIdx.x] = 0u;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red_pat[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw && w < 16; ++w) v = red_pat[w] > v ? red_pat[w] : v;
    if (v) {  // (no look: the workgroup would sit on the load's latency; one fire-and-forget atomic per workgroup)
      const uint32_t bits = __float_as_uint((float)__builtin_bit_cast(_Float16, (unsigned short)v));
      (void)__hip_atomic_fetch_max(slots + b, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The same bookkeeping on fp32 values a kernel is ABOUT to split (when touching the split halves again would cost registers):
// running maximum of |x| as an IEEE bit pattern - an integer maximum, so a NaN (any sign) is kept, not dropped.
__device__ __forceinline__ uint32_t absmax_bits(uint32_t m, float a, float b) {
  const uint32_t x = __float_as_uint(a) & 0x7fffffffu, y = __float_as_uint(b) & 0x7fffffffu;
  const uint32_t t = x > y ? x : y;
  return m > t ? m : t;  // (v_max3_u32)
}
__device__ __forceinline__ void fold_bits(uint32_t* slots, int b_lo, int b_hi, uint32_t bits) {
  bits = wave_umax(bits);
  if ((threadIdx.x & 63) == 0 && bits) {
    for (int b = b_lo; b <= b_hi; ++b)
      if (bits > slot_peek(slots + b)) atomicMax(slots + b, bits);
  }
}

// The power of two that puts the largest magnitude recorded in n range slots (IEEE bit patterns) into [2^13, 2^14): what the
// training kernels multiply their staged operands by before splitting them into halves (exact, taken out again afterwards).
// All-zero slots: 1.  An inf / NaN maximum: NaN - the consumer's whole output turns NaN instead of quietly wrong.
// (n <= 64 words, read one per lane and reduced across the wave: a scalar loop over 48 words was a chain of 48 dependent
// loads at the head of every workgroup - a third of a 30 us workgroup in the training step's convs.)
__device__ __forceinline__ float scale_of(uint32_t mx) {  // mx: IEEE bit pattern of a maximum of magnitudes
  if (mx >= 0x7f800000u) return __uint_as_float(0x7fc00000u);
  if (!mx) return 1.f;
  int eb = 13 - ((int)(mx >> 23) - 127) + 127;
  eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
  return __uint_as_float((uint32_t)eb << 23);
}
__device__ __forceinline__ float range_scale(const uint32_t* slots, int n) {
  const int lane = threadIdx.x & 63;
  return scale_of(wave_umax(lane < n ? slots[lane] : 0u));
}

}  // namespace p16
}  // namespace segmif
