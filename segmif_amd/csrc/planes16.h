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
  const f2 back = __builtin_convertvector(a, f2);
  const f2 res = {(x0 - back[0]) * LSCALE, (x1 - back[1]) * LSCALE};
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

__device__ __forceinline__ float abs_max8(const float* y, float m) {
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(y[e]));
  return m;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Fold this wave's maximum (non-negative, or NaN) into the slot.  The plain load first: a tensor has a few hundred
// thousand waves and all but a handful carry a value below the running maximum - one atomic per wave on a single
// address serialises in the L2 (measured 10x on segmif_planes16_from_f32).
__device__ __forceinline__ void fold_max(uint32_t* slot, float lane_value) {
  const float m = wave_max(lane_value);
  if ((threadIdx.x & 63) == 0) {
    const uint32_t bits = __float_as_uint(m);  // monotonic for m >= 0; NaN reads as larger than inf
    if (bits > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, bits);
  }
}

}  // namespace p16
}  // namespace segmif
