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
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>

#include "segmif_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KT = 32;                      // keys per tile
constexpr int KPITCH = 3 * 128 + 16;        // K image row (one key): [plane][64 positions] bf16 + 16 B (conflict-free ds_read_b128)
constexpr int VPITCH = 3 * 64 + 16;         // V^T image row (one head dimension): [plane][32 key positions] bf16 + 16 B
constexpr int K_BYTES = KT * KPITCH;        // 12800
constexpr int V_BYTES = 64 * VPITCH;        // 13312
constexpr int IMG = 26624;                  // K image + V^T image, rounded up to 26 x 1 KB (one LDS-DMA wave instruction each)
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
__device__ __forceinline__ int slot_to_index(int pos) {  // position 16 s + 8 h + j -> 16 s + 4 h + (j & 3) + 8 (j >> 2)
  const int s = pos >> 4, hh = (pos >> 3) & 1, j = pos & 7;
  return 16 * s + 4 * hh + (j & 3) + 8 * (j >> 2);
}
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One workgroup per (key tile, head, batch): the tile's K rows and V^T rows, split, in LDS-image order.
__global__ __launch_bounds__(256) void sr_attention_pack_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                                unsigned char* __restrict__ img, int Nk, int ldkv, int ntiles) {
  const int kt = blockIdx.x, head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int tid = threadIdx.x;
  const float* kb = k + (long long)b * Nk * ldkv + head * 64;
  const float* vb = v + (long long)b * Nk * ldkv + head * 64;
  unsigned char* dst = img + (((long long)b * heads + head) * ntiles + kt) * IMG;
  for (int u = tid; u < KT * 32; u += 256) {  // K: row = key, positions pp, pp + 1 = adjacent head dimensions
    const int row = u >> 5, pp = 2 * (u & 31);
    const int key = kt * KT + row, d = slot_to_index(pp);
    f32x2 val = {0.f, 0.f};
    if (key < Nk) val = *reinterpret_cast<const f32x2*>(kb + (long long)key * ldkv + d);
    uint32_t a, bb, c;
    split3(val[0], val[1], a, bb, c);
    unsigned char* o = dst + row * KPITCH + pp * 2;
    *reinterpret_cast<uint32_t*>(o) = a;
    *reinterpret_cast<uint32_t*>(o + 128) = bb;
    *reinterpret_cast<uint32_t*>(o + 256) = c;
  }
  for (int u = tid; u < 64 * 16; u += 256) {  // V^T: row = head dimension, positions pp, pp + 1 = adjacent keys
    const int d = u & 63, pp = 2 * (u >> 6);
    const int key = kt * KT + slot_to_index(pp);
    const float v0 = key < Nk ? vb[(long long)key * ldkv + d] : 0.f;
    const float v1 = key + 1 < Nk ? vb[(long long)(key + 1) * ldkv + d] : 0.f;
    uint32_t a, bb, c;
    split3(v0, v1, a, bb, c);
    unsigned char* o = dst + K_BYTES + d * VPITCH + pp * 2;
    *reinterpret_cast<uint32_t*>(o) = a;
    *reinterpret_cast<uint32_t*>(o + 64) = bb;
    *reinterpret_cast<uint32_t*>(o + 128) = c;
  }
}

// Workgroup = 4 waves = 128 queries of one (batch, head); key tiles double-buffered in LDS (one barrier per tile).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void sr_attention_split_kernel(const float* __restrict__ q, const unsigned char* __restrict__ img,
                                                                 float* __restrict__ out, int N, int Nk, int ldq, int ldo,
                                                                 float scale, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][IMG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool q_ok = qi < N;
  const unsigned char* src = img + ((long long)b * heads + head) * ntiles * IMG;

  // 26 wave-sized (1 KB) DMA instructions per tile: waves 0, 1 issue 7, waves 2, 3 issue 6
  auto stage = [&](int kt, int buf) {
    const unsigned char* s = src + (long long)kt * IMG;
    unsigned char* d = smem + buf * IMG;
    for (int i = wave; i < IMG / 1024; i += 4) dma16(s + i * 1024 + lane * 16, d + i * 1024);
  };
  stage(0, 0);

  Op3 qp[4];  // the query row, split: K-step s = head dimensions 16 s + {4h .. 4h+3, 8 + 4h .. 8 + 4h+3}
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
      qp[s] = split8(lo * (scale * 1.44269504088896340736f), hi * (scale * 1.44269504088896340736f));
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) stage(kt + 1, cur ^ 1);  // buffer cur^1 was last read in iteration kt-1, before its closing barrier
    const unsigned char* Kt = smem + cur * IMG;
    const unsigned char* Vt = Kt + K_BYTES;

    // ---- S^T = K (scale log2(e) Q)^T: scores in the base-2 exponent domain -----------------------
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      u32x4 kf[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) kf[k] = *reinterpret_cast<const u32x4*>(Kt + r * KPITCH + k * 128 + (16 * st + 8 * h) * 2);
      s = mma6(kf, qp[st].p, s);
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
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = __builtin_amdgcn_exp2f(s[e] - m_new);
      psum += s[e];
    }
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {  // some lane's maximum moved: rescale the running sums
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
    // ---- O^T += V^T P^T: registers 8 sp .. 8 sp + 7 of s are the K-slots (keys) of step sp ------------
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const Op3 pk = split8(f32x4{s[8 * sp], s[8 * sp + 1], s[8 * sp + 2], s[8 * sp + 3]},
                            f32x4{s[8 * sp + 4], s[8 * sp + 5], s[8 * sp + 6], s[8 * sp + 7]});
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x4 vf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
          vf[k] = *reinterpret_cast<const u32x4*>(Vt + (dt * 32 + r) * VPITCH + k * 64 + (16 * sp + 8 * h) * 2);
        o[dt] = mma6(vf, pk.p, o[dt]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile kt + 1 has landed
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    float* orow = out + ((long long)b * N + qi) * ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g + 4 * h) = w;
      }
  }
}

}  // namespace

extern "C" int64_t segmif_sr_attention_split_workspace(int B, int heads, int Nk) {
  if (B <= 0 || heads <= 0 || Nk <= 0) return 0;
  return (int64_t)B * heads * ((Nk + KT - 1) / KT) * IMG;
}

extern "C" int segmif_sr_attention_split_f32(const float* q, const float* k, const float* v, float* out, void* workspace, int B,
                                             int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale,
                                             void* stream) {
  if (!q || !k || !v || !out || !workspace || B <= 0 || heads <= 0 || N <= 0 || Nk <= 0 || hd != 64) return SEGMIF_EINVAL;
  if ((ldq | ldo) & 3 || (ldkv & 1)) return SEGMIF_EINVAL;
  if (((uintptr_t)q | (uintptr_t)out | (uintptr_t)workspace) & 15) return SEGMIF_EINVAL;
  if (((uintptr_t)k | (uintptr_t)v) & 7) return SEGMIF_EINVAL;
  const int ntiles = (Nk + KT - 1) / KT;
  hipStream_t s = (hipStream_t)stream;
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)sr_attention_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipLaunchKernelGGL(sr_attention_pack_kernel, dim3((unsigned)ntiles, (unsigned)heads, (unsigned)B), dim3(256), 0, s, k, v,
                     (unsigned char*)workspace, Nk, ldkv, ntiles);
  hipLaunchKernelGGL(sr_attention_split_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)heads, (unsigned)B), dim3(256),
                     2 * IMG, s, q, (const unsigned char*)workspace, out, N, Nk, ldq, ldo, scale, ntiles);
  return (int)hipGetLastError();
}
