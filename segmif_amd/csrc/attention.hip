// Fused spatial-reduction attention for the MiT encoder (core/mix_transformer.py:107-111):
//   O = softmax(Q K^T * scale) V   per (batch, head), Nk = a few hundred reduced keys.
//
// gfx950 design (fp32 MFMA 32x32x2, wave64): everything is computed TRANSPOSED so that a query is
// a lane, not a register row:
//   S^T[key][q] = K[key][:] . Q[q][:]      A = K tile from LDS, B = Q held in registers
//   O^T[d][q]  += V^T[d][key] * P^T[key][q] A = V tile from LDS, B = the lane's own P registers
// With the 32x32 C/D map (col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)) lane (q, h) owns
// 16 of the 32 keys of a tile for query q; its partner lane^32 owns the other 16.  Softmax is
// therefore per lane (15 in-register max/adds + one cross-half shuffle), the running max / sum
// and the O rescale factor are lane scalars, and P feeds the PV MFMA as its B operand with no
// cross-lane movement at all (MFMA step v consumes key pair {key(v,0), key(v,1)}).
// The N x Nk score matrix (23 MB per block per image at stage 1, 268 MB at 1024^2) never exists.
//
// Block = 4 waves = 128 queries of one (batch, head); K/V tiles of 32 keys are staged through a
// double-buffered LDS ring with register prefetch (one barrier per tile).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "segmif_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int HD>
__global__ __launch_bounds__(256) void sr_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ out,
                                                           int N, int Nk, int ldq, int ldkv, int ldo, float scale) {
  constexpr int KT = 32;  // keys per tile
  constexpr int KP = HD + 4;  // K row pitch: conflict-free ds_read_b128
  constexpr int VP = HD;  // V row pitch: lanes read consecutive d
  constexpr int QT = HD / 8;  // float4 Q fragments per lane
  constexpr int DT = HD / 32;  // 32-wide d sub-tiles of O^T
  constexpr int UPR = HD / 4;  // float4 units per K/V row
  constexpr int LU = KT * UPR / 256;  // units per thread per matrix (HD=64: 2, HD=32: 1)
  __shared__ __attribute__((aligned(16))) float Ks[2][KT * KP];
  __shared__ __attribute__((aligned(16))) float Vs[2][KT * VP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ql = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * 128 + wave * 32 + ql;
  const bool q_ok = qi < N;

  const float* qrow = q + ((long long)b * N + (q_ok ? qi : 0)) * ldq + head * HD + 4 * h;
  f32x4 qf[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q_ok) qf[t] = *reinterpret_cast<const f32x4*>(qrow + 8 * t);
  }

  const float* kbase = k + (long long)b * Nk * ldkv + head * HD;
  const float* vbase = v + (long long)b * Nk * ldkv + head * HD;
  const int lrow = tid / UPR, lkq = tid % UPR;
  constexpr int RPP = 256 / UPR;
  f32x4 rk[LU], rv[LU];
  auto gload = [&](int kt) {
#pragma unroll
    for (int j = 0; j < LU; ++j) {
      const int key = kt * KT + lrow + j * RPP;
      rk[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      rv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (key < Nk) {
        rk[j] = *reinterpret_cast<const f32x4*>(kbase + (long long)key * ldkv + lkq * 4);
        rv[j] = *reinterpret_cast<const f32x4*>(vbase + (long long)key * ldkv + lkq * 4);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < LU; ++j) {
      const int r = lrow + j * RPP;
      *reinterpret_cast<f32x4*>(&Ks[buf][r * KP + lkq * 4]) = rk[j];
      *reinterpret_cast<f32x4*>(&Vs[buf][r * VP + lkq * 4]) = rv[j];
    }
  };

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (Nk + KT - 1) / KT;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) gload(kt + 1);

    // ---- S^T = K Q^T ---------------------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    const float* ka = &Ks[cur][ql * KP + 4 * h];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ka + 8 * t);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[t][e], s, 0, 0, 0);
    }
    // ---- per-lane online softmax over this lane's 16 keys (+ partner half) -----------------
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * KT + (e & 3) + 8 * (e >> 2) + 4 * h;
      s[e] = key < Nk ? s[e] * scale : -INFINITY;
      mx = fmaxf(mx, s[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = expf(s[e] - m_new);
      psum += s[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    // ---- O^T += V^T P^T ------------------------------------------------------------------
    const float* va = &Vs[cur][ql];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int krow = (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[krow * VP + 32 * dt], s[e], o[dt], 0, 0, 0);
    }
    if (kt + 1 < ntiles) sstore(cur ^ 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    float* orow = out + ((long long)b * N + qi) * ldo + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g + 4 * h) = w;
      }
  }
}

}  // namespace

extern "C" int segmif_sr_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int heads,
                                       int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || heads <= 0 || N <= 0 || Nk <= 0) return SEGMIF_EINVAL;
  if ((ldq | ldkv | ldo) & 3) return SEGMIF_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return SEGMIF_EINVAL;
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)heads, (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  if (hd == 64)
    hipLaunchKernelGGL(sr_attention_kernel<64>, grid, dim3(256), 0, s, q, k, v, out, N, Nk, ldq, ldkv, ldo, scale);
  else if (hd == 32)
    hipLaunchKernelGGL(sr_attention_kernel<32>, grid, dim3(256), 0, s, q, k, v, out, N, Nk, ldq, ldkv, ldo, scale);
  else
    return SEGMIF_EINVAL;
  return (int)hipGetLastError();
}
