// Backward of the spatial-reduction attention (core/mix_transformer.py:107-111: attn = softmax(q k^T scale); x = attn v) without
// the score matrix (round 5).  Round 4's backward wrote P and dP as (B, heads, N, Nk) fp32 tensors - 4 GB per
// segmentation step at 8 images -, ran two row-softmax passes over them and five batched GEMM / weight-gradient launches per
// attention call: 8.8 ms of a 50 ms step (profiles/r04_segtrain_kernel_stats.txt).  Here the scores are recomputed tile by
// tile in registers, flash-attention style, on the EXACT-fp32 matrix pipe (v_mfma_f32_32x32x2_f32: the training path's
// arithmetic), head_dim 64:
//
//   attn_bwd_dq_kernel   lane = query (the products run transposed, like the forward kernel): per 32-key tile
//                        S^T = K Q^T, dP^T = V dO^T, W = e^(S - m) (dP - D) with the running maximum m of an online softmax and
//                        D_i = dO_i . O_i, dQ^T += K^T W (rescaled when m moves); at the end dq = scale / l * dQ, and the row
//                        statistics LSE_i = m + log l (base 2) and D_i are written for the second kernel.
//   attn_bwd_dkv_kernel  lane = key: a wave owns 32 keys (K and V rows in registers) and walks a CHUNK of queries:
//                        S = Q K^T, P = 2^(S - LSE), dP = dO V^T, dS = scale P (dP - D), dV^T += dO^T P, dK^T += Q^T dS - the
//                        accumulators are the next product's B operand register for register (k-slot = accumulator register).
//                        One partial (dk | dv) per chunk; attn_bwd_reduce_kernel sums the chunks in a fixed order
//                        (deterministic: no floating-point atomics anywhere).
// K/V (first kernel) and Q/dO (second) tiles are staged in LDS with 68-float rows (conflict-free 16-byte reads of 32
// consecutive rows), double buffered, loaded by the whole workgroup one tile ahead.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "igemm_common.h"
#include "segmif_hip.h"

namespace segmif {
namespace {

constexpr int ABT = 32;        // keys / queries per tile
constexpr int ABP = 68;        // LDS row pitch in floats (64 + 4)
constexpr int AB_TILE = ABT * ABP;  // floats per staged 32 x 64 tile

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int v = 0; v < 16; ++v) z[v] = 0.f;
  return z;
}

// accumulator register v of lane (r, hh) holds row (v & 3) + 8 (v >> 2) + 4 hh of the 32 x 32 tile, column r
__device__ __forceinline__ int acc_row(int v, int hh) { return (v & 3) + 8 * (v >> 2) + 4 * hh; }

struct AttnBwdK {
  const float* q; const float* k; const float* v; const float* o; const float* dout;
  float* dq; float* dkv;       // dkv: [nchunk][B][Nk][2C] partials (nchunk == 1: the result itself)
  float* lse; float* dsum;     // [B][heads][N]: base-2 log-sum-exp of the scaled scores, D_i = dO_i . O_i
  int N, Nk, C, heads, ldkv, chunk, nchunk, B;
  float scale;
};

// 256 threads stage one 32 x 64 tile (rows row0 .. row0 + 31 of a (rows, pitch ld) matrix, 64 floats from column col0) into LDS
__device__ __forceinline__ void stage_tile(const float* __restrict__ base, long long ld, int row0, int nrows, int tid, f32x4* regs) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = tid + 256 * j, row = u >> 4, c4 = (u & 15) * 4;
    regs[j] = row0 + row < nrows ? *reinterpret_cast<const f32x4*>(base + (long long)(row0 + row) * ld + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void store_tile(float* dst, int tid, const f32x4* regs) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = tid + 256 * j, row = u >> 4, c4 = (u & 15) * 4;
    *reinterpret_cast<f32x4*>(dst + row * ABP + c4) = regs[j];
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdK p) {
  __shared__ __attribute__((aligned(16))) float sm[2][2][AB_TILE];  // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool q_ok = qi < p.N;
  const long long qrow = ((long long)b * p.N + (q_ok ? qi : 0)) * p.C + head * 64 + 32 * hh;
  const float* kb = p.k + (long long)b * p.Nk * p.ldkv + head * 64;
  const float* vb = p.v + (long long)b * p.Nk * p.ldkv + head * 64;
  // this lane's half of its query's rows: dims 32 hh .. 32 hh + 31 (k-slot (s, hh) <-> dim 32 hh + s)
  float qreg[32], doreg[32];
  float dsum = 0.f;
  const float sl2 = p.scale * 1.44269504088896340736f;  // scores in the base-2 exponent domain
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = a, o = a;
    if (q_ok) {
      a = *reinterpret_cast<const f32x4*>(p.q + qrow + 4 * g);
      d = *reinterpret_cast<const f32x4*>(p.dout + qrow + 4 * g);
      o = *reinterpret_cast<const f32x4*>(p.o + qrow + 4 * g);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qreg[4 * g + e] = a[e] * sl2;
      doreg[4 * g + e] = d[e];
      dsum = fmaf(d[e], o[e], dsum);
    }
  }
  dsum += __shfl_xor(dsum, 32);

  const int ntiles = (p.Nk + ABT - 1) / ABT;
  f32x4 kr[2], vr[2];
  stage_tile(kb, p.ldkv, 0, p.Nk, tid, kr);
  stage_tile(vb, p.ldkv, 0, p.Nk, tid, vr);
  store_tile(sm[0][0], tid, kr);
  store_tile(sm[0][1], tid, vr);
  __syncthreads();

  f32x16 dqa[2] = {zero16(), zero16()};
  float m_run = -INFINITY, l_run = 0.f;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) {  // next tile's rows: requested now, stored after this tile's products
      stage_tile(kb, p.ldkv, (kt + 1) * ABT, p.Nk, tid, kr);
      stage_tile(vb, p.ldkv, (kt + 1) * ABT, p.Nk, tid, vr);
    }
    const float* Kt = sm[cur][0];
    const float* Vt = sm[cur][1];
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const f32x4 ka = *reinterpret_cast<const f32x4*>(Kt + r * ABP + 32 * hh + 4 * g);
      const f32x4 va = *reinterpret_cast<const f32x4*>(Vt + r * ABP + 32 * hh + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = mfma_f32(ka[e], qreg[4 * g + e], s);     // S^T[key][query] (x scale log2 e)
        dp = mfma_f32(va[e], doreg[4 * g + e], dp);  // dP^T[key][query]
      }
    }
    if ((kt + 1) * ABT > p.Nk) {  // last, partial tile: keys past the end get no weight
#pragma unroll
      for (int v = 0; v < 16; ++v) s[v] = kt * ABT + acc_row(v, hh) < p.Nk ? s[v] : -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int v = 1; v < 16; ++v) mx = fmaxf(mx, s[v]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
    float psum = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float pv = __builtin_amdgcn_exp2f(s[v] - m_new);
      psum += pv;
      s[v] = pv * (dp[v] - dsum);  // W[key][query]
    }
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {  // some lane's maximum moved: rescale the running sums
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) dqa[dt][v] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
    // dQ^T[d][query] += sum_key K[key][d] W[key][query]: k-slot (step v, hh) <-> key acc_row(v, hh) = this lane's register v
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float* krow = Kt + acc_row(v, hh) * ABP + r;
      dqa[0] = mfma_f32(krow[0], s[v], dqa[0]);
      dqa[1] = mfma_f32(krow[32], s[v], dqa[1]);
    }
    __syncthreads();  // every wave is done with buffer cur ^ 1's previous contents ... (they were read in iteration kt - 1)
    if (kt + 1 < ntiles) {
      store_tile(sm[cur ^ 1][0], tid, kr);
      store_tile(sm[cur ^ 1][1], tid, vr);
    }
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (q_ok) {
    const float inv = p.scale / l_tot;
    float* drow = p.dq + ((long long)b * p.N + qi) * p.C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(drow + 32 * dt + 8 * g + 4 * hh) =
            f32x4{dqa[dt][4 * g] * inv, dqa[dt][4 * g + 1] * inv, dqa[dt][4 * g + 2] * inv, dqa[dt][4 * g + 3] * inv};
    if (hh == 0) {
      const long long si = ((long long)b * p.heads + head) * p.N + qi;
      p.lse[si] = m_run + log2f(l_tot);
      p.dsum[si] = dsum;
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdK p) {
  __shared__ __attribute__((aligned(16))) float sm[2][2][AB_TILE];  // [buffer][Q | dO]
  __shared__ __attribute__((aligned(16))) float st[2][2][ABT];      // [buffer][LSE | D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z / p.heads, head = blockIdx.z - b * p.heads;
  const int chunk = blockIdx.y;
  const int kt = blockIdx.x * 4 + wave;         // this wave's key tile
  const int key = kt * ABT + r;
  const bool k_ok = key < p.Nk;
  const int q0 = chunk * p.chunk;
  const int q1 = q0 + p.chunk < p.N ? q0 + p.chunk : p.N;
  const float* qb = p.q + (long long)b * p.N * p.C + head * 64;
  const float* dob = p.dout + (long long)b * p.N * p.C + head * 64;
  const float* lseb = p.lse + ((long long)b * p.heads + head) * p.N;
  const float* dsb = p.dsum + ((long long)b * p.heads + head) * p.N;
  const float sl2 = p.scale * 1.44269504088896340736f;
  float kreg[32], vreg[32];
  {
    const long long krow = ((long long)b * p.Nk + (k_ok ? key : 0)) * p.ldkv + head * 64 + 32 * hh;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
      if (k_ok) {
        a = *reinterpret_cast<const f32x4*>(p.k + krow + 4 * g);
        c = *reinterpret_cast<const f32x4*>(p.v + krow + 4 * g);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kreg[4 * g + e] = a[e];
        vreg[4 * g + e] = c[e];
      }
    }
  }
  const int ntiles = (q1 - q0 + ABT - 1) / ABT;
  f32x4 qr[2], dr[2];
  float lr = 0.f;
  auto stage = [&](int t) {
    const int row0 = q0 + t * ABT;
    stage_tile(qb, p.C, row0, q1, tid, qr);
    stage_tile(dob, p.C, row0, q1, tid, dr);
    if (tid < 64) {  // LSE (lanes 0..31) and D (32..63) of the tile's queries; past the end: +inf / 0 -> P = 0
      const int qi = row0 + (tid & 31);
      lr = qi < q1 ? (tid < 32 ? lseb[qi] : dsb[qi]) : (tid < 32 ? INFINITY : 0.f);
    }
  };
  auto store = [&](int buf) {
    store_tile(sm[buf][0], tid, qr);
    store_tile(sm[buf][1], tid, dr);
    if (tid < 64) st[buf][tid >> 5][tid & 31] = lr;
  };
  stage(0);
  store(0);
  __syncthreads();
  f32x16 dka[2] = {zero16(), zero16()}, dva[2] = {zero16(), zero16()};
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) stage(t + 1);
    const float* Qt = sm[cur][0];
    const float* Dt = sm[cur][1];
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const f32x4 qa = *reinterpret_cast<const f32x4*>(Qt + r * ABP + 32 * hh + 4 * g);
      const f32x4 da = *reinterpret_cast<const f32x4*>(Dt + r * ABP + 32 * hh + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = mfma_f32(qa[e], kreg[4 * g + e], s);    // S[query][key]
        dp = mfma_f32(da[e], vreg[4 * g + e], dp);  // dP[query][key]
      }
    }
    // register v <-> query acc_row(v, hh): its LSE and D sit at st[..][4 hh + 8 (v >> 2) + (v & 3)]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(&st[cur][0][8 * g + 4 * hh]);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(&st[cur][1][8 * g + 4 * hh]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v = 4 * g + e;
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[v], sl2, -l4[e]));  // softmax weight of (query, key); 0 past the end
        s[v] = k_ok ? pv : 0.f;                                           // P
        dp[v] = s[v] * (dp[v] - d4[e]) * p.scale;                         // dS
      }
    }
    // dV^T[d][key] += dO[query][d] P[query][key];  dK^T[d][key] += Q[query][d] dS[query][key]   (k-slot = register v)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float* drow = Dt + acc_row(v, hh) * ABP + r;
      const float* qrow = Qt + acc_row(v, hh) * ABP + r;
      dva[0] = mfma_f32(drow[0], s[v], dva[0]);
      dva[1] = mfma_f32(drow[32], s[v], dva[1]);
      dka[0] = mfma_f32(qrow[0], dp[v], dka[0]);
      dka[1] = mfma_f32(qrow[32], dp[v], dka[1]);
    }
    __syncthreads();
    if (t + 1 < ntiles) store(cur ^ 1);
    __syncthreads();
  }
  if (k_ok) {
    float* out = p.dkv + (((long long)chunk * p.B + b) * p.Nk + key) * (2LL * p.C) + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(out + 32 * dt + 8 * g + 4 * hh) =
            f32x4{dka[dt][4 * g], dka[dt][4 * g + 1], dka[dt][4 * g + 2], dka[dt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(out + p.C + 32 * dt + 8 * g + 4 * hh) =
            f32x4{dva[dt][4 * g], dva[dt][4 * g + 1], dva[dt][4 * g + 2], dva[dt][4 * g + 3]};
      }
  }
}

// dkv[i] = sum over chunks of part[c][i], in chunk order (fp32, deterministic)
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long n4,
                                                              int nchunk) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
  for (int c = 1; c < nchunk; ++c) a += reinterpret_cast<const f32x4*>(part)[(long long)c * n4 + i];
  reinterpret_cast<f32x4*>(out)[i] = a;
}

}  // namespace
}  // namespace segmif

using namespace segmif;

// Queries per chunk of the dk / dv kernel.  Its workgroups (4 waves = 4 key tiles; two fit a CU) should number about one
// round of the chip: with 256-query chunks a stage-3 call of the segmentation step launched 600 of them - a full round and a
// 17 % one - and took two rounds' time (tools/attn_bwd_bench.py).  -> chunk length, a multiple of the 32-query tile.
static int attn_bwd_chunk_len(int B, int heads, int N, int Nk) {
  const long long per_chunk = (long long)((Nk + 127) / 128) * B * heads;
  long long nchunk = (512 + per_chunk / 2) / per_chunk;
  const long long max_chunks = (N + 31) / 32;
  nchunk = nchunk < 1 ? 1 : (nchunk > max_chunks ? max_chunks : nchunk);
  return (int)(((N + nchunk - 1) / nchunk + 31) / 32 * 32);
}

extern "C" int segmif_sr_attention_bwd_chunks(int B, int heads, int N, int Nk) {
  if (B <= 0 || heads <= 0 || N <= 0 || Nk <= 0) return 0;
  const int len = attn_bwd_chunk_len(B, heads, N, Nk);
  return (N + len - 1) / len;
}

extern "C" int64_t segmif_sr_attention_bwd_workspace_floats(int B, int heads, int N, int Nk, int C) {
  if (B <= 0 || heads <= 0 || N <= 0 || Nk <= 0 || C <= 0) return 0;
  const int nchunk = segmif_sr_attention_bwd_chunks(B, heads, N, Nk);
  return 2LL * B * heads * N + 4 + (nchunk > 1 ? (int64_t)nchunk * B * Nk * 2 * C : 0);  // (+ 4: the partials start 16-byte aligned)
}

extern "C" int segmif_sr_attention_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* dout, float* dq,
                                           float* dkv, float* workspace, int B, int heads, int N, int Nk, int hd, int ldkv, float scale,
                                           void* stream) {
  if (!q || !k || !v || !out || !dout || !dq || !dkv || !workspace || B <= 0 || heads <= 0 || N <= 0 || Nk <= 0 || hd != 64)
    return SEGMIF_EINVAL;
  const int C = heads * 64;
  if (ldkv < 2 * C || (ldkv & 3)) return SEGMIF_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dkv | (uintptr_t)workspace) & 15)
    return SEGMIF_EINVAL;
  AttnBwdK p;
  p.q = q; p.k = k; p.v = v; p.o = out; p.dout = dout; p.dq = dq;
  p.N = N; p.Nk = Nk; p.C = C; p.heads = heads; p.ldkv = ldkv; p.B = B; p.scale = scale;
  p.chunk = attn_bwd_chunk_len(B, heads, N, Nk);
  p.nchunk = (N + p.chunk - 1) / p.chunk;
  p.lse = workspace;
  p.dsum = workspace + (long long)B * heads * N;
  float* part = workspace + 2LL * B * heads * N;
  part = reinterpret_cast<float*>(((uintptr_t)part + 15) & ~(uintptr_t)15);
  p.dkv = p.nchunk > 1 ? part : dkv;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)heads, (unsigned)B), dim3(256), 0, s, p);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)((Nk + 127) / 128), (unsigned)p.nchunk, (unsigned)(B * heads)), dim3(256), 0, s, p);
  if (p.nchunk > 1) {
    const long long n4 = (long long)B * Nk * 2 * C / 4;
    hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, dkv, n4, p.nchunk);
  }
  return (int)hipGetLastError();
}
