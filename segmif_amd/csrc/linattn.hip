// Linear ("efficient") cross attention of the fusion net's CrossPath
// (core/model_fusion.py:263-288 CrossAttention, :303-328 CrossAttention2).
//
// The reference computes, per image and head (d = 8, 8 heads, N = H*W = 307 200 tokens):
//     ctx = softmax_{dim=-2}( (K^T V) * d^-1/2 )      an 8x8 matrix
//     x   = Q @ ctx                                    (N x 8) @ (8 x 8)
// and then concatenates two such results and applies end_proj (128 -> 64).  No N x N matrix exists,
// so the work is a long reduction over N followed by a tiny matrix.  We split it as
//   1. segmif_linattn_partial_f32: row-block partial sums of K^T V for all heads (bandwidth bound:
//      one pass over kv), accumulated in fp32 over 32-row chunks and in fp64 across chunks — the
//      reduction feeds a softmax, so its rounding is the accuracy-critical step;
//   2. segmif_linattn_fold_f32: fp64 sum of the partials, softmax over the k index, and folding of
//      the block-diagonal context into the end_proj weight:
//          Weff[b][n][kofs + h*d + i] = sum_j ctx[b][h][i][j] * Wend[n][wofs + h*d + j]
//      so that Q @ ctx followed by cat + end_proj becomes ONE dense GEMM over [y3 | u_i] with a
//      per-image 64x128 weight (segmif_igemm_f32, two-source mode) — z/v (2 x 157 MB per image)
//      are never written.
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>

#include "segmif_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LA_ROWS = 1024;  // rows per block
constexpr int LA_CHUNK = 32;  // rows per LDS chunk / fp32 accumulation run

// heads*d == 64, d == 8: 512 (h, i, j) entries, two per thread.
__global__ __launch_bounds__(256) void linattn_partial_kernel(const float* __restrict__ kv, double* __restrict__ partial,
                                                              long long N, int ldkv, int nblk) {
  __shared__ __attribute__((aligned(16))) float rows[LA_CHUNK][128 + 4];
  const int tid = threadIdx.x;
  const int blk = blockIdx.x, b = blockIdx.y;
  const int hh = tid >> 5, i = (tid >> 2) & 7, j0 = (tid & 3) * 2;
  const long long r0 = (long long)blk * LA_ROWS;
  const float* base = kv + ((long long)b * N) * ldkv;
  double acc0 = 0.0, acc1 = 0.0;
  const int lr = tid >> 3, lq = tid & 7;  // loader: 32 rows x 8 threads, 4 float4 each
  for (int c = 0; c < LA_ROWS / LA_CHUNK; ++c) {
    const long long row = r0 + c * LA_CHUNK + lr;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N) val = *reinterpret_cast<const float4*>(base + row * ldkv + (lq + 8 * u) * 4);
      *reinterpret_cast<float4*>(&rows[lr][(lq + 8 * u) * 4]) = val;
    }
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < LA_CHUNK; ++r) {
      const float kk = rows[r][hh * 8 + i];
      s0 = fmaf(kk, rows[r][64 + hh * 8 + j0], s0);
      s1 = fmaf(kk, rows[r][64 + hh * 8 + j0 + 1], s1);
    }
    acc0 += (double)s0;
    acc1 += (double)s1;
    __syncthreads();
    if (r0 + (c + 1) * LA_CHUNK >= N) break;
  }
  double* dst = partial + ((long long)b * nblk + blk) * 512 + hh * 64 + i * 8 + j0;
  dst[0] = acc0;
  dst[1] = acc1;
}

__global__ __launch_bounds__(1024) void linattn_fold_kernel(const double* __restrict__ partial,
                                                            const float* __restrict__ wend, float* __restrict__ weff,
                                                            int nblk, int Nout, int ldw, int wofs, int ldweff, int kofs,
                                                            float scale) {
  __shared__ double part[4][512];
  __shared__ double ctx[512];  // [h][i][j]
  const int tid = threadIdx.x, b = blockIdx.x;
  const int e = tid & 255, slice = tid >> 8;
  const double* p = partial + (long long)b * nblk * 512;
  double a0 = 0.0, a1 = 0.0;
  for (int k = slice; k < nblk; k += 4) {  // fixed order per slice: deterministic
    a0 += p[(long long)k * 512 + e];
    a1 += p[(long long)k * 512 + 256 + e];
  }
  part[slice][e] = a0;
  part[slice][256 + e] = a1;
  __syncthreads();
  if (tid < 512) ctx[tid] = (((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]) * (double)scale;
  __syncthreads();
  if (tid < 64) {  // one (h, j) column per thread: softmax over i (dim = -2)
    const int hh = tid >> 3, j = tid & 7;
    double mx = -1e300;
    for (int i = 0; i < 8; ++i) mx = fmax(mx, ctx[hh * 64 + i * 8 + j]);
    double ev[8], sum = 0.0;
    for (int i = 0; i < 8; ++i) {
      ev[i] = exp(ctx[hh * 64 + i * 8 + j] - mx);
      sum += ev[i];
    }
    for (int i = 0; i < 8; ++i) ctx[hh * 64 + i * 8 + j] = ev[i] / sum;
  }
  __syncthreads();
  // Weff[b][n][kofs + c] for c = h*8 + i in [0, 64)
  for (int o = tid; o < Nout * 64; o += 1024) {
    const int n = o >> 6, c = o & 63, hh = c >> 3, i = c & 7;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf((float)ctx[hh * 64 + i * 8 + j], wend[(long long)n * ldw + wofs + hh * 8 + j], acc);
    weff[((long long)b * Nout + n) * ldweff + kofs + c] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused kv projection + K^T V partial reduction: kv = y @ Wkv^T is formed tile by tile on the
// fp32 matrix pipe, parked in LDS and reduced per head without ever being written to HBM
// (2 x 157 MB per image per context otherwise).  One block = 1024 rows = 8 sub-tiles of 128 rows.
// ---------------------------------------------------------------------------------------------
constexpr int KVP = 68;   // pitch of the y / Wkv tiles (64 + 4: conflict-free ds_read_b128)
constexpr int KVT = 132;  // pitch of the kv tile

__global__ __launch_bounds__(256) void linattn_kvpartial_kernel(const float* __restrict__ y,
                                                                const float* __restrict__ wkv,
                                                                double* __restrict__ partial, long long N, int ldy,
                                                                int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;               // [128][KVP]  Wkv rows (k | v outputs) x 64 inputs
  float* As = Ws + 128 * KVP;     // [128][KVP]  y tile
  float* KVs = As + 128 * KVP;    // [128][KVT]  kv tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int blk = blockIdx.x, b = blockIdx.y;
  const long long r0 = (long long)blk * LA_ROWS;
  const float* base = y + (long long)b * N * ldy;
  const int lrow = tid >> 4, lq = (tid & 15) * 4;  // loader: 16 rows x 16 float4 per pass, 8 passes

#pragma unroll
  for (int j = 0; j < 8; ++j)
    *reinterpret_cast<f32x4*>(Ws + (lrow + 16 * j) * KVP + lq) =
        *reinterpret_cast<const f32x4*>(wkv + (lrow + 16 * j) * 64 + lq);

  f32x4 ra[8];
  auto gload = [&](int sub) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long row = r0 + sub * 128 + lrow + 16 * j;
      ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < N) ra[j] = *reinterpret_cast<const f32x4*>(base + row * ldy + lq);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(As + (lrow + 16 * j) * KVP + lq) = ra[j];
  };

  const int hh = tid >> 5, ei = (tid >> 2) & 7, j0 = (tid & 3) * 2;
  double acc0 = 0.0, acc1 = 0.0;
  const int nsub = (int)((((N - r0) < LA_ROWS ? (N - r0) : LA_ROWS) + 127) / 128);
  gload(0);
  sstore();
  __syncthreads();
  for (int sub = 0; sub < nsub; ++sub) {
    if (sub + 1 < nsub) gload(sub + 1);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const float* a_base = As + (wm * 64 + r) * KVP + 4 * h;
    const float* b_base = Ws + (wn * 64 + r) * KVP + 4 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x4 a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * KVP + 8 * t);
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * KVP + 8 * t);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s2], bb[j][s2], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v)
          KVs[(wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * KVT + wn * 64 + j * 32 + r] = acc[i][j][v];
    __syncthreads();  // kv tile complete; every wave is done reading As
    if (sub + 1 < nsub) sstore();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int rr = 0; rr < 128; ++rr) {
      const float kk = KVs[rr * KVT + hh * 8 + ei];
      const float2 vv = *reinterpret_cast<const float2*>(KVs + rr * KVT + 64 + hh * 8 + j0);
      s0 = fmaf(kk, vv.x, s0);
      s1 = fmaf(kk, vv.y, s1);
    }
    acc0 += (double)s0;
    acc1 += (double)s1;
    __syncthreads();  // reduce done before the next tile overwrites KVs; next As visible
  }
  double* dst = partial + ((long long)b * nblk + blk) * 512 + hh * 64 + ei * 8 + j0;
  dst[0] = acc0;
  dst[1] = acc1;
}

// ---------------------------------------------------------------------------------------------
// (r6) Any head geometry with heads * d <= 64 and d <= 8 - the reference's ablation networks build their interaction
// modules at dim 32 (8 heads of 4: model_fusion.py:639-640, :832, :867).  Same arithmetic as the two kernels above (fp32 inside a
// 32-row run, fp64 across runs and blocks, fixed order), entry e = (h d + i) d + j owned by thread e (and e + 256): these
// problems are small (32-channel maps), so one generic kernel instead of one instantiation per geometry.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linattn_partial_generic_kernel(const float* __restrict__ kv, double* __restrict__ partial,
                                                                      long long N, int ldkv, int nblk, int heads, int d) {
  __shared__ __attribute__((aligned(16))) float rows[LA_CHUNK][128 + 4];
  const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
  const int C = heads * d, E = C * d;
  const long long r0 = (long long)blk * LA_ROWS;
  const float* base = kv + ((long long)b * N) * ldkv;
  int ki[2], vj[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u, hi = e / d;       // hi = h d + i
    ki[u] = e < E ? hi : 0;
    vj[u] = e < E ? C + (hi / d) * d + e % d : 0;  // V column h d + j
  }
  double acc[2] = {0.0, 0.0};
  for (int c = 0; c < LA_ROWS / LA_CHUNK; ++c) {
    for (int u = tid; u < LA_CHUNK * 2 * C; u += 256) {
      const int r = u / (2 * C), col = u % (2 * C);
      const long long row = r0 + c * LA_CHUNK + r;
      rows[r][col] = row < N ? base[row * ldkv + col] : 0.f;
    }
    __syncthreads();
    float s[2] = {0.f, 0.f};
#pragma unroll 8
    for (int r = 0; r < LA_CHUNK; ++r) {
      s[0] = fmaf(rows[r][ki[0]], rows[r][vj[0]], s[0]);
      s[1] = fmaf(rows[r][ki[1]], rows[r][vj[1]], s[1]);
    }
    acc[0] += (double)s[0];
    acc[1] += (double)s[1];
    __syncthreads();
    if (r0 + (c + 1) * LA_CHUNK >= N) break;
  }
  double* dst = partial + ((long long)b * nblk + blk) * E;
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + 256 * u < E) dst[tid + 256 * u] = acc[u];
}

__global__ __launch_bounds__(512) void linattn_fold_generic_kernel(const double* __restrict__ partial, const float* __restrict__ wend,
                                                                   float* __restrict__ weff, int nblk, int Nout, int ldw, int wofs,
                                                                   int ldweff, int kofs, float scale, int heads, int d) {
  __shared__ double ctx[512];  // [h][i][j]
  const int tid = threadIdx.x, b = blockIdx.x;
  const int C = heads * d, E = C * d;
  if (tid < E) {
    const double* p = partial + (long long)b * nblk * E + tid;
    double a = 0.0;
    for (int k = 0; k < nblk; ++k) a += p[(long long)k * E];  // fixed order: deterministic
    ctx[tid] = a * (double)scale;
  }
  __syncthreads();
  if (tid < C) {  // one (h, j) column per thread: softmax over i (dim = -2)
    const int hh = tid / d, j = tid % d;
    double mx = -1e300, ev[8], sum = 0.0;
    for (int i = 0; i < d; ++i) mx = fmax(mx, ctx[(hh * d + i) * d + j]);
    for (int i = 0; i < d; ++i) {
      ev[i] = exp(ctx[(hh * d + i) * d + j] - mx);
      sum += ev[i];
    }
    for (int i = 0; i < d; ++i) ctx[(hh * d + i) * d + j] = ev[i] / sum;
  }
  __syncthreads();
  for (int o = tid; o < Nout * C; o += 512) {  // Weff[b][n][kofs + h d + i] = sum_j ctx[h][i][j] Wend[n][wofs + h d + j]
    const int n = o / C, c = o % C, hh = c / d;
    float acc = 0.f;
    for (int j = 0; j < d; ++j) acc = fmaf((float)ctx[c * d + j], wend[(long long)n * ldw + wofs + hh * d + j], acc);
    weff[((long long)b * Nout + n) * ldweff + kofs + c] = acc;
  }
}

// (r6) Backward of segmif_linattn_fold_f32 for the training path (heads = d = 8): given ktv = K^T V per head (fp64, [h][i][j]), the
// end_proj weight and dWeff, one workgroup per image forms
//     ctx = softmax_i(ktv scale);   dctx[h][i][j] = sum_n dWeff[n][kofs + 8h + i] Wend[n][wofs + 8h + j]
//     dktv[h][i][j] = scale ctx_ij (dctx_ij - sum_i' ctx_i'j dctx_i'j)                       (softmax over i = dim -2)
//     dWend_part[b][n][wofs + 8h + j] = sum_i dWeff[n][kofs + 8h + i] ctx[h][i][j]           (summed over images by the caller)
// - the softmax / einsum / cat that CrossPath's training path ran as torch ops on (B, 8, 8, 8) tensors (core/model_fusion.py:281-286,
// :316-326, :357-360 under autograd).  fp64 where the forward is.
__global__ __launch_bounds__(512) void linattn_fold_bwd_kernel(const double* __restrict__ ktv, const float* __restrict__ wend, int ldw,
                                                               int wofs, const float* __restrict__ dweff, int ldweff, int kofs,
                                                               float scale, double* __restrict__ dktv, float* __restrict__ dwend_part,
                                                               int ldp, int Nout) {
  __shared__ double ctx[512], dctx[512];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int hh = tid >> 6, i = (tid >> 3) & 7, j = tid & 7;
  ctx[tid] = ktv[(long long)b * 512 + tid] * (double)scale;
  __syncthreads();
  if (tid < 64) {  // one (h, j) column per thread: softmax over i
    const int h2 = tid >> 3, j2 = tid & 7;
    double mx = -1e300, ev[8], sum = 0.0;
    for (int q = 0; q < 8; ++q) mx = fmax(mx, ctx[h2 * 64 + q * 8 + j2]);
    for (int q = 0; q < 8; ++q) {
      ev[q] = exp(ctx[h2 * 64 + q * 8 + j2] - mx);
      sum += ev[q];
    }
    for (int q = 0; q < 8; ++q) ctx[h2 * 64 + q * 8 + j2] = ev[q] / sum;
  }
  const float* dw = dweff + (long long)b * Nout * ldweff + kofs;
  double acc = 0.0;
  for (int n = 0; n < Nout; ++n) acc += (double)dw[(long long)n * ldweff + hh * 8 + i] * (double)wend[(long long)n * ldw + wofs + hh * 8 + j];
  dctx[tid] = acc;
  __syncthreads();
  double dot = 0.0;
  for (int q = 0; q < 8; ++q) dot += ctx[hh * 64 + q * 8 + j] * dctx[hh * 64 + q * 8 + j];
  dktv[(long long)b * 512 + tid] = (double)scale * ctx[tid] * (dctx[tid] - dot);
  float* dp = dwend_part + (long long)b * Nout * ldp + wofs;
  for (int o = tid; o < Nout * 64; o += 512) {
    const int n = o >> 6, c = o & 63, h2 = c >> 3, j2 = c & 7;
    double a = 0.0;
    for (int q = 0; q < 8; ++q) a += (double)dw[(long long)n * ldweff + h2 * 8 + q] * ctx[h2 * 64 + q * 8 + j2];
    dp[(long long)n * ldp + c] = (float)a;
  }
}

}  // namespace

extern "C" int segmif_linattn_num_blocks(int64_t N) { return (int)((N + LA_ROWS - 1) / LA_ROWS); }

extern "C" int segmif_linattn_partial_f32(const float* kv, double* partial, int B, int64_t N, int heads, int d,
                                          int ldkv, void* stream) {
  if (!kv || !partial || B <= 0 || N <= 0 || heads <= 0 || d <= 0 || d > 8 || heads * d > 64 || ldkv < 2 * heads * d) return SEGMIF_EINVAL;
  const int nblk = segmif_linattn_num_blocks(N);
  if (heads != 8 || d != 8) {  // (r6) the ablation networks' geometries (dim 32: 8 heads of 4)
    if (((uintptr_t)kv & 3) || ((uintptr_t)partial & 7)) return SEGMIF_EINVAL;
    hipLaunchKernelGGL(linattn_partial_generic_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, kv,
                       partial, (long long)N, ldkv, nblk, heads, d);
    return (int)hipGetLastError();
  }
  if (ldkv < 128 || (ldkv & 3) || ((uintptr_t)kv & 15) || ((uintptr_t)partial & 7)) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(linattn_partial_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, kv,
                     partial, (long long)N, ldkv, nblk);
  return (int)hipGetLastError();
}

extern "C" int segmif_linattn_fold_f32(const double* partial, const float* wend, float* weff, int B, int nblk,
                                       int heads, int d, int Nout, int ldw, int wofs, int ldweff, int kofs, float scale,
                                       void* stream) {
  if (!partial || !wend || !weff || B <= 0 || nblk <= 0 || heads <= 0 || d <= 0 || d > 8 || heads * d > 64 || Nout <= 0) return SEGMIF_EINVAL;
  if (heads != 8 || d != 8) {
    hipLaunchKernelGGL(linattn_fold_generic_kernel, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, partial, wend, weff, nblk,
                       Nout, ldw, wofs, ldweff, kofs, scale, heads, d);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(linattn_fold_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream,
                     partial, wend, weff, nblk, Nout, ldw, wofs, ldweff, kofs, scale);
  return (int)hipGetLastError();
}

extern "C" int segmif_linattn_kvpartial_f32(const float* y, const float* wkv, double* partial, int B, int64_t N,
                                            int heads, int d, int ldy, void* stream) {
  if (!y || !wkv || !partial || B <= 0 || N <= 0 || heads != 8 || d != 8 || ldy < 64 || (ldy & 3)) return SEGMIF_EINVAL;
  if ((((uintptr_t)y | (uintptr_t)wkv) & 15) || ((uintptr_t)partial & 7)) return SEGMIF_EINVAL;
  const int nblk = segmif_linattn_num_blocks(N);
  constexpr size_t smem = (size_t)(2 * 128 * KVP + 128 * KVT) * sizeof(float);
  static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)linattn_kvpartial_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipLaunchKernelGGL(linattn_kvpartial_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), smem, (hipStream_t)stream,
                     y, wkv, partial, (long long)N, ldy, nblk);
  return (int)hipGetLastError();
}

extern "C" int segmif_linattn_fold_bwd_f32(const double* ktv, const float* wend, int ldw, int wofs, const float* dweff, int ldweff,
                                           int kofs, float scale, double* dktv, float* dwend_part, int ldp, int B, int Nout,
                                           void* stream) {
  if (!ktv || !wend || !dweff || !dktv || !dwend_part || B <= 0 || Nout <= 0 || wofs < 0 || kofs < 0 || ldw < wofs + 64 ||
      ldweff < kofs + 64 || ldp < wofs + 64)
    return SEGMIF_EINVAL;
  if (((uintptr_t)ktv | (uintptr_t)dktv) & 7) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(linattn_fold_bwd_kernel, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, ktv, wend, ldw, wofs, dweff, ldweff,
                     kofs, scale, dktv, dwend_part, ldp, Nout);
  return (int)hipGetLastError();
}
