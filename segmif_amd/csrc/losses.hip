// Fusion losses of train_fusion (train.py:363-383) as fused HIP kernels, forward and backward:
//   Fusionloss_grad3 (core/loss.py:506-517):  MSE(mask_0, fused) + 1.1 (1 - SSIM(fused, mask_0)), SSIM = pytorch_ssim
//                                             (pytorch_ssim/__init__.py:19-43: 11x11 Gaussian window, sigma 1.5, zero padding)
//   Fusionloss3      (core/loss.py:459-476):  L1(mask_0, fused) + L1(Sobelxy(mask_0), Sobelxy(fused)),  Sobelxy :634-650
// Single-channel (B, 1, H, W) images.  The five window convolutions of SSIM (and the three of its backward) run in the
// separable blur kernel of rowops.hip; everything around them — the product planes, the SSIM map with its reduction, the
// derivative planes, the gradient assembly, the Sobel stencils and their adjoint — is here.  Reductions are two-pass and
// deterministic: per-block partial sums in double, then one fixed-order sum.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "segmif_hip.h"

namespace {

constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

__device__ __forceinline__ void block_sum2(double a, double b, double* out /* [2] per block */) {
  __shared__ double sa[256], sb[256];
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sa[threadIdx.x] += sa[threadIdx.x + s];
      sb[threadIdx.x] += sb[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sa[0];
    out[1] = sb[0];
  }
}

// planes [5][n]: g, m, g^2, m^2, g m
__global__ __launch_bounds__(256) void ssim_prep_kernel(const float* __restrict__ g, const float* __restrict__ m, float* __restrict__ st,
                                                        long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = g[i], b = m[i];
  st[i] = a;
  st[n + i] = b;
  st[2 * n + i] = a * a;
  st[3 * n + i] = b * b;
  st[4 * n + i] = a * b;
}

// bl: the five blurred planes.  partial[blk] = {sum ssim_map, sum (m - g)^2}; der [3][n]: dS/dmu1, dS/de11, dS/de12
__global__ __launch_bounds__(256) void ssim_map_kernel(const float* __restrict__ bl, const float* __restrict__ g,
                                                       const float* __restrict__ m, float* __restrict__ der,
                                                       double* __restrict__ partial, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double s = 0.0, q = 0.0;
  if (i < n) {
    const float mu1 = bl[i], mu2 = bl[n + i];
    const float s11 = bl[2 * n + i] - mu1 * mu1, s22 = bl[3 * n + i] - mu2 * mu2, s12 = bl[4 * n + i] - mu1 * mu2;
    const float A = 2.f * mu1 * mu2 + C1, Bq = 2.f * s12 + C2, Cq = mu1 * mu1 + mu2 * mu2 + C1, D = s11 + s22 + C2;
    const float inv = 1.f / (Cq * D);
    s = (double)(A * Bq * inv);
    const float d = m[i] - g[i];
    q = (double)(d * d);
    if (der) {
      const float S = A * Bq * inv;
      // dS/dmu1 at fixed window moments: through A, B (s12 = e12 - mu1 mu2), C and D (s11 = e11 - mu1^2)
      der[i] = 2.f * mu2 * (Bq - A) * inv - 2.f * mu1 * S / Cq + 2.f * mu1 * S / D;
      der[n + i] = -S / D;
      der[2 * n + i] = 2.f * A * inv;
    }
  }
  block_sum2(s, q, partial + 2 * (long long)blockIdx.x);
}

// grad = cs (bd[0] + 2 g bd[1] + m bd[2]) + cm (g - m)     bd: the three blurred derivative planes
__global__ __launch_bounds__(256) void ssim_grad_kernel(const float* __restrict__ bd, const float* __restrict__ g,
                                                        const float* __restrict__ m, float* __restrict__ grad, long long n,
                                                        const float* __restrict__ upstream, float cs, float cm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float u = upstream[0];
  const float a = g[i], b = m[i];
  grad[i] = u * (cs * (bd[i] + 2.f * a * bd[n + i] + b * bd[2 * n + i]) + cm * (a - b));
}

__device__ __forceinline__ float at(const float* p, int H, int W, int y, int x) {
  return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? p[(long long)y * W + x] : 0.f;
}

__device__ __forceinline__ void sobel(const float* p, int H, int W, int y, int x, float& gx, float& gy) {
  const float a = at(p, H, W, y - 1, x - 1), b = at(p, H, W, y - 1, x), c = at(p, H, W, y - 1, x + 1);
  const float d = at(p, H, W, y, x - 1), f = at(p, H, W, y, x + 1);
  const float g = at(p, H, W, y + 1, x - 1), hh = at(p, H, W, y + 1, x), k = at(p, H, W, y + 1, x + 1);
  gx = (c + 2.f * f + k) - (a + 2.f * d + g);   // kernelx = [[-1,0,1],[-2,0,2],[-1,0,1]] (cross-correlation)
  gy = (a + 2.f * b + c) - (g + 2.f * hh + k);  // kernely = [[1,2,1],[0,0,0],[-1,-2,-1]]
}

__device__ __forceinline__ float sgn(float v) { return (float)(v > 0.f) - (float)(v < 0.f); }

// partial[blk] = {sum |m - g|, sum |S(m) - S(g)|}; pxy [2][n] (optional): t sign(gx(g)), t sign(gy(g)), t = dsum|S(m)-S(g)|/dS(g)
__global__ __launch_bounds__(256) void sobel_l1_kernel(const float* __restrict__ g, const float* __restrict__ m,
                                                       float* __restrict__ pxy, double* __restrict__ partial, int H, int W,
                                                       long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (i < n) {
    const long long img = i / ((long long)H * W);
    const int rem = (int)(i - img * H * W), y = rem / W, x = rem - y * W;
    const float* gp = g + img * H * W;
    const float* mp = m + img * H * W;
    float gx, gy, mx, my;
    sobel(gp, H, W, y, x, gx, gy);
    sobel(mp, H, W, y, x, mx, my);
    const float sg = fabsf(gx) + fabsf(gy), sm = fabsf(mx) + fabsf(my);
    s1 = (double)fabsf(mp[rem] - gp[rem]);
    s2 = (double)fabsf(sm - sg);
    if (pxy) {
      const float t = -sgn(sm - sg);
      pxy[i] = t * sgn(gx);
      pxy[n + i] = t * sgn(gy);
    }
  }
  block_sum2(s1, s2, partial + 2 * (long long)blockIdx.x);
}

// grad = u ( c sign(g - m) + c * [adjoint of the two Sobel correlations applied to pxy] ),  c = 1 / n
__global__ __launch_bounds__(256) void sobel_l1_bwd_kernel(const float* __restrict__ pxy, const float* __restrict__ g,
                                                           const float* __restrict__ m, float* __restrict__ grad, int H, int W,
                                                           long long n, const float* __restrict__ upstream, float c) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long img = i / ((long long)H * W);
  const int rem = (int)(i - img * H * W), y = rem / W, x = rem - y * W;
  const float* px = pxy + img * H * W;
  const float* py = pxy + n + img * H * W;
  // out(y', x') used in(y' + dy, x' + dx) with weight k[dy][dx]; so in(y, x) receives k[dy][dx] p(y - dy, x - dx)
  float acc = 0.f;
  acc += -1.f * at(px, H, W, y + 1, x + 1) + 1.f * at(px, H, W, y + 1, x - 1);   // kernelx row dy = -1: [-1, 0, 1]
  acc += -2.f * at(px, H, W, y, x + 1) + 2.f * at(px, H, W, y, x - 1);           //          row dy =  0: [-2, 0, 2]
  acc += -1.f * at(px, H, W, y - 1, x + 1) + 1.f * at(px, H, W, y - 1, x - 1);   //          row dy = +1: [-1, 0, 1]
  acc += 1.f * at(py, H, W, y + 1, x + 1) + 2.f * at(py, H, W, y + 1, x) + 1.f * at(py, H, W, y + 1, x - 1);    // kernely row -1
  acc += -1.f * at(py, H, W, y - 1, x + 1) - 2.f * at(py, H, W, y - 1, x) - 1.f * at(py, H, W, y - 1, x - 1);  //         row +1
  grad[i] = upstream[0] * c * (sgn(g[i] - m[i]) + acc);
}

// ---- LapLoss2 (lap_loss.py:100-118; constructed at core/loss.py:509) ---------------------------------------------------
// Three "Laplacian" levels d_k(img) = img - G_k * img with G_k the k x k (k = 3, 5, 7), sigma = 2 Gaussian of
// lap_loss.py:39-80 (zero padding k // 2, normalised to sum 1; exp(-(x^2 + y^2) / 2 sigma^2) factorises, so the 2-D window
// is the outer product of the normalised 1-D windows); loss = 10 (L1_3 + L1_5) + L1_7 with
// L1_k = mean | d_k(gen) - max(d_k(ir), d_k(vis)) |.  One pass over the 7 x 7 neighbourhood serves all three windows.
struct LapTaps {
  float g3[3], g5[5], g7[7];
};

__device__ __forceinline__ void lap_levels(const float* p, int H, int W, int y, int x, const LapTaps& t, float& d3, float& d5, float& d7) {
  float b3 = 0.f, b5 = 0.f, b7 = 0.f;
#pragma unroll
  for (int dy = -3; dy <= 3; ++dy) {
    float r3 = 0.f, r5 = 0.f, r7 = 0.f;
#pragma unroll
    for (int dx = -3; dx <= 3; ++dx) {
      const float v = at(p, H, W, y + dy, x + dx);
      r7 += t.g7[dx + 3] * v;
      if (dx >= -2 && dx <= 2) r5 += t.g5[dx + 2] * v;
      if (dx >= -1 && dx <= 1) r3 += t.g3[dx + 1] * v;
    }
    b7 += t.g7[dy + 3] * r7;
    if (dy >= -2 && dy <= 2) b5 += t.g5[dy + 2] * r5;
    if (dy >= -1 && dy <= 1) b3 += t.g3[dy + 1] * r3;
  }
  const float c = p[(long long)y * W + x];
  d3 = c - b3;
  d5 = c - b5;
  d7 = c - b7;
}

// partial[blk] = {sum 10 |a3| + 10 |a5| + |a7|, 0}; sign3 [3][n] (optional): sign(a_k), a_k = d_k(gen) - max(d_k(ir), d_k(vis))
__global__ __launch_bounds__(256) void laploss2_kernel(const float* __restrict__ g, const float* __restrict__ ir,
                                                       const float* __restrict__ vis, float* __restrict__ sign3,
                                                       double* __restrict__ partial, int H, int W, long long n, LapTaps t) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double s = 0.0;
  if (i < n) {
    const long long img = i / ((long long)H * W);
    const int rem = (int)(i - img * H * W), y = rem / W, x = rem - y * W;
    float a[3], b[3], c[3];
    lap_levels(g + img * H * W, H, W, y, x, t, a[0], a[1], a[2]);
    lap_levels(ir + img * H * W, H, W, y, x, t, b[0], b[1], b[2]);
    lap_levels(vis + img * H * W, H, W, y, x, t, c[0], c[1], c[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float e = a[k] - fmaxf(b[k], c[k]);
      s += (double)((k < 2 ? 10.f : 1.f) * fabsf(e));
      if (sign3) sign3[k * n + i] = sgn(e);
    }
  }
  block_sum2(s, 0.0, partial + 2 * (long long)blockIdx.x);
}

// grad = u / n * sum_k c_k (s_k - G_k * s_k)   (the windows are symmetric: the adjoint of a zero-padded blur is that blur)
__global__ __launch_bounds__(256) void laploss2_bwd_kernel(const float* __restrict__ sign3, float* __restrict__ grad, int H, int W,
                                                           long long n, const float* __restrict__ upstream, float inv_n, LapTaps t) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long img = i / ((long long)H * W);
  const int rem = (int)(i - img * H * W), y = rem / W, x = rem - y * W;
  float d3, d5, d7, q;
  lap_levels(sign3 + img * H * W, H, W, y, x, t, d3, q, q);
  lap_levels(sign3 + n + img * H * W, H, W, y, x, t, q, d5, q);
  lap_levels(sign3 + 2 * n + img * H * W, H, W, y, x, t, q, q, d7);
  grad[i] = upstream[0] * inv_n * (10.f * d3 + 10.f * d5 + d7);
}

LapTaps lap_taps() {
  LapTaps t;
  float* dst[3] = {t.g3, t.g5, t.g7};
  for (int k = 0; k < 3; ++k) {
    const int size = 3 + 2 * k;
    double w[7], sum = 0.0;
    for (int j = 0; j < size; ++j) {
      const double d = j - (size - 1) / 2.0;
      w[j] = exp(-d * d / (2.0 * 2.0 * 2.0));
      sum += w[j];
    }
    for (int j = 0; j < size; ++j) dst[k][j] = (float)(w[j] / sum);
  }
  return t;
}

// ---- the shared scalar PReLU of the fusion net (core/model_fusion.py:1038) on the training path: y = z > 0 ? z : a z kept
// apart from the conv so that the backward reads the branch off the PRE-activation (any slope, also <= 0) ----------------------
template <int V>
__global__ __launch_bounds__(256) void prelu_kernel(const float* __restrict__ z, const float* __restrict__ slope, float* __restrict__ y,
                                                    long long nv) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nv) return;
  const float a = slope[0];
  float v[V], o[V];
  if (V == 4) *reinterpret_cast<float4*>(v) = reinterpret_cast<const float4*>(z)[i];
  else v[0] = z[i];
#pragma unroll
  for (int e = 0; e < V; ++e) o[e] = v[e] > 0.f ? v[e] : a * v[e];
  if (V == 4) reinterpret_cast<float4*>(y)[i] = *reinterpret_cast<float4*>(o);
  else y[i] = o[0];
}

// dz = dy (z > 0 ? 1 : a); partial[blk] = {sum over z <= 0 of dy z, 0}   (torch's PReLU backward, incl. its z == 0 branch)
template <int V>
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                        const float* __restrict__ slope, float* __restrict__ dz,
                                                        double* __restrict__ partial, long long nv, int upr, long long lddy) {
  // upr > 0: dy is a rows view (upr V-element units per row, pitch lddy floats) - a channel slice of a wider gradient buffer
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double s = 0.0;
  if (i < nv) {
    const float a = slope[0];
    float g[V], v[V], o[V];
    const float* gp = upr ? dy + (i / upr) * lddy + (i % upr) * V : dy + i * V;
    if (V == 4) {
      *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(gp);
      *reinterpret_cast<float4*>(v) = reinterpret_cast<const float4*>(z)[i];
    } else {
      g[0] = gp[0];
      v[0] = z[i];
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      o[e] = v[e] > 0.f ? g[e] : a * g[e];
      t += v[e] > 0.f ? 0.f : g[e] * v[e];
    }
    if (V == 4) reinterpret_cast<float4*>(dz)[i] = *reinterpret_cast<float4*>(o);
    else dz[i] = o[0];
    s = (double)t;
  }
  block_sum2(s, 0.0, partial + 2 * (long long)blockIdx.x);
}

__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out) { out[0] = (float)in[0]; }

// out[0..1] = fixed-order sums of the two columns of partial (nblk x 2)
__global__ __launch_bounds__(256) void reduce2_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) {
    a += partial[2 * (long long)k];
    b += partial[2 * (long long)k + 1];
  }
  block_sum2(a, b, out);
}

}  // namespace

extern "C" int segmif_loss_blocks(int64_t n) { return (int)((n + 255) / 256); }

extern "C" int segmif_ssim_prep_f32(const float* gen, const float* mask, float* stack5, int64_t n, void* stream) {
  if (!gen || !mask || !stack5 || n <= 0) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(ssim_prep_kernel, dim3((unsigned)segmif_loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, gen, mask, stack5,
                     (long long)n);
  return (int)hipGetLastError();
}

extern "C" int segmif_ssim_map_f32(const float* blurred5, const float* gen, const float* mask, float* der3, double* partial,
                                   double* sums2, int64_t n, void* stream) {
  if (!blurred5 || !gen || !mask || !partial || !sums2 || n <= 0) return SEGMIF_EINVAL;
  const int nblk = segmif_loss_blocks(n);
  hipLaunchKernelGGL(ssim_map_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, blurred5, gen, mask, der3, partial,
                     (long long)n);
  hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, sums2);
  return (int)hipGetLastError();
}

extern "C" int segmif_ssim_grad_f32(const float* blurred_der3, const float* gen, const float* mask, float* grad, int64_t n,
                                    const float* upstream, float coef_ssim, float coef_mse, void* stream) {
  if (!blurred_der3 || !gen || !mask || !grad || !upstream || n <= 0) return SEGMIF_EINVAL;
  hipLaunchKernelGGL(ssim_grad_kernel, dim3((unsigned)segmif_loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, blurred_der3, gen,
                     mask, grad, (long long)n, upstream, coef_ssim, coef_mse);
  return (int)hipGetLastError();
}

extern "C" int segmif_sobel_l1_f32(const float* gen, const float* mask, float* pxy2, double* partial, double* sums2, int planes,
                                   int H, int W, void* stream) {
  if (!gen || !mask || !partial || !sums2 || planes <= 0 || H <= 0 || W <= 0) return SEGMIF_EINVAL;
  const long long n = (long long)planes * H * W;
  const int nblk = segmif_loss_blocks(n);
  hipLaunchKernelGGL(sobel_l1_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, gen, mask, pxy2, partial, H, W, n);
  hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, sums2);
  return (int)hipGetLastError();
}

extern "C" int segmif_sobel_l1_bwd_f32(const float* pxy2, const float* gen, const float* mask, float* grad, int planes, int H,
                                       int W, const float* upstream, void* stream) {
  if (!pxy2 || !gen || !mask || !grad || !upstream || planes <= 0 || H <= 0 || W <= 0) return SEGMIF_EINVAL;
  const long long n = (long long)planes * H * W;
  hipLaunchKernelGGL(sobel_l1_bwd_kernel, dim3((unsigned)segmif_loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, pxy2, gen,
                     mask, grad, H, W, n, upstream, 1.0f / (float)n);
  return (int)hipGetLastError();
}

extern "C" int segmif_laploss2_f32(const float* gen, const float* ir, const float* vis, float* sign3, double* partial,
                                   double* sums2, int planes, int H, int W, void* stream) {
  if (!gen || !ir || !vis || !partial || !sums2 || planes <= 0 || H <= 0 || W <= 0) return SEGMIF_EINVAL;
  const long long n = (long long)planes * H * W;
  const int nblk = segmif_loss_blocks(n);
  hipLaunchKernelGGL(laploss2_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, gen, ir, vis, sign3, partial, H, W, n,
                     lap_taps());
  hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, sums2);
  return (int)hipGetLastError();
}

extern "C" int segmif_laploss2_bwd_f32(const float* sign3, float* grad, int planes, int H, int W, const float* upstream,
                                       void* stream) {
  if (!sign3 || !grad || !upstream || planes <= 0 || H <= 0 || W <= 0) return SEGMIF_EINVAL;
  const long long n = (long long)planes * H * W;
  hipLaunchKernelGGL(laploss2_bwd_kernel, dim3((unsigned)segmif_loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, sign3, grad, H,
                     W, n, upstream, 1.0f / (float)n, lap_taps());
  return (int)hipGetLastError();
}

static bool prelu_vec(const void* p0, const void* p1, const void* p2, int64_t n) {
  return !(n & 3) && !(((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15);
}

extern "C" int segmif_prelu_f32(const float* z, const float* slope, float* y, int64_t n, void* stream) {
  if (!z || !slope || !y || n <= 0) return SEGMIF_EINVAL;
  if (prelu_vec(z, y, y, n))
    hipLaunchKernelGGL(prelu_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, slope, y,
                       (long long)(n / 4));
  else
    hipLaunchKernelGGL(prelu_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, slope, y,
                       (long long)n);
  return (int)hipGetLastError();
}

extern "C" int segmif_prelu_bwd_blocks(int64_t n) { return (int)((n + 255) / 256); }  // upper bound for both paths

static int prelu_bwd_dispatch(const float* dy, const float* z, const float* slope, float* dz, double* partial, float* dslope,
                              int64_t n, int C, int64_t lddy, void* stream) {
  if (!dy || !z || !slope || !dz || !partial || !dslope || n <= 0) return SEGMIF_EINVAL;
  int nblk;
  if (prelu_vec(dy, z, dz, n) && !(C & 3) && !(lddy & 3)) {
    nblk = (int)((n / 4 + 255) / 256);
    hipLaunchKernelGGL(prelu_bwd_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dy, z, slope, dz, partial,
                       (long long)(n / 4), C / 4, (long long)lddy);
  } else {
    nblk = (int)((n + 255) / 256);
    hipLaunchKernelGGL(prelu_bwd_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dy, z, slope, dz, partial,
                       (long long)n, C, (long long)lddy);
  }
  // partial holds 2 * (segmif_prelu_bwd_blocks(n) + 1) doubles: the pair after the last block's receives the fixed-order sum
  hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, partial + 2 * (long long)nblk);
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, partial + 2 * (long long)nblk, dslope);
  return (int)hipGetLastError();
}

extern "C" int segmif_prelu_bwd_f32(const float* dy, const float* z, const float* slope, float* dz, double* partial,
                                    float* dslope, int64_t n, void* stream) {
  return prelu_bwd_dispatch(dy, z, slope, dz, partial, dslope, n, 0, 0, stream);
}

extern "C" int segmif_prelu_bwd_rows_f32(const float* dy, int64_t lddy, const float* z, const float* slope, float* dz, double* partial,
                                         float* dslope, int64_t rows, int C, void* stream) {
  if (rows <= 0 || C <= 0 || lddy < C) return SEGMIF_EINVAL;
  return prelu_bwd_dispatch(dy, z, slope, dz, partial, dslope, rows * C, C, lddy, stream);
}
