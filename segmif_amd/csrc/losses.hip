// Fusion losses of train_fusion (train.py:363-383) as fused HIP kernels, forward and backward:
//   Fusionloss_grad3 (core/loss.py:506-517):  MSE(mask_0, fused) + 1.1 (1 - SSIM(fused, mask_0)), SSIM = pytorch_ssim
//                                             (pytorch_ssim/__init__.py:19-43: 11x11 Gaussian window, sigma 1.5, zero padding)
//   Fusionloss3      (core/loss.py:459-476):  L1(mask_0, fused) + L1(Sobelxy(mask_0), Sobelxy(fused)),  Sobelxy :634-650
// Single-channel (B, 1, H, W) images.  The five window convolutions of SSIM (and the three of its backward) run in the
// separable blur kernel of rowops.hip; everything around them — the product planes, the SSIM map with its reduction, the
// derivative planes, the gradient assembly, the Sobel stencils and their adjoint — is here.  Reductions are two-pass and
// deterministic: per-block partial sums in double, then one fixed-order sum.
#include <hip/hip_runtime.h>
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
