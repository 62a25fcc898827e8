// Evaluation-side kernels: the confusion matrix behind the mIoU the reference's scripts
// print (test_segmentation.py:176 sklearn confusion_matrix(labels=[0..8]) -> util/util.py:31-55), and the
// uint8 write-out of a fused image (test_fusion.py:112-120: uint8(255 x), global min/max over the batch,
// rescale, uint8) — so neither needs a device-to-host copy of full-resolution tensors.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "segmif_hip.h"

namespace {

constexpr int MAXK = 32;

// conf[t][p] += 1 for every element with 0 <= t < K and 0 <= p < K (elements outside the label set are
// ignored, as sklearn does when `labels` is given).  Block-private LDS histogram, one atomic per bin.
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ pred, const int64_t* __restrict__ label,
                                                        long long* __restrict__ conf, long long n, int K) {
  __shared__ unsigned int h[MAXK * MAXK];
  for (int i = threadIdx.x; i < K * K; i += 256) h[i] = 0u;
  __syncthreads();
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const long long t = label[i];
    const int p = pred[i];
    if (t >= 0 && t < K && p >= 0 && p < K) atomicAdd(&h[(int)t * K + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += 256)
    if (h[i]) atomicAdd(reinterpret_cast<unsigned long long*>(conf) + i, (unsigned long long)h[i]);
}

__device__ __forceinline__ int to_u8(float x) {  // np.uint8(float32) for values in [0, 255]: truncation
  return (int)(255.0f * x);
}

// mm[0] = min, mm[1] = max of uint8(255 x) over the whole tensor (mm initialised to {255, 0})
__global__ __launch_bounds__(256) void u8_minmax_kernel(const float* __restrict__ x, int* __restrict__ mm, long long n) {
  int lo = 255, hi = 0;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int a = to_u8(x[i]);
    lo = min(lo, a);
    hi = max(hi, a);
  }
  for (int o = 32; o; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(mm, lo);
    atomicMax(mm + 1, hi);
  }
}

// NCHW fp32 in [0,1] -> NHWC uint8: uint8(255.0 * ((a - min) / (max - min))) with a = uint8(255 x), the
// division and the product in float64 like numpy's; max == min (numpy: 0/0 = NaN -> 0) gives 0.
__global__ __launch_bounds__(256) void u8_quantize_kernel(const float* __restrict__ x, uint8_t* __restrict__ out,
                                                          const int* __restrict__ mm, int B, int C, long long HW) {
  const long long total = (long long)B * C * HW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pix = (i / C) % HW;
  const long long b = i / ((long long)C * HW);
  const int lo = mm[0], hi = mm[1];
  const int a = to_u8(x[(b * C + c) * HW + pix]);
  uint8_t q = 0;
  if (hi > lo) q = (uint8_t)(255.0 * ((double)(a - lo) / (double)(hi - lo)));
  out[i] = q;
}

// NHWC uint8 -> NCHW fp32 / 255 (IEEE division, as numpy's float32 array / 255.0)
__global__ __launch_bounds__(256) void u8_dequantize_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int C,
                                                            long long HW, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // output index (b, c, pix)
  if (i >= total) return;
  const long long pix = i % HW;
  const int c = (int)((i / HW) % C);
  const long long b = i / ((long long)C * HW);
  out[i] = __fdiv_rn((float)in[(b * HW + pix) * C + c], 255.0f);
}

}  // namespace

extern "C" int segmif_dequantize_u8(const uint8_t* in_nhwc, float* out_nchw, int B, int C, int64_t HW, void* stream) {
  if (!in_nhwc || !out_nchw || B <= 0 || C <= 0 || HW <= 0) return SEGMIF_EINVAL;
  const long long total = (long long)B * C * HW;
  hipLaunchKernelGGL(u8_dequantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in_nhwc,
                     out_nchw, C, (long long)HW, total);
  return (int)hipGetLastError();
}

extern "C" int segmif_confusion_i32(const int32_t* pred, const int64_t* label, int64_t* conf, int64_t n, int K,
                                    void* stream) {
  if (!pred || !label || !conf || n < 0 || K <= 0 || K > MAXK) return SEGMIF_EINVAL;
  if (n == 0) return 0;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, label,
                     (long long*)conf, (long long)n, K);
  return (int)hipGetLastError();
}

extern "C" int segmif_quantize_u8(const float* x_nchw, uint8_t* out_nhwc, int32_t* minmax, int B, int C, int64_t HW,
                                  void* stream) {
  if (!x_nchw || !out_nhwc || !minmax || B <= 0 || C <= 0 || HW <= 0) return SEGMIF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // {min, max} start at {255, 0}: two device-side fills (stream ordered, graph capturable; no host buffer involved)
  hipError_t e = hipMemsetD32Async((hipDeviceptr_t)minmax, 255, 1, s);
  if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)(minmax + 1), 0, 1, s);
  if (e != hipSuccess) return (int)e;
  const long long n = (long long)B * C * HW;
  long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(u8_minmax_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, x_nchw, minmax, n);
  hipLaunchKernelGGL(u8_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x_nchw, out_nhwc, minmax, B, C,
                     (long long)HW);
  return (int)hipGetLastError();
}
