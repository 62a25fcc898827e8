// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 from ONE or TWO waves per SIMD as a function
// of the number of independent accumulation chains (dependent-issue latency of the matrix pipe).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % CHAINS], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int v = 0; v < 16; ++v) s += acc[c][v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int threads, float* d) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, d, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 16 * (threads / 256);  // waves per SIMD x MFMAs per wave
  const double tf = (double)blocks * (threads / 64) * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("chains %d, %d wave(s)/SIMD: %.3f ms, %.0f TF/s, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", CHAINS,
         threads / 256, ms, tf, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  run<1>(256, d); run<2>(256, d); run<3>(256, d); run<4>(256, d); run<8>(256, d);
  run<1>(512, d); run<2>(512, d); run<4>(512, d);
  return 0;
}
