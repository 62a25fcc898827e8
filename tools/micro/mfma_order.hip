// Micro-benchmark: matrix-pipe occupancy of v_mfma_f32_32x32x16_bf16 as a function of the ORDER in which two
// accumulation chains are fed (registers only, order pinned with sched_barrier).  Motivation: in the DRDB conv
// kernels SQ_VALU_MFMA_BUSY_CYCLES reads 42.7 cycles per MFMA although a pure chain costs 32.
//   V0: one chain                      acc0 x12
//   V1: blocks of six                  acc0 x6, acc1 x6
//   V2: alternate, shared B operand    (acc0 <- a0,b ; acc1 <- a1,b) x6      <- round-1 step order
//   V3: alternate, shared A operand
//   V4: alternate, nothing shared
//   V5: blocks of three                acc0 x3, acc1 x3, acc0 x3, acc1 x3
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)

template <int V>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* seed) {
  f32x16 acc0, acc1;
  for (int v = 0; v < 16; ++v) { acc0[v] = 0.f; acc1[v] = 0.f; }
  bf16x8 a[3], b[3], c[3], d[3];
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 8; ++i) {
      a[j][i] = (__bf16)(seed[threadIdx.x % 64 + i + j] * 0.01f);
      b[j][i] = (__bf16)(seed[i + 3 * j + 1] * 0.02f);
      c[j][i] = (__bf16)(seed[i + 5 * j + 2] * 0.03f);
      d[j][i] = (__bf16)(seed[i + 7 * j + 3] * 0.04f);
    }
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int t = 0; t < 12; ++t) { MF(acc0, a[t % 3], b[(t / 3) % 3]); }
    } else if (V == 1) {
#pragma unroll
      for (int t = 0; t < 6; ++t) { MF(acc0, a[t % 3], b[t / 2]); }
#pragma unroll
      for (int t = 0; t < 6; ++t) { MF(acc1, c[t % 3], b[t / 2]); }
    } else if (V == 2) {
#pragma unroll
      for (int t = 0; t < 6; ++t) { MF(acc0, a[t % 3], b[t / 2]); MF(acc1, c[t % 3], b[t / 2]); }
    } else if (V == 3) {
#pragma unroll
      for (int t = 0; t < 6; ++t) { MF(acc0, a[t % 3], b[t / 2]); MF(acc1, a[t % 3], d[t / 2]); }
    } else if (V == 4) {
#pragma unroll
      for (int t = 0; t < 6; ++t) { MF(acc0, a[t % 3], b[t / 2]); MF(acc1, c[t % 3], d[t / 2]); }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { MF(acc0, a[t], b[u]); }
#pragma unroll
        for (int t = 0; t < 3; ++t) { MF(acc1, c[t], b[u]); }
      }
    }
  }
  float s = 0.f;
  for (int v = 0; v < 16; ++v) s += acc0[v] + acc1[v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(int threads, float* d, const float* seed) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, 10, seed);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 12 * (threads / 256);
  printf("V%d, %d wave(s)/SIMD: %.3f ms, %.1f ns per MFMA per SIMD\n", V, threads / 256, ms, ms * 1e6 / mfma_per_simd);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = (float)((i * 37) % 19 - 9) + 0.37f * i;
  float* seed; (void)hipMalloc(&seed, sizeof(h)); (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(256, d, seed); run<1>(256, d, seed); run<2>(256, d, seed); run<3>(256, d, seed); run<4>(256, d, seed); run<5>(256, d, seed);
  run<0>(512, d, seed); run<1>(512, d, seed); run<2>(512, d, seed); run<4>(512, d, seed);
  return 0;
}
