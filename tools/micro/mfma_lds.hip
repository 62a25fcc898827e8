// Micro-benchmark: does LDS fragment traffic inflate the matrix pipe's occupancy per MFMA?
// 12 MFMAs per "step" on two accumulators; per step R ds_read_b128 results replace operand registers
// (double-buffered like the conv kernels).  ORDER 0: alternate acc0/acc1 (shared B); 1: six on acc0 then six on acc1.
// USE 1: the MFMAs consume the freshly read registers; 0: the reads land in registers nobody multiplies.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int R, int ORDER, int USE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = seed[i % 128] * 1e-3f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int v = 0; v < 16; ++v) { acc0[v] = 0.f; acc1[v] = 0.f; }
  const int lane = threadIdx.x & 63;
  const unsigned char* base = lds + (threadIdx.x >> 6) * 16384 + lane * 16;  // contiguous: conflict free
  bf16x8 F[2][3], W[2][3], W2[2][3], X[2][6];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    F[0][j] = F[1][j] = *reinterpret_cast<const bf16x8*>(base + j * 1024);
    W[0][j] = W[1][j] = *reinterpret_cast<const bf16x8*>(base + (3 + j) * 1024);
    W2[0][j] = W2[1][j] = *reinterpret_cast<const bf16x8*>(base + (6 + j) * 1024);
  }
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PW[6] = {0, 1, 2, 0, 1, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int cur = st & 1, nxt = cur ^ 1;
      const unsigned char* p = base + ((it * 2 + st) & 3) * 1024;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + j * 1024);
        if (USE) {
          if (j < 3) F[nxt][j] = v; else W[nxt][j - 3] = v;
        } else {
          X[nxt][j] = v;
        }
      }
      if (ORDER == 0) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[cur][PW[t]], F[cur][PA[t]], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2[cur][PW[t]], F[cur][PA[t]], acc1, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[cur][PW[t]], F[cur][PA[t]], acc0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 6; ++t) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2[cur][PW[t]], F[cur][PA[t]], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (t < R) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!USE) {
#pragma unroll
        for (int j = 0; j < R; ++j) asm volatile("" ::"v"(X[nxt][j]));
      }
    }
  }
  float s = 0.f;
  for (int v = 0; v < 16; ++v) s += acc0[v] + acc1[v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, int ORDER, int USE>
void run(float* d, const float* seed, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<R, ORDER, USE>), dim3(blocks), dim3(256), 0, 0, d, 10, seed);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<R, ORDER, USE>), dim3(blocks), dim3(256), 0, 0, d, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 24 * (blocks / 256);
  printf("reads/step %d order %d use %d, %d wave(s)/SIMD: %.3f ms, %.1f ns per MFMA per SIMD\n", R, ORDER, USE, blocks / 256, ms,
         ms * 1e6 / mfma_per_simd);
}

int main() {
  float* d; (void)hipMalloc(&d, 512 * 256 * 4);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = (float)((i * 37) % 19 - 9) + 0.37f * i;
  float* seed; (void)hipMalloc(&seed, sizeof(h)); (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  for (int blocks = 256; blocks <= 512; blocks += 256) {
    run<0, 0, 1>(d, seed, blocks); run<3, 0, 1>(d, seed, blocks); run<6, 0, 1>(d, seed, blocks); run<6, 1, 1>(d, seed, blocks);
    run<6, 0, 0>(d, seed, blocks); run<6, 1, 0>(d, seed, blocks);
  }
  return 0;
}
