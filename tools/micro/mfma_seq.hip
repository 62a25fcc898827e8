// Micro-benchmark: the exact MFMA sequence of one chunk of csrc/conv3x3_planes.hip (12 steps, 108 MFMAs, two
// accumulators, operands from a 2 x 3 activation set and a 3 x 3 weight ring held in registers), to find out why
// SQ_VALU_MFMA_BUSY_CYCLES reads 42.7 cycles per MFMA in the conv kernels against 32 in a plain chain.
// ENV 0: launch_bounds(256) only; 1: + amdgpu_waves_per_eu(2,2); 2: + 69 KB dynamic LDS (two workgroups per CU)
// SEQ 0: conv order; 1: PA/PW order in which consecutive products share an operand; 2: one product per accumulator pair
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_seq.hip -o /tmp/mfma_seq && /tmp/mfma_seq
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SEQ>
__device__ __forceinline__ void body(float* out, int iters, const float* seed) {
  f32x16 acc[2];
  for (int v = 0; v < 16; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
  bf16x8 F[2][3], Wr[3][3];
  for (int i = 0; i < 8; ++i) {
    for (int j = 0; j < 6; ++j) F[j / 3][j % 3][i] = (__bf16)(seed[(threadIdx.x + i * 7 + j * 13) % 128] * 0.01f);
    for (int j = 0; j < 9; ++j) Wr[j / 3][j % 3][i] = (__bf16)(seed[(i * 5 + j * 11 + 3) % 128] * 0.02f);
  }
  constexpr int PA0[6] = {2, 1, 0, 1, 0, 0}, PW0[6] = {0, 1, 2, 0, 1, 0};
  constexpr int PA1[6] = {2, 1, 1, 0, 0, 0}, PW1[6] = {0, 0, 1, 1, 2, 0};  // neighbours share a plane of one operand
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 12; ++st) {
      const int m = st & 3;
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pa = SEQ == 1 ? PA1[t] : PA0[t], pw = SEQ == 1 ? PW1[t] : PW0[t];
        if (m < 3) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[m][pw], F[st & 1][pa], acc[0], 0, 0, 0);
        if (m > 0) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[m - 1][pw], F[st & 1][pa], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int v = 0; v < 16; ++v) s += acc[0][v] + acc[1][v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SEQ> __global__ __launch_bounds__(256) void k0(float* out, int iters, const float* seed) { body<SEQ>(out, iters, seed); }
template <int SEQ> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k1(float* out, int iters, const float* seed) {
  extern __shared__ float sm[];
  if (iters < 0) sm[threadIdx.x] = 0.f;
  body<SEQ>(out, iters, seed);
}

template <typename K>
void run(const char* name, K fn, size_t smem, float* d, const float* seed) {
  const int iters = 400, blocks = 512;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (smem > 65536) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), smem, 0, d, 10, seed);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), smem, 0, d, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.3f ms, %.1f ns per MFMA per SIMD (2 waves/SIMD)\n", name, ms, ms * 1e6 / (iters * 108.0 * 2));
}

int main() {
  float* d; (void)hipMalloc(&d, 512 * 256 * 4);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = (float)((i * 37) % 19 - 9) + 0.37f * i;
  float* seed; (void)hipMalloc(&seed, sizeof(h)); (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  run("env0 seq0", k0<0>, 0, d, seed);
  run("env0 seq1", k0<1>, 0, d, seed);
  run("env1 seq0", k1<0>, 0, d, seed);
  run("env2 seq0", k1<0>, 69632, d, seed);
  run("env2 seq1", k1<1>, 69632, d, seed);
  return 0;
}
