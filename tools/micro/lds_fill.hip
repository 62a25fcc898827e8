// Micro-benchmark: how fast can one CU pull L2-resident data into LDS?
//   MODE 0: LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction)
//   MODE 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// One workgroup per CU (WAVES waves), each wave moves its share of a TILE-byte block per round, `depth` rounds in flight
// before a vmcnt wait; the source block (per workgroup) stays in L2.  Prints bytes per clock per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_fill.hip -o /tmp/lds_fill && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FOOT = 65536;
template <int MODE, int WAVES, int UNROLL>
__global__ __launch_bounds__(WAVES * 64) void k(const unsigned char* src, int rounds, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* s = src + (size_t)blockIdx.x * (1 << 18);  // FOOT bytes per workgroup, cycled: 256 x 32 KB = 8 MB stays in L2
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rounds; ++r) {
    const unsigned char* sr = s + ((r * 4096) & (FOOT - 1));
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) {
        const int i = j * WAVES + wave;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sr + i * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 v[UNROLL];
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) v[j] = *reinterpret_cast<const f32x4*>(sr + (j * WAVES + wave) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) *reinterpret_cast<volatile f32x4*>(lds + (j * WAVES + wave) * 1024 + lane * 16) = v[j];
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  acc = *reinterpret_cast<const f32x4*>(lds + tid * 16);
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int MODE, int WAVES, int UNROLL>
void run(const unsigned char* src, float* sink, const char* name) {
  const int rounds = 2000, grid = 256;
  const size_t smem = (size_t)UNROLL * WAVES * 1024;
  hipFuncSetAttribute((const void*)k<MODE, WAVES, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, WAVES, UNROLL><<<grid, WAVES * 64, smem>>>(src, 10, sink);
  hipEventRecord(a);
  k<MODE, WAVES, UNROLL><<<grid, WAVES * 64, smem>>>(src, rounds, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)rounds * UNROLL * WAVES * 1024;
  printf("%-28s waves %d x %d KB in flight: %7.3f ms  %6.1f GB/s per CU  = %5.1f B/clk at 2.4 GHz (chip %5.2f TB/s)\n", name, WAVES,
         UNROLL * WAVES, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 2.4, bytes * grid / ms / 1e9);
}

int main() {
  unsigned char* src; float* sink;
  hipMalloc(&src, (size_t)258 << 20); hipMemset(src, 1, (size_t)258 << 20); hipMalloc(&sink, 64);
  run<0, 4, 7>(src, sink, "LDS-DMA");
  run<0, 8, 7>(src, sink, "LDS-DMA");
  run<0, 8, 13>(src, sink, "LDS-DMA");
  run<0, 16, 8>(src, sink, "LDS-DMA");
  run<1, 4, 7>(src, sink, "load + ds_write_b128");
  run<1, 8, 7>(src, sink, "load + ds_write_b128");
  run<1, 8, 13>(src, sink, "load + ds_write_b128");
  run<1, 16, 8>(src, sink, "load + ds_write_b128");
  return 0;
}
