// (r6) How fast does ONE wave per SIMD issue straight-line vector-ALU code?  The conv kernels' epilogues (hundreds of unrolled VALU
// instructions run by one team of four waves while the other team sleeps at a barrier) measure ~12-15 ticks per instruction
// (profiles/r06_tail_epilogue_timeline.txt).  This probe times .rept blocks of independent instructions with s_memtime, second pass
// (instruction cache warm), for 1 / 2 waves per SIMD:   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 512
#define STR2(x) #x
#define STR(x) STR2(x)
#define BLOCK(name, body)                                                                          \
  __global__ void name(unsigned long long* out) {                                                  \
    unsigned long long t0 = 0, t1 = 0;                                                             \
    float a = threadIdx.x, b = 1.5f, c = 2.5f;                                                     \
    for (int pass = 0; pass < 2; ++pass) {                                                         \
      __syncthreads();                                                                             \
      t0 = __builtin_amdgcn_s_memtime();                                                           \
      asm volatile("s_waitcnt lgkmcnt(0)\n .rept " STR(N) "\n" body "\n .endr\n" : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v42", "v43", "vcc"); \
      t1 = __builtin_amdgcn_s_memtime();                                                           \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    }                                                                                              \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;              \
    if (a == 12345.f) out[0] = 0;                                                                  \
  }
BLOCK(k_mov, "v_mov_b32 v40, %1")
BLOCK(k_fma, "v_fma_f32 v40, %1, %2, %1")
BLOCK(k_pkfma, "v_pk_fma_f32 v[40:41], v[40:41], v[42:43], v[40:41]")
BLOCK(k_cnd, "v_cmp_le_f32_e32 vcc, 0, %1\n v_cndmask_b32_e32 v40, %1, %2, vcc")
BLOCK(k_dep, "v_fma_f32 %0, %0, %1, %2")
BLOCK(k_cvt, "v_cvt_pk_f16_f32 v40, %1, %2")
template <typename K>
void run(const char* name, K k, int instr_per_rep) {
  unsigned long long* d; (void)hipMalloc(&d, 4096 * 8);
  for (int threads : {256, 512}) {
    for (int grid : {1, 256}) {
      (void)hipMemset(d, 0, 4096 * 8);
      hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d);
      unsigned long long h[8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("%-8s %d waves/SIMD, grid %3d: %6.2f ticks per instruction (wave 0: %llu ticks for %d)\n", name, threads / 256, grid,
             (double)h[0] / (N * instr_per_rep), h[0], N * instr_per_rep);
    }
  }
  (void)hipFree(d);
}
#ifndef TEAMS
int main() {
  run("v_mov", k_mov, 1);
  run("v_fma", k_fma, 1);
  run("pk_fma", k_pkfma, 1);
  run("cmp+cnd", k_cnd, 2);
  run("dep fma", k_dep, 1);
  run("cvt_pk", k_cvt, 1);
  return 0;
}
#endif
// ---- second question: does a team of four waves parked at s_barrier (or holding LDS-DMA loads in flight) slow the other team's VALU? ----
// build with -DTEAMS: 512 threads; waves 4-7 optionally issue `ndma` LDS-DMA instructions each, then wait at the workgroup barrier;
// waves 0-3 run the v_fma block, then join the barrier.
#ifdef TEAMS
__global__ __launch_bounds__(512) void k_teams(unsigned long long* out, const unsigned char* src, int ndma) {
  extern __shared__ unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long t0 = 0, t1 = 0;
  float a = threadIdx.x, b = 1.5f, c = 2.5f;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (wave >= 4) {
      for (int i = 0; i < ndma; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)blockIdx.x * 64 + (wave - 4) * 16 + i) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + ((wave - 4) * 16 + i) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      t0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)\n .rept " STR(N) "\n v_fma_f32 v40, %1, %2, %1\n .endr\n" : "+v"(a) : "v"(b), "v"(c) : "v40");
      t1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  if (wave < 4 && lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (a == 12345.f) out[0] = 0;
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 4096 * 8);
  unsigned char* src; (void)hipMalloc(&src, (size_t)256 * 64 * 1024);
  (void)hipFuncSetAttribute((const void*)k_teams, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int ndma : {0, 4, 16}) {
    hipLaunchKernelGGL(k_teams, dim3(256), dim3(512), 65536, 0, d, src, ndma);
    unsigned long long h[8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("teams: other team issues %2d LDS-DMA instructions per wave then waits at the barrier: %6.2f ticks per v_fma\n", ndma, (double)h[0] / N);
  }
  return 0;
}
#endif
