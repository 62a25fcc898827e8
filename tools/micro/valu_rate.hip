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
int main() {
  run("v_mov", k_mov, 1);
  run("v_fma", k_fma, 1);
  run("pk_fma", k_pkfma, 1);
  run("cmp+cnd", k_cnd, 2);
  run("dep fma", k_dep, 1);
  run("cvt_pk", k_cvt, 1);
  return 0;
}
