// Prints HW_REG_LDS_ALLOC for co-resident workgroups (which LDS slot did this block get?).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  extern __shared__ float s[];
  s[threadIdx.x] = 1.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
  for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 1024 * 8);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80640);
  hipLaunchKernelGGL(k, dim3(1024), dim3(256), 80640, 0, d);
  unsigned h[2048]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 1024; i += 37) printf("block %4d: LDS_ALLOC=0x%08x xcc=0x%x\n", i, h[2 * i], h[2 * i + 1]);
  int z = 0; for (int i = 0; i < 1024; ++i) z += (h[2 * i] & 0xfff) == 0;
  printf("blocks with base 0: %d of 1024\n", z);
  return 0;
}
