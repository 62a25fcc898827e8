// Private to segmif_amd/csrc: kernel-side argument block shared by the implicit-GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "segmif_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace segmif {

struct IgemmK {
  const float* in;
  const float* in2;
  const float* wt;
  const float* bias;
  const float* res;
  const float* prelu;
  float* out;
  long long M;
  int N, K, Kp;
  int lda, lda2, K1, ldo, ldr;
  int H, W, Cin, KH, KW, stride, pad, dil, OH, OW;
  int act;
  long long in_zs, in2_zs, wt_zs, out_zs, res_zs;
  int nz2;  // z = zb * nz2 + z2; second-level strides below (0 when unused)
  long long in_zs2, wt_zs2, out_zs2, res_zs2;
  int ldw;  // weight row pitch (floats), normally Kp
  int splitk;  // > 1: gridDim.z = splitk, raw partial sums go to ws[split][M][N] (reduce + epilogue in a 2nd kernel)
  int ksteps_per_split;
  float* ws;
  const float* ln_gamma;  // fused LayerNorm over the N = 64 output columns (tiles whose wave tile spans 64 columns)
  const float* ln_beta;
  float ln_eps;
  int ntm, ntn;
  unsigned char* planes;  // optional planes copy of the output (conv3x3_planes.hip format): chunks [pl_chunk0, pl_chunk0 + N/16)
  int pl_Hp, pl_Wp, pl_chunks, pl_chunk0;
  int pl_f16;             // planes are f16x3 half pairs (64 bytes per pixel) instead of bf16 triples (96)
  uint32_t* pl_amax;      // f16x3: range slots receiving max |output| (planes16.h) or null
  int pl_amax_images;     // > 1: one slot per image (M = images x OH x OW), else everything reports to pl_amax[0]
  const float* mask;  // out = mask[m][n] > 0 ? y : 0, applied after the residual: the split 3x3 tile (DRDB backward) and the
  int ldm;            // dense tiles' 16-byte epilogue (CrossPath backward: a gradient written through the consumer's ReLU mask)
  long long mask_zs;  // z stride of the mask (batched weights: one image per z)
  int split_f16;             // split 3x3 tile: wt is an f16x3 image (segmif_conv3x3_split16_pack), in_amax must be given
  const uint32_t* in_amax;   // range slots (bit patterns of max |x|) of the input's channel blocks; the kernel takes their maximum
  int in_amax_n;
  uint32_t* out_amax;        // or null: out_amax_n words (a power of two) receiving max |output|, spread by workgroup index
  int out_amax_n;
  int vec4;  // epilogue may use 16-byte accesses: N, ldo, ldr, z strides multiples of 4 and out / res / bias / ws 16-byte aligned
};

__device__ __forceinline__ float gelu_exact(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}


// halo-tiled 3x3 stride-1 convolution (conv3x3.hip); variant 0: 16-channel chunks, 1: 8-channel chunks
bool conv3x3_halo_eligible(const IgemmK& k);
int conv3x3_halo_launch(const IgemmK& k, int variant, hipStream_t stream);
// same geometry on the bf16 matrix pipe with 3-way split operands (conv3x3_split.hip); k.wt = split-packed weights
int conv3x3_split_launch(const IgemmK& k, hipStream_t stream);

}  // namespace segmif
