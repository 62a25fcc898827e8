// Halo-tiled 3x3 stride-1 convolution (dilation 1 or 2) on the exact-fp32 matrix pipe — the DRDB
// dilated convs (core/model_fusion.py:121-157: 59 % of a pair's FLOPs) and the plain 3x3 convs of
// the fusion net.
//
// Why a second kernel: the generic implicit GEMM (igemm.hip) re-gathers every input pixel once per
// tap, i.e. 9x through L2 -> L1 -> LDS; with only 32 output channels to amortise it over, that
// gather — not the matrix pipe — bounds the DRDB convs (44-55 % of the fp32 MFMA peak measured in
// round 1, against 70 % for the 64-output conv2).  Here a workgroup owns an 8 x 32 patch of output
// pixels, stages the (8+2d) x (32+2d) input halo for one channel chunk in LDS ONCE, and reads the
// nine taps' A fragments from it at shifted addresses: 1.7x (halo) instead of 9x gather traffic.
//
// Mapping (256 threads = 4 waves): wave w owns patch rows 2w, 2w+1; an MFMA 32x32 sub-tile is one
// patch row (32 consecutive pixels) x 32 output channels, so fragment lanes read consecutive
// pixels (row pitch CK+4 dwords -> conflict-free ds_read_b128).  One LDS buffer, register
// prefetch: chunk c+1 streams into registers under chunk c's 9 * CK/2 MFMAs per sub-tile.
#include <hip/hip_runtime.h>
#include "device_once.h"
#include <stdint.h>

#include "igemm_common.h"
#include "segmif_hip.h"

namespace segmif {
namespace {

constexpr int TH = 8, TW = 32;

template <int NOUT, int CK, int DIL>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const IgemmK p, int tiles_x, int tiles_y) {
  constexpr int CKP = CK + 4;
  constexpr int UPP = CK / 4;  // float4 units per pixel / per weight row
  constexpr int HH = TH + 2 * DIL, HW = TW + 2 * DIL, HP = HH * HW;
  constexpr int A_UNITS = HP * UPP, B_UNITS = 9 * NOUT * UPP;
  constexpr int AJ = (A_UNITS + 255) / 256, BJ = (B_UNITS + 255) / 256;
  constexpr int TN = NOUT / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;  // [HP][CKP]
  float* Bs = smem + HP * CKP;  // [9][NOUT][CKP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware remap: each XCD owns a contiguous run of tiles (x fastest, then y, then image), so
  // vertically adjacent patches — which share 2*DIL halo rows — hit the same L2.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;  // (gridDim.y = output-channel tiles)
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int nbase = blockIdx.y * NOUT;  // output-channel tile (N > 64: the halo is re-read per tile, from L2)
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int b = bid / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  const float* __restrict__ in = p.in + (long long)b * p.H * p.W * p.lda;

  // per-thread gather slots (fixed across channel chunks)
  int a_pix[AJ];  // pixel index inside the image, -1 = zero (padding or unused slot)
  int a_dst[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int u = tid + 256 * j;
    a_pix[j] = -1;
    a_dst[j] = -1;
    if (u < A_UNITS) {
      const int pp = u / UPP, q4 = u - pp * UPP;
      const int hy = pp / HW, hx = pp - hy * HW;
      const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
      a_dst[j] = pp * CKP + q4 * 4;
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) a_pix[j] = gy * p.W + gx;
    }
  }
  int b_src[BJ], b_dst[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int u = tid + 256 * j;
    b_src[j] = -1;
    b_dst[j] = -1;
    if (u < B_UNITS) {
      const int row = u / UPP, q4 = u - row * UPP;  // row = tap * NOUT + n
      const int tap = row / NOUT, n = row - tap * NOUT;
      b_dst[j] = row * CKP + q4 * 4;
      if (nbase + n < p.N) b_src[j] = (nbase + n) * p.Kp + tap * p.Cin + q4 * 4;  // rows beyond N stay zero
    }
  }
  const int a_q4 = (tid % UPP) * 4;

  f32x4 ra[AJ], rb[BJ];
  auto gload = [&](int c0) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a_pix[j] >= 0) ra[j] = *reinterpret_cast<const f32x4*>(in + (long long)a_pix[j] * p.lda + c0 + a_q4);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      rb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b_src[j] >= 0) rb[j] = *reinterpret_cast<const f32x4*>(p.wt + b_src[j] + c0);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      if (a_dst[j] >= 0) *reinterpret_cast<f32x4*>(As + a_dst[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      if (b_dst[j] >= 0) *reinterpret_cast<f32x4*>(Bs + b_dst[j]) = rb[j];
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int nchunks = p.Cin / CK;
  gload(0);
  for (int c = 0; c < nchunks; ++c) {
    sstore();
    __syncthreads();
    if (c + 1 < nchunks) gload((c + 1) * CK);
    const float* a_lane = As + ((2 * wave) * HW + r) * CKP + 4 * h;
    const float* b_lane = Bs + r * CKP + 4 * h;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* a_tap = a_lane + (ky * DIL * HW + kx * DIL) * CKP;
        const float* b_tap = b_lane + (ky * 3 + kx) * NOUT * CKP;
#pragma unroll
        for (int t = 0; t < CK / 8; ++t) {
          f32x4 a[2], bb[TN];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_tap + i * HW * CKP + 8 * t);
#pragma unroll
          for (int j = 0; j < TN; ++j) bb[j] = *reinterpret_cast<const f32x4*>(b_tap + j * 32 * CKP + 8 * t);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], bb[j][s], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const float slope = (p.act == SEGMIF_ACT_PRELU) ? *p.prelu : 0.f;
  const long long img = (long long)b * p.H * p.W;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = y0 + 2 * wave + i;
    if (oy >= p.H) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nbase + j * 32 + r;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int ox = x0 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (ox >= p.W) continue;
        const long long m = img + (long long)oy * p.W + ox;
        float y = acc[i][j][v] + bv;
        if (p.act == SEGMIF_ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == SEGMIF_ACT_PRELU) y = y >= 0.f ? y : slope * y;
        else if (p.act == SEGMIF_ACT_GELU) y = gelu_exact(y);
        if (p.res) y += p.res[m * p.ldr + n];
        p.out[m * p.ldo + n] = y;
      }
    }
  }
}

template <int NOUT, int CK, int DIL>
int launch(const IgemmK& k, hipStream_t stream) {
  constexpr int HP = (TH + 2 * DIL) * (TW + 2 * DIL);
  constexpr size_t smem = (size_t)(HP + 9 * NOUT) * (CK + 4) * sizeof(float);
  auto fn = conv3x3_halo_kernel<NOUT, CK, DIL>;
  if (smem > 64 * 1024) {
    static segmif::PerDeviceFlag raised_flag;
  bool& raised = raised_flag.here();
    if (!raised) {
      hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      raised = true;
    }
  }
  const int tiles_x = (k.W + TW - 1) / TW, tiles_y = (k.H + TH - 1) / TH;
  const long long B = k.M / ((long long)k.H * k.W);
  dim3 grid((unsigned)(B * tiles_x * tiles_y), (unsigned)((k.N + NOUT - 1) / NOUT));
  hipLaunchKernelGGL(fn, grid, dim3(256), smem, stream, k, tiles_x, tiles_y);
  return (int)hipGetLastError();
}

}  // namespace

bool conv3x3_halo_eligible(const IgemmK& k) {
  return k.KH == 3 && k.KW == 3 && k.stride == 1 && (k.dil == 1 || k.dil == 2) && k.pad == k.dil &&
         k.OH == k.H && k.OW == k.W && k.Cin % 16 == 0 && (k.lda % 4) == 0 && (k.N <= 256) && !k.in2 &&
         k.in_zs == 0 && k.wt_zs == 0 && k.M % ((long long)k.H * k.W) == 0 && (long long)k.H * k.W < (1ll << 31) &&
         !(((uintptr_t)k.in | (uintptr_t)k.wt) & 15);
}

int conv3x3_halo_launch(const IgemmK& k, int variant, hipStream_t s) {
  const bool wide = k.N > 32;
  if (variant == 0) {
    if (k.dil == 2) return wide ? launch<64, 16, 2>(k, s) : launch<32, 16, 2>(k, s);
    return wide ? launch<64, 16, 1>(k, s) : launch<32, 16, 1>(k, s);
  }
  if (k.dil == 2) return wide ? launch<64, 8, 2>(k, s) : launch<32, 8, 2>(k, s);
  return wide ? launch<64, 8, 1>(k, s) : launch<32, 8, 1>(k, s);
}

}  // namespace segmif
