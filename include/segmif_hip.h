/*
 * segmif_hip.h — C ABI of libsegmif_hip.so: the MI355X (gfx950) kernels behind SegMiF's
 * fusion + segmentation hot path.
 *
 * The upstream reference (JinyuanLiu-CV/SegMiF) is pure Python: it has no FFI of its own and
 * delegates all arithmetic to aten ops.  The entry points below are therefore what a binding
 * for this path binds *instead of* those aten calls; each one cites the reference call site(s)
 * it replaces (paths relative to the reference root).  segmif_amd/_lib.py is the ctypes binding;
 * INTEGRATION.md shows how the reference's core/ package is swapped for segmif_amd.core.
 *
 * Conventions
 *  - all tensors are fp32, device-resident, activation layout NHWC == tokens (B, H*W, C);
 *    a "row" is one pixel/token; leading dimensions (ld*) are in floats.
 *  - no allocation, no global state, re-entrant per stream; work is enqueued on `stream`
 *    (a hipStream_t passed as void*).  Returns 0 on success, a hipError_t / negative
 *    SEGMIF_E* code otherwise.
 *  - pointers must be 16-byte aligned where a vector path applies (documented per call).
 */
#ifndef SEGMIF_HIP_H
#define SEGMIF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGMIF_ABI_VERSION 3
#define SEGMIF_EINVAL (-22)
#define SEGMIF_ENOSYS (-38)

/* activation codes for fused epilogues */
enum { SEGMIF_ACT_NONE = 0, SEGMIF_ACT_RELU = 1, SEGMIF_ACT_PRELU = 2, SEGMIF_ACT_GELU = 3 };

int segmif_abi_version(void);
/* name of the device the library sees (debug aid); returns 0 and fills buf. */
int segmif_device_name(char* buf, int len);

/*
 * Implicit-GEMM convolution / linear on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *
 *   out[z][m][n] = epi( sum_k A(m,k) * Wt[z][n][k] + bias[n] )
 *   epi(y) = act(y)                      res == NULL
 *          = res[z][m][n] + act(y)       res != NULL
 *
 * A(m,k) is gathered on the fly (no im2col buffer):
 *   conv:  m = (b, oy, ox) over B*OH*OW, k = (ky, kx, c) tap-major / channel-minor,
 *          A = in[b][oy*stride - pad + ky*dil][ox*stride - pad + kx*dil][c]  (0 outside),
 *          `in` is NHWC with pixel pitch lda (lets a conv read the first Cin channels of a
 *          wider concat buffer) ;
 *   dense: A = in[m*lda + k]  (KH = KW = 1, stride 1, pad 0 — nn.Linear / 1x1 conv);
 *   dense, two sources (k < K1 from in, k >= K1 from in2 with pitch lda2) when in2 != NULL.
 * Wt is [N][Kp] row-major, Kp = K rounded up to 16 and zero-filled (see segmif_pack_conv_weight).
 * out pitch ldo (lets a conv write its channels in place into a concat buffer).
 *
 * Replaces: nn.Conv2d / nn.Linear / F.conv2d call sites on the path —
 *   core/mix_transformer.py:47,51 (fc1/fc2), :96,102,112 (q/kv/proj), :100 (sr conv), :193 (patch
 *   embed); core/segformer_head.py:23,77,80; core/model_fusion.py:135-155 (DRDB convs, with the
 *   torch.cat of :137-153 folded into ldo), :351-353,:359-360 (CrossPath linears), :1051-1065
 *   (conv1/2/21/22/3/4 with the shared PReLU of :1038).
 */
typedef struct SegmifIgemm {
  const float* in;     /* activations */
  const float* in2;    /* optional second dense source (NULL otherwise) */
  const float* wt;     /* packed weights [N][Kp] */
  const float* bias;   /* [N] or NULL */
  const float* res;    /* residual [M][ldr] or NULL */
  const float* prelu;  /* device scalar slope, used when act == SEGMIF_ACT_PRELU */
  float* out;
  int64_t M;           /* rows per z-slice: B*OH*OW */
  int32_t N, K;        /* K = KH*KW*Cin (un-padded) */
  int32_t lda, lda2, K1, ldo, ldr;
  /* conv geometry (dense: KH=KW=1, stride=1, pad=0, dil=1, H=OH, W=OW, Cin=K) */
  int32_t H, W, Cin, KH, KW, stride, pad, dil, OH, OW;
  int32_t act;
  /* batching over gridDim.z (e.g. per-image weights): element strides, 0 = shared */
  int32_t nz;
  int64_t in_zstride, in2_zstride, wt_zstride, out_zstride, res_zstride;
  int32_t tile;        /* -1 = auto; otherwise index into the tile table (bench / tests) */
  /* optional second batching level: z = zb * nz2 + z2 (e.g. zb = image, z2 = attention head) */
  int32_t nz2;         /* 0 / 1 = unused */
  int32_t ldw;         /* weight row pitch in floats; 0 = Kp (packed). Lets K / V slices of a kv tensor act as weights */
  int64_t in_zstride2, wt_zstride2, out_zstride2, res_zstride2;
  /* optional split-K scratch: segmif_igemm_workspace_floats(desc) floats (0 = none needed). Without it the
   * same problem runs un-split. Partials are summed in a fixed order: results do not depend on the split. */
  float* workspace;
  int64_t workspace_floats;
  /* optional fused LayerNorm over the output row (N must be 64, act NONE): out = LN(res + A.W^T + bias)
   * — CrossPath's norm(x + end_proj(...)) (core/model_fusion.py:359-360) in the GEMM epilogue */
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  /* optional second copy of the output as "planes" chunks (see segmif_planes_* below): conv output (B, OH, OW, N) with
   * N % 16 == 0 written as chunks [planes_chunk0, planes_chunk0 + N/16) of a planes buffer of planes_chunks chunk
   * images per batch element, geometry of segmif_planes_dims(OH, OW).  Needs the 16-byte epilogue (N, ldo multiples of 4,
   * aligned out), nz <= 1 and a convolution (not a plain 1x1 / Linear problem); conv1 of Fusion_Network3_ac hands its
   * result to the first DRDB this way.
   * planes_f16 != 0: the buffer is an f16x3 one (segmif_planes16_*), max |output| is folded into planes_amax[image]
   * (planes_amax_images == B: one range slot per batch element) or into planes_amax[0] (planes_amax_images <= 1). */
  void* planes_out;
  int32_t planes_chunks, planes_chunk0;
  int32_t planes_f16;
  uint32_t* planes_amax;   /* or NULL */
  int32_t planes_amax_images;
  /* optional ReLU mask applied LAST (after the residual): out = relu_mask[m][n] > 0 ? act(A.W^T + bias) + res : 0.
   * The DRDB backward in gather form (core/model_fusion.py:134-157 differentiated): the conv that completes a block's
   * gradient writes it through that block's ReLU mask (relu_mask = the block's forward output) straight into the slot
   * the next conv reads - no separate mask pass.  Served by the split 3x3 tile (14) and by the 16-byte epilogue of the
   * implicit-GEMM tiles (N, ldo, ld_mask multiples of 4, aligned pointers, no fused LayerNorm / planes copy / second batch
   * level; a masked problem is never split along K): CrossPath's backward writes each gradient through the ReLU of the
   * channel_proj half it belongs to (core/model_fusion.py:351-353).  Anything else is SEGMIF_EINVAL. */
  const float* relu_mask;  /* [M][ld_mask] or NULL */
  int32_t ld_mask;
  /* split 3x3 tile (14) on f16x3 arithmetic (r4, the training path): wt = segmif_conv3x3_split16_pack image, activations split
   * into half pairs in the kernel after scaling by the power of two that puts max |input| in [2^13, 2^14).  split_in_amax:
   * split_in_amax_n (1..64) device words holding the IEEE bit pattern of max |x| per channel block of the input (written by
   * split_out_amax of the producing launches or by segmif_amax_f32; the kernel takes their maximum; all zero = scale 1;
   * inf / NaN = every output NaN).  split_out_amax (either arithmetic, or NULL): split_out_amax_n words (a power of two
   * <= 64; 0 = 1) receiving the atomic max of |output| bit patterns, one atomic per workgroup spread over the words by
   * workgroup index - the consumer passes all of them as (part of) its split_in_amax. */
  int32_t split_f16;
  const uint32_t* split_in_amax;
  int32_t split_in_amax_n;
  uint32_t* split_out_amax;
  int32_t split_out_amax_n;
  /* segmif_wgrad_f32 with split_f16 != 0 (3x3 stride-1 convs, the two-team kernel): the same f16x3 arithmetic for the weight
   * gradient - split_in_amax covers the input's channel blocks, wgrad_dy_amax those of dY */
  const uint32_t* wgrad_dy_amax;
  int32_t wgrad_dy_amax_n;
  int64_t mask_zstride;    /* relu_mask with nz > 1 (per-image weights): element stride between batch slices */
} SegmifIgemm;

int segmif_igemm_f32(const SegmifIgemm* desc, void* stream);
int64_t segmif_igemm_workspace_floats(const SegmifIgemm* desc);
/* number of tile configurations and a printable name for each (bench / tests) */
int segmif_igemm_num_tiles(void);
const char* segmif_igemm_tile_name(int tile);

/*
 * Repack an OIHW conv weight (Cout, Cin, KH, KW) — or a Linear weight (N, K) with KH=KW=1 — into
 * the [N][Kp] tap-major/channel-minor layout segmif_igemm_f32 consumes. dst holds N*Kp floats,
 * Kp = ((KH*KW*Cin + 15) / 16) * 16; the tail is zero-filled.
 */
int segmif_pack_conv_weight(const float* src_oihw, float* dst, int N, int Cin, int KH, int KW, void* stream);

/*
 * Split-bf16 ("bf16x6") form of a packed 3x3 weight, for tile 14 of segmif_igemm_f32: every fp32
 * weight becomes three bf16 planes (w = w0 + w1 + w2, 24 significand bits), laid out
 * [n-tile][16-channel chunk][tap][n][plane][16] so a workgroup streams one chunk's nine taps as one
 * contiguous block.  The kernel splits the activations the same way on the fly and sums the six
 * products >= 2^-16 on v_mfma_f32_32x32x16_bf16: fp32-class accuracy at 2.7x the fp32 MFMA rate
 * (DRDB convs, core/model_fusion.py:121-157).  `packed` is the segmif_pack_conv_weight image
 * (row pitch ldw floats); `out` holds segmif_conv3x3_split_weight_bytes(N, Cin) bytes.  Cin % 16 == 0.
 * With tile = 14 the descriptor's `wt` points at this image; geometry limits are those of the halo
 * tiles (3x3, stride 1, dilation 1|2, pad = dilation, N <= 256), anything else returns SEGMIF_EINVAL.
 */
int64_t segmif_conv3x3_split_weight_bytes(int N, int Cin);
int segmif_conv3x3_split_pack(const float* packed, int N, int Cin, int ldw, void* out, void* stream);
/* f16x3 form of the same image (r4): rows scaled by a power of two, planes W0 | Wl | 2^-11 W0 as halves, then one float per
 * padded output channel (the scale's inverse).  Re-packed per call on the training path (weights change every step). */
int64_t segmif_conv3x3_split16_weight_bytes(int N, int Cin);
int segmif_conv3x3_split16_pack(const float* packed, int N, int Cin, int ldw, void* out, void* stream);
/* max |x| of a rows view (rows x C floats, pitch ld; C, ld multiples of 4, 16-byte aligned) folded into slots[0 .. nslots)
 * (nslots a power of two <= 64: one atomic per block, spread over the words by block index; the maximum over the words
 * is the tensor's) as IEEE bit patterns by atomic integer max (a NaN stays on top); the caller zeroes the words first */
int segmif_amax_f32(const float* x, int64_t rows, int C, int ld, uint32_t* slots, int nslots, void* stream);

/*
 * Dense GEMM with bf16x6 arithmetic (csrc/gemm_split.hip): out = res + act(A W^T + bias), A (M, K) fp32 rows (pitch lda,
 * 16-byte aligned), K % 32 == 0, any M / N.  `w` is the segmif_gemm_split_pack image of the (N, K) weight (row pitch ldw
 * floats), segmif_gemm_split_weight_bytes(N, K) bytes.  fp32-class accuracy (three bf16 terms per operand, six products)
 * at 2.7x the fp32 MFMA rate: the MiT encoder's and SegFormer head's nn.Linear layers
 * (core/mix_transformer.py:46-53, :94-115; core/segformer_head.py:23).
 */
typedef struct SegmifGemmSplit {
  const float* a; const void* w; const float* bias; const float* res; const float* prelu; float* out;
  int64_t M; int32_t N, K, lda, ldo, ldr, act;
  /* patch mode (patch_k > 0): a k x k convolution with stride patch_st and zero padding patch_pad as this GEMM with no gather
   * pass - Attention's spatial-reduction conv (core/mix_transformer.py:73-75, :98-101; k = stride = sr, pad 0) and the
   * overlapping patch embeds of stages 2-4 (:171-172; k 3, stride 2, pad 1).  `a` is a dense NHWC image batch
   * (B, patch_H, patch_W, C) with lda = C, C % 32 == 0; row m = (b, oy, ox) of A is the patch at (st oy - pad, st ox - pad) in
   * (ky, kx, c) order, taps outside the image read as zeros; K = k * k * C, M = B * OH * OW with OH = (H + 2 pad - k) / st + 1;
   * `w` is the image of the conv weight in the same order (segmif_pack_conv_weight's [N][Kp] rows).  0 = plain rows. */
  int32_t patch_k, patch_st, patch_pad, patch_H, patch_W;
} SegmifGemmSplit;
int64_t segmif_gemm_split_weight_bytes(int N, int K);
int segmif_gemm_split_pack(const float* w, int N, int K, int ldw, void* out, void* stream);
int segmif_gemm_split_f32(const SegmifGemmSplit* desc, void* stream);
/*
 * (r5) The f16x3 GEMM with BOTH operands pre-split (csrc/gemm_pairs.hip): out = res + act(A W^T + bias) where A arrives in
 * PAIRS format - a row of K values is K / 16 groups of 64 bytes, [16 hi halves | 16 lo halves], x = hi + 2^-11 lo - written
 * once by its producer (segmif_layernorm_pairs_f32, segmif_dwconv3x3_gelu_pairs_f32, the attention kernel's pairs output,
 * segmif_pairs_from_f32) with the same byte count as fp32, and W is the two-plane image of segmif_gemm_pairs_pack.  Both go
 * HBM -> LDS by LDS-DMA; the kernel does no vector arithmetic on its operands.  Replaces the aten addmm of nn.Linear in
 * core/mix_transformer.py:46-53 (fc1 / fc2), :94-115 (q / kv / proj) and, in patch mode, Attention's spatial-reduction conv
 * (:73-75, :98-101) for the tall problems of stages 2-4.  Range: the PRODUCER of A folds max |x| into the guard slots.
 *   lda_bytes: row pitch of A in bytes (>= 4 K, a multiple of 16); patch mode (patch_k > 0): A is a dense NHWC pairs image
 *   (B, patch_H, patch_W, patch_C), patch_C % 16 == 0, row m = (b, oy, ox) the k x k patch at (st oy - pad, st ox - pad) in
 *   (ky, kx, c) order, taps outside the image read as zeros, K = k k C;  tile_rows: 0 = chosen by size, 128 | 256 forces one.
 */
typedef struct SegmifGemmPairs {
  const void* a; const void* w; const float* bias; const float* res; const float* prelu; float* out;
  int64_t M; int64_t lda_bytes;
  int32_t N, K, ldo, ldr, act;
  int32_t patch_k, patch_st, patch_pad, patch_H, patch_W, patch_C;
  int32_t tile_rows;
} SegmifGemmPairs;
int64_t segmif_gemm_pairs_weight_bytes(int N, int K);
int segmif_gemm_pairs_pack(const float* w, int N, int K, int ldw, void* out, void* stream);
int segmif_gemm_pairs_f32(const SegmifGemmPairs* desc, void* stream);
/* fp32 rows (rows x C, pitch ldx floats) <-> pairs rows (pitch in bytes); C % 16 == 0.  amax (or NULL): range slots receiving
 * max |x| (planes16 patterns), amax_images whole images of rows / amax_images rows each. */
int segmif_pairs_from_f32(const float* x, int64_t ldx, void* y, int64_t ldy_bytes, int64_t rows, int C, uint32_t* amax,
                          int amax_images, void* stream);
int segmif_pairs_to_f32(const void* x, int64_t ldx_bytes, float* y, int64_t ldy, int64_t rows, int C, void* stream);
/*
 * (r5) Backward of the spatial-reduction attention without the score matrix (csrc/attention_bwd.hip; the autograd of
 * core/mix_transformer.py:107-111 - softmax(q k^T scale) v): q, out, dout, dq are (B, N, heads * 64) rows, k / v the two
 * halves of a (B, Nk, ldkv >= 2 heads 64) tensor, dkv receives (B, Nk, 2 heads 64) = (dk | dv).  Scores are recomputed per
 * 32 x 32 tile on the exact-fp32 matrix pipe; dk / dv are formed per chunk of queries (segmif_sr_attention_bwd_chunks of them:
 * about one round of workgroups on the chip) and summed in a fixed order
 * (deterministic).  workspace: segmif_sr_attention_bwd_workspace_floats floats (row statistics + the chunk partials).
 */
int segmif_sr_attention_bwd_chunks(int B, int heads, int N, int Nk);
int64_t segmif_sr_attention_bwd_workspace_floats(int B, int heads, int N, int Nk, int C);
int segmif_sr_attention_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* dout, float* dq,
                                float* dkv, float* workspace, int B, int heads, int N, int Nk, int hd, int ldkv, float scale,
                                void* stream);

/* Producers of PAIRS rows (r5): LayerNorm (core/mix_transformer.py:152-155 norm1 / norm2, :113 the norm after the sr conv),
 * depthwise 3x3 + bias + GELU (:46-53, :376-387) and the fused attention kernel (:107-111) with their result written as half
 * pairs - 16-channel groups of [16 hi | 16 lo], the fp32 row's byte count - and max |y| folded into the f16x3 range slot of the
 * row's image (amax: amax_images words, rows / amax_images rows per image; NULL = no report).  Same arithmetic as
 * segmif_layernorm_f32 / segmif_dwconv3x3_gelu_f32 / segmif_sr_attention_split16_f32; C % 16 == 0.  The LayerNorm's waves are
 * short and numerous: it takes amax_sub (a power of two) consecutive ROWS of slots, amax_pitch words apart (>= amax_images: the
 * guard's own image count, which differs from amax_images when the launch reports to column 0 only), and spreads its
 * reports over them (a slot access serialises on its memory channel). */
int segmif_layernorm_pairs_f32(const float* x, const float* gamma, const float* beta, void* y, int64_t rows, int C, int ldx,
                               int ldy, float eps, uint32_t* amax, int amax_images, int amax_sub, int amax_pitch, void* stream);
int segmif_dwconv3x3_gelu_pairs_f32(const float* x, const float* w9, const float* bias, void* y, int B, int H, int W, int C,
                                    uint32_t* amax, int amax_images, void* stream);
int segmif_sr_attention_split16_pairs_f32(const float* q, const float* k, const float* v, void* out, void* workspace, int B,
                                          int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale,
                                          uint32_t* amax, uint32_t* out_amax, int amax_images, void* stream);

/* The same GEMM with f16x3 arithmetic (half pairs x three products, see segmif_planes16_* below): A is split in the kernel,
 * max |A| of the staged rows is folded into the range slots amax[0 .. amax_images) (NULL = off): the M rows are
 * amax_images whole images of M / amax_images rows, image i reports to amax[i] (amax_images = 1: one slot) - the caller
 * must re-run the images whose slot left [2^-13, 65504) on segmif_gemm_split_f32; `w` is the segmif_gemm_split16_pack
 * image (the bf16 image's layout with scaled half planes, then one float 2^-e(n) per padded output column). */
int64_t segmif_gemm_split16_weight_bytes(int N, int K);
int segmif_gemm_split16_pack(const float* w, int N, int K, int ldw, void* out, void* stream);
int segmif_gemm_split16_f32(const SegmifGemmSplit* desc, uint32_t* amax, int amax_images, void* stream);

/*
 * "Planes" activations: the bf16x6 operand split done ONCE by the producer instead of in every
 * consumer (csrc/conv3x3_planes.hip).  A planes buffer holds `chunks` 16-channel chunk images per batch
 * element, [b][chunk][Hp][Wp][plane 0..2][16] bf16 (x = p0 + p1 + p2, 24 significand bits), with
 * Hp = ceil(H/8)*8 + 4, Wp = ceil(W/32)*32 + 4 (segmif_planes_dims): a zero border of 2 pixels plus the
 * round-up to whole 8 x 32 patches, which segmif_planes_zero_border clears once per buffer (kernels only
 * ever write the H x W interior).  Position j of a chunk holds channel 16*chunk + (j&3) + 4*(j>>3) +
 * 8*((j>>2)&1) (the MFMA accumulator's hand-out order).  segmif_planes_from_f32 converts `nconv` chunks
 * (16*nconv leading channels of the fp32 rows x, pixel pitch ldx floats, 16-byte aligned) into chunks
 * [chunk0, chunk0 + nconv).  segmif_planes_pack_weight turns a segmif_pack_conv_weight image (row pitch
 * ldw floats; taps = 9 for a 3x3 conv, 1 for a 1x1 conv / Linear; N % 32 == 0, Cin % 16 == 0) into the
 * matching split weight image of segmif_planes_weight_bytes(N, Cin, taps) bytes.
 *
 * segmif_conv3x3_planes_bf16x6: 3x3 stride-1 "same" convolution (dilation 1 | 2) with 32 output channels
 * over the first `cin` channels of a planes buffer: bias + act (NONE | RELU | PRELU), written as two more
 * planes chunks (planes_out / out_chunk0; may be the input buffer, chunks >= cin/16) and / or as fp32 rows
 * (out, pitch ldo).  Six bf16 MFMA products per fp32-equivalent MAC: fp32-class accuracy.  With w1 != NULL
 * the kernel also evaluates   out1 = res + act1( W1 . [in(cin) | act(conv)(32)] + bias1 )   — a 1x1 conv with
 * 64 outputs over the input channels and the conv's own result (w1: segmif_planes_pack_weight(N = 64,
 * Cin = cin + 32, taps = 1)) — i.e. a whole DRDB tail (core/model_fusion.py:153-157: Dcov5, torch.cat,
 * conv 224 -> 64, ReLU, residual) in one launch, the concat never reaching HBM.
 * Replaces core/model_fusion.py:135-157 in inference.
 */
typedef struct SegmifConvPlanes {
  const void* planes_in;
  void* planes_out;    /* or NULL */
  const void* wt;      /* segmif_planes_pack_weight(N = 32, Cin = cin, taps = 9) */
  const float* bias;   /* [32] or NULL, 16-byte aligned */
  const float* prelu;
  float* out;          /* optional fp32 rows [B*H*W][ldo] */
  int32_t ldo;
  int32_t B, H, W, cin, dil;
  int32_t in_chunks;   /* chunk images per batch element in planes_in */
  int32_t out_chunks, out_chunk0;
  int32_t act;
  const void* w1;      /* fused 1x1 tail (NULL = off) */
  const float* bias1;  /* [64] or NULL */
  const float* res;    /* [B*H*W][ldr] or NULL */
  float* out1;         /* [B*H*W][ldo1], 64 channels */
  int32_t ldr, ldo1, act1;
  /* f16x3 fused tail with res == NULL: != 0 takes the residual from the conv's OWN input, chunks 0..3 of planes_in
   * (x = hi + 2^-11 lo: the DRDB's input as its producer split it, 23 significand bits) - the fp32 copy of that tensor then
   * needs no writer and no reader (core/model_fusion.py:157: out = conv(cat) + x).  Ignored by the bf16x6 entry point. */
  int32_t res_from_planes;
} SegmifConvPlanes;

int segmif_planes_dims(int H, int W, int* Hp, int* Wp);
int64_t segmif_planes_bytes(int B, int H, int W, int chunks);
int segmif_planes_zero_border(void* planes, int B, int H, int W, int chunks, void* stream);
int segmif_planes_from_f32(const float* x, int ldx, void* planes, int B, int H, int W, int chunks, int chunk0, int nconv,
                           void* stream);
int64_t segmif_planes_weight_bytes(int N, int Cin, int taps);
int segmif_planes_pack_weight(const float* packed, int N, int Cin, int taps, int ldw, void* out, void* stream);
int segmif_conv3x3_planes_bf16x6(const SegmifConvPlanes* desc, void* stream);

/*
 * "f16x3" planes: the same buffers, kernels and descriptor with HALF-precision pairs instead of bf16 triples -
 * [b][chunk][Hp][Wp][plane 0..1][16] halves, x = p0 + 2^-11 p1 (p0 = RN16(x), p1 = RN16(2^11 (x - p0)): 23 significand
 * bits + rounding, one bit short of fp32) - and weight rows scaled by a power of two into the half's range and stored as
 * W0 | W - W0 | 2^-11 W0 followed by the N row factors (segmif_planes16_weight_bytes = the bf16 image + 4 N bytes).
 * THREE MFMA products per fp32-equivalent MAC instead of six and 2/3 of the activation bytes, at the error level of
 * the bf16x6 kernels (tests/test_gpu_kernels.py) - provided the activations lie in the half's exponent range: every
 * producer (segmif_planes16_from_f32, the conv's own planes / fused-tail output) folds max |x| of what it wrote into
 * a range slot (atomic max on the IEEE bit pattern of a non-negative float - NaN compares above inf; NULL = off):
 * amax[b] for batch element b when amax_images == B, amax[0] when amax_images == 1.  The caller must read the slots
 * back and re-run, on the bf16x6 entry points, the images whose value left [2^-13, 65504) (0 = an all-zero tensor is
 * fine): above, a half overflows; below, the pair keeps fewer than 23 bits.  segmif_amd.ops.run_guarded does this once
 * per pair forward, per image (ops.Planes16Guard).
 * Same geometry (segmif_planes_dims), channel order and argument rules as the bf16 entry points above.
 */
int64_t segmif_planes16_bytes(int B, int H, int W, int chunks);
int segmif_planes16_zero_border(void* planes, int B, int H, int W, int chunks, void* stream);
int segmif_planes16_from_f32(const float* x, int ldx, void* planes, int B, int H, int W, int chunks, int chunk0, int nconv,
                             uint32_t* amax, int amax_images, void* stream);
/* (r6) The fusion net's first convs, conv1_ir / conv1_vis (core/model_fusion.py:1029-1030, applied at :1051-1053 / :1055-1057 to
 * ir[:, 0:1] / vis[:, 0:1]): 3 x 3 'same' conv from ONE channel to 64 + bias + activation (SEGMIF_ACT_*; prelu = the shared scalar
 * slope) as a store-bound stencil.  x: (B, H, W) fp32, w: the raw (64, 1, 3, 3) weight, bias: 64 floats or NULL.  Outputs, at least
 * one: `planes` - an f16x3 planes buffer of `chunks` chunks per image receiving chunks [chunk0, chunk0 + 4) (half pairs, max |y|
 * folded into the range slots `amax`: amax_images = B words, one per image, or 1) - and / or `out`, (B, H, W, 64) fp32 rows with
 * pixel pitch ldo.  Replaces the scalar-gather implicit GEMM for Cin = 1. */
int segmif_conv3x3_c1_f16x3(const float* x, const float* w, const float* bias, const float* prelu, int act, void* planes,
                            int chunks, int chunk0, float* out, int ldo, int B, int H, int W, uint32_t* amax, int amax_images,
                            void* stream);
int64_t segmif_planes16_weight_bytes(int N, int Cin, int taps);
int segmif_planes16_pack_weight(const float* packed, int N, int Cin, int taps, int ldw, void* out, void* stream);
int segmif_conv3x3_planes_f16x3(const SegmifConvPlanes* desc, uint32_t* amax, int amax_images, void* stream);

/*
 * Weight gradient of the same problem: dW[n][k] = sum_m dY[m][n] * A(m,k), contraction over rows on
 * fp32 MFMA, deterministic two-pass reduction (per-chunk partials, then an fp64 sum) written in the
 * parameter's own layout: dw[n*dw_sn + k'*...] — OIHW for convs, (N, K) for linears; for dense
 * problems the output element (n, k) goes to dw[n*dw_sn + k*dw_sk] (dw_sn = K, dw_sk = 1 normally;
 * other strides let attention backward write dK / dV in place).  `desc` describes the FORWARD problem
 * (in, lda, geometry, M, N, K; nz / in_zstride / out_zstride reused as batch count, input and dW batch
 * strides).  workspace: segmif_wgrad_workspace_size(M, N, K) * max(nz,1) floats.  accumulate: dw += .
 * Backward of every nn.Conv2d / nn.Linear listed under segmif_igemm_f32.
 */
int64_t segmif_wgrad_workspace_size(int64_t M, int N, int K);
int segmif_wgrad_f32(const SegmifIgemm* desc, const float* dy, int ldy, int64_t dy_zstride, float* dw,
                     int64_t dw_sn, int64_t dw_sk, float* dbias /* optional [N]: sum_m dY, fused */,
                     float* workspace, int accumulate, void* stream);
/* The same for a dense problem batched over TWO levels (batch z = z1 * desc->nz2 + z2; e.g. image x attention head):
 * input / dY / dW offsets z1 * {in_zstride, dy_zstride, out_zstride} + z2 * {in_zstride2, dy_zstride2, out_zstride2}.
 * One launch forms dK^T (or dV^T) of every head of every image in the spatial-reduction attention's backward
 * (core/mix_transformer.py:107-111 under autograd).  workspace: segmif_wgrad_workspace_size(M, N, K) * nz * nz2 floats. */
int segmif_wgrad_batched2_f32(const SegmifIgemm* desc, const float* dy, int ldy, int64_t dy_zstride, int64_t dy_zstride2,
                              float* dw, int64_t dw_sn, int64_t dw_sk, float* workspace, int accumulate, void* stream);
/* column sums of a rows x N matrix (bias gradients), two-pass, fp64 accumulation.
 * workspace: segmif_colsum_blocks(rows) * N doubles. */
int segmif_colsum_blocks(int64_t rows);
int segmif_colsum_f32(const float* x, float* out, double* workspace, int64_t rows, int N, int ldx,
                      int accumulate, void* stream);
/* dx = dy * act'(.): ref = activation OUTPUT for ReLU / PReLU (slope > 0), PRE-activation for GELU. */
int segmif_act_bwd_f32(const float* dy, const float* ref, float* dx, int64_t rows, int C, int ldy, int ldr,
                       int ldx, int act, const float* slope, void* stream);
/* the same with the activation output given as ref - ref2 (r4: a DRDB keeps x and x + relu(.) - the mask source is their
 * difference, formed in registers instead of by a full-resolution subtraction pass; core/model_fusion.py:153-157) */
int segmif_act_bwd2_f32(const float* dy, const float* ref, const float* ref2, float* dx, int64_t rows, int C, int ldy,
                        int ldr, int ldr2, int ldx, int act, const float* slope, void* stream);

/*
 * LayerNorm over the last dimension: y = (x - mean) / sqrt(var + eps) * gamma + beta,
 * biased variance, two-pass in registers.  rows x C, pitches ldx/ldy. C % 4 == 0, C <= 1024.
 * Replaces nn.LayerNorm at core/mix_transformer.py:101,152-153,196,320,328,336,344 and
 * core/model_fusion.py:359-360 (eps 1e-6 / 1e-5: SURVEY F6).
 */
int segmif_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                         int64_t rows, int C, int ldx, int ldy, float eps, void* stream);
/* Residual form of the training path (r4; core/mix_transformer.py:171-177: x = x + drop_path(f(norm(x)))):
 *   sum = x + scale[row / rows_per_image] * branch  (scale NULL = 1: the per-sample stochastic-depth factor mask / keep)
 *   y   = LayerNorm(sum)
 * in one pass - the residual add, the DropPath multiply and the next LayerNorm were three. */
int segmif_add_layernorm_f32(const float* x, const float* branch, const float* scale, int64_t rows_per_image,
                             const float* gamma, const float* beta, float* sum, float* y, int64_t rows, int C, int ldx,
                             int ldb, int lds, int ldy, float eps, void* stream);

/*
 * Mix-FFN middle: depthwise 3x3 (pad 1) + bias + exact-erf GELU on tokens viewed as an
 * NHWC image (B, H, W, C); w9 is the depthwise weight repacked to [9][C].  C % 4 == 0.
 * Replaces DWConv.forward + nn.GELU: core/mix_transformer.py:48-49, :381-387.
 */
int segmif_dwconv3x3_gelu_f32(const float* x, const float* w9, const float* bias, float* y,
                              int B, int H, int W, int C, void* stream);
/* DWConv.forward called on its own (core/mix_transformer.py:381-387): depthwise 3x3 + bias, no activation. */
int segmif_dwconv3x3_bias_f32(const float* x, const float* w9, const float* bias, float* y,
                              int B, int H, int W, int C, void* stream);

/*
 * 3x3 stride-1 "same" convolution from 32 channels to ONE + bias + act (NONE | RELU | PRELU): conv22 of Fusion_Network3_ac
 * (core/model_fusion.py:1042, :1065), a bandwidth-bound stencil on the vector ALU (the matrix-pipe kernels pad the single
 * output channel to a 32-wide tile).  x: (B, H, W, >= 32) rows with pixel pitch ldx (16-byte aligned, ldx % 4 == 0);
 * wt: [9][32] tap-major = the segmif_pack_conv_weight image of the (1, 32, 3, 3) weight; y: (B, H, W) dense.
 */
int segmif_conv3x3_c32to1_f32(const float* x, int ldx, const float* wt, const float* bias, const float* prelu, int act, float* y,
                              int B, int H, int W, void* stream);

/*
 * Bilinear resize, align_corners=False, NHWC: (B, IH, IW, C) -> (B, OH, OW, C) written with
 * pixel pitch ldo at channel offset 0 of `y` (lets the SegFormer head write straight into its
 * concat buffer).  16-byte vector path when C, ldx, ldo are multiples of 4; scalar otherwise (9-class logits).
 * Replaces F.interpolate at core/mix_transformer.py:364-373, core/segformer_head.py:67-73,
 * core/model_fusion.py:1095 and test_segmentation.py:170.
 */
int segmif_bilinear_nhwc_f32(const float* x, float* y, int B, int IH, int IW, int OH, int OW, int C,
                             int ldx, int ldo, void* stream);
/* out = act(base + bias + sum of up to three bilinear resizes (align_corners=False) to OH x OW), all NHWC with C
 * channels (C % 4 == 0; the sources are dense, base / out may be pitched); act NONE or RELU; NULL base / bias /
 * x_i are skipped.  The SegFormer head with linear_fuse applied per scale before the resize
 * (segformer_head.py:67-77 commuted: a 1x1 conv commutes with a bilinear resize) ends in this kernel. */
int segmif_upsum_act_nhwc_f32(const float* base, int ldb, const float* x0, int ih0, int iw0, const float* x1, int ih1,
                              int iw1, const float* x2, int ih2, int iw2, const float* bias, float* out, int ldo, int B,
                              int OH, int OW, int C, int act, void* stream);

/*
 * Fused spatial-reduction attention: O = softmax(Q K^T * scale) V per (batch, head), never
 * materialising the N x Nk score matrix.  q: (B, N, ldq) with head h at columns [h*hd, (h+1)*hd);
 * k and v: (B, Nk, ldkv) (the kv Linear output: k = kv, v = kv + C); out: (B, N, ldo).
 * hd in {32, 64}.  QK^T and PV run on fp32 MFMA, softmax is per-lane (online, exact expf).
 * Replaces core/mix_transformer.py:107-111.
 */
int segmif_sr_attention_f32(const float* q, const float* k, const float* v, float* out,
                            int B, int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo,
                            float scale, void* stream);
/* The same function on the bf16 matrix pipe (csrc/attention_split.hip): operands split three ways into bf16, six MFMA
 * products per fp32 product (fp32-class, like segmif_conv3x3_planes_bf16x6).  hd == 64 only.  K and V are split once per
 * call into `workspace` (segmif_sr_attention_split_workspace(B, heads, Nk) bytes, 16-byte aligned, device memory),
 * which the attention workgroups then read by LDS-DMA. */
int64_t segmif_sr_attention_split_workspace(int B, int heads, int Nk);
int segmif_sr_attention_split_f32(const float* q, const float* k, const float* v, float* out, void* workspace,
                                  int B, int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo,
                                  float scale, void* stream);
/* The same with f16x3 arithmetic (half pairs x three f16 MFMA products, see segmif_planes16_*): K and V^T are packed as
 * scaled half planes with one power-of-two scale per (key tile, head, image), Q and the probabilities are split in registers.
 * amax[image] (amax_images == B) or amax[0] (1) receives max |scale log2(e) Q| (NULL = off): the caller re-runs images whose
 * slot left [2^-13, 65504) on segmif_sr_attention_split_f32.  Same workspace size. */
int segmif_sr_attention_split16_f32(const float* q, const float* k, const float* v, float* out, void* workspace, int B,
                                    int heads, int N, int Nk, int hd, int ldq, int ldkv, int ldo, float scale,
                                    uint32_t* amax, int amax_images, void* stream);

/*
 * Linear ("efficient") cross attention context, step 1: per (batch, head) partial sums of
 * K^T V over row blocks.  kv: (B, N, 2*C), C = heads*d: k = cols [0,C), v = [C,2C).  heads = d = 8 (C = 64: the configuration
 * Fusion_Network3_ac instantiates) runs the tuned kernels; (r6) any other geometry with C <= 64, d <= 8 - the ablation networks'
 * dim-32 modules, core/model_fusion.py:363-429, :626-661 - a generic one with the same arithmetic.
 * partial: (B, nblk, heads*d*d) DOUBLES with nblk = segmif_linattn_num_blocks(N) (fp32 inside a
 * 32-row run, fp64 across runs: the sum feeds a softmax).
 * Step 2 (segmif_linattn_fold_f32) reduces the partials in fp64, applies
 * softmax over the k-index (dim=-2) of (K^T V)*scale, and folds the block-diagonal context into
 * the following end_proj weight:  Weff[b][n][kofs + h*d + i] = sum_j ctx[b][h][i][j] * Wend[n][wofs + h*d + j].
 * Replaces core/model_fusion.py:281-286 and :316-326 (ctx and q@ctx) together with the cat +
 * end_proj of :357-360, which becomes one dense igemm over [y3 | u_i] with per-image weights.
 */
int segmif_linattn_num_blocks(int64_t N);
int segmif_linattn_partial_f32(const float* kv, double* partial, int B, int64_t N, int heads, int d,
                               int ldkv, void* stream);
/* Fused form of (kv Linear without bias) + segmif_linattn_partial_f32: y is the (B, N, 64) input of the
 * kv projection (pitch ldy), wkv its raw (128, 64) row-major weight; kv is never written to HBM. */
int segmif_linattn_kvpartial_f32(const float* y, const float* wkv, double* partial, int B, int64_t N,
                                 int heads, int d, int ldy, void* stream);
int segmif_linattn_fold_f32(const double* partial, const float* wend, float* weff, int B, int nblk,
                            int heads, int d, int Nout, int ldw, int wofs, int ldweff, int kofs,
                            float scale, void* stream);
/* (r6) Backward of segmif_linattn_fold_f32 (training path, heads = d = 8, C = 64): ktv (B, 8, 8, 8) = the per-head K^T V the fold
 * read (fp64; the sum of its partials), wend / wofs / kofs / scale as in the forward, dweff (B, Nout, ldweff) the gradient of the folded
 * weight.  Writes dktv (B, 8, 8, 8) fp64 - through the softmax over the k index - and, per image, this fold's share of d end_proj:
 * dwend_part[b][n][wofs + c] (pitch ldp floats; the caller sums over b).  Replaces the torch softmax / einsum / cat of the training
 * path's context fold (core/model_fusion.py:281-286, :316-326, :357-360 under autograd). */
int segmif_linattn_fold_bwd_f32(const double* ktv, const float* wend, int ldw, int wofs, const float* dweff, int ldweff, int kofs,
                                float scale, double* dktv, float* dwend_part, int ldp, int B, int Nout, void* stream);

/*
 * CrossPath in inference without its 128-wide intermediates (csrc/crosspath.hip; core/model_fusion.py:329-361).
 *  - segmif_crosspath_gram_f32: per-image Gram matrix G = sum_n y_n y_n^T of y_n = ReLU(W x_n + bias), W: 64 rows x 64
 *    (row-major, the needed half of a channel_proj weight), x: (B, N, 64) rows with pixel pitch ldx.  Written as
 *    segmif_crosspath_gram_blocks(N) partial matrices per image, 3072 doubles each (tiles (0,0), (0,1), (1,1) of the
 *    symmetric 64 x 64 matrix, 32 x 32 row-major); fp32 inside a 32-pixel run, fp64 across runs.
 *  - segmif_crosspath_fold_f32: K^T V = Wk G Wv^T per head from the Gram partials ([Wk; Wv] = the raw (128, 64) kv
 *    weight), softmax over the k index of (K^T V) * scale, folded into end_proj exactly like segmif_linattn_fold_f32.
 *    cond (or NULL): B words, one per image; the launch raises cond[b] (integer atomic max over fp32 bit patterns) to
 *    kappa = max over its 64 softmax columns of max_i |p_i (dL_i - sum_k p_k dL_k)|, dL the derivative of the logits with
 *    respect to a relative perturbation G -> G o (1 + eps S) of the Gram entries under one fixed symmetric +-1 pattern S - how
 *    far this softmax moves per unit RELATIVE error of its input.  The host's f16x3 guard reads it beside the range slots: an
 *    image whose kappa passes a calibrated bound is computed again with the 3x3 convs in exact fp32 (round 5;
 *    core/model_fusion.py:281-286, :316-326).
 *  - segmif_crosspath_tail_f32: out = LayerNorm_64(x_i + Weff_b . [ReLU(W3 x_3 + b3) | ReLU(Wi x_i + bi)] + bend):
 *    channel_proj halves, the context-folded end_proj (Weff: (B, 64, 128)), residual and norm in one pass over the
 *    tokens; optionally also emits out as planes chunks 0..3 (conv3x3_planes format, H * W == N) for the next DRDB.
 * Replaces core/model_fusion.py:351-360 together with :281-286 / :316-326 (the three kv Linears and K^T V).
 */
typedef struct SegmifCrossTail {
  const float* x3; const float* xi;     /* (B, N, 64) rows, pitches ld3 / ldi */
  const float* w3; const float* b3;     /* 64 x 64 rows of channel_proj3 (y half) and their bias (or NULL) */
  const float* wi; const float* bi;     /* 64 x 64 rows of channel_proj_i (u half) and their bias (or NULL) */
  const float* weff;                    /* (B, 64, 128) */
  const float* bend;                    /* end_proj bias [64] or NULL */
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  float* out;                           /* may be NULL when planes_out is the only consumer (no fp32 copy is written) */
  int32_t ld3, ldi, ldo;
  int32_t B; int64_t N;
  void* planes_out; int32_t H, W, planes_chunks;   /* optional planes copy of out (NULL = off) */
  int32_t planes_f16;                               /* != 0: an f16x3 buffer (segmif_planes16_*) */
  uint32_t* planes_amax;                            /* f16x3: range slot(s) for max |out|, or NULL */
  int32_t planes_amax_images;                       /* == B: planes_amax[image]; <= 1: planes_amax[0] */
  /* (r5) x3_ih > 0: x_3 is NOT a full-resolution tensor.  x3 then points at the (B, x3_ih * x3_iw, ld3) LOW-resolution map
   * of channel_proj3's y half ALREADY APPLIED (w3 x + b3, no ReLU; w3 / b3 here are ignored), and the kernel resizes it to
   * H x W (bilinear, align_corners = False: segmif_bilinear_nhwc_f32's arithmetic) row by row as it consumes it - a resize
   * is a convex combination per channel, so it commutes with the Linear.  H, W must be set (H * W == N).  Replaces
   * F.interpolate at core/mix_transformer.py:364-373 for this consumer: the resized tensor never exists. */
  int32_t x3_ih, x3_iw;
  /* (r6) arith_f16 != 0: the kernel's own contractions on f16x3 operands (half pairs x power-of-two-scaled half planes, three MFMA
   * products per MAC instead of six bf16 ones) - for the lazy (x3_ih > 0), planes-only (out == NULL), f16x3-planes launch, inside a
   * guarded scope: arith_amax = the range slot(s) receiving max |x_i| and max |interpolated y_3| (arith_amax_images == B: one per
   * image; <= 1: arith_amax[0]). */
  int32_t arith_f16; uint32_t* arith_amax; int32_t arith_amax_images;
} SegmifCrossTail;

int segmif_crosspath_gram_blocks(int64_t N);
int segmif_crosspath_gram_f32(const float* x, int ldx, const float* w, const float* bias, double* partial, int B, int64_t N,
                              void* stream);
/* (r5) the Gram partials of relu(resize(s_low -> H x W)) for s_low = the (B, ih * iw, lds) low-resolution map of a channel_proj
 * half ALREADY APPLIED (W x + c, no ReLU): same (B, segmif_crosspath_gram_blocks(H * W), 3072) fp64 output as
 * segmif_crosspath_gram_f32 on the resized tensor (bilinear, align_corners = False), which is never formed.  Needs
 * W % 4 == 0 and an enlargement by three or more (3 iw <= W), else SEGMIF_EINVAL (the caller resizes first).  Replaces
 * F.interpolate (core/mix_transformer.py:364-373) + channel_proj3 (core/model_fusion.py:353) for cross_attn's context. */
int segmif_crosspath_gram_lazy_f32(const float* s_low, int lds, int ih, int iw, int H, int W, double* partial, int B, void* stream);
/* (r6) total (B, 3072) = the sum over an image's nblk Gram partials, in a fixed order, by 12 workgroups per image; the result is a
 * one-partial image (nblk = 1) for segmif_crosspath_fold_f32, whose single workgroup per image otherwise spends its launch pulling
 * 1.5 - 3 MB of partials through one CU. */
int segmif_crosspath_gram_sum_f64(const double* partial, int nblk, double* total, int B, void* stream);
int segmif_crosspath_fold_f32(const double* partial, int nblk, const float* wkv, const float* wend, float* weff, int B,
                              int Nout, int ldw, int wofs, int ldweff, int kofs, float scale, uint32_t* cond, void* stream);
int segmif_crosspath_tail_f32(const SegmifCrossTail* desc, void* stream);

/*
 * Pointwise helpers (bandwidth-bound, NCHW <-> NHWC at the module boundary).
 *  - segmif_seg_normalize: (x*255 - mean_c)/std_c on a (B,3,H,W) NCHW image -> NHWC (B,H,W,3)
 *    (core/model_fusion.py:1083-1085).
 *  - segmif_nchw_to_nhwc / segmif_nhwc_to_nchw: layout transposes (B, C, HW) <-> (B, HW, C).
 *  - segmif_fuse_ycrcb: fused RGB = clamp01(YCrCb2RGB([Yf, Cr(vis), Cb(vis)])) with
 *    vis NCHW RGB, yf (B,1,H,W); out NCHW (core/model_fusion.py:69-111 + test_fusion.py:102-111).
 */
int segmif_seg_normalize_f32(const float* x_nchw, float* y_nhwc, int B, int H, int W, void* stream);
int segmif_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int64_t HW, int ldo, void* stream);
int segmif_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int64_t HW, int ldx, void* stream);
int segmif_fuse_ycrcb_f32(const float* vis_nchw, const float* yf, float* out_nchw, int B, int64_t HW, void* stream);
/* (r6) two-source pointwise pass over rows views (rows x C, pitches in floats, C % 4 == 0, 16-byte aligned) - the glue of the
 * reference's ablation networks: mode 0 y = a + b (Fusion_Network3_Add, core/model_fusion.py:745-752); mode 1
 * y = silu(a) + silu(b) (Fusion_Network3_Average's att_i(x) + att_j(seg): AttentionModule ends in sigmoid(z) * z, :769-770,
 * :807-813); mode 2 y = silu(a) (AttentionModule alone; b may be NULL) */
int segmif_pointwise2_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int64_t rows, int C, int mode,
                          void* stream);
/* RGB2YCrCb / YCrCb2RGB themselves (core/model_fusion.py:69-91, :93-111; called at train.py:355, :365) and their backward, on
 * planar (B, 3, HW) images: mode 0 RGB -> YCrCb; mode 1 YCrCb -> RGB (ysrc != NULL supplies channel 0 from a (B, 1, HW)
 * tensor: train.py:362-364's clone + slice assignment folded in); mode 2 backward of 0 (in = d/dYCrCb -> d/dRGB); mode 3
 * backward of 1 (in = d/dRGB -> d/d[Y, Cr, Cb], nout = 1: d/dY only, (B, 1, HW)).  nout = 3 otherwise. */
int segmif_color3_f32(const float* in, const float* ysrc, float* out, int B, int64_t HW, int mode, int nout, void* stream);
/* argmax over C of NHWC logits -> int32 labels (test_segmentation.py:174); ties -> lowest index */
int segmif_argmax_nhwc_i32(const float* x, int32_t* labels, int64_t rows, int C, int ldx, void* stream);
/* (r6) test_segmentation.py:169-174 in one pass: labels = argmax_c bilinear(x -> OH x OW) for NHWC logits x (B, IH, IW, C), pixel pitch
 * ldx (align_corners = False, segmif_bilinear_nhwc_f32's arithmetic; ties -> lowest index): the resized logits are never written. */
int segmif_bilinear_argmax_i32(const float* x, int32_t* labels, int B, int IH, int IW, int OH, int OW, int C, int ldx, void* stream);
/* conf[t*K + p] += #{i : label[i] == t, pred[i] == p}, both inside [0, K) (K <= 32) — the accumulation of
 * test_segmentation.py:176-177 (sklearn confusion_matrix with labels=[0..K-1]: rows = ground truth,
 * columns = prediction, samples outside the label set ignored); conf is NOT cleared */
int segmif_confusion_i32(const int32_t* pred, const int64_t* label, int64_t* conf, int64_t n, int K, void* stream);
/* test_fusion.py:112-120 on device: a = uint8(255 x), min / max of a over the WHOLE batch, then
 * uint8(255.0 * ((a - min) / (max - min))) in float64, written NHWC (the reference's transpose(0,2,3,1));
 * x is (B, C, HW) fp32 already clamped to [0, 1]; minmax: 2 int32 of scratch (returns {min, max}) */
int segmif_quantize_u8(const float* x_nchw, uint8_t* out_nhwc, int32_t* minmax, int B, int C, int64_t HW, void* stream);
/* the way back, as test_segmentation.py's loader reads those PNGs (TaskFusion_dataset2.py:84-88): NHWC uint8 ->
 * NCHW fp32 = float(u) / 255 with an IEEE division (PairForward(uint8_roundtrip=True)) */
int segmif_dequantize_u8(const uint8_t* in_nhwc, float* out_nchw, int B, int C, int64_t HW, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Training path (backward of the ops above; autograd in the reference: loss.backward() at
 * train.py:226, :384).  Contractions reuse segmif_igemm_f32 (input gradients, with transposed /
 * rotated weights) and segmif_wgrad_f32; the entries below cover the rest.  Every reduction is
 * two-pass and deterministic.
 * ---------------------------------------------------------------------------------------------- */

/* LayerNorm backward: dx, plus per-block partial rows [dgamma | dbeta] (nblk x 2C floats) that the
 * caller sums with segmif_colsum_f32.  nblk = segmif_layernorm_bwd_blocks(rows, C). */
int segmif_layernorm_bwd_blocks(int64_t rows, int C);
int segmif_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx, float* partial,
                             int64_t rows, int C, int ldx, int ldy, int lddx, float eps, void* stream);
/* backward of segmif_add_layernorm_f32: dx = LayerNorm-backward(dy at x = the saved sum) + dres (the gradient arriving at the
 * sum itself; NULL = none), dbranch = scale[row / rows_per_image] * dx (NULL with scale NULL: the branch's gradient IS dx).
 * dx may alias dres. */
int segmif_layernorm_bwd_add_f32(const float* x, const float* dy, const float* gamma, const float* dres, int lddres,
                                 const float* scale, int64_t rows_per_image, float* dx, float* dbranch, int lddbr,
                                 float* partial, int64_t rows, int C, int ldx, int ldy, int lddx, float eps, void* stream);

/* Mix-FFN middle backward.  Part 1: z = dwconv(h)+b is recomputed, dz = dy * gelu'(z) is stored and
 * per-block partials [9 taps | bias][C] are written (segmif_dwconv_bwd_partial_rows rows x 10C floats;
 * sum with segmif_colsum_f32).  Part 2: dh = segmif_dwconv3x3_plain_f32(dz, taps flipped). C % 128 == 0. */
int64_t segmif_dwconv_bwd_partial_rows(int B, int H, int W);
int segmif_dwconv3x3_gelu_bwd_f32(const float* h, const float* w9, const float* bias, const float* dy, float* dz,
                                  float* partial, int B, int H, int W, int C, void* stream);
int segmif_dwconv3x3_plain_f32(const float* x, const float* w9, float* y, int B, int H, int W, int C, void* stream);
/* parameter-gradient partials of the bare depthwise conv (segmif_dwconv3x3_bias_f32): same [9 taps | bias][C] rows as above
 * with dz = dy; the input gradient is segmif_dwconv3x3_plain_f32(dy, taps flipped). */
int segmif_dwconv3x3_bias_bwd_f32(const float* h, const float* w9, const float* dy, float* partial, int B, int H, int W,
                                  int C, void* stream);

/* adjoint of segmif_bilinear_nhwc_f32 (gather form): dy (B,OH,OW,C) -> dx (B,IH,IW,C) */
int segmif_bilinear_nhwc_bwd_f32(const float* dy, float* dx, int B, int IH, int IW, int OH, int OW, int C,
                                 int lddy, int lddx, void* stream);

/* attention training path (scores materialised): p = softmax(s*scale) in place; ds = p*(dp - sum(p*dp))*scale in place.
 * Columns [L, ld) of every row (ld <= 1024: pitch padding a later GEMM contracts over) are set to zero by both. */
int segmif_row_softmax_f32(float* s, int64_t rows, int L, int ld, float scale, void* stream);
int segmif_row_softmax_bwd_f32(const float* p, float* dp, int64_t rows, int L, int ld, float scale, void* stream);

/* softmax cross-entropy with ignore_index over NHWC logits (train.py:156,224; model_fusion.py:1096):
 * partial[2*blk] = sum of losses, partial[2*blk+1] = valid count (doubles, segmif_softmax_ce_blocks(rows)
 * blocks); dlogits (optional) = softmax - onehot, 0 for ignored rows, NOT divided by the count. */
int segmif_softmax_ce_blocks(int64_t rows);
int segmif_softmax_ce_f32(const float* logits, const int64_t* labels, float* dlogits, double* partial,
                          int64_t rows, int C, int ld, int ldd, int ignore_index, void* stream);

/* input gradient of a strided convolution (overlap patch embed, sr conv); wd = weight as [tap][n][c] */
int segmif_conv_dgrad_strided_f32(const float* dy, const float* wd, float* dx, int B, int H, int W, int Cin, int N,
                                  int KH, int KW, int stride, int pad, int OH, int OW, int lddy, int lddx, void* stream);

/* second half of the same gradient computed as GEMM + gather: cols (B, OH, OW, k*k*C) = dY . W^T with columns ordered
 * (ky, kx, c) -> dx (B, H, W, C), dx[y][x] = sum of the taps ky = (y + pad) mod stride (+ stride ..), likewise kx, read from
 * output pixel ((y + pad - ky) / stride, (x + pad - kx) / stride).  Replaces the scalar kernel above wherever the GEMM's
 * scratch (rows x k*k*C floats) is affordable (every strided conv of the path: overlap patch embeds). */
int segmif_col2im_f32(const float* cols, float* dx, int B, int H, int W, int C, int k, int stride, int pad, int OH, int OW,
                      void* stream);

/* Train-mode BatchNorm2d on NHWC rows (segformer_head.py:50-55: linear_fuse.bn with batch statistics).
 * segmif_bn_colstats_f32: mode 0 -> out[C] = sum_rows (x - mu)^2 ; mode 1 -> out[2C] = [sum dz | sum dz*xhat]
 * (fp64; partial holds ceil(rows/256) * (mode ? 2C : C) doubles).  segmif_bn_apply_f32: y = relu?(x*scale + shift).
 * segmif_bn_bwd_apply_f32: dx = scale * (dz - a - xhat * b) with a = mean dz, b = mean dz*xhat. */
int segmif_bn_colstats_f32(const float* x, const float* dz, const float* mu, const float* rstd, double* partial,
                           double* out, int64_t rows, int C, int mode, void* stream);
int segmif_bn_apply_f32(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C,
                        int relu, void* stream);
int segmif_bn_bwd_apply_f32(const float* x, const float* dz, const float* mu, const float* rstd, const float* scale,
                            const float* a, const float* b, float* dx, int64_t rows, int C, void* stream);

/* 11-tap separable Gaussian blur, zero padded, over (planes, H, W): the SSIM window of
 * pytorch_ssim/__init__.py:8-43 (five of these per SSIM evaluation; self-adjoint, so also its backward).
 * taps11 is a HOST pointer to the 11 window weights. */
int segmif_gauss_blur11_f32(const float* x, float* y, int planes, int H, int W, const float* taps11, void* stream);

/* Fusion losses of train_fusion as fused kernels (csrc/losses.hip), single-channel images of n = planes*H*W pixels:
 *  Fusionloss_grad3 (core/loss.py:506-517) = MSE + 1.1 (1 - SSIM), SSIM of pytorch_ssim/__init__.py:19-43:
 *    segmif_ssim_prep_f32   stack5 = [gen, mask, gen^2, mask^2, gen*mask]  (then segmif_gauss_blur11_f32 over 5*planes)
 *    segmif_ssim_map_f32    sums2 = {sum of the SSIM map, sum (mask - gen)^2} (partial: 2*segmif_loss_blocks(n) doubles)
 *                           and, if der3 != NULL, the planes dS/dmu1, dS/dE[gen^2], dS/dE[gen*mask] for the backward
 *    segmif_ssim_grad_f32   grad = upstream * (coef_ssim * (B0 + 2 gen B1 + mask B2) + coef_mse * (gen - mask)), B = the
 *                           blurred der3 planes (the Gaussian window is symmetric: its adjoint is the same blur)
 *  Fusionloss3 (core/loss.py:459-476, Sobelxy :634-650) = L1(mask, gen) + L1(Sobel(mask), Sobel(gen)):
 *    segmif_sobel_l1_f32    sums2 = {sum |mask - gen|, sum |S(mask) - S(gen)|}; pxy2 (optional) = the backward's two planes
 *    segmif_sobel_l1_bwd_f32  grad = upstream / n * (sign(gen - mask) + adjoint Sobel stencils of pxy2)
 * `upstream` is a device scalar (d loss / d this term). */
int segmif_loss_blocks(int64_t n);
int segmif_ssim_prep_f32(const float* gen, const float* mask, float* stack5, int64_t n, void* stream);
int segmif_ssim_map_f32(const float* blurred5, const float* gen, const float* mask, float* der3, double* partial, double* sums2,
                        int64_t n, void* stream);
int segmif_ssim_grad_f32(const float* blurred_der3, const float* gen, const float* mask, float* grad, int64_t n,
                         const float* upstream, float coef_ssim, float coef_mse, void* stream);
int segmif_sobel_l1_f32(const float* gen, const float* mask, float* pxy2, double* partial, double* sums2, int planes, int H, int W,
                        void* stream);
int segmif_sobel_l1_bwd_f32(const float* pxy2, const float* gen, const float* mask, float* grad, int planes, int H, int W,
                            const float* upstream, void* stream);

/* LapLoss2 (lap_loss.py:100-118, built by Fusionloss_grad3 at core/loss.py:509): d_k(img) = img - G_k * img for the three
 * zero-padded Gaussians G_3, G_5, G_7 (sigma 2, lap_loss.py:39-80), loss = 10 (L1_3 + L1_5) + L1_7,
 * L1_k = mean | d_k(gen) - max(d_k(ir), d_k(vis)) | over n = planes*H*W pixels.
 *   segmif_laploss2_f32      sums2[0] = sum of 10 |a_3| + 10 |a_5| + |a_7| (divide by n); sign3 (optional, [3][n]) = sign(a_k)
 *   segmif_laploss2_bwd_f32  grad = upstream / n * sum_k c_k (s_k - G_k * s_k)   (gradient w.r.t. gen only)
 * partial: 2*segmif_loss_blocks(n) doubles. */
int segmif_laploss2_f32(const float* gen, const float* ir, const float* vis, float* sign3, double* partial, double* sums2,
                        int planes, int H, int W, void* stream);
int segmif_laploss2_bwd_f32(const float* sign3, float* grad, int planes, int H, int W, const float* upstream, void* stream);

/* The fusion net's shared scalar PReLU (core/model_fusion.py:1038) on the training path, kept apart from the conv so
 * that the backward reads the branch off the pre-activation z (any slope, also <= 0): y = z > 0 ? z : a z;
 * dz = dy (z > 0 ? 1 : a), dslope[0] = sum over z <= 0 of dy z (fp64 two-pass).  16-byte path when n % 4 == 0 and the
 * pointers are aligned, scalar otherwise.  partial: 2 * (segmif_prelu_bwd_blocks(n) + 1) doubles. */
int segmif_prelu_f32(const float* z, const float* slope, float* y, int64_t n, void* stream);
int segmif_prelu_bwd_blocks(int64_t n);
int segmif_prelu_bwd_f32(const float* dy, const float* z, const float* slope, float* dz, double* partial, float* dslope,
                         int64_t n, void* stream);
/* the same with dy a rows view (rows x C, pitch lddy floats): a channel slice of a wider gradient buffer, read in place */
int segmif_prelu_bwd_rows_f32(const float* dy, int64_t lddy, const float* z, const float* slope, float* dz, double* partial,
                              float* dslope, int64_t rows, int C, void* stream);

/* multi-tensor AdamW (utils/optimizer.py:6 -> torch.optim.AdamW arithmetic). table: device array of
 * {float* p; const float* g; float* m; float* v; int64 n; float lr; float wd;} entries
 * (segmif_adamw_entry_bytes() each); chunk_entry / chunk_off map each block to (entry, offset).
 * bias_corr1 = 1 - beta1^step and bias_corr2_sqrt = sqrt(1 - beta2^step) are evaluated by the caller in double. */
int segmif_adamw_entry_bytes(void);
int segmif_adamw_f32(const void* table, const int32_t* chunk_entry, const int64_t* chunk_off, int nchunks,
                     int chunk_elems, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2_sqrt,
                     void* stream);

/*
 * Mix-FFN of a MiT block as ONE kernel (csrc/mixffn.hip), C = 64 | 128 (stages 1-2 of mit_b1 .. b5):
 *     out = x + fc2(GELU(dwconv3x3(fc1(LayerNorm(x)))))     core/mix_transformer.py:46-53, :376-387, :152-155
 * on (B, H*W, C) tokens; a workgroup owns a 12 x 16 pixel tile, the 4C-wide hidden tensor stays in its LDS / registers.
 * f16x3 arithmetic (half pairs x three products, see segmif_planes16_*): `wimg` is the segmif_mixffn_pack image
 * (segmif_mixffn_weight_bytes(C) bytes) of everything per hidden channel - the fc1 (4C, C) and fc2 (C, 4C) weights, fc1's bias,
 * the depthwise weight as [9][4C] (tap-major) and its bias -, one contiguous block per 32 hidden channels;
 * amax_a / amax_g: range slots of the two tensors the kernel splits (LN(x) and the GELU output), indexed by image when
 * amax_images == B (NULL = off); the caller re-runs tripped images on the bf16x6 chain (segmif_layernorm_f32,
 * segmif_gemm_split_f32, segmif_dwconv3x3_gelu_f32).  out must not alias x.
 */
typedef struct SegmifMixFfn {
  const float* x; float* out; const void* wimg;
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  const float* b2;   /* fc2 bias [C] */
  int32_t B, H, W, C;
  uint32_t* amax_a; uint32_t* amax_g; int32_t amax_images;
} SegmifMixFfn;
int64_t segmif_mixffn_weight_bytes(int C);
int segmif_mixffn_pack(const float* w1 /* (4C, C) */, const float* b1, const float* dw9 /* [9][4C] */, const float* dw_bias,
                       const float* w2 /* (C, 4C) */, int C, void* out, void* stream);
int segmif_mixffn_f16x3(const SegmifMixFfn* desc, void* stream);

/*
 * Gradient exchange of the data-parallel training steps behind the C ABI (SURVEY.md section 8(b) lists
 * segmif_comm_{init, allreduce, destroy}; section 8(e): one fp32 sum / average all-reduce per step over RCCL / xGMI; the
 * reference has no distributed code at all).  Thin entry points over RCCL for a host that is not Python - this
 * repo's Python host uses torch.distributed (backend "nccl" IS RCCL) in segmif_amd/parallel.py and does not route through
 * them.  librccl is looked up at first use (the copy the process already holds first), never at load time.
 *   segmif_comm_available   0 when RCCL could be resolved (version = ncclGetVersion), SEGMIF_ENOSYS otherwise
 *   segmif_comm_unique_id   rank 0 obtains the 128-byte rendezvous id; the host hands it to every rank by its own means
 *   segmif_comm_init        one communicator per process / GPU (binds to the caller's current HIP device)
 *   segmif_comm_allreduce_f32  recv = sum (average != 0: mean) over ranks of send, in place allowed, on `stream`
 * Return 0, SEGMIF_EINVAL / SEGMIF_ENOSYS, or 1000 + ncclResult_t.
 */
int segmif_comm_available(int* version);
int segmif_comm_unique_id(void* id, int64_t bytes /* >= 128 */);
int segmif_comm_init(void** comm, int world, int rank, const void* id, int64_t bytes);
int segmif_comm_world(void* comm, int* world, int* rank);
int segmif_comm_allreduce_f32(void* comm, const float* send, float* recv, int64_t count, int average, void* stream);
int segmif_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* SEGMIF_HIP_H */
