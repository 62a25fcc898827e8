"""ctypes binding of libsegmif_hip.so (C ABI in include/segmif_hip.h).

There is no fallback: if the gfx950 library is missing this module raises, it never routes
work to torch ops or to the CPU oracle.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libsegmif_hip.so")

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_GELU = 0, 1, 2, 3


class SegmifIgemm(ctypes.Structure):
    _fields_ = [
        ("in_", c_void_p), ("in2", c_void_p), ("wt", c_void_p), ("bias", c_void_p), ("res", c_void_p),
        ("prelu", c_void_p), ("out", c_void_p),
        ("M", c_int64), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("lda2", c_int32), ("K1", c_int32), ("ldo", c_int32), ("ldr", c_int32),
        ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("KH", c_int32), ("KW", c_int32),
        ("stride", c_int32), ("pad", c_int32), ("dil", c_int32), ("OH", c_int32), ("OW", c_int32),
        ("act", c_int32), ("nz", c_int32),
        ("in_zstride", c_int64), ("in2_zstride", c_int64), ("wt_zstride", c_int64),
        ("out_zstride", c_int64), ("res_zstride", c_int64),
        ("tile", c_int32), ("nz2", c_int32), ("ldw", c_int32),
        ("in_zstride2", c_int64), ("wt_zstride2", c_int64), ("out_zstride2", c_int64), ("res_zstride2", c_int64),
        ("workspace", c_void_p), ("workspace_floats", c_int64),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
        ("planes_out", c_void_p), ("planes_chunks", c_int32), ("planes_chunk0", c_int32),
        ("planes_f16", c_int32), ("planes_amax", c_void_p), ("planes_amax_images", c_int32),
        ("relu_mask", c_void_p), ("ld_mask", c_int32),
        ("split_f16", c_int32), ("split_in_amax", c_void_p), ("split_in_amax_n", c_int32), ("split_out_amax", c_void_p), ("split_out_amax_n", c_int32),
        ("wgrad_dy_amax", c_void_p), ("wgrad_dy_amax_n", c_int32), ("mask_zstride", c_int64),
    ]


class SegmifConvPlanes(ctypes.Structure):
    _fields_ = [
        ("planes_in", c_void_p), ("planes_out", c_void_p), ("wt", c_void_p), ("bias", c_void_p), ("prelu", c_void_p),
        ("out", c_void_p), ("ldo", c_int32),
        ("B", c_int32), ("H", c_int32), ("W", c_int32), ("cin", c_int32), ("dil", c_int32),
        ("in_chunks", c_int32), ("out_chunks", c_int32), ("out_chunk0", c_int32), ("act", c_int32),
        ("w1", c_void_p), ("bias1", c_void_p), ("res", c_void_p), ("out1", c_void_p),
        ("ldr", c_int32), ("ldo1", c_int32), ("act1", c_int32), ("res_from_planes", c_int32),
    ]


class SegmifGemmSplit(ctypes.Structure):
    _fields_ = [("a", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("prelu", c_void_p), ("out", c_void_p),
                ("M", c_int64), ("N", c_int32), ("K", c_int32), ("lda", c_int32), ("ldo", c_int32), ("ldr", c_int32),
                ("act", c_int32), ("patch_k", c_int32), ("patch_st", c_int32), ("patch_pad", c_int32), ("patch_H", c_int32),
                ("patch_W", c_int32)]


class SegmifGemmPairs(ctypes.Structure):
    _fields_ = [("a", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("prelu", c_void_p), ("out", c_void_p),
                ("M", c_int64), ("lda_bytes", c_int64),
                ("N", c_int32), ("K", c_int32), ("ldo", c_int32), ("ldr", c_int32), ("act", c_int32),
                ("patch_k", c_int32), ("patch_st", c_int32), ("patch_pad", c_int32), ("patch_H", c_int32), ("patch_W", c_int32),
                ("patch_C", c_int32), ("tile_rows", c_int32)]


class SegmifCrossTail(ctypes.Structure):
    _fields_ = [
        ("x3", c_void_p), ("xi", c_void_p), ("w3", c_void_p), ("b3", c_void_p), ("wi", c_void_p), ("bi", c_void_p),
        ("weff", c_void_p), ("bend", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
        ("out", c_void_p), ("ld3", c_int32), ("ldi", c_int32), ("ldo", c_int32),
        ("B", c_int32), ("N", c_int64),
        ("planes_out", c_void_p), ("H", c_int32), ("W", c_int32), ("planes_chunks", c_int32),
        ("planes_f16", c_int32), ("planes_amax", c_void_p), ("planes_amax_images", c_int32),
        ("x3_ih", c_int32), ("x3_iw", c_int32),
        ("arith_f16", c_int32), ("arith_amax", c_void_p), ("arith_amax_images", c_int32),
    ]


class SegmifMixFfn(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("wimg", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p),
                ("ln_eps", c_float), ("b2", c_void_p),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32),
                ("amax_a", c_void_p), ("amax_g", c_void_p), ("amax_images", c_int32)]


# name -> (restype, argtypes); must list every symbol include/segmif_hip.h declares
SIGNATURES = {
    "segmif_abi_version": (c_int, []),
    "segmif_device_name": (c_int, [c_char_p, c_int]),
    "segmif_igemm_f32": (c_int, [POINTER(SegmifIgemm), c_void_p]),
    "segmif_igemm_workspace_floats": (c_int64, [POINTER(SegmifIgemm)]),
    "segmif_igemm_num_tiles": (c_int, []),
    "segmif_igemm_tile_name": (c_char_p, [c_int]),
    "segmif_pack_conv_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_conv3x3_split_weight_bytes": (c_int64, [c_int, c_int]),
    "segmif_conv3x3_split_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_conv3x3_split16_weight_bytes": (c_int64, [c_int, c_int]),
    "segmif_conv3x3_split16_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_amax_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "segmif_gemm_split_weight_bytes": (c_int64, [c_int, c_int]),
    "segmif_gemm_split_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_gemm_split_f32": (c_int, [POINTER(SegmifGemmSplit), c_void_p]),
    "segmif_gemm_split16_weight_bytes": (c_int64, [c_int, c_int]),
    "segmif_gemm_split16_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_gemm_split16_f32": (c_int, [POINTER(SegmifGemmSplit), c_void_p, c_int, c_void_p]),
    "segmif_planes_dims": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "segmif_planes_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "segmif_planes_zero_border": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_planes_from_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_planes_weight_bytes": (c_int64, [c_int, c_int, c_int]),
    "segmif_planes_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_conv3x3_planes_bf16x6": (c_int, [POINTER(SegmifConvPlanes), c_void_p]),
    "segmif_planes16_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "segmif_planes16_zero_border": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_planes16_from_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "segmif_planes16_weight_bytes": (c_int64, [c_int, c_int, c_int]),
    "segmif_planes16_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_conv3x3_planes_f16x3": (c_int, [POINTER(SegmifConvPlanes), c_void_p, c_int, c_void_p]),
    "segmif_gemm_pairs_weight_bytes": (c_int64, [c_int, c_int]),
    "segmif_gemm_pairs_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_gemm_pairs_f32": (c_int, [POINTER(SegmifGemmPairs), c_void_p]),
    "segmif_pairs_from_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "segmif_pairs_to_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "segmif_layernorm_pairs_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                         c_int, c_int, c_int, c_void_p]),
    "segmif_dwconv3x3_gelu_pairs_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                              c_void_p]),
    "segmif_sr_attention_split16_pairs_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                    c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "segmif_sr_attention_bwd_chunks": (c_int, [c_int, c_int, c_int, c_int]),
    "segmif_sr_attention_bwd_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "segmif_sr_attention_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                          c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "segmif_crosspath_gram_blocks": (c_int, [c_int64]),
    "segmif_crosspath_gram_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "segmif_crosspath_gram_lazy_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "segmif_crosspath_gram_sum_f64": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "segmif_crosspath_fold_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_float, c_void_p, c_void_p]),
    "segmif_crosspath_tail_f32": (c_int, [POINTER(SegmifCrossTail), c_void_p]),
    "segmif_color3_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "segmif_mixffn_weight_bytes": (c_int64, [c_int]),
    "segmif_mixffn_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "segmif_mixffn_f16x3": (c_int, [POINTER(SegmifMixFfn), c_void_p]),
    "segmif_comm_available": (c_int, [POINTER(c_int)]),
    "segmif_comm_unique_id": (c_int, [c_void_p, c_int64]),
    "segmif_comm_init": (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p, c_int64]),
    "segmif_comm_world": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "segmif_comm_allreduce_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "segmif_comm_destroy": (c_int, [c_void_p]),
    "segmif_upsum_act_nhwc_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                          c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_confusion_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "segmif_quantize_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "segmif_dequantize_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "segmif_wgrad_workspace_size": (c_int64, [c_int64, c_int, c_int]),
    "segmif_wgrad_f32": (c_int, [POINTER(SegmifIgemm), c_void_p, c_int, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                 c_void_p, c_int, c_void_p]),
    "segmif_wgrad_batched2_f32": (c_int, [POINTER(SegmifIgemm), c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                          c_void_p, c_int, c_void_p]),
    "segmif_colsum_blocks": (c_int, [c_int64]),
    "segmif_colsum_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "segmif_act_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p]),
    "segmif_act_bwd2_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p]),
    "segmif_add_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "segmif_layernorm_bwd_add_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                             c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "segmif_prelu_bwd_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_void_p]),
    "segmif_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "segmif_dwconv3x3_gelu_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_dwconv3x3_bias_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_dwconv3x3_bias_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_laploss2_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "segmif_laploss2_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_prelu_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "segmif_prelu_bwd_blocks": (c_int, [c_int64]),
    "segmif_prelu_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "segmif_conv3x3_c32to1_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "segmif_bilinear_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_sr_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_float, c_void_p]),
    "segmif_sr_attention_split_workspace": (c_int64, [c_int, c_int, c_int]),
    "segmif_sr_attention_split_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                              c_int, c_int, c_int, c_float, c_void_p]),
    "segmif_sr_attention_split16_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                              c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "segmif_linattn_num_blocks": (c_int, [c_int64]),
    "segmif_linattn_partial_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "segmif_linattn_kvpartial_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "segmif_linattn_fold_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_float, c_void_p]),
    "segmif_layernorm_bwd_blocks": (c_int, [c_int64, c_int]),
    "segmif_layernorm_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                         c_int, c_float, c_void_p]),
    "segmif_dwconv_bwd_partial_rows": (c_int64, [c_int, c_int, c_int]),
    "segmif_dwconv3x3_gelu_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                              c_int, c_int, c_void_p]),
    "segmif_dwconv3x3_plain_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "segmif_bilinear_nhwc_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p]),
    "segmif_row_softmax_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "segmif_row_softmax_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "segmif_softmax_ce_blocks": (c_int, [c_int64]),
    "segmif_softmax_ce_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "segmif_conv_dgrad_strided_f32": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 13 + [c_void_p]),
    "segmif_col2im_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "segmif_bn_colstats_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                       c_void_p]),
    "segmif_bn_apply_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "segmif_bn_bwd_apply_f32": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_void_p]),
    "segmif_gauss_blur11_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_float), c_void_p]),
    "segmif_loss_blocks": (c_int, [c_int64]),
    "segmif_ssim_prep_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "segmif_ssim_map_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "segmif_ssim_grad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_void_p]),
    "segmif_sobel_l1_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "segmif_sobel_l1_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "segmif_adamw_entry_bytes": (c_int, []),
    "segmif_adamw_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                               c_void_p]),
    "segmif_seg_normalize_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "segmif_nchw_to_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "segmif_nhwc_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "segmif_fuse_ycrcb_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "segmif_conv3x3_c1_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                      c_int, c_int, c_void_p, c_int, c_void_p]),
    "segmif_linattn_fold_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
                                          c_int, c_int, c_void_p]),
    "segmif_pointwise2_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "segmif_argmax_nhwc_i32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "segmif_bilinear_argmax_i32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises HipLibraryMissing — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SEGMIF_HIP_LIB", LIB_PATH)  # override: kernel-tuning builds of the same ABI
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found: build it with `python -m segmif_amd.build` "
            "(hipcc --offload-arch=gfx950). segmif_amd has no CPU or torch fallback.")
    # torch FIRST: it ships its own libamdhip64, and the HIP runtime this library binds to must be the one torch initialises.
    # Loaded the other way round (python __graft_entry__.py smoke: build() binds the symbols before anything imported torch) the
    # kernels' library came up on the system ROCm's runtime and its first launch failed with hipErrorNoDevice (r5).
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.segmif_abi_version() != 3:
        raise HipLibraryMissing("libsegmif_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


_SYNC_DEBUG = bool(os.environ.get("SEGMIF_SYNC_DEBUG"))
_TRACE = bool(os.environ.get("SEGMIF_TRACE"))
_trace_n = [0]


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}")
    if _TRACE:
        import sys
        _trace_n[0] += 1
        print(f"[segmif {_trace_n[0]}] {what}", file=sys.stderr, flush=True)
    if _SYNC_DEBUG:  # debugging aid: localise an asynchronous GPU fault to the launch that caused it
        import sys
        import torch
        print(f"[segmif] {what} ...", end="", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print(" ok", file=sys.stderr, flush=True)
