"""ctypes binding of librcf_hip.so (the C ABI declared in include/rcf_hip.h).

The product has NO CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import c_int, c_uint, c_long, c_float, c_double, c_void_p, c_size_t, c_char_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librcf_hip.so")
# The 16-bit storage type is a build parameter of the library (csrc/rcf_common.h): librcf_hip.so stores bf16, librcf_hip_f16.so
# IEEE fp16 (the same sources; the four that touch 16-bit tensors compiled with -DRCF_HALF_F16).  ACTIVE names the one `call`
# talks to; rcf_amd.ops.half_storage(dtype) switches it for the duration of a model's forward / backward.
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.path.join(HERE, "librcf_hip_f16.so")}
ACTIVE = "bf16"


class RcfHipError(RuntimeError):
    pass


class ConvShape(ctypes.Structure):
    """mirror of rcf_conv_shape"""
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Ho", "Wo", "Cout", "R", "S", "stride", "pad", "dil",
                                     "x_pitch", "y_pitch")] + \
               [(n, ctypes.c_void_p) for n in ("amax_x", "amax_w", "amax_dy", "w_pairs", "w_pairs_t", "w_pairs2",
                                                "w_pairs2_t", "amax_y")] + \
               [("flags", ctypes.c_uint), ("struct_bytes", ctypes.c_uint)]


class BnBwdIn(ctypes.Structure):
    """mirror of rcf_bn_bwd_in"""
    _fields_ = [("x", ctypes.c_void_p), ("x_pitch", c_int), ("relu_mask", ctypes.c_void_p), ("mean", ctypes.c_void_p),
                ("invstd", ctypes.c_void_p)]


# include/rcf_hip.h RCF_CONV_* flag bits
CONV_WGRAD_TILE_128 = 0x1
CONV_X_PLANES, CONV_DY_PLANES = 0x2, 0x4
WARP_PER_PIXEL = 0x100
BN_SWEEP_OFF, BN_SWEEP_ALWAYS, BN_Y_PLANES_ONLY, BN_DX_PLANES = 0x1, 0x2, 0x4, 0x8
CONV_H2P_NEVER, CONV_H2P_ALWAYS = 0x8, 0x10
CONV_NO_WGRAD_XCD, CONV_NO_COLMAP, CONV_KORDER_NATURAL, CONV_NO_THIN, CONV_NT_STORES = 0x20, 0x40, 0x80, 0x100, 0x200


def CONV_FP32_MFMA(v):
    return ((int(v) & 3) + 1) << 12


class BnFinalize(ctypes.Structure):
    """struct rcf_bn_finalize of include/rcf_hip.h"""
    _fields_ = [("count", ctypes.c_double), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
                ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p), ("running_mean", ctypes.c_void_p),
                ("running_var", ctypes.c_void_p), ("num_batches_tracked", ctypes.c_void_p)]


class FoldFinalize(ctypes.Structure):
    """struct rcf_fold_finalize of include/rcf_hip.h"""
    _fields_ = [("count", ctypes.c_double), ("eps", ctypes.c_float), ("momentum", ctypes.c_float)] + \
               [(n, ctypes.c_void_p) for n in ("gamma", "beta", "mean", "invstd", "scale", "shift", "running_mean", "running_var",
                                               "num_batches_tracked")]


class BnResNorm(ctypes.Structure):
    """struct rcf_bn_res_norm of include/rcf_hip.h"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("mean", "invstd", "gamma", "beta")]


class BnBwdSecond(ctypes.Structure):
    """struct rcf_bn_bwd_second of include/rcf_hip.h"""
    _fields_ = [("x", ctypes.c_void_p), ("x_pitch", c_int), ("dx", ctypes.c_void_p), ("dx_pitch", c_int)] + \
               [(n, ctypes.c_void_p) for n in ("mean", "invstd", "gamma", "sums2", "sums2_local", "dgamma", "dbeta", "amax_out", "amax_x")]


class ConvRegion(ctypes.Structure):
    """mirror of rcf_conv_region"""
    _fields_ = [(n, c_int) for n in ("y0", "x0", "h", "w", "band")]


class FlowHeadCfg(ctypes.Structure):
    """mirror of rcf_flowhead_cfg"""
    _fields_ = [(n, c_int) for n in ("B", "C", "h", "w", "logits_pitch", "nf", "D", "robust", "tanh_residual")] + \
               [(n, c_float) for n in ("eps", "q", "clamp_t", "res_scale", "div_coeff", "w_seg", "w_entropy")] + \
               [("n_targets", c_int), ("target_channel", c_int)] + \
               [(n, c_float * 2) for n in ("t_wpos", "t_wneg", "t_weight", "t_thresh")] + \
               [("w_compact", c_float), ("compact_channel", c_int), ("w_sharpen", c_float), ("t_sharpen", c_float),
                ("sharpen_mode", c_int)]


P = c_void_p
_CS = ctypes.POINTER(ConvShape)
_FH = ctypes.POINTER(FlowHeadCfg)
_CR = ctypes.POINTER(ConvRegion)
_BF = ctypes.POINTER(BnFinalize)

# name -> (restype, argtypes); every int-returning entry point is status-checked by `call`
PROTOS = {
    "rcf_version": (c_char_p, []),
    "rcf_conv2d_fwd_f32": (c_int, [P, P, P, P, _CS, c_int, c_float, c_int, P]),
    "rcf_conv2d_fwd_region_f32": (c_int, [P, P, P, P, _CS, _CR, c_int, c_float, c_int, P]),
    "rcf_conv2d_dgrad_region_f32": (c_int, [P, P, P, _CS, _CR, c_int, P, c_size_t, P]),
    "rcf_conv2d_wgrad_region_workspace_bytes": (c_size_t, [_CS, _CR]),
    "rcf_conv2d_wgrad_region_f32": (c_int, [P, P, P, _CS, _CR, c_int, P, c_size_t, P]),
    "rcf_gemm_nt_f32": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P, P, P, P,
                                P]),
    "rcf_gemm_nt_batched_f32": (c_int, [P, c_int, c_long, c_long, P, c_int, c_long, c_long, P, c_int, c_long, c_long, c_int,
                                        c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "rcf_attention_fwd_f32": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P, P]),
    "rcf_layernorm_f32": (c_int, [P, c_int, P, c_int, c_long, c_int, P, P, c_float, P, P]),
    "rcf_softmax_rows_f32": (c_int, [P, c_long, c_long, c_int, c_float, P]),
    "rcf_transpose2d_f32": (c_int, [P, c_long, P, c_long, c_int, c_int, P]),
    "rcf_l2_normalize_rows_f32": (c_int, [P, c_long, P, c_long, c_long, c_int, P]),
    "rcf_affinity_threshold_f32": (c_int, [P, c_long, c_int, c_float, c_float, P]),
    "rcf_ncut_value_grad_f32": (c_int, [P, c_long, c_int, P, P, P, c_int, P, P, P]),
    "rcf_clamp01_f32": (c_int, [P, c_int, P]),
    "rcf_split_rect_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_absmax_f32": (c_int, [P, c_long, c_int, c_int, P, P]),
    "rcf_conv_weight_pairs_f32": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_uint, P]),
    "rcf_conv_weight_pairs_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rcf_conv2d_dgrad_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_dgrad_bnsums_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_dgrad_bnsums_ok": (c_int, [_CS]),
    "rcf_conv2d_dgrad_bnsums_f32": (c_int, [P, P, P, _CS, c_int, ctypes.POINTER(BnBwdIn), P, P, c_size_t, P]),
    "rcf_conv2d_dgrad_add_f32": (c_int, [P, P, P, _CS, c_int, P, c_int, P, ctypes.POINTER(BnBwdIn), P, P, c_size_t, P]),
    "rcf_relu_mask_copy_mp": (c_int, [P, c_int, c_int, P, P, c_int, c_long, c_int, c_int, P]),
    "rcf_conv2d_fwd_stats_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_fwd_stats_f32": (c_int, [P, P, P, _CS, P, P, c_size_t, P]),
    "rcf_conv_weight_pairs_t_f32": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_uint, P]),
    "rcf_conv_weight_pairs2_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "rcf_conv_weight_pairs2_f32": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_uint, P]),
    "rcf_conv_kernel_of": (c_int, [_CS, _CR, c_int]),
    "rcf_conv_pairs2_useful": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "rcf_conv_weights_prepare_f32": (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, c_uint, P]),
    "rcf_conv_weights_prepare_bf16": (c_int, [P, c_int, P, c_int, c_int, c_uint, P]),
    "rcf_conv2d_fwd_bnstats_f32": (c_int, [P, P, P, _CS, P, _BF, P, c_size_t, P]),
    "rcf_conv2d_fwd_bnstats_bf16": (c_int, [P, P, P, c_int, _CS, P, _BF, P, c_size_t, P]),
    "rcf_sum_partials_f64": (c_int, [P, c_int, c_int, P, P, P]),
    "rcf_conv2d_dgrad_f32": (c_int, [P, P, P, _CS, c_int, P, c_size_t, P]),
    "rcf_conv2d_wgrad_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_wgrad_f32": (c_int, [P, P, P, _CS, c_int, P, c_size_t, P]),
    "rcf_bn_stats_workspace_bytes": (c_size_t, [c_long, c_int]),
    "rcf_bn_stats_f32": (c_int, [P, c_long, c_int, c_int, P, P, c_size_t, P]),
    "rcf_bn_finalize_f32": (c_int, [P, c_double, c_int, c_float, c_float, P, P, P, P, P]),
    "rcf_bn_apply_f32": (c_int, [P, c_int, P, c_int, P, c_int, c_long, c_int, P, P, P, P, c_int, P, c_long, P, P, P]),
    "rcf_bn_invstd_from_var_f32": (c_int, [P, c_int, c_float, P, P]),
    "rcf_bn_bwd_reduce_f32": (c_int, [P, c_int, P, c_int, P, c_int, c_long, c_int, P, P, c_int, P, P, c_long, P, P,
                                      c_size_t, P]),
    "rcf_bn_bwd_apply_f32": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_long, c_int, P, P, P,
                                     c_int, P, P, c_long, P, P, c_double, P, P, P, P]),
    "rcf_maxpool3x3s2_fwd_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_maxpool3x3s2_bwd_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_resize_bilinear_nhwc_fwd_f32": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_resize_bilinear_nhwc_fwd_frame_f32": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                       c_int, P]),
    "rcf_resize_bilinear_nhwc_bwd_frame_f32": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                       c_int, c_int, P]),
    "rcf_resize_bilinear_nhwc_bwd_f32": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_int, P]),
    "rcf_resize_bilinear_nchw_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_nchw_to_nhwc_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_nhwc_to_nchw_f32": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "rcf_copy2d_f32": (c_int, [P, c_long, P, c_long, c_long, c_int, c_int, P]),
    "rcf_copy2d_batched_f32": (c_int, [P, c_long, c_long, c_long, P, c_long, c_long, c_long, c_long, c_int, c_int, c_int,
                                       c_int, P]),
    "rcf_colsum_f32": (c_int, [P, c_long, c_int, c_int, P, c_int, P, c_size_t, P]),
    "rcf_flow_warp_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_flow_warp_bwd_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_occu_mask_backward_f32": (c_int, [P, P, c_float, P, c_int, c_int, c_int, P]),
    "rcf_occu_mask_bidirection_f32": (c_int, [P, P, P, c_float, c_float, c_int, c_int, c_int, P]),
    "rcf_warp_l1_residual_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_photometric_loss_f32": (c_int, [P, P, P, c_float, c_float, P, P, c_int, c_int, c_int, c_int, P]),
    "rcf_crf_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rcf_crf_soft": (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, P, P, P,
                             P, c_size_t, P]),
    "rcf_crf_soft_ex": (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_int, P, P, P,
                                P, c_size_t, P]),
    "rcf_crf_soft_f32": (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_int, P, P, P,
                                 P, c_size_t, P]),
    "rcf_crf_hard": (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_int,
                             P, P, P, P, c_size_t, P]),
    "rcf_crf_prepare": (c_int, [P, P, P, P, c_int, c_float, P, P, P, c_int, c_int, c_int, P]),
    "rcf_flowhead_workspace_bytes": (c_size_t, [_FH]),
    "rcf_flowhead_prepare_f32": (c_int, [_FH, P, P, P, P, c_size_t, P]),
    "rcf_flowhead_fwd_f32": (c_int, [_FH] + [P] * 15 + [P, c_size_t, P]),
    "rcf_flowhead_bwd_f32": (c_int, [_FH, P, P, P, P, P, P, c_float] + [P] * 7 + [P, c_size_t, P]),
    "rcf_lrelu_bwd_f32": (c_int, [P, P, P, c_long, c_float, P]),
    "rcf_adam_step_f32": (c_int, [P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "rcf_ema_update_f32": (c_int, [P, P, c_long, c_float, P]),
    "rcf_ema_update_multi": (c_int, [P, c_int, c_float, c_float, P]),
    "rcf_fill_f32": (c_int, [P, c_long, c_float, P]),
    "rcf_dropout2d_scale_f32": (c_int, [P, c_long, c_float, ctypes.c_ulonglong, P]),
    # mixed-precision (bf16 storage) forms
    "rcf_bn_stats_mp": (c_int, [P, c_int, c_long, c_int, c_int, P, P, c_size_t, P]),
    "rcf_bn_apply_mp": (c_int, [P, c_int, c_int, P, c_int, P, c_int, c_int, c_long, c_int, P, P, P, P, c_int, P, c_long, P,
                                P, P, P, P, c_uint, P]),
    "rcf_bn_apply_res_mp": (c_int, [P, c_int, c_int, P, c_int, P, P, c_int, c_int, c_long, c_int, P, P, P, P, c_int, P, c_long, P,
                                    P, P, P, P, c_uint, P]),
    "rcf_bn_bwd_reduce2_mp": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_long, c_int, P, P, P, P, P, P, P, c_size_t, c_uint, P]),
    "rcf_bn_bwd_apply2_mp": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_long, c_int, P, P, P, P, P, P, c_double, P, P, P, P, P, P,
                                     c_uint, P]),
    "rcf_bn_bwd_reduce_mp": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_long, c_int, P, P, c_int, P, P, c_long,
                                     P, P, c_size_t, c_uint, P]),
    "rcf_bn_bwd_apply_mp": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int, c_long, c_int, P,
                                    P, P, c_int, P, P, c_long, P, P, c_double, P, P, P, P, P, c_uint, P]),
    "rcf_colsum_mp": (c_int, [P, c_int, c_long, c_int, c_int, P, c_int, P, c_size_t, P]),
    "rcf_maxpool3x3s2_fwd_mp": (c_int, [P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_maxpool3x3s2_bwd_mp": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_resize_bilinear_nhwc_fwd_mp": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_int, P]),
    "rcf_resize_bilinear_nhwc_bwd_mp": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_int, c_int, P]),
    "rcf_copy2d_batched_mp": (c_int, [P, c_long, c_long, c_long, P, c_long, c_long, c_long, c_int, c_long, c_int, c_int,
                                      c_int, c_int, P]),
    "rcf_copy2d_mp": (c_int, [P, c_int, c_long, P, c_int, c_long, c_long, c_int, c_int, P]),
    "rcf_split_rect_mp": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rcf_conv_weight_bf16_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "rcf_conv_weight_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_uint, P]),
    "rcf_conv2d_fwd_stats_bf16_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_fwd_bf16": (c_int, [P, P, P, P, c_int, _CS, _CR, c_int, c_float, c_int, P, P, c_size_t, P]),
    "rcf_conv2d_dgrad_bf16_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_dgrad_bf16": (c_int, [P, P, P, _CS, _CR, c_int, P, c_size_t, P]),
    "rcf_conv2d_wgrad_bf16_workspace_bytes": (c_size_t, [_CS, _CR]),
    "rcf_conv2d_wgrad_bf16": (c_int, [P, P, P, _CS, _CR, c_int, P, c_size_t, P]),
    "rcf_fold_fwd_scratch_bytes": (c_size_t, [c_int, c_int]),
    "rcf_fold_fwd_f32": (c_int, [P, P, P, P, P, ctypes.POINTER(FoldFinalize), P, c_size_t, c_double, c_int, c_int, P]),
    "rcf_fold_finalize_f32": (c_int, [P, c_int, ctypes.POINTER(FoldFinalize), P]),
    "rcf_conv_relu_bits_bytes": (c_size_t, [c_long, c_int]),
    "rcf_conv2d_fwd_affine_bf16": (c_int, [P, P, P, P, P, c_int, c_int, P, P, _CS, P]),
    "rcf_relu_mask_colsum_bf16_workspace_bytes": (c_size_t, [c_long, c_int]),
    "rcf_relu_mask_colsum_bf16": (c_int, [P, c_int, P, c_int, P, c_int, c_long, c_int, P, P, c_size_t, P]),
    "rcf_conv2d_dgrad_masked_bf16_workspace_bytes": (c_size_t, [_CS]),
    "rcf_conv2d_dgrad_masked_bf16": (c_int, [P, P, P, _CS, c_int, P, c_int, P, P, P, c_size_t, P]),
    "rcf_fold_bwd_sums_f32": (c_int, [P, P, P, P, P, P, c_int, c_int, P]),
    "rcf_fold_bwd_scratch_bytes": (c_size_t, [c_int, c_int]),
    "rcf_fold_wg_bf16": (c_int, [P, P, P, c_int, c_int, P]),
    "rcf_fold_bwd_prepare_f32": (c_int, [P] * 6 + [c_double] + [P] * 9 + [c_size_t, c_int, c_int, P]),
    "rcf_eval_iou_counts_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, P]),
    "rcf_aug_frames_u8": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, P, P]),
    "rcf_aug_flows_f32": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P]),
    "rcf_aug_masks_u8": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P]),
}
F32, BF16 = 0, 1          # storage type codes (RCF_F32 / RCF_BF16)

_libs = {}


def load(which=None):
    """Load librcf_hip.so (or, which = "f16" / ACTIVE == "f16", librcf_hip_f16.so); raises RcfHipError when it has not been built
    (no fallback)."""
    which = which or ACTIVE
    lib = _libs.get(which)
    if lib is None:
        path = LIB_PATHS[which]
        if not os.path.exists(path):
            raise RcfHipError(f"{path} is missing: run `python __graft_entry__.py` (build) first; "
                              "there is no CPU fallback for the product path")
        lib = ctypes.CDLL(path)
        for name, (res, args) in PROTOS.items():
            fn = getattr(lib, name, None)
            if fn is None:
                continue            # reported by missing_symbols(); calling it raises
            fn.restype = res
            fn.argtypes = args
        _libs[which] = lib
    return lib


def missing_symbols(which=None):
    lib = load(which)
    return [n for n in PROTOS if not hasattr(lib, n)]


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    fn = getattr(load(), name, None)
    if fn is None:
        raise RcfHipError(f"{os.path.basename(LIB_PATHS[ACTIVE])} does not export {name}")
    rc = fn(*args)
    if PROTOS[name][0] is c_int and rc != 0:
        raise RcfHipError(f"{name} failed with status {rc}" + (" (bad argument)" if rc == -1 else
                                                              " (workspace too small)" if rc == -2 else " (HIP error)"))
    return rc
