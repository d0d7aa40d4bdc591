"""ctypes binding of libspcl_hip.so (the C ABI declared in include/spcl_hip.h).

PyTorch is used only for device memory and streams: every call passes raw device pointers and the current
HIP stream.  There is NO CPU or eager-PyTorch fallback: a missing library or a non-GPU tensor raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libspcl_hip.so")
SPCL_F32, SPCL_BF16 = 0, 1

_lib = None

# name -> (restype, argtypes); kept in the order of include/spcl_hip.h
_P = c_void_p
_SIGNATURES = {
    "spcl_abi_version": (c_int, []),
    "spcl_last_error": (c_char_p, []),
    "spcl_supcon_workspace_bytes": (c_size_t, [c_int, c_int]),
    "spcl_supcon_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_int, c_float, c_int, _P, _P, _P]),
    "spcl_supcon_backward": (c_int, [_P, _P, c_int, c_int, c_float, c_int, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_supcon_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "spcl_supcon_unit_gradient_block": (c_int, [c_int, c_int, _P, _P]),
    "spcl_supcon_materialize": (c_int, [_P, _P, c_int, c_int, c_float, c_int, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_supcon_xpos_workspace_bytes": (c_size_t, [c_int, c_int]),
    "spcl_supcon_xpos_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "spcl_supcon_xpos_backward": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    "spcl_proj_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int,
                                  _P, _P, _P, _P, _P]),
    "spcl_proj_backward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P,
                                   _P, _P, _P, _P, _P, _P, _P]),
    "spcl_proj_heads_forward": (c_int, [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int,
                                        _P, _P, _P, _P, _P]),
    "spcl_proj_heads_backward": (c_int, [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P,
                                         _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_proj_heads_backward_pooled": (c_int, [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P,
                                                _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_adaptive_pool2d_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "spcl_adaptive_pool2d_backward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                              _P]),
    "spcl_l2norm_rows_forward": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "spcl_l2norm_rows_backward": (c_int, [_P, _P, c_size_t, c_int, _P, _P]),
    "spcl_conv_packed_elems": (c_size_t, [c_int, c_int, c_int, c_int]),
    "spcl_conv_pack_weights": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "spcl_conv_pack_weights_both": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "spcl_conv_pack_weights_block": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P, c_int, _P]),
    "spcl_conv_pack_weights_block_at": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "spcl_conv_pack_weights_multi": (c_int, [_P, c_int, c_int, _P]),
    "spcl_conv_pack_weights_multi_acorr": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "spcl_conv_pack_weights_multi_zero": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "spcl_conv_num_tiles": (c_int, [c_int, c_int, c_int]),
    "spcl_conv_stat_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv_set_gemm": (None, [c_int]),
    "spcl_conv_set_f32_split": (None, [c_int]),
    "spcl_conv_get_f32_split": (c_int, []),
    "spcl_conv_cat_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_forward_cat": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "spcl_conv_split_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_forward_split": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "spcl_conv_split_bnstats_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_dgrad_split_bnstats": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P,
                                                 _P]),
    "spcl_conv_up2_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_forward_up2": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "spcl_conv3x3_wgrad_up2": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "spcl_conv3x3_wgrad_cat": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "spcl_bn_stats_elems": (c_size_t, [c_int, c_int]),
    "spcl_conv3x3_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P,
                                     _P]),
    "spcl_supcon_forward_heads": (c_int, [c_int, _P, _P, c_size_t, _P, c_int, c_int, c_float, c_int, _P, c_int, _P,
                                          c_size_t, _P, _P]),
    "spcl_supcon_rows_supported": (c_int, [c_int, c_int]),
    "spcl_supcon_forward_rows": (c_int, [c_int, _P, _P, c_size_t, _P, c_int, c_int, c_float, c_int, _P, c_int, _P,
                                         c_size_t, _P, _P]),
    "spcl_supcon_backward_heads": (c_int, [c_int, _P, c_int, c_int, c_float, c_int, _P, _P, c_size_t, _P, c_size_t, _P,
                                           _P, _P, _P, c_size_t, _P]),
    "spcl_conv_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                   _P, _P, _P, _P]),
    "spcl_conv_wgrad_batched_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv_wgrad_batched_workspace_bytes": (c_size_t, [_P, c_int]),
    "spcl_conv3x3_wgrad_batched": (c_int, [_P, c_int, c_int, _P, _P]),
    "spcl_wgrad_tail_capture": (c_int, [_P]),
    "spcl_conv3x3_wgrad_batched_tails": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    "spcl_bn_finalize": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_bn_eval_affine": (c_int, [c_int, c_int, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P]),
    "spcl_bn_eval_affine_multi": (c_int, [_P, c_int, _P]),
    "spcl_bnrelu_pool_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "spcl_bnrelu_up2_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "spcl_bnrelu_pool_forward_strided": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "spcl_bnrelu_pool_backward_strided": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                                                  _P, c_int, _P, _P, _P, _P, _P]),
    "spcl_bnrelu_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "spcl_bnrelu_pool_backward": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P,
                                          c_int, _P, _P, _P, _P, _P]),
    "spcl_bnrelu_backward_up2": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P,
                                         _P]),
    "spcl_bnrelu_backward_bcast": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P,
                                           _P, _P, _P]),
    "spcl_bnrelu_image_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "spcl_bnrelu_backward_image_wgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P,
                                                 c_int, _P, _P, _P, _P, _P]),
    "spcl_accumulate_scalars": (c_int, [c_int, _P, _P, _P, _P]),
    "spcl_stage_bytes": (c_int, [_P, _P, c_size_t, _P]),
    "spcl_conv_dgrad_bnstats_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_dgrad_bnstats": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_conv_dgrad_poolstats_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_dgrad_poolstats": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P,
                                             _P, _P, _P]),
    "spcl_bnrelu_pool_backward_rows": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P,
                                               c_int, _P, _P, _P, _P, _P]),
    "spcl_bnrelu_backward_rows": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P,
                                          c_int, _P, _P, _P, _P, _P, _P]),
    "spcl_image_autocorr_rows": (c_int, [c_int, c_int, c_int]),
    "spcl_image_autocorr": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "spcl_conv_dgrad_bnstats_image_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_forward_image_acorr_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_forward_image_acorr": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "spcl_conv16_bwd_fused_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv16_bwd_fused_splits": (c_int, [c_int, c_int, c_int]),
    "spcl_conv16_bwd_fused": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int,
                                      _P, _P, c_int, _P, _P]),
    "spcl_bnrelu_backward_wgrows_image3": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                                                   c_int, _P, _P, _P, _P, _P]),
    "spcl_conv3x3_dgrad_bnstats_image": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P,
                                                 _P, _P]),
    "spcl_bnrelu_image3_workspace_bytes": (c_size_t, [c_int]),
    "spcl_bnrelu_backward_rows_image3": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                                                 c_int, _P, _P, _P, _P, _P]),
    "spcl_conv1x1_forward": (c_int, [_P, c_int, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P]),
    "spcl_conv1x1_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "spcl_conv1x1_backward": (c_int, [_P, _P, c_int, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "spcl_conv1x1_forward_bn": (c_int, [_P, c_int, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "spcl_conv1x1_bwd_rows": (c_int, [c_size_t]),
    "spcl_conv1x1_backward_bn": (c_int, [_P, _P, c_int, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_softmax_forward": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "spcl_softmax_backward": (c_int, [_P, _P, c_size_t, c_int, _P, _P]),
    "spcl_kl_workspace_bytes": (c_size_t, []),
    "spcl_kl_div_forward": (c_int, [_P, _P, c_size_t, c_int, c_float, _P, _P, _P]),
    "spcl_kl_div_backward": (c_int, [_P, _P, c_size_t, c_int, c_float, _P, _P, _P]),
    "spcl_sup_loss_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P]),
    "spcl_one_hot": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "spcl_argmax_classes": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "spcl_dice_counts": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "spcl_upsample2x_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "spcl_upsample2x_backward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "spcl_concat2_channels": (c_int, [_P, _P, _P, c_int, c_size_t, c_int, c_int, _P]),
    "spcl_split2_channels": (c_int, [_P, _P, _P, c_int, c_size_t, c_int, c_int, _P]),
    "spcl_augment_views": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P]),
    "spcl_augment_views_recipe": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "spcl_augment_views_recipe_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "spcl_rows_linear_forward": (c_int, [_P, c_int, c_long, c_int, _P, _P, c_int, c_int, c_int, _P, _P]),
    "spcl_rows_linear_forward_act": (c_int, [_P, c_int, c_long, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "spcl_adaptive_avgpool2d_backward_act": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "spcl_rows_linear_backward_input": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, c_long, _P]),
    "spcl_rows_linear_backward_weight_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "spcl_rows_linear_backward_weight": (c_int, [_P, c_int, _P, c_int, c_long, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "spcl_augment_views_recipe_ws": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_size_t,
                                             _P]),
    "spcl_resize_bilinear_pil": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, _P]),
    "spcl_augment_views_pil": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P]),
    "spcl_flip_batch": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "spcl_flip_pair": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "spcl_flip_pair_stage": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, c_size_t, _P]),
    "spcl_profile_enable": (c_int, [c_int]),
    "spcl_profile_count": (c_int, []),
    "spcl_profile_get": (c_int, [c_int, c_char_p, c_int, _P, _P, _P]),
    "spcl_radam_step": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_double, c_double, c_double, c_double, _P, _P]),
    "spcl_radam_step_scalars": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_double, c_double, c_double, c_double, _P,
                                        c_int, _P, _P, _P, _P]),
    "spcl_radam_step_scaled": (c_int, [_P, _P, c_double, _P, _P, c_size_t, _P, _P, c_double, c_double, c_double, c_double,
                                       _P, c_int, _P, _P, _P, _P]),
    "spcl_bnrelu_gap_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "spcl_bnrelu_gap_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "spcl_bnrelu_backward_fill_acc": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P,
                                              _P, _P, _P]),
    "spcl_copy_pair": (c_int, [_P, _P, c_size_t, _P, _P, c_size_t, _P]),
    "spcl_bnrelu_backward_rows_acc": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P,
                                              _P, _P, _P]),
    "spcl_radam_apply_staged": (c_int, [_P, _P, c_double, _P, _P, c_size_t, _P, _P, c_double, c_double, c_double, c_double,
                                        c_int, _P, _P, _P, _P]),
    # BatchNorm sums as fixed-point accumulator blocks (csrc/bn_acc.hpp)
    "spcl_bn_acc_elems": (c_size_t, [c_int]),
    "spcl_conv_bn_acc_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_forward_acc": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_bnrelu_pool_forward_acc": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "spcl_conv_dgrad_bnstats_acc_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_dgrad_bnstats_acc": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spcl_conv_dgrad_poolstats_acc_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "spcl_conv3x3_dgrad_poolstats_acc": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P,
                                                 _P, _P, _P, _P]),
    "spcl_bnrelu_backward_acc": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P,
                                         _P]),
}


class WgradItem(ctypes.Structure):
    """``spcl_wgrad_item`` of include/spcl_hip.h (one layer of a batched weight-gradient launch)"""
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("in_scale", c_void_p), ("in_shift", c_void_p),
                ("dw_oihw", c_void_p), ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("CinS", c_int),
                ("Cout", c_int), ("CoutS", c_int), ("in_mode", c_int), ("x2", c_void_p), ("x_up2", c_int)]


class PackItem(ctypes.Structure):
    """``spcl_pack_item`` of include/spcl_hip.h (one layer of a multi-layer weight pack)"""
    _fields_ = [("w_oihw", c_void_p), ("fwd", c_void_p), ("dgrad", c_void_p), ("Cin", c_int), ("Cout", c_int),
                ("H", c_int), ("W", c_int)]


BN_EVAL_MAX = 32  # == SPCL_BN_EVAL_MAX


class BnEvalItem(ctypes.Structure):
    """``spcl_bn_eval_item`` of include/spcl_hip.h (one BatchNorm of a multi-layer eval-affine launch)"""
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("st", c_void_p), ("C", c_int), ("CS", c_int), ("eps", c_float)]


class BnAcc(ctypes.Structure):
    """``spcl_bn_acc`` of include/spcl_hip.h (a BatchNorm whose statistics travel as a fixed-point accumulator block)"""
    _fields_ = [("acc", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p),
                ("running_var", c_void_p), ("num_batches_tracked", c_void_p), ("st", c_void_p), ("momentum", c_float),
                ("eps", c_float), ("count", c_float), ("C", c_int), ("CS", c_int)]


PACK_MULTI_MAX = 24


class WgradTail(ctypes.Structure):
    """``spcl_wgrad_tail`` of include/spcl_hip.h (a weight gradient's pending final sum)"""
    _fields_ = [("partial", c_void_p), ("dw", c_void_p), ("kind", c_int), ("nsplit", c_int), ("nblk_ci", c_int),
                ("nblk_co", c_int), ("CIB", c_int), ("COB", c_int), ("Cin", c_int), ("Cout", c_int)]


ABI_VERSION = 9  # == SPCL_ABI_VERSION of include/spcl_hip.h (tests/test_abi.py compares them); lib() refuses any other library
WGRAD_BATCH_MAX = 16
WGRAD_TAILS_MAX = 16
_NO_STATUS = ("spcl_abi_version", "spcl_conv3x3_forward_image_acorr_rows", "spcl_image_autocorr_rows", "spcl_conv_dgrad_bnstats_image_supported", "spcl_conv16_bwd_fused_supported", "spcl_conv16_bwd_fused_splits", "spcl_conv_num_tiles", "spcl_conv_stat_rows", "spcl_conv_set_gemm", "spcl_conv_set_f32_split", "spcl_conv_get_f32_split", "spcl_supcon_unit_gradient_block", "spcl_conv_cat_supported", "spcl_conv_up2_supported", "spcl_conv_split_supported", "spcl_conv_split_bnstats_supported", "spcl_conv1x1_bwd_rows", "spcl_profile_count", "spcl_conv_dgrad_bnstats_supported", "spcl_conv_dgrad_poolstats_supported",
              "spcl_conv_wgrad_batched_supported", "spcl_conv_bn_acc_supported", "spcl_conv_dgrad_bnstats_acc_supported",
              "spcl_conv_dgrad_poolstats_acc_supported", "spcl_supcon_rows_supported", "spcl_bnrelu_gap_supported")


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load libspcl_hip.so (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `python self-paced-contrastive-learning_amd/build.py` "
                "(there is no CPU / eager fallback for the hot path)")
        L = ctypes.CDLL(LIB_PATH)
        L.spcl_abi_version.restype, L.spcl_abi_version.argtypes = c_int, []
        have = L.spcl_abi_version()
        if have != ABI_VERSION:
            raise NativeLibraryError(
                f"{LIB_PATH} has C-ABI version {have}, this binding was written for {ABI_VERSION} (include/spcl_hip.h "
                "SPCL_ABI_VERSION): a stale build -- rebuild with `python self-paced-contrastive-learning_amd/build.py`")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:
                continue  # reported by tests/test_abi.py; calling it raises below
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def call(name: str, *args):
    L = lib()
    fn = getattr(L, name, None)
    if fn is None:
        raise NativeLibraryError(f"libspcl_hip.so does not export {name}")
    rc = fn(*args)
    if fn.restype is c_int and name not in _NO_STATUS and rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.spcl_last_error().decode()}")
    return rc


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("self-paced-contrastive-learning_amd runs on MI355X only: got a CPU tensor "
                               "(the HIP path has no CPU fallback)")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_array(tensors):
    """host array of device pointers (None -> NULL), for the entry points that take K tensors of one kind"""
    return (c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return SPCL_F32
    if dt == torch.bfloat16:
        return SPCL_BF16
    raise TypeError(f"unsupported activation dtype {dt}")
