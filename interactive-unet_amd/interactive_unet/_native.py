"""ctypes binding of libiunet.so (the C ABI declared in include/iunet.h).

The library is pure HIP (no torch types in its signatures): tensors cross the boundary
as raw device pointers + sizes + the HIP stream to order the work on.  There is NO CPU
fallback: if the shared library is missing or a call fails, this module raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IUNET_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libiunet.so')   # IUNET_LIB: A/B builds

_lib = None

c_void_p, c_int, c_ll, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float

# name -> argtypes (all functions return int status except where noted)
_SIGS = {
    'iunet_abi_version': [],
    'iunet_conv3_num_tiles': [c_int] * 5,
    'iunet_conv3_stats_parts': [c_int] * 7,
    'iunet_pack_conv3': [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_pack_first_conv': [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'iunet_pack_convT': [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'iunet_pack_batch': [c_void_p, c_int, c_int, c_void_p],
    'iunet_conv3_fwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_conv3_pick_layout': [c_int] * 7,
    'iunet_conv3_tile_pairs': [c_int] * 7,
    'iunet_conv3_compact_ok': [c_int] * 9,
    'iunet_conv3_fwd_act': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_first_conv_fwd': [c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_void_p, c_void_p,
                             c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_maxpool_fwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_convT_fwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p,
                        c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_head_fwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                       ctypes.POINTER(c_ll), c_float, c_int, c_int, c_int, c_int, c_int, c_void_p],
    # ---- fp32 parity mode
    'iunet_f32_pack_conv': [c_void_p] * 7 + [c_float, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_f32_conv_fwd': [c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_f32_maxpool_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_f32_head_fwd': [c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                           ctypes.POINTER(c_ll), c_float, c_int, c_int, c_int, c_int, c_int, c_void_p],
    # ---- fp32 parity form of the training step (csrc/train_f32.hip)
    'iunet_f32_bn_stats': [c_void_p, c_ll, c_int, c_int, c_ll, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'iunet_f32_bn_relu_fwd': [c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_f32_bn_relu_bwd': [c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_ll, c_void_p],
    'iunet_f32_maxpool_bwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_f32_wgrad_splits': [c_int] * 7,
    'iunet_f32_wgrad': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_f32_head_loss_num_parts': [c_int, c_ll],
    'iunet_f32_head_loss_fwd': [c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_void_p, c_int, c_ll, c_void_p],
    'iunet_f32_head_loss_bwd': [c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_ll,
                                c_void_p, c_ll, c_int, c_ll, c_void_p],
    'iunet_f32_channel_sum': [c_void_p, c_ll, c_void_p, c_int, c_int, c_ll, c_void_p],
    # ---- fp16x2 split precision (the tolerance-meeting mode on the 16-bit matrix cores)
    'iunet_x2_prep': [c_void_p] * 9 + [c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2_first_conv_fwd': [c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p,
                                c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2_conv3_fwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2_maxpool_fwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2_convT_fwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2_head_fwd': [c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p,
                          ctypes.POINTER(c_ll), c_float, c_int, c_int, c_int, c_int, c_int, c_void_p],
    # ---- fp16x2 with the cross terms on the fp8 matrix cores (csrc/conv3_x2m.hip)
    'iunet_x2m_prep': [c_void_p] * 9 + [c_float, c_float, c_float, c_int, c_int, c_void_p],
    'iunet_x2m_prep_nd': [c_int] + [c_void_p] * 9 + [c_float, c_float, c_float, c_int, c_int, c_void_p],
    'iunet_x2_prep_batch': [c_void_p, c_int, c_int, c_void_p],
    'iunet_x2m_prep_batch': [c_void_p, c_int, c_int, c_void_p],
    'iunet_x2m_conv_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2m_conv_pool_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2m_first_stage_fwd': [c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_ll, c_int, c_void_p, c_ll,
                                  c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2m_conv_head_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_ll), c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2m_make8': [c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2m_first_conv_fwd': [c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                                 c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2m_convT_fwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                            c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2_conv3_fwd_flag': [c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_x2m_maxpool_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_x2m_conv3_fwd': [c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    # ---- handle level (csrc/net.hip): the whole forward sequenced in C++
    'iunet_f32_gn_relu_fwd': [c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_x2_gn_relu_fwd': [c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_ll, c_void_p, c_void_p],
    'iunet_x2m_gn_relu_fwd': [c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_ll, c_void_p, c_void_p],
    'iunet_logit_diff': [c_void_p, c_void_p, c_ll, c_void_p, c_void_p],
    'iunet_net_create': [c_int, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.POINTER(c_void_p)],
    'iunet_net_create_ex': [c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, ctypes.POINTER(c_void_p)],
    'iunet_net_num_tensors': [c_void_p],
    'iunet_net_param': [c_void_p, c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_ll), ctypes.POINTER(c_ll)],
    'iunet_net_load': [c_void_p, c_void_p, c_void_p, c_void_p],
    'iunet_net_forward': [c_void_p, c_void_p, c_int, ctypes.POINTER(c_ll), c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                          c_void_p, ctypes.POINTER(c_ll), c_float, c_int, c_void_p],
    'iunet_net_eval_step': [c_void_p, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                            c_void_p, c_void_p, c_void_p],
    'iunet_net_forward_argmax': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    # ---- training state on the device + the training step as one C call (csrc/train_net.hip)
    'iunet_train_state_init': [c_void_p, c_float, c_int, c_void_p],
    'iunet_head_loss_bwd_dev': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p],
    'iunet_head_grad_scatter': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    'iunet_adamw_step_dev': [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float, c_float, c_void_p, c_int, c_float, c_void_p],
    'iunet_train_create': [c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)],
    'iunet_train_create_ex': [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)],
    'iunet_train_num_tensors': [c_void_p],
    'iunet_train_param': [c_void_p, c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_ll), ctypes.POINTER(c_ll)],
    'iunet_train_num_bn': [c_void_p],
    'iunet_train_bind': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p],
    'iunet_train_repack': [c_void_p, c_void_p],
    'iunet_train_forward_backward': [c_void_p, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p],
    'iunet_train_forward_backward_hooks': [c_void_p, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'iunet_train_update': [c_void_p, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p],
    'iunet_train_step': [c_void_p, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                         c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p],
    # ---- fp8 matrix cores (config C5)
    'iunet_f8_pack_conv3': [c_void_p] * 5 + [c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'iunet_conv3_f8_fwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_conv3_f8_fwd_q': [c_int, c_int, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_first_conv_fwd_q': [c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_void_p, c_void_p,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_maxpool_q_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_convT_fwd_q': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p,
                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_gather_block': [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_blend_accumulate': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p],
    'iunet_normalize_quantize': [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_float, c_void_p],
    'iunet_div_f32': [c_void_p, c_ll, c_float, c_void_p],
    'iunet_colorize': [c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p],
    'iunet_slice_gather': [c_void_p, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int),
                           ctypes.POINTER(c_int), c_int, c_int, c_int, c_void_p, c_void_p],
    'iunet_slice_scatter': [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_double), c_int, c_int, c_void_p, c_void_p,
                            c_void_p],
    'iunet_augment_batch': [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'iunet_zoom_nearest_table': [c_int, ctypes.c_double, ctypes.POINTER(c_int), c_int],
    'iunet_zoom_nearest_u8': [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_ll), c_void_p, ctypes.POINTER(c_ll),
                              ctypes.POINTER(c_int), c_void_p, c_void_p],
    # ---- training
    'iunet_bn_finalize': [c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                          c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'iunet_bn_relu_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_bn_relu_pool_fwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p,
                               c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_bn_bwd_num_parts': [c_int, c_ll],
    'iunet_conv3_dgrad_bnstats': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_conv3_dgrad_bnstats_lay': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_head_bn_bwd_ok': [c_int, c_int],
    'iunet_head_bn_bwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p],
    'iunet_bn_relu_bwd_apply': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_gn_num_parts': [c_int, c_ll],
    'iunet_gn_finalize': [c_void_p, c_int, c_int, c_int, c_ll, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'iunet_gn_relu_fwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_gn_relu_fwd_rows': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_gn_relu_pool_fwd_rows': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_float, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_conv3_sample_stats_rows': [c_int] * 9,
    'iunet_conv3_fwd_sample_stats': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_void_p],
    'iunet_conv3_dgrad_sample_bnstats': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_gn_relu_bwd_rows': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_gn_relu_bwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_gn_relu_pool_fwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_float, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_gn_relu_pool_bwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_bn_relu_bwd': [c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_bn_relu_pool_bwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_int, c_int, c_void_p],
    'iunet_maxpool_bwd': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int,
                          c_int, c_int, c_void_p],
    'iunet_head_loss_num_parts': [c_int, c_ll],
    'iunet_head_loss_bwd_num_parts': [c_int, c_ll, c_int, c_int],
    'iunet_head_loss_fwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p],
    'iunet_head_loss_bwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                            c_void_p, c_float, c_void_p, c_ll, c_void_p, c_int, c_ll, c_void_p],
    'iunet_head_loss_fwd_act': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p],
    'iunet_head_loss_fwd_act_ps': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p],
    'iunet_head_gn_bwd': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p],
    'iunet_head_loss_bwd_act': [c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                c_void_p, c_float, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_void_p],
    'iunet_reduce_slab': [c_void_p, c_int, c_ll, c_void_p, c_float, c_int, c_void_p],
    'iunet_check_finite': [c_void_p, c_ll, c_void_p, c_void_p],
    'iunet_adamw_step': [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float, c_float,
                         c_int, c_float, c_void_p, c_void_p],
    'iunet_conv3_wgrad_blocks': [c_int] * 7,
    'iunet_conv3_wgrad': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_float,
                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_conv3_wgrad_act': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                              c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_pack_convT_dgrad': [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'iunet_convT_dgrad': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_void_p],
    'iunet_convT_wgrad_blocks': [c_int] * 7,
    'iunet_convT_wgrad': [c_int, c_int, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'iunet_first_conv_wgrad_blocks': [c_int] * 5,
    'iunet_first_conv_wgrad_bn': [c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_void_p, c_ll, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_void_p],
    'iunet_first_conv_wgrad': [c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_ll), c_void_p, c_ll, c_void_p, c_void_p,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
}
# functions that return a size / count instead of a status
_INT_RETURN = ['iunet_pack_desc_bytes', 'iunet_augment_desc_bytes', 'iunet_x2_prep_desc_bytes']
_INT_RETURN_ARGS = {'iunet_zoom_nearest_len': [c_int, ctypes.c_double], 'iunet_x2_convT_kc': [c_int], 'iunet_x2m_head_fusable': [c_int, c_int], 'iunet_x2m_pool_fusable': [c_int, c_int], 'iunet_x2m_first_stage_fusable': [c_int] * 6, 'iunet_x2_pack_mode': [c_int], 'iunet_f8_pack_order': [c_int, c_int]}
_LL_RETURN = {'iunet_gn_precise_slab_bytes': [c_int, c_int, c_ll], 'iunet_net_eval_scratch_bytes': [c_void_p, c_int, c_int, c_int, c_int], 'iunet_x2m_w8_bytes': [c_int] * 2, 'iunet_x2m_w8_bytes_nd': [c_int] * 3, 'iunet_train_num_params': [c_void_p], 'iunet_train_packed_bytes': [c_void_p], 'iunet_train_workspace_bytes': [c_void_p, c_int, c_int, c_int, c_int], 'iunet_conv3_wgrad_slab_floats': [c_int] * 7, 'iunet_f32_pack_conv_elems': [c_int] * 3, 'iunet_f8_pack_conv3_bytes': [c_int] * 3, 'iunet_conv3_f8_workspace_elems': [c_int] * 7, 'iunet_pack_conv3_elems': [c_int] * 4,
              'iunet_pack_first_conv_elems': [c_int] * 3, 'iunet_slice_scatter_workspace_bytes': [c_int],
              'iunet_net_num_params': [c_void_p], 'iunet_net_packed_bytes': [c_void_p], 'iunet_net_workspace_bytes': [c_void_p] + [c_int] * 4}


class NativeError(RuntimeError):
    pass


def lib():
    """Load libiunet.so once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise NativeError(f'{LIB_PATH} is missing: build it with __graft_entry__.build() '
                              f'(interactive-unet_amd/csrc/build.sh); there is no CPU fallback')
        l = ctypes.CDLL(LIB_PATH)
        l.iunet_last_error.restype = ctypes.c_char_p
        l.iunet_last_error.argtypes = []
        for name, args in _SIGS.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.argtypes = args
            fn.restype = c_int
        for name in _INT_RETURN:
            getattr(l, name).restype = c_int
            getattr(l, name).argtypes = []
        for name, args in _INT_RETURN_ARGS.items():
            getattr(l, name).restype = c_int
            getattr(l, name).argtypes = args
        for name, args in _LL_RETURN.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = c_ll
        l.iunet_net_destroy.argtypes, l.iunet_net_destroy.restype = [c_void_p], None
        l.iunet_train_destroy.argtypes, l.iunet_train_destroy.restype = [c_void_p], None
        _lib = l
    return _lib


def exported_symbols():
    return ['iunet_last_error', 'iunet_net_destroy', 'iunet_train_destroy'] + list(_SIGS) + list(_LL_RETURN) + list(_INT_RETURN) + list(_INT_RETURN_ARGS)


def check(status):
    if status != 0:
        raise NativeError(f'libiunet error {status}: {lib().iunet_last_error().decode()}')


def call(name, *args):
    check(getattr(lib(), name)(*args))


def pack_conv3_elems(cout, cin, taps, mode=0):
    return int(lib().iunet_pack_conv3_elems(cout, cin, taps, mode))


class PackDesc(ctypes.Structure):
    """One layer of iunet_pack_batch (mirror of csrc/pack_batch.hip: PackDesc)."""
    _fields_ = [('w', c_void_p), ('gamma', c_void_p), ('beta', c_void_p), ('mean', c_void_p), ('var', c_void_p),
                ('bias_out', c_void_p), ('dst', c_void_p), ('total', c_ll), ('Cout', c_int), ('Cin', c_int),
                ('taps', c_int), ('kind', c_int), ('dgrad', c_int), ('dtype', c_int), ('eps', c_float), ('pad_', c_int),
                ('qscale', c_void_p)]


def make_desc(w, dst, cout, cin, taps, kind, dtype, dgrad=0, bn=None, bias_out=None, eps=1e-5, qscale=None):
    d = PackDesc()
    d.w, d.dst, d.total = w.data_ptr(), dst.data_ptr(), dst.numel()
    d.Cout, d.Cin, d.taps, d.kind, d.dgrad, d.dtype, d.eps = cout, cin, taps, kind, int(dgrad), DTYPE_CODE[dtype], eps
    if bn is not None:
        d.gamma, d.beta, d.mean, d.var = [t.data_ptr() for t in bn]
        d.bias_out = None if bias_out is None else bias_out.data_ptr()
    d.qscale = None if qscale is None else qscale.data_ptr()
    return d


class PackTable:
    """Descriptor table of iunet_pack_batch in device memory; `sources` keeps the tensors whose addresses it holds."""

    def __init__(self, descs, device, sources=()):
        assert lib().iunet_pack_desc_bytes() == ctypes.sizeof(PackDesc), 'PackDesc layout mismatch with libiunet'
        arr = (PackDesc * len(descs))(*descs)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = host.to(device)
        self.n = len(descs)
        self.sources = list(sources)
        self.quant_max_cout = max([d.Cout for d in descs if d.qscale] + [0])

    def run(self):
        call('iunet_pack_batch', ptr(self.dev), self.n, self.quant_max_cout, stream())


class X2PrepDesc(ctypes.Structure):
    """One operator of iunet_x2_prep_batch / iunet_x2m_prep_batch (mirror of csrc/x2_prep_desc.h)."""
    _fields_ = [('w', c_void_p), ('out', c_void_p), ('w8', c_void_p), ('oscale', c_void_p), ('bias_out', c_void_p), ('gamma', c_void_p),
                ('beta', c_void_p), ('mean', c_void_p), ('var', c_void_p), ('bias_in', c_void_p), ('eps', c_float), ('act_in', c_float),
                ('act_out', c_float), ('Cout', c_int), ('Cin', c_int), ('taps', c_int), ('kind', c_int), ('kc', c_int), ('row0', c_int)]


def make_x2_prep_desc(w, out, oscale, bias_out, cout, cin, taps, kind, kc, act_in, act_out, bn=None, bias_in=None, w8=None, eps=1e-5):
    d = X2PrepDesc()
    d.w, d.out, d.oscale, d.bias_out = w.data_ptr(), out.data_ptr(), oscale.data_ptr(), bias_out.data_ptr()
    d.w8 = None if w8 is None else w8.data_ptr()
    if bn is not None:
        d.gamma, d.beta, d.mean, d.var = [t.data_ptr() for t in bn]
    d.bias_in = None if bias_in is None else bias_in.data_ptr()
    d.eps, d.act_in, d.act_out = eps, act_in, act_out
    d.Cout, d.Cin, d.taps, d.kind, d.kc = cout, cin, taps, kind, kc
    return d


class X2PrepTable:
    """Descriptor table of iunet_x2_prep_batch (x2m=False) / iunet_x2m_prep_batch (x2m=True) in device memory."""

    def __init__(self, descs, device, x2m):
        assert lib().iunet_x2_prep_desc_bytes() == ctypes.sizeof(X2PrepDesc), 'X2PrepDesc layout mismatch with libiunet'
        rows = 0
        for d in descs:
            d.row0 = rows
            rows += d.Cout
        arr = (X2PrepDesc * len(descs))(*descs)
        self.dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.n, self.rows = len(descs), rows
        self.fn = 'iunet_x2m_prep_batch' if x2m else 'iunet_x2_prep_batch'

    def run(self):
        call(self.fn, ptr(self.dev), self.n, self.rows, stream())


class PackedConv:
    """A stage conv's weights in the fragment order(s) its launches may need: layout 1 (K16, the
    LDS-fed Cout-32 structure) always, layout 0 also when Cout is a multiple of 64.  `dgrad`: the
    data-gradient operator (roles of cin / cout swapped)."""

    def __init__(self, cout, cin, taps, dtype, device, dgrad=False):
        self.cout, self.cin, self.taps, self.dg = cout, cin, taps, int(bool(dgrad))
        self.out_ch = cin if dgrad else cout
        self.dt = DTYPE_CODE[dtype]
        self.buf = {1: torch.empty(pack_conv3_elems(cout, cin, taps, 2 | self.dg), dtype=dtype, device=device)}
        # layout 0 is only ever picked for 2-D launches with more than 64 input channels (iunet_conv3_pick_layout: every 3-D conv
        # and the narrow 2-D ones run on the K16 operator): packing it for the others was half of the per-step pack work
        in_ch = cout if dgrad else cin
        # layout 3: the compact K16 order (the padding-free step of conv3_v4.hip: 3^3 filters with streamed weights, every 3^2 filter)
        # beside the padded one, which the launches that do not qualify keep using (fused BatchNorm-backward sums)
        compact2d = taps == 9 and not os.environ.get('IUNET_NO_COMPACT2D')
        if self.out_ch % 64 == 0 and taps == 9 and in_ch > 64 and not compact2d:
            self.buf[0] = torch.empty(pack_conv3_elems(cout, cin, taps, self.dg), dtype=dtype, device=device)
        if ((taps == 27 and in_ch > 32) or compact2d) and not os.environ.get('IUNET_NO_COMPACT'):
            self.buf[3] = torch.empty(pack_conv3_elems(cout, cin, taps, 6 | self.dg), dtype=dtype, device=device)

    def pack(self, w, scale=None):
        for lay, b in self.buf.items():
            call('iunet_pack_conv3', self.dt, ptr(w), ptr(scale), ptr(b), self.cout, self.cin, self.taps,
                 (6 if lay == 3 else 2 if lay == 1 else 0) | self.dg, stream())

    def descs(self, w, bn=None, bias_out=None, eps=1e-5, qscale=None):
        """Descriptors of all layouts for iunet_pack_batch (the first one also writes the folded bias)."""
        out = []
        for k, (lay, b) in enumerate(sorted(self.buf.items(), reverse=True)):
            out.append(make_desc(w, b, self.cout, self.cin, self.taps, 6 if lay == 3 else 1 if lay == 1 else 0, b.dtype, self.dg, bn,
                                 bias_out if k == 0 else None, eps, qscale))
        return out

    def pick(self, nd, N, D, H, W, act=False, bw=False):
        """(layout, buffer) for a launch on this grid.  act: the launch applies a fused input activation (iunet_conv3_fwd_act);
        bw: it accumulates the BatchNorm-backward sums (iunet_conv3_dgrad_bnstats)."""
        in_ch = self.cout if self.dg else self.cin
        lay = lib().iunet_conv3_pick_layout(nd, N, D, H, W, in_ch, self.out_ch)
        # the fused sums exist where layout 2 is the grid's choice (and, in 2-D, on the compact operator of those launches): a launch that
        # asks for them elsewhere (2-D, more than 64 input channels) runs plain, so it may as well run on the compact operator
        bw = bool(bw) and lay == 2
        if 3 in self.buf and lib().iunet_conv3_compact_ok(nd, N, D, H, W, in_ch, self.out_ch, int(bool(act)), int(bw)):
            return 3, self.buf[3]
        if lay == 0 and 0 not in self.buf:
            lay = 1
        return lay, self.buf[1 if lay == 2 else lay]      # layout 2 runs on the K16 operator of layout 1


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ll_array(vals):
    return (c_ll * len(vals))(*[int(v) for v in vals])


def int_array(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])


DTYPE_CODE = {torch.float16: 0, torch.bfloat16: 1}
IN_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.uint8: 2, torch.bfloat16: 3}
