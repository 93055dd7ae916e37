"""ctypes binding of librnet_hip.so (C ABI declared in include/rnet_hip.h).

The library is the product: there is no CPU fallback.  `lib()` raises if the shared object
is missing, and every call raises `RnetError` on a non-zero status.
"""
from __future__ import annotations

import ctypes
import logging
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_uint32,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RNET_HIP_LIB: alternative build of the same ABI (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("RNET_HIP_LIB") or os.path.join(_HERE, "librnet_hip.so")
LIB_PATH_F16 = os.environ.get("RNET_HIP_LIB_F16") or os.path.join(_HERE, "librnet_hip_f16.so")

RN_DT_F32, RN_DT_BF16 = 0, 1
RN_OK, RN_EINVAL, RN_ENOMEM, RN_EHIP, RN_ECOMM, RN_EUNSUPPORTED = 0, -1, -2, -3, -4, -5   # rn_status
RN_ACT_NONE, RN_ACT_RELU, RN_ACT_RELU6, RN_ACT_SWISH = 0, 1, 2, 3
RN_CONV_MAX_SEGMENTS = 10
ABI_VERSION = 8
# f32 kernels of the dtype=float32 prediction convs (detection_head.py:80-88) as split-bf16 planes (rn_conv_segment.w_terms)
PRED_W_TERMS = int(os.environ.get("RNET_PRED_W_TERMS", "2"))
ACT_IDS = {None: RN_ACT_NONE, "none": RN_ACT_NONE, "relu": RN_ACT_RELU, "relu6": RN_ACT_RELU6,
           "swish": RN_ACT_SWISH}


class RnetError(RuntimeError):
    pass


class DgradPack(Structure):   # rn_dgrad_pack
    _fields_ = [("w_ohwi", c_void_p), ("w_packed", c_void_p), ("R", c_int32), ("S", c_int32), ("Cin", c_int32),
                ("Cout", c_int32), ("Cout_pad", c_int32), ("pad_", c_int32)]


class ExampleInfo(Structure):   # rn_example_info
    _fields_ = [("image_offset", c_uint64), ("image_length", c_uint64), ("image_id", c_int64),
                ("n_xmins", c_int32), ("n_ymins", c_int32), ("n_xmaxs", c_int32), ("n_ymaxs", c_int32),
                ("n_classes", c_int32), ("pad_", c_int32)]


class LaunchOpts(Structure):   # rn_launch_opts: per-call options of the MFMA kernels (all zero = the dispatcher's choice)
    _fields_ = [("conv_tile", c_int32), ("conv_no_halo", c_int32), ("conv_big_min_tiles", c_int32),
                ("max_workgroups", c_int32), ("reserved_cus", c_int32), ("wgrad_kernel", c_int32),
                ("wgrad_target_blocks", c_int32), ("ablate", c_int32), ("splitk_target_blocks", c_int32)]

    def __init__(self, **kw):
        super().__init__()
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError(f"rn_launch_opts has no field {k!r}")
            setattr(self, k, int(v))

    def copy(self, **kw):
        o = LaunchOpts(**{k: getattr(self, k) for k, _ in self._fields_})
        for k, v in kw.items():
            setattr(o, k, int(v))
        return o


class Bottleneck64Problem(Structure):   # rn_bottleneck64_problem
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("w_packed", c_void_p), ("affine", c_void_p), ("N", c_int32), ("H", c_int32),
                ("W", c_int32), ("Cx", c_int32), ("opts", LaunchOpts)]


class ConvSegment(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("residual", c_void_p), ("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32),
                ("pix_stride", c_int32), ("Ho", c_int32), ("Wo", c_int32), ("Cout", c_int32), ("bn_partial", c_void_p),
                ("bias", c_void_p), ("w_terms", c_int32), ("w_pair", c_int32), ("bn_bwd_y", c_void_p),
                ("bn_bwd_fwd", c_void_p)]


class ConvProblem(Structure):
    _fields_ = [("R", c_int32), ("S", c_int32), ("stride_h", c_int32), ("stride_w", c_int32),
                ("pad_top", c_int32), ("pad_left", c_int32), ("act", c_int32), ("out_dtype", c_int32),
                ("num_segments", c_int32), ("seg", ConvSegment * RN_CONV_MAX_SEGMENTS), ("opts", LaunchOpts),
                ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_int64)]


class WgradSegment(Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("N", c_int32), ("H", c_int32), ("W", c_int32),
                ("Cin", c_int32), ("Ho", c_int32), ("Wo", c_int32), ("Cout", c_int32),
                ("dy_pix_stride", c_int32), ("x_pix_stride", c_int32)]


class WgradProblem(Structure):
    _fields_ = [("R", c_int32), ("S", c_int32), ("stride_h", c_int32), ("stride_w", c_int32),
                ("pad_top", c_int32), ("pad_left", c_int32), ("num_segments", c_int32),
                ("seg", WgradSegment * RN_CONV_MAX_SEGMENTS), ("opts", LaunchOpts)]


class DwSegment(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("residual", c_void_p),
                ("N", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("Ho", c_int32), ("Wo", c_int32)]


class DwProblem(Structure):
    _fields_ = [("k", c_int32), ("stride", c_int32), ("pad_top", c_int32), ("pad_left", c_int32), ("act", c_int32),
                ("num_segments", c_int32), ("seg", DwSegment * RN_CONV_MAX_SEGMENTS)]


class BnSegment(Structure):
    _fields_ = [("y", c_void_p), ("z", c_void_p), ("residual", c_void_p), ("dz", c_void_p), ("dy", c_void_p),
                ("dres", c_void_p), ("sums", c_void_p), ("fwd", c_void_p), ("bsums", c_void_p),
                ("gamma", c_void_p), ("beta", c_void_p), ("moving_mean", c_void_p), ("moving_var", c_void_p),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("P", c_int64), ("C", c_int32),
                ("dres_accumulate", c_int32), ("sample_scale", c_void_p), ("rows_per_sample", c_int64),
                ("ext_chunks", c_int32), ("ext_chunks_bwd", c_int32), ("act_mask", c_void_p),
                ("dy_colsum_partial", c_void_p)]


class BnProblem(Structure):
    _fields_ = [("num_segments", c_int32), ("act", c_int32), ("bessel", c_int32), ("eps", c_float),
                ("momentum", c_float), ("count_scale", c_float), ("seg", BnSegment * RN_CONV_MAX_SEGMENTS)]


_PP = POINTER(c_void_p)
_SIGNATURES = {
    "rn_last_error": (c_char_p, []),
    "rn_abi_version": (c_int, []),
    "rn_storage_dtype": (c_int, []),
    "rn_device_ok": (c_int, []),
    "rn_anchors_generate": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, POINTER(c_float),
                                    POINTER(c_float), c_int, POINTER(c_float), c_int, POINTER(c_int64),
                                    c_void_p]),
    "rn_match_workspace_bytes": (c_size_t, [c_int, c_int]),
    "rn_anchor_match_encode": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                       c_float, POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "rn_loss_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "rn_retinanet_loss_fwd_bwd": (c_int, [_PP, _PP, _PP, _PP, POINTER(c_int64), c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                          c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_retinanet_loss_fwd_bwd_bf16": (c_int, [_PP, _PP, _PP, _PP, c_int, c_int, c_int, POINTER(c_int64), c_int, c_int,
                                               c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                               c_float, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_decode_boxes": (c_int, [_PP, POINTER(c_int64), c_int, c_int, c_void_p, POINTER(c_float), c_float,
                                c_float, c_void_p, c_void_p]),
    "rn_sigmoid_scores": (c_int, [_PP, POINTER(c_int64), c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_topk_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "rn_topk_per_class": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "rn_detect_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "rn_detect_per_class": (c_int, [_PP, POINTER(c_int64), c_int, c_int, c_int, c_void_p, c_int, c_float,
                                    c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "rn_nms_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rn_nms_per_class": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_rowmax_argmax": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "rn_conv2d_nhwc_fwd": (c_int, [POINTER(ConvProblem), c_void_p]),
    "rn_conv_cout_pad": (c_int, [c_int]),
    "rn_conv_tile_rows": (c_int, [POINTER(ConvProblem)]),
    "rn_conv_bn_row_blocks": (c_int, [POINTER(ConvProblem), c_int]),
    "rn_probe_spin": (c_int, [c_int, c_void_p]),
    "rn_conv_kernel_id": (c_int, [POINTER(ConvProblem)]),
    "rn_conv_splitk_workspace_bytes": (c_size_t, [POINTER(ConvProblem)]),
    "rn_conv_splitk_workspace_max_bytes": (c_size_t, []),
    "rn_conv_cin_pad": (c_int, [c_int]),
    "rn_depthwise_conv2d_nhwc_fwd": (c_int, [POINTER(DwProblem), c_void_p]),
    "rn_pack_depthwise_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rn_pack_depthwise_weight_flip": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rn_depthwise_wgrad_workspace_bytes": (c_size_t, [POINTER(DwProblem)]),
    "rn_depthwise_conv2d_nhwc_wgrad": (c_int, [POINTER(DwProblem), c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_squeeze_excite_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_void_p, c_size_t, c_void_p]),
    "rn_squeeze_excite_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_coco_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_int, c_int, c_int,
                                   c_int, c_void_p, c_void_p, c_void_p]),
    "rn_pack_conv_weight_ohwi": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_se_workspace_bytes": (c_size_t, [c_int, c_int]),
    "rn_squeeze_excite_inplace": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_void_p, c_size_t, c_void_p]),
    "rn_pack_conv_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_conv_pair_rows": (c_int, [c_int]),
    "rn_pack_conv_weight_pair": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_pack_conv_weight_split": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_void_p]),
    "rn_pack_stem_weight": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "rn_stem_padded_width": (c_int, [c_int]),
    "rn_pack_stem_input": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_pack_stem_weight_rs": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_pack_image_nhwc4": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_wgrad_workspace_bytes": (c_size_t, [POINTER(WgradProblem)]),
    "rn_wgrad_kernel_id": (c_int, [POINTER(WgradProblem)]),
    "rn_conv2d_nhwc_wgrad": (c_int, [POINTER(WgradProblem), c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "rn_wgrad_group_workspace_bytes": (c_size_t, [POINTER(POINTER(WgradProblem)), c_int]),
    "rn_wgrad_group_fused": (c_int, [POINTER(POINTER(WgradProblem)), c_int]),
    "rn_conv2d_nhwc_wgrad_group": (c_int, [POINTER(POINTER(WgradProblem)), c_int, _PP, c_float, c_void_p, c_size_t,
                                           c_void_p]),
    "rn_pack_conv_weight_dgrad": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_pack_conv_weight_dgrad_batch": (c_int, [POINTER(DgradPack), c_int, c_void_p]),
    "rn_cast_pad_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rn_reduce_rows_f32": (c_int, [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "rn_upsample_zero2x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rn_scatter_add2x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rn_depth_to_space2x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rn_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rn_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rn_bn_workspace_bytes": (c_size_t, [POINTER(BnProblem)]),
    "rn_bn_workspace_init": (c_int, [POINTER(BnProblem), c_void_p, c_size_t, c_void_p]),
    "rn_bn_partial_offset_bytes": (c_size_t, [POINTER(BnProblem), c_int]),
    "rn_bn_bwd_partial_offset_bytes": (c_size_t, [POINTER(BnProblem), c_int]),
    "rn_bn_stats": (c_int, [POINTER(BnProblem), c_void_p, c_size_t, c_void_p]),
    "rn_bn_stats_finalize": (c_int, [POINTER(BnProblem), c_void_p, c_size_t, c_void_p]),
    "rn_bn_finalize": (c_int, [POINTER(BnProblem), c_void_p]),
    "rn_bn_apply": (c_int, [POINTER(BnProblem), c_void_p]),
    "rn_bn_bwd_reduce": (c_int, [POINTER(BnProblem), c_void_p, c_size_t, c_void_p]),
    "rn_bn_bwd_apply": (c_int, [POINTER(BnProblem), c_void_p]),
    "rn_bn_bwd_colsum_chunks": (c_int, [POINTER(BnProblem), c_int]),
    "rn_maxpool2d_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rn_fpn_topdown_bwd_level": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "rn_balance_features_bwd_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "rn_balance_features_bwd": (c_int, [_PP, _PP, _PP, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p]),
    "rn_optim_chunk": (c_int, []),
    "rn_optim_workspace_bytes": (c_size_t, [c_int, c_int]),
    "rn_optim_clip": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_float, c_float, c_float,
                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_optim_clip_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "rn_optim_clip_factors": (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "rn_optim_clip_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "rn_optim_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "rn_prepare_image": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, POINTER(c_float),
                                 POINTER(c_float), c_float, c_void_p]),
    "rn_bottleneck64_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rn_bottleneck64_packed_bytes": (c_size_t, [c_int]),
    "rn_bottleneck64_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rn_bottleneck64_fwd": (c_int, [POINTER(Bottleneck64Problem), c_void_p]),
    "rn_stem_conv_bn_relu_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 14 + [c_void_p]),
    "rn_maxpool2d_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_void_p]),
    "rn_fpn_topdown": (c_int, [_PP, _PP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rn_balance_features": (c_int, [_PP, _PP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rn_create": (c_int, [c_int, POINTER(c_void_p)]),
    "rn_destroy": (c_int, [c_void_p]),
    "rn_handle_device": (c_int, [c_void_p]),
    "rn_handle_num_cus": (c_int, [c_void_p]),
    "rn_handle_set_launch_opts": (c_int, [c_void_p, POINTER(LaunchOpts)]),
    "rn_handle_get_launch_opts": (c_int, [c_void_p, POINTER(LaunchOpts)]),
    "rn_handle_comm_init": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int]),
    "rn_handle_comm": (c_void_p, [c_void_p, c_int]),
    "rn_handle_comm_destroy": (c_int, [c_void_p, c_int]),
    "rn_comm_available": (c_int, []),
    "rn_comm_unique_id_bytes": (c_int, []),
    "rn_comm_unique_id": (c_int, [c_void_p]),
    "rn_comm_init": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "rn_comm_destroy": (c_int, [c_void_p]),
    "rn_allreduce_bucket": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rn_allreduce_small": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "rn_probe_mfma_flops": (c_longlong, [c_int]),
    "rn_probe_mfma": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rn_jpeg_info": (c_int, [c_void_p, c_size_t, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "rn_jpeg_decode": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t]),
    "rn_jpeg_idct_islow": (c_int, [c_void_p, c_void_p]),
    # host functions (TFRecord input format)
    "rn_crc32c": (c_uint32, [c_void_p, c_size_t]),
    "rn_crc32c_masked": (c_uint32, [c_void_p, c_size_t]),
    "rn_tfrecord_scan": (c_longlong, [c_void_p, c_size_t, c_void_p, c_void_p, c_longlong, c_int, c_int,
                                      POINTER(c_size_t)]),
    "rn_tfrecord_frame": (c_size_t, [c_void_p, c_size_t, c_void_p]),
    "rn_example_parse": (c_int, [c_void_p, c_size_t, POINTER(ExampleInfo), c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int]),
    "rn_example_serialize": (c_size_t, [c_void_p, c_size_t, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                        c_size_t]),
}

_libs = {}


def exported_symbols():
    """Names include/rnet_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(_SIGNATURES)


def lib(f16=False):
    """librnet_hip.so (bfloat16 storage) or, f16=True, librnet_hip_f16.so: the same sources built with -DRN_F16 (IEEE
    half storage + v_mfma_f32_32x32x16_f16) for the `mixed_float16` configs.  Kernels without 16-bit tensors (anchors,
    matching, loss, post-processing, pre-processing) are identical in both."""
    f16 = bool(f16)
    if f16 not in _libs:
        path = LIB_PATH_F16 if f16 else LIB_PATH
        if not os.path.exists(path):
            raise RnetError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch first: it ships its own HIP runtime, and a process must end up with ONE (librnet_hip.so then binds to
        # the copy that is already loaded; the other order gives two runtimes and "no ROCm-capable device" on launch)
        import torch  # noqa: F401
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.rn_abi_version() != ABI_VERSION:
            raise RnetError(f"{path}: ABI version {handle.rn_abi_version()}, this binding needs {ABI_VERSION}: rebuild")
        if handle.rn_storage_dtype() != (1 if f16 else 0):
            raise RnetError(f"{path}: built for the other 16-bit storage type: rebuild")
        _libs[f16] = handle
    return _libs[f16]


class Handle:
    """rn_handle (include/rnet_hip.h): per-device context of one engine — device id, CU count, the
    engine's default rn_launch_opts, and the native communicators created through it (destroyed with it)."""

    def __init__(self, library, device_id, opts=None):
        self.lib = library
        self.h = c_void_p()
        check(library.rn_create(int(device_id), ctypes.byref(self.h)), "rn_create")
        if opts is not None:
            self.set_launch_opts(opts)

    def set_launch_opts(self, opts):
        check(self.lib.rn_handle_set_launch_opts(self.h, ctypes.byref(opts)), "rn_handle_set_launch_opts")

    def launch_opts(self):
        o = LaunchOpts()
        check(self.lib.rn_handle_get_launch_opts(self.h, ctypes.byref(o)), "rn_handle_get_launch_opts")
        return o

    @property
    def device(self):
        return self.lib.rn_handle_device(self.h)

    @property
    def num_cus(self):
        return self.lib.rn_handle_num_cus(self.h)

    def comm_init(self, slot, unique_id, rank, world):
        check(self.lib.rn_handle_comm_init(self.h, int(slot), unique_id, int(rank), int(world)), "rn_handle_comm_init")
        return c_void_p(self.lib.rn_handle_comm(self.h, int(slot)))

    def close(self):
        if self.h:
            self.lib.rn_destroy(self.h)
            self.h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check(status: int, what: str = ""):
    if status != 0:
        msgs = [h.rn_last_error() for h in _libs.values()]   # the failing call went to one of the loaded builds
        msg = b" | ".join(m for m in msgs if m)
        raise RnetError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else c_void_p(t.data_ptr())


def ptr_array(tensors):
    """Host array of addresses for a `const T* const*` parameter (None -> NULL array)."""
    if tensors is None:
        return None
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return ctypes.cast(arr, _PP)


def f32_array(values):
    if values is None:
        return None
    return (c_float * len(values))(*[float(v) for v in values])


def i64_array(values):
    return (c_int64 * len(values))(*[int(v) for v in values])


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def wait_blocks(lib_, a, probe, main, helper, spin_us=600, pre=None):
    """True when an event-wait pending on torch stream `a` stalls `probe()` — a callable that enqueues something short
    from stream `main` (a kernel, a c10d collective) — i.e. when whatever `probe` uses shares a hardware queue with `a`.
    HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) by creation order.  Kernels of two
    streams on one queue still overlap, but a `wait_event` is a barrier packet: everything submitted to that QUEUE
    afterwards — the other stream's kernels included — waits until it is satisfied.  Round 5 measured what that does to
    the data-parallel step (profiles/r05_ab): SyncBN all-reduces hop main -> c10d's stream -> main, the weight-gradient
    stream is full of "wait for the main stream" packets, and when the two shared a queue every message waited for the
    weight gradients queued before it: +6.4 ms per step inside bench.py, +1.2 ms in a fresh process (other queue map).
    `helper`: a third stream that idles for `spin_us` and then releases the wait; `pre()`: enqueued behind the wait, in front
    of the probe (keep it warm: its host time counts against the 0.5 * spin_us threshold).  Synchronises the device."""
    import time
    import torch
    # the timed interval contains the HOST time of pre() and probe() — a first-call lazy init or a slow c10d
    # enqueue read as "blocked".  So: one untimed warm call of both (always issued — `probe` may be a collective, every
    # rank must run the same sequence), its host time measured, and the spin stretched so that the threshold (half the
    # spin) stays at least 4 x that enqueue time.  The stretch is rank-local but only changes a kernel's duration, never
    # the number of collectives.
    t0 = time.perf_counter()
    if pre is not None:
        pre()
    probe()
    host_us = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize(a.device)
    t0 = time.perf_counter()
    if pre is not None:
        pre()
    probe()
    host_us = min(host_us, (time.perf_counter() - t0) * 1e6)   # the second call is the warm one
    spin_us = int(min(max(spin_us, 8.0 * host_us), 20000))
    torch.cuda.synchronize(a.device)
    gate = torch.cuda.Event()
    check(lib_.rn_probe_spin(int(spin_us), c_void_p(helper.cuda_stream)), "rn_probe_spin")
    gate.record(helper)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)                          # BEFORE the barrier goes in: on a shared queue a later record waits behind it too
    a.wait_event(gate)                       # a's queue now holds a barrier that stays shut for ~spin_us
    if pre is not None:
        pre()
    probe()
    e1.record(main)
    torch.cuda.synchronize(a.device)
    took_us = e0.elapsed_time(e1) * 1e3
    blocked = took_us > 0.5 * spin_us
    logging.getLogger("retinanet").debug("wait_blocks: spin %d us, host enqueue %.0f us, probe took %.0f us -> %s",
                                         spin_us, host_us, took_us, "blocked" if blocked else "free")
    return blocked


def concurrent_stream(lib_, device, others, attempts=12, probes=(), agree=None):
    """A new torch stream on `device` whose pending event-waits stall neither a kernel on any stream of `others` (and vice
    versa) nor any of `probes` (callables enqueued from others[0], e.g. a small c10d all-reduce) — or the last candidate
    when none of `attempts` qualifies.  Which streams share a hardware queue depends on the process's history (how many
    streams exist): probe instead of assuming.  RNET_STREAM_PROBE=0: first stream, unprobed.  Returns (stream, ok).
    When `probes` are collectives every rank must take the same decisions: each candidate runs every probe (no short cut)
    and `agree(ok)` — a MIN over the ranks — settles the verdict, so all ranks try the same number of candidates."""
    import torch
    if os.environ.get("RNET_STREAM_PROBE", "1") == "0" or not others:
        return torch.cuda.Stream(device), None
    held, s = [], None
    with torch.cuda.device(device):
        helper = torch.cuda.Stream(device)
        held.append(helper)
        main = others[0]
        for _ in range(attempts):
            s = torch.cuda.Stream(device)
            ok = True
            for o in others:
                kern_o = lambda o=o: check(lib_.rn_probe_spin(1, c_void_p(o.cuda_stream)), "rn_probe_spin")
                kern_s = lambda s=s: check(lib_.rn_probe_spin(1, c_void_p(s.cuda_stream)), "rn_probe_spin")
                # (the timing events sit on the stream the probe kernel goes to)
                ok = ok and not wait_blocks(lib_, s, kern_o, o, helper) and not wait_blocks(lib_, o, kern_s, s, helper)
                if not ok:
                    break
            for pr in probes:
                blocked = wait_blocks(lib_, s, pr, main, helper)      # always issued: `pr` may be a collective
                ok = ok and not blocked
            if agree is not None:
                ok = bool(agree(ok))
            if ok:
                return s, True
            held.append(s)     # keep the rejected ones alive so the pool hands out a different stream next time
    return s, False


def new_splitk_workspace(lib_, device, default_on=False):
    """The zero-filled split-K workspace (rn_conv_problem.splitk_ws) of ONE engine: its conv launches are ordered on one
    stream, so they share it.  Attach it to every rn_conv_problem when the problem is created — the dispatcher looks at
    it (rn_conv_kernel_id / rn_conv_tile_rows).
    What it buys, measured on MI355X (round 4): the LAST ROUND of a big launch split along K does not
    pay — a part hands its tile over as 256 KB of fp32 accumulators, which costs what the half tile of MFMA work saves — and
    the dispatcher no longer does it (big 3x3 launches run 512 x 128 tiles, whole).  SMALL launches do gain: a 3x3 layer
    of fewer tiles than the 256-row kernels normally take (ResNet stage 4 at batch 8: 26 tiles on 256 CUs) runs on the halo
    kernel with every tile cut into up to four parts instead of on the 128-row kernel: 68 -> 48 us, batch-8 inference
    3.74 -> 3.69 ms.  The inference engine therefore attaches one by default; the training engine (two streams: a part
    that polls for its partner holds a CU the other stream may be waiting for) only under RNET_SPLITK=1, where it
    measured 30.60 -> 30.57 ms.  RNET_SPLITK=0 / 1 overrides either default."""
    import torch
    on = os.environ.get("RNET_SPLITK", "1" if default_on else "0") == "1"
    if not on:
        return None
    return torch.zeros((int(lib_.rn_conv_splitk_workspace_max_bytes()),), dtype=torch.uint8, device=device)


def pair_form_kernel(lib_, batch, k, stride, pad, cin, cout, shapes, opts):
    """Kernel id (rn_conv_kernel_id) an f32 conv launch would get with its two split-bf16 weight planes stacked along Cout
    (rn_conv_segment.w_pair) — 0: the 256- / 512-row kernels do not take it, keep the planes along Cin (w_terms).
    shapes: one (H, W, pix_stride, Ho, Wo) per segment of the grouped launch.  Shapes only, no tensor is touched."""
    if PRED_W_TERMS != 2 or os.environ.get("RNET_PRED_PAIR", "1") == "0" or len(shapes) > RN_CONV_MAX_SEGMENTS:
        return 0
    p = ConvProblem()
    p.opts = opts
    p.R = p.S = k
    p.stride_h = p.stride_w = stride
    p.pad_top = p.pad_left = pad
    p.act, p.out_dtype, p.num_segments = RN_ACT_NONE, RN_DT_F32, len(shapes)
    for i, (H, W, ps, Ho, Wo) in enumerate(shapes):
        s = p.seg[i]
        s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout = batch, H, W, cin, ps, Ho, Wo, cout
        s.w_pair = 1
    kid = lib_.rn_conv_kernel_id(ctypes.byref(p))
    return kid if kid > 0 else 0


def attach_splitk_workspace(problem, ws):
    if ws is not None:
        problem.splitk_ws, problem.splitk_ws_bytes = ws.data_ptr(), ws.numel()
    return problem
