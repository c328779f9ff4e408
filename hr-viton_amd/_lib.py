"""ctypes binding of the C ABI declared in include/hrviton_hip.h.

The product path has no CPU fallback: if the gfx950 library is missing or a
kernel launch fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhrviton_hip.so")

HRV_MAX_SRC = 4
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


class hrv_src_t(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("cstride", C.c_int32), ("coff", C.c_int32),
                ("up_shift", C.c_int32), ("pre_act", C.c_int32), ("C_real", C.c_int32)]


class hrv_spade_epi_t(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_cstride", C.c_int32), ("x_coff", C.c_int32), ("C", C.c_int32),
                ("_pad", C.c_int32), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("noise_z", C.c_void_p),
                ("noise_scale", C.c_void_p), ("g1p_out", C.c_void_p)]


class hrv_norm_bwd_t(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("x", C.c_void_p), ("x_cstride", C.c_int32), ("x_coff", C.c_int32),
                ("noise_z", C.c_void_p), ("noise_scale", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("g1p", C.c_void_p), ("g1p_cstride", C.c_int32), ("g1p_coff", C.c_int32),
                ("dout", C.c_void_p), ("dout_cstride", C.c_int32), ("dout_coff", C.c_int32),
                ("dnh", C.c_void_p), ("dnh_cstride", C.c_int32), ("dnh_coff", C.c_int32),
                ("dgb", C.c_void_p), ("dgb_cstride", C.c_int32), ("dgb_coff", C.c_int32),
                ("dx", C.c_void_p), ("dx_cstride", C.c_int32), ("dx_coff", C.c_int32),
                ("dx_accumulate", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
                ("dns_accumulate", C.c_int32), ("dnoise_scale", C.c_void_p), ("workspace", C.c_void_p),
                ("dgb_bf16", C.c_int32), ("out_bf16", C.c_int32), ("dx_bf16", C.c_int32), ("g1p_bf16", C.c_int32),
                ("dnh_bf16", C.c_int32), ("dout_bf16", C.c_int32),
                ("x_up_channels", C.c_int32), ("x2", C.c_void_p), ("x2_cstride", C.c_int32), ("x2_coff", C.c_int32)]


class hrv_sn_job_t(C.Structure):
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p), ("u_keep", C.c_void_p),
                ("v_keep", C.c_void_p), ("R", C.c_int32), ("K", C.c_int32)]


class hrv_thin_conv_t(C.Structure):
    _fields_ = [("src", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("src_channels", C.c_int32),
                ("src_cstride", C.c_int32), ("src_coff", C.c_int32),
                ("w_oihw", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("sigma", C.c_void_p), ("wscale", C.c_float), ("mode", C.c_int32), ("shift", C.c_void_p),
                ("residual", C.c_void_p), ("res_cstride", C.c_int32), ("res_coff", C.c_int32), ("res_bf16", C.c_int32),
                ("res_mode", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32), ("out_bf16", C.c_int32)]


class hrv_conv_cout1_t(C.Structure):
    _fields_ = [("x", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("x_cstride", C.c_int32), ("x_coff", C.c_int32), ("w_oihw", C.c_void_p), ("sigma", C.c_void_p),
                ("wscale", C.c_float), ("bias", C.c_void_p), ("K", C.c_int32), ("pad", C.c_int32),
                ("y", C.c_void_p), ("y_cstride", C.c_int32), ("y_coff", C.c_int32),
                ("dx", C.c_void_p), ("dx_cstride", C.c_int32), ("dx_coff", C.c_int32),
                ("add", C.c_void_p), ("add_cstride", C.c_int32), ("add_coff", C.c_int32),
                ("workspace", C.c_void_p), ("round_bf16", C.c_int32)]


class hrv_spade_gb_t(C.Structure):
    _fields_ = [("mode", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("src", C.c_void_p), ("src_cstride", C.c_int32), ("src_coff", C.c_int32), ("w_packed", C.c_void_p),
                ("C", C.c_int32), ("Cp", C.c_int32), ("hid", C.c_int32), ("x_f32", C.c_int32),
                ("x", C.c_void_p), ("x_cstride", C.c_int32), ("x_coff", C.c_int32),
                ("stat_stride", C.c_int32), ("g1p_bf16", C.c_int32),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("noise_z", C.c_void_p), ("noise_scale", C.c_void_p),
                ("bias_gamma", C.c_void_p), ("bias_beta", C.c_void_p), ("g1p", C.c_void_p),
                ("act", C.c_int32), ("act_slope", C.c_float),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32), ("out_f32", C.c_int32),
                ("_pad", C.c_int32), ("mask", C.c_void_p), ("mask_cstride", C.c_int32), ("mask_coff", C.c_int32)]


class hrv_spade_fused_t(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("seg", C.c_void_p), ("seg_H", C.c_int32), ("seg_W", C.c_int32), ("seg_shift", C.c_int32), ("x_f32", C.c_int32),
                ("w_packed", C.c_void_p),
                ("x", C.c_void_p), ("x_cstride", C.c_int32), ("x_coff", C.c_int32),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("noise_z", C.c_void_p), ("noise_scale", C.c_void_p),
                ("bias_gamma", C.c_void_p), ("bias_beta", C.c_void_p), ("g1p", C.c_void_p),
                ("act", C.c_int32), ("act_slope", C.c_float),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("actv", C.c_void_p), ("actv_cstride", C.c_int32), ("actv_coff", C.c_int32),
                ("x_up_channels", C.c_int32), ("x2", C.c_void_p), ("x2_cstride", C.c_int32), ("x2_coff", C.c_int32)]


class hrv_conv_p2_t(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("src", C.c_void_p), ("src_cstride", C.c_int32), ("src_coff", C.c_int32), ("Cout", C.c_int32),
                ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("act", C.c_int32), ("act_slope", C.c_float),
                ("mask", C.c_void_p), ("mask_cstride", C.c_int32), ("mask_coff", C.c_int32), ("mask_slope", C.c_float), ("out_f32", C.c_int32),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("res_f32", C.c_int32), ("residual", C.c_void_p), ("res_cstride", C.c_int32), ("res_coff", C.c_int32),
                ("res_after_mask", C.c_int32)]


class hrv_conv_s2_t(C.Structure):
    _fields_ = [("mode", C.c_int32),
                ("N", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("K", C.c_int32),
                ("src", C.c_void_p), ("src_cstride", C.c_int32), ("src_coff", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("cols", C.c_int32), ("Cph", C.c_int32),
                ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("act", C.c_int32), ("act_slope", C.c_float),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32), ("out_f32", C.c_int32),
                ("residual", C.c_void_p), ("res_cstride", C.c_int32), ("res_coff", C.c_int32), ("res_f32", C.c_int32),
                ("mask", C.c_void_p), ("mask_cstride", C.c_int32), ("mask_coff", C.c_int32), ("mask_slope", C.c_float)]


class hrv_s2_pack_job_t(C.Structure):
    _fields_ = [("mode_flags", C.c_int32), ("K", C.c_int32), ("cols", C.c_int32), ("Cph", C.c_int32),
                ("w", C.c_void_p), ("sigma", C.c_void_p), ("wscale", C.c_float), ("_pad", C.c_int32), ("out", C.c_void_p)]


class hrv_conv2d_t(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("nsrc", C.c_int32), ("src", hrv_src_t * HRV_MAX_SRC),
                ("w_packed", C.c_void_p), ("w_oihw", C.c_void_p), ("Cout", C.c_int32), ("tile_cfg", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p),
                ("res_cstride", C.c_int32), ("res_coff", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("spade", C.POINTER(hrv_spade_epi_t)), ("out_up_shift", C.c_int32), ("_pad2", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("pad_w_plus1", C.c_int32), ("free_extent", C.c_int32), ("out_step", C.c_int32),
                ("out_off_h", C.c_int32), ("out_off_w", C.c_int32), ("out_H", C.c_int32), ("out_W", C.c_int32),
                ("res_mode", C.c_int32), ("mixed_flags", C.c_int32), ("_pad3", C.c_int32)]


class hrv_flow_warp_t(C.Structure):
    _fields_ = [("src", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("src_cstride", C.c_int32), ("src_coff", C.c_int32), ("flow", C.c_void_p),
                ("fh", C.c_int32), ("fw", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("rh", C.c_float), ("rw", C.c_float), ("norm_x", C.c_float), ("norm_y", C.c_float),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32), ("flow_up", C.c_void_p)]


class hrv_flow_warp_bwd_t(C.Structure):
    _fields_ = [("src", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("src_cstride", C.c_int32), ("src_coff", C.c_int32), ("flow_up", C.c_void_p),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("norm_x", C.c_float), ("norm_y", C.c_float),
                ("dout", C.c_void_p), ("dout_cstride", C.c_int32), ("dout_coff", C.c_int32),
                ("dsrc", C.c_void_p), ("dsrc_cstride", C.c_int32), ("dsrc_coff", C.c_int32),
                ("dflow", C.c_void_p), ("dflow_accumulate", C.c_int32), ("_pad", C.c_int32)]


# every symbol include/hrviton_hip.h declares: (restype, argtypes)
_i32, _i64, _f, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
_ip = C.POINTER(C.c_int32)
SYMBOLS = {
    "hrv_version": (C.c_char_p, []),
    "hrv_last_error": (C.c_char_p, []),
    "hrv_device_check": (C.c_int, []),
    "hrv_set_reserved_cus": (C.c_int, [_i32]),
    "hrv_persistent_cus": (C.c_int, []),
    "hrv_diag_set_tlog": (C.c_int, [_vp, _i64]),
    "hrv_diag_reload_env": (C.c_int, []),
    "hrv_conv2d_pick_tile": (C.c_int, [_i64, _i32]),
    "hrv_conv2d_tile_bn": (C.c_int, [_i32]),
    "hrv_conv2d_tile_bm": (C.c_int, [_i32]),
    "hrv_conv2d_tile_row_bytes": (C.c_int, [_i32]),
    "hrv_conv2d_packed_elems": (_i64, [_i32, _i32, _i32, _i32, _ip, _i32]),
    "hrv_conv2d_pack_weight_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _vp]),
    "hrv_conv2d_pack_weight_dev_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _i32, _i32, _i32, _i32,
                                                 _i32, _f, _vp, _vp, _ip, _vp]),
    "hrv_conv2d_pack_weight_dev_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _i32, _i32, _i32, _i32,
                                                  _i32, _f, _vp, _vp, _ip, _vp]),
    "hrv_conv2d_pack_record_bytes": (_i32, []),
    "hrv_conv2d_pack_weight_record": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _i32, _i32, _i32, _i32, _i32, _f,
                                                _vp, _i32, _vp, _i32, _i32, _vp, _ip, _vp, _ip]),
    "hrv_conv2d_pack_weight_multi": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "hrv_conv2d_wgrad_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i64]),
    "hrv_conv2d_wgrad_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                            _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32,
                                            _vp, _i32, _vp]),
    "hrv_conv2d_wgrad_bf16mma_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                                    _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                                    _i64, _vp, _i32, _vp, _i32, _vp]),
    "hrv_conv2d_wgrad_bf16mma_st_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                                       _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                                       _vp, _i64, _vp, _i32, _vp, _i32, _i32, _vp]),
    "hrv_colsum_nhwc_f32": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _vp]),
    "hrv_norm_bwd_workspace_elems": (_i64, [_i32, _i32, _i32, _i32]),
    "hrv_spade_norm_bwd2_nhwc_f32": (C.c_int, [C.POINTER(hrv_norm_bwd_t), C.POINTER(hrv_norm_bwd_t), _vp]),
    "hrv_spade_norm_bwd_nhwc_f32": (C.c_int, [C.POINTER(hrv_norm_bwd_t), _vp]),
    "hrv_thin_conv_supported": (C.c_int, [_i32, _i32, _i32, _i32]),
    "hrv_thin_conv_bf16": (C.c_int, [C.POINTER(hrv_thin_conv_t), _vp]),
    "hrv_loss_f32": (C.c_int, [_vp, _vp, _i64, _i32, _f, _f, _vp, _vp, _vp, _i32, _vp]),
    "hrv_loss_bf16in_f32": (C.c_int, [_vp, _vp, _i64, _i32, _f, _f, _vp, _vp, _vp, _i32, _vp]),
    "hrv_scale_f32": (C.c_int, [_vp, _i64, _f, _vp, _vp]),
    "hrv_act_bwd_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _i32, _f, _vp]),
    "hrv_tanh_bwd_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "hrv_add_slice_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "hrv_downsum2x2_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "hrv_avgpool3x3s2_bwd_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "hrv_maxpool2x2_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_maxpool2x2_nhwc_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_maxpool2x2_bwd_nhwc_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_maxpool2x2_bwd_relu_nhwc_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_maxpool2x2_bwd_relu_nhwc_xbf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_maxpool2x2_bwd_relu_nhwc_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_add_slice_nhwc_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "hrv_adam_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i32, _f, _vp]),
    "hrv_adam_hyper_f32": (C.c_int, [_vp, _vp, _f, _f, _vp, _vp]),
    "hrv_adam_dev_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _vp]),
    "hrv_spectral_norm_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _f, _vp, _vp, _vp]),
    "hrv_spectral_norm_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "hrv_spectral_norm_batched_f32": (C.c_int, [_vp, _i32, _i32, _f, _vp, _vp]),
    "hrv_conv2d_pack_weight_pair_dev": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _i32, _i32,
                                                  _i32, _vp, _ip, _vp]),
    "hrv_spade_vec_prep_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "hrv_spade_vec_prep_multi_f32": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hrv_shared_taps_prep_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "hrv_shared_taps_grad_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "hrv_conv2d_workspace_bytes": (_i64, [C.POINTER(hrv_conv2d_t)]),
    "hrv_conv2d_nhwc_f32": (C.c_int, [C.POINTER(hrv_conv2d_t), _vp]),
    "hrv_conv2d_packed_elems_bf16": (_i64, [_i32, _i32, _i32, _i32, _ip, _i32]),
    "hrv_conv2d_pack_weight_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _vp]),
    "hrv_conv2d_nhwc_bf16": (C.c_int, [C.POINTER(hrv_conv2d_t), _vp]),
    "hrv_conv2d_naive_nhwc_f32": (C.c_int, [C.POINTER(hrv_conv2d_t), _vp]),
    "hrv_tapsum_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32,
                                      _vp]),
    "hrv_instnorm_workspace_elems": (_i64, [_i32, _i32, _i32, _i32]),
    "hrv_instnorm_stats_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp,
                                              _vp]),
    "hrv_instnorm_stats2_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hrv_instnorm_stats2_up_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _f, _vp,
                                                  _vp, _vp, _vp, _vp, _vp]),
    "hrv_instnorm_stats_nhwc_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp,
                                               _vp]),
    "hrv_instnorm_apply_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f, _vp, _i32,
                                              _i32, _vp]),
    "hrv_avgpool3x3s2_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "hrv_mul_channel_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _vp]),
    "hrv_gauss_blur_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "hrv_parse_argmax_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _i32, _vp]),
    "hrv_resize_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_occlusion_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i64, _vp]),
    "hrv_nchw_to_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "hrv_nchw_f32_to_nhwc_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "hrv_nhwc_bf16_to_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_nhwc_to_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_space_to_depth2_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_depth_to_space2_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_conv_cout1_fwd_f32": (C.c_int, [C.POINTER(hrv_conv_cout1_t), _vp]),
    "hrv_conv_cout1_dgrad_f32": (C.c_int, [C.POINTER(hrv_conv_cout1_t), _vp]),
    "hrv_conv_cout1_wgrad_slabs": (_i32, [_i32, _i32, _i32]),
    "hrv_conv_cout1_wgrad_f32": (C.c_int, [C.POINTER(hrv_conv_cout1_t), _vp, _i32, _vp, _i32, _vp]),
    "hrv_concat_nhwc_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "hrv_resize_bilinear_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f,
                                               _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "hrv_flow_warp_nhwc_f32": (C.c_int, [C.POINTER(hrv_flow_warp_t), _vp]),
    "hrv_bn_finalize_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp]),
    "hrv_affine_act_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _f, _vp, _i32,
                                          _i32, _vp]),
    "hrv_bn_bwd_workspace_elems": (_i64, [_i32]),
    "hrv_bn_bwd_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                      _i32, _vp, _vp, _i32, _vp]),
    "hrv_resize_bilinear_bwd_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f, _vp, _i32, _i32,
                                                   _i32, _i32, _i32, _vp]),
    "hrv_resize_nearest_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_resize_nearest_nchw_bwd_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_resize_nearest_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "hrv_resize_nearest_bwd_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "hrv_flow_warp_bwd_nhwc_f32": (C.c_int, [C.POINTER(hrv_flow_warp_bwd_t), _vp]),
    "hrv_grid_sample_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "hrv_grid_sample_nchw_bwd_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "hrv_softmax_nchw_f32": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp]),
    "hrv_softmax_nchw_bwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "hrv_cross_entropy_nchw_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _f, _vp, _vp, _vp, _vp]),
    "hrv_tapsum_bwd_nhwc_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "hrv_tap_expand_nhwc": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_mul_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "hrv_spade_gb_packed_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "hrv_spade_gb_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "hrv_spade_gb_pack_dev": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "hrv_spade_gb_bf16": (C.c_int, [C.POINTER(hrv_spade_gb_t), _vp]),
    "hrv_conv_p2_packed_bytes": (C.c_int64, [_i32, _i32]),
    "hrv_conv_p2_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "hrv_conv_p2_pack_dev": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _vp, _f, _vp, _vp]),
    "hrv_conv_p2_bf16": (C.c_int, [C.POINTER(hrv_conv_p2_t), _vp]),
    "hrv_conv_s2_packed_bytes": (C.c_int64, [_i32, _i32, _i32]),
    "hrv_conv_s2_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "hrv_conv_s2_pack_dev": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _vp, _f, _vp, _vp]),
    "hrv_conv_s2_pack_multi_dev": (C.c_int, [_i32, C.POINTER(hrv_s2_pack_job_t), _vp]),
    "hrv_conv_s2_bf16": (C.c_int, [C.POINTER(hrv_conv_s2_t), _vp]),
    "hrv_space_to_depth2_cells_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "hrv_instnorm_apply_nhwc_bf16out": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f, _vp, _i32, _i32, _vp]),
    "hrv_scale_bf16": (C.c_int, [_vp, _i64, _f, _vp, _vp]),
    "hrv_split3_nhwc_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "hrv_conv2d_wgrad_s2_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "hrv_pad_width_nhwc_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "hrv_spade_fused_packed_bytes": (C.c_int64, [_i32]),
    "hrv_spade_fused_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "hrv_spade_fused_pack_dev": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "hrv_spade_fused_bf16": (C.c_int, [C.POINTER(hrv_spade_fused_t), _vp]),
    "hrv_tv_loss_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
}

_lib = None


class HrvError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and bind every declared symbol (loud on failure)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HrvError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def reload_env():
    """The library caches its HRV_* environment switches on first use (no getenv() in a launch path); a process that changes one
    while it runs -- tests, tools/conv_bench.py -- calls this afterwards."""
    if _lib is not None:
        _lib.hrv_diag_reload_env()


def check(rc: int, what: str):
    if rc != 0:
        msg = load().hrv_last_error().decode("utf-8", "replace")
        raise HrvError(f"{what} failed (rc={rc}): {msg}")
