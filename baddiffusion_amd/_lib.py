"""ctypes binding of libbd_hip.so (C ABI declared in include/bd_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (never a silent PyTorch/CPU substitute).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first so libbd_hip.so binds to torch's HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbd_hip.so")

i32, i64, f32, f64, vp, sz = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_size_t


class PoisonQsampleDesc(C.Structure):
    _fields_ = [("B", i32), ("C", i32), ("H", i32), ("W", i32),
                ("images_f32", vp), ("images_u8", vp), ("is_poison", vp), ("trigger", vp), ("target_img", vp),
                ("noise", vp), ("timesteps", vp), ("alphas", vp), ("alphas_cumprod", vp), ("vmin", f32),
                ("x_noisy", vp), ("ld_noisy", i64), ("target", vp), ("ld_target", i64),
                ("R_out", vp), ("x0_out", vp), ("mask_out", vp), ("image_out", vp), ("row_index", vp), ("flip", vp)]


class QsampleDesc(C.Structure):
    _fields_ = [("B", i32), ("C", i32), ("H", i32), ("W", i32), ("x0", vp), ("R", vp), ("noise", vp),
                ("timesteps", vp), ("alphas", vp), ("alphas_cumprod", vp),
                ("x_noisy", vp), ("ld_noisy", i64), ("target", vp), ("ld_target", i64)]


class DdpmStepDesc(C.Structure):
    _fields_ = [("n", i64), ("model_output", vp), ("sample", vp), ("noise", vp), ("prev_sample", vp),
                ("pred_original", vp), ("alphas_cumprod", vp), ("t", i32), ("prev_t", i32), ("variance_type", i32),
                ("clip_sample", i32), ("clip_sample_range", f32), ("clip_defense", i32), ("clip_defense_range", f32)]


class DdimStepDesc(C.Structure):
    _fields_ = [("n", i64), ("model_output", vp), ("sample", vp), ("noise", vp), ("prev_sample", vp),
                ("pred_original", vp), ("alphas_cumprod", vp), ("t", i32), ("prev_t", i32),
                ("final_alpha_cumprod", f32), ("eta", f32), ("clip_sample", i32), ("clip_sample_range", f32)]


class GnFwdDesc(C.Structure):
    _fields_ = [("B", i32), ("HW", i32), ("C", i32), ("G", i32), ("eps", f32), ("silu", i32),
                ("x", vp), ("ldx", i64), ("gamma", vp), ("beta", vp), ("y", vp), ("ldy", i64),
                ("mean", vp), ("rstd", vp), ("workspace", vp), ("workspace_bytes", sz),
                ("y_split", vp), ("ldys", i64), ("stats", vp), ("stats_splits", i32)]


class GnBwdDesc(C.Structure):
    _fields_ = [("B", i32), ("HW", i32), ("C", i32), ("G", i32), ("silu", i32),
                ("x", vp), ("ldx", i64), ("gamma", vp), ("beta", vp), ("mean", vp), ("rstd", vp),
                ("dy", vp), ("lddy", i64), ("dx", vp), ("lddx", i64), ("accumulate_dx", i32),
                ("dgamma", vp), ("dbeta", vp), ("workspace", vp), ("workspace_bytes", sz),
                ("dx_colsum", vp), ("ld_colsum", i64), ("dx_split", vp), ("lddxs", i64),
                ("dx_add", vp), ("ld_add", i64), ("param_partials", vp), ("dx_split_c0", i32), ("dx_split_c1", i32)]


class GnParamItem(C.Structure):
    _fields_ = [("partials", vp), ("C", i32), ("dgamma", vp), ("dbeta", vp)]


class Operand(C.Structure):
    _fields_ = [("kind", i32), ("kc", i32), ("p", vp), ("ld", i64), ("bs_outer", i64), ("bs_inner", i64),
                ("C", i32), ("Hs", i32), ("Ws", i32), ("Ho", i32), ("Wo", i32), ("stride", i32),
                ("pad_t", i32), ("pad_l", i32), ("ups", i32), ("split", vp)]


class IgemmDesc(C.Structure):
    _fields_ = [("A", Operand), ("B", Operand), ("M", i32), ("N", i32), ("K", i32),
                ("batch_outer", i32), ("batch_inner", i32),
                ("C", vp), ("ldc", i64), ("c_bs_outer", i64), ("c_bs_inner", i64),
                ("alpha", f32), ("out_scale", f32), ("bias", vp),
                ("rowbias", vp), ("ld_rowbias", i64), ("rows_per_group", i32),
                ("residual", vp), ("ldr", i64), ("accumulate", i32), ("ksplit", i32),
                ("workspace", vp), ("workspace_bytes", sz), ("tile", i32), ("mode", i32),
                ("a_colsum", vp)]


class ConvFwdDesc(C.Structure):
    _fields_ = [("B", i32), ("Hs", i32), ("Ws", i32), ("Cin", i32), ("Cout", i32), ("stride", i32),
                ("pad_t", i32), ("pad_l", i32), ("ups", i32), ("Ho", i32), ("Wo", i32),
                ("x", vp), ("ldx", i64), ("w", vp), ("bias", vp), ("rowbias", vp), ("ld_rowbias", i64),
                ("residual", vp), ("ldr", i64), ("out_scale", f32), ("y", vp), ("ldy", i64),
                ("workspace", vp), ("workspace_bytes", sz), ("mode", i32), ("w_split", vp)]


class ConvDgradDesc(C.Structure):
    _fields_ = [("B", i32), ("Hs", i32), ("Ws", i32), ("Cin", i32), ("Cout", i32), ("stride", i32),
                ("pad_t", i32), ("pad_l", i32), ("ups", i32), ("Ho", i32), ("Wo", i32),
                ("dy", vp), ("lddy", i64), ("w", vp), ("dx", vp), ("lddx", i64), ("accumulate", i32),
                ("workspace", vp), ("workspace_bytes", sz), ("mode", i32), ("w_split", vp)]


class ConvWgradDesc(C.Structure):
    _fields_ = [("B", i32), ("Hs", i32), ("Ws", i32), ("Cin", i32), ("Cout", i32), ("stride", i32),
                ("pad_t", i32), ("pad_l", i32), ("ups", i32), ("Ho", i32), ("Wo", i32),
                ("x", vp), ("ldx", i64), ("dy", vp), ("lddy", i64), ("dw", vp),
                ("workspace", vp), ("workspace_bytes", sz), ("mode", i32), ("db", vp)]


class ConvPsDesc(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("W", i32), ("K", i32), ("N", i32), ("direction", i32),
                ("x_split", vp), ("ldx", i64), ("w_split", vp), ("bias", vp), ("rowbias", vp), ("ld_rowbias", i64),
                ("residual", vp), ("ldr", i64), ("out_scale", f32), ("y", vp), ("ldy", i64), ("accumulate", i32),
                ("workspace", vp), ("workspace_bytes", sz), ("gn_part", vp), ("gn_groups", i32)]


class ConvPsWgradDesc(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("x_split", vp), ("ldx", i64),
                ("dy_split", vp), ("lddy", i64), ("dw", vp), ("db", vp), ("workspace", vp), ("workspace_bytes", sz)]


class UpsampleConvDesc(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("x_split", vp), ("ldx", i64), ("dy_split", vp),
                ("lddy", i64), ("e_split", vp), ("et_split", vp), ("bias", vp), ("y", vp), ("ldy", i64), ("dx", vp), ("lddx", i64),
                ("accumulate", i32), ("dw", vp), ("db", vp), ("workspace", vp), ("workspace_bytes", sz)]


class ConvS2DgradDesc(C.Structure):
    _fields_ = [("B", i32), ("Ho", i32), ("Wo", i32), ("Cin", i32), ("Cout", i32), ("pad", i32), ("dy_split", vp), ("lddy", i64),
                ("wT_split", vp), ("dx", vp), ("lddx", i64), ("accumulate", i32)]


class GemmSpDesc(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("batch", i32), ("a", vp), ("lda", i64), ("a_bs", i64), ("a_kmajor", i32),
                ("b", vp), ("ldb", i64), ("b_bs", i64), ("b_kmajor", i32), ("c", vp), ("ldc", i64), ("c_bs", i64),
                ("c_split", vp), ("ldcs", i64), ("cs_bs", i64), ("bias", vp), ("residual", vp), ("ldr", i64), ("r_bs", i64),
                ("alpha", f32), ("out_scale", f32), ("accumulate", i32), ("a_colsum", vp), ("workspace", vp), ("workspace_bytes", sz)]


class AttnSpDesc(C.Structure):
    _fields_ = [("B", i32), ("heads", i32), ("N", i32), ("dh", i32), ("qkv_split", vp), ("ld", i64), ("scale", f32), ("o_split", vp),
                ("ldo", i64), ("pt_split", vp), ("do_split", vp), ("lddo", i64), ("dst_split", vp), ("dqkv_split", vp), ("lddqkv", i64)]


class Conv2dDesc(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("w", vp), ("bias", vp), ("y", vp), ("ldy", i64), ("B", i32), ("H", i32), ("W", i32), ("Cin", i32),
                ("Cout", i32), ("KH", i32), ("KW", i32), ("stride_h", i32), ("stride_w", i32), ("pad_h", i32), ("pad_w", i32), ("relu", i32),
                ("w_kc", i32)]


class UnetConfig(C.Structure):
    _fields_ = [("sample_size", i32), ("in_channels", i32), ("out_channels", i32), ("num_blocks", i32),
                ("block_out_channels", i32 * 8), ("down_attn", i32 * 8), ("up_attn", i32 * 8),
                ("layers_per_block", i32), ("downsample_padding", i32), ("flip_sin_to_cos", i32),
                ("freq_shift", f32), ("norm_eps", f32), ("norm_num_groups", i32), ("attention_head_dim", i32),
                ("mid_block_scale_factor", f32), ("compute_mode", i32)]


# name -> (restype, argtypes).  Every symbol declared in include/bd_hip.h must appear here
# (tests/test_abi.py cross-checks the header against this table and the built library).
SIGNATURES = {
    "bd_last_error": (C.c_char_p, []),
    "bd_version": (i32, []),
    "bd_poison_qsample": (i32, [C.POINTER(PoisonQsampleDesc), vp]),
    "bd_qsample": (i32, [C.POINTER(QsampleDesc), vp]),
    "bd_nchw_to_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i64, vp]),
    "bd_nhwc_to_nchw": (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    "bd_ddpm_step": (i32, [C.POINTER(DdpmStepDesc), vp]),
    "bd_ddim_step": (i32, [C.POINTER(DdimStepDesc), vp]),
    "bd_to_image": (i32, [vp, i32, i64, i32, i32, i32, i32, vp, vp, vp]),
    "bd_timestep_embedding": (i32, [vp, i32, i32, i32, i32, f32, vp, vp]),
    "bd_gn_workspace_bytes": (sz, [i32, i32]),
    "bd_gn_fwd_takes_stats": (i32, [i32, i32, i32, i32]),
    "bd_gn_fwd": (i32, [C.POINTER(GnFwdDesc), vp]),
    "bd_gn_bwd": (i32, [C.POINTER(GnBwdDesc), vp]),
    "bd_lincomb": (i32, [i32, C.POINTER(vp), C.POINTER(f32), i64, i32, f32, vp, vp]),
    "bd_anp_apply": (i32, [vp, i64, vp, vp, vp, i32, i64, vp, vp]),
    "bd_anp_grad": (i32, [vp, vp, vp, i32, i64, vp, vp, vp, vp, vp]),
    "bd_ssim_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "bd_ssim": (i32, [vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, f32, vp, vp, sz, vp]),
    "bd_conv2d_nhwc": (i32, [C.POINTER(Conv2dDesc), vp]),
    "bd_pool2d_nhwc": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "bd_resize_bilinear_nhwc": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, f32, vp]),
    "bd_global_avgpool_nhwc": (i32, [vp, i64, vp, i32, i32, i32, vp]),
    "bd_gn_bwd_defers": (i32, [i32, i32, i32, i32]),
    "bd_gn_bwd_params": (i32, [C.POINTER(GnParamItem), i32, i32, vp]),
    "bd_igemm_workspace_bytes": (sz, [C.POINTER(IgemmDesc)]),
    "bd_igemm": (i32, [C.POINTER(IgemmDesc), vp]),
    "bd_split_bf16": (i32, [vp, i64, vp, vp]),
    "bd_split_rows": (i32, [vp, i64, i64, i32, vp, i64, vp]),
    "bd_split_wt": (i32, [vp, i32, i32, vp, vp]),
    "bd_split_rows_ups2": (i32, [vp, i64, i32, i32, i32, i32, vp, i64, vp]),
    "bd_conv3x3_ps": (i32, [C.POINTER(ConvPsDesc), vp]),
    "bd_conv3x3_ps_workspace_bytes": (sz, [C.POINTER(ConvPsDesc)]),
    "bd_conv3x3_ps_gn_splits": (i32, [i32, i32, i32, i32, i32, i32]),
    "bd_upsample_weights": (i32, [vp, i32, i32, vp, vp, vp]),
    "bd_upsample_conv_fwd": (i32, [C.POINTER(UpsampleConvDesc), vp]),
    "bd_upsample_conv_dgrad": (i32, [C.POINTER(UpsampleConvDesc), vp]),
    "bd_upsample_conv_wgrad": (i32, [C.POINTER(UpsampleConvDesc), vp]),
    "bd_upsample_conv_wgrad_workspace_bytes": (sz, [C.POINTER(UpsampleConvDesc)]),
    "bd_upsample_conv_dgrad_workspace_bytes": (sz, [C.POINTER(UpsampleConvDesc)]),
    "bd_conv3x3_s2_dgrad_ps": (i32, [C.POINTER(ConvS2DgradDesc), vp]),
    "bd_gemm_sp_workspace_bytes": (sz, [C.POINTER(GemmSpDesc)]),
    "bd_gemm_sp": (i32, [C.POINTER(GemmSpDesc), vp]),
    "bd_attn_sp_supported": (i32, [i32, i32]),
    "bd_attn_sp_fwd": (i32, [C.POINTER(AttnSpDesc), vp]),
    "bd_attn_sp_bwd": (i32, [C.POINTER(AttnSpDesc), vp]),
    "bd_conv3x3_ps_wgrad": (i32, [C.POINTER(ConvPsWgradDesc), vp]),
    "bd_conv3x3_ps_wgrad_workspace_bytes": (sz, [C.POINTER(ConvPsWgradDesc)]),
    "bd_conv3x3_fwd": (i32, [C.POINTER(ConvFwdDesc), vp]),
    "bd_conv3x3_dgrad": (i32, [C.POINTER(ConvDgradDesc), vp]),
    "bd_conv3x3_wgrad": (i32, [C.POINTER(ConvWgradDesc), vp]),
    "bd_conv3x3_workspace_bytes": (sz, [i32] * 8),
    "bd_colsum": (i32, [vp, i64, i64, i32, i64, vp, i64, i32, vp]),
    "bd_sum2x2": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]),
    "bd_softmax_fwd": (i32, [vp, vp, i64, i32, vp]),
    "bd_softmax_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "bd_silu_fwd": (i32, [vp, vp, i64, vp]),
    "bd_silu_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "bd_reduce_workspace_bytes": (sz, []),
    "bd_loss_fwd_bwd": (i32, [vp, i64, vp, i64, i64, i32, i32, f32, vp, vp, i64, vp, vp]),
    "bd_sumsq": (i32, [vp, i64, vp, vp, vp]),
    "bd_adam_clip": (i32, [vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, f64, i32, vp, vp]),
    "bd_prof_enable": (i32, [i32]),
    "bd_prof_enabled": (i32, []),
    "bd_prof_reset": (i32, []),
    "bd_prof_num_classes": (i32, []),
    "bd_prof_get": (i32, [i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
    "bd_mfma_probe": (i32, [i32, i32, i32, C.POINTER(f64), vp]),
    "bd_axpy": (i32, [vp, vp, i64, f32, i32, vp]),
    "bd_adam_clip_dev": (i32, [vp, vp, vp, vp, i64, vp, f64, vp, f64, f64, f64, vp, vp]),
    "bd_unet_create": (i32, [C.POINTER(UnetConfig), C.POINTER(vp)]),
    "bd_unet_destroy": (None, [vp]),
    "bd_unet_set_compute_mode": (i32, [vp, i32]),
    "bd_unet_set_deferred_join": (i32, [vp, i32]),
    "bd_unet_stream_wait_aux": (i32, [vp, vp]),
    "bd_unet_num_params": (i64, [vp]),
    "bd_unet_num_tensors": (i32, [vp]),
    "bd_unet_param_info": (i32, [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i32), i64 * 4, C.POINTER(i32)]),
    "bd_unet_workspace_bytes": (sz, [vp, i32, i32]),
    "bd_unet_forward": (i32, [vp, i32, i32, vp, vp, i64, vp, i32, vp, i64, vp, sz, vp]),
    "bd_unet_backward": (i32, [vp, i32, vp, vp, i64, vp, i64, vp, vp, sz, vp]),
    "bd_unet_num_segments": (i32, [vp]),
    "bd_unet_set_aux_stream": (i32, [vp, i32]),
    "bd_unet_set_static_weights": (i32, [vp, i32]),
    "bd_unet_reset_static_cache": (i32, [vp]),
    "bd_tune_set": (i32, [C.c_char_p, i32]),
    "bd_unet_segment_range": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
    "bd_unet_segment_num_ranges": (i32, [vp, i32]),
    "bd_unet_segment_range_k": (i32, [vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]),
    "bd_unet_backward_segment": (i32, [vp, i32, i32, vp, vp, i64, vp, i64, vp, vp, sz, vp, C.POINTER(i64), C.POINTER(i64)]),
}

_lib = None


def load():
    """Load libbd_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m baddiffusion_amd.build` (hipcc, gfx950). "
            "There is no CPU / PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().bd_last_error()
        raise RuntimeError(f"libbd_hip {what} failed ({status}): {msg.decode() if msg else ''}")


def ptr(t):
    """data_ptr of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    """hipStream_t of torch's current stream, as void*."""
    return torch.cuda.current_stream().cuda_stream
