"""Loads the gfx950 kernel library and declares its C ABI (include/mi_ddpm.h) for ctypes.

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_LIB = None


def library_path() -> str:
    return os.environ.get("MI_DDPM_LIB", os.path.join(_PKG_ROOT, "lib", "libmi_ddpm.so"))


class MiConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "N", "IH", "IW", "OH", "OW", "K", "Nc", "KH", "KW", "stride", "pad", "transposed", "w_kn", "mode",
        "K1", "ldx", "ldx2", "ldy", "ldr", "accumulate")]


class MiWgradDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "N", "GH", "GW", "DH", "DW", "Ci", "Cj", "KH", "KW", "stride", "pad", "gather_i", "mode", "I1",
        "ldp", "ldp2", "ldq")]


class MiGnDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("HW", C.c_int), ("C", C.c_int), ("G", C.c_int), ("eps", C.c_float),
                ("ldx", C.c_int), ("ldy", C.c_int), ("ldr", C.c_int)]


class MiRowSum(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("pad_", C.c_int)]


ROWSUM_MAX = 16
_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> argtypes; every function returns int (0 = ok) except the two noted below
SIGNATURES = {
    "mi_conv_igemm": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "mi_conv_igemm_bf16w": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "mi_conv_igemm_bf16w_io": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P],
    "mi_conv_igemm_bf16w_io_supported": [C.POINTER(MiConvDesc)],
    "mi_conv_gt_supported": [C.POINTER(MiConvDesc)],
    "mi_conv_gt_tile": [C.POINTER(MiConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mi_conv_gt": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _P],
    "mi_conv_gt_dual": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P],
    "mi_conv_igemm_tile": [C.POINTER(MiConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mi_conv3x3_bf16w": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "mi_conv3x3_bf16w_supported": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_bf16w_uses_splitk": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_bf16w_tile": [C.POINTER(MiConvDesc), _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mi_conv3x3_wgrad_tile": [C.POINTER(MiWgradDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mi_debug_wgrad3x3_phase": [_I],
    "mi_conv3x3_wgrad_tr_supported": [C.POINTER(MiWgradDesc)],
    "mi_conv3x3_wgrad_tr_splits": [C.POINTER(MiWgradDesc)],
    "mi_conv3x3_wgrad_tr": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _P, _Z, _P],
    "mi_conv3x3_wgrad_tr_batch": [_I, C.POINTER(MiWgradDesc), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, _Z, _P],
    "mi_debug_wgrad_tr_phase": [_I],
    "mi_conv1x1_wgrad_tr_supported": [C.POINTER(MiWgradDesc), _I],
    "mi_conv1x1_wgrad_tr_batch": [_I, C.POINTER(MiWgradDesc), C.POINTER(_I), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                  C.POINTER(_P), _P, _Z, _P],
    "mi_debug_wgrad1x1_tr_phase": [_I],
    "mi_conv_s2_wgrad_tr_supported": [C.POINTER(MiWgradDesc)],
    "mi_conv_s2_wgrad_tr_batch": [_I, C.POINTER(MiWgradDesc), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, _Z, _P],
    "mi_conv_s2_wgrad_f32_supported": [C.POINTER(MiWgradDesc)],
    "mi_conv_s2_wgrad_f32": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _Z, _P],
    "mi_debug_wgrad_s2_tr_phase": [_I],
    "mi_debug_spin": [_I, _I, _P, _Z, _I, _P],
    "mi_debug_clock_probe": [_I, _I, _P, _P],
    "mi_pack_weights_tile": [],
    "mi_conv3x3_bf16w_io_gnsums": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "mi_gn_coef_from_sums": [_I, _I, _I, _I, _F, _P, _P, _P, _P, _I, _P, _P, _P],
    "mi_gn_mish_apply_sums": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P],
    "mi_conv3x3_bf16w_io_dual": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "mi_conv3x3_gn_mish_sums": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _I, _P],
    "mi_debug_wgrad1x1_tr_blocks": [_I],
    "mi_debug_wgrad_tr_blocks": [_I],
    "mi_conv3x3_bf16w_io": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P],
    "mi_conv3x3_wgrad_io": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _P, _P, _Z, _I, _P],
    "mi_gn_mish_fwd_io": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _I, _P, _P, _P, _I, _P],
    "mi_gn_mish_fwd_dual": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P],
    "mi_f32_to_bf16": [_Z, _I, _P, _I, _P, _I, _P],
    "mi_f32_to_bf16_colsum": [_Z, _I, _P, _I, _P, _I, _P, _P, _Z, _P],
    "mi_gn_mish_bwd_io": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P],
    "mi_pack_weights_bf16": [_I, _P, _I, _P, _P, _P, _P, _P, _P],
    "mi_conv3x3_pw_supported": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_tile": [C.POINTER(MiConvDesc)],
    "mi_debug_conv_pw_tile": [_I],
    "mi_debug_conv_pw_auto256": [_I],
    "mi_debug_conv1x1_pw_nloop": [_I],
    "mi_conv3x3_pw_f32_tile": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_f32": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "mi_conv1x1_pw_x32_supported": [C.POINTER(MiConvDesc)],
    "mi_conv1x1_pw_x32": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P],
    "mi_conv_pw_set_splitk_workspace": [_P, C.c_size_t],
    "mi_conv3x3_pw_splitk": [C.POINTER(MiConvDesc), _I, _I],
    "mi_debug_conv_pw_splitk": [_I],
    "mi_ln_conv1x1_pw_supported": [C.POINTER(MiConvDesc)],
    "mi_ln_conv1x1_pw": [C.POINTER(MiConvDesc), _P, _P, _P, C.c_float, _P, _P, _P, _P],
    "mi_ln_conv1x1_pw_dual": [C.POINTER(MiConvDesc), _P, _P, _P, C.c_float, _P, _P, _P, _P, _I, _P],
    "mi_conv1x1_pw_f32_supported": [C.POINTER(MiConvDesc)],
    "mi_conv1x1_pw_f32": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "mi_pack_weights_f32frag": [_I, _P, _I, _P, _P, _P, _P],
    "mi_conv3x3_pw_x32_supported": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_x32_tile": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_x32": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "mi_conv3x3_pw": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P],
    "mi_conv3x3_pw_gn_mish_supported": [C.POINTER(MiConvDesc)],
    "mi_conv1x1_pw_supported": [C.POINTER(MiConvDesc)],
    "mi_conv1x1_pw": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P, _I, _P],
    "mi_conv3x3_pw_gnsums": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "mi_conv3x3_pw_gn_mish": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _P],
    "mi_conv3x3_pw_gn_mish_sums": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _I, _P],
    "mi_conv3x3_pw_x32_gn_mish_supported": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_x32_gn_mish_tile": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_gn_mish_tile": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_pw_x32_gn_mish": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _P],
    "mi_conv3x3_pw_x32_gn_mish_sums": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _I, _P],
    "mi_gn_stats_coef": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "mi_conv3x3_gn_mish_supported": [C.POINTER(MiConvDesc)],
    "mi_conv3x3_gn_mish_tile": [C.POINTER(MiConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mi_conv3x3_gn_mish": [C.POINTER(MiConvDesc), _P, _P, _P, _P, _P, _I, _P],
    "mi_conv3x3_small_cin_fwd": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P],
    "mi_conv3x3_small_cin_wgrad": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P],
    "mi_conv1x1_small_cout": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "mi_conv1x1_small_cout_ws": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P, _Z, _P],
    "mi_conv1x1_small_cout_io": [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, _Z, _P],
    "mi_linattn_fold_fwd": [_I, _I, _I, _P, _P, _I, _P, _P],
    "mi_conv1x1_pw_batched": [C.POINTER(MiConvDesc), _P, _P, _I, _P, _P, _P, _I, _P, _I, _P],
    "mi_conv1x1_small_cout_bwd": [_I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, C.c_size_t, _P],
    "mi_conv1x1_small_cout_gn_supported": [_I, _I, _I],
    "mi_conv1x1_small_cout_gn_fwd": [_I, _I, _I, _I, _P, _I, _P, _P, _P, _I, C.c_float, _P, _P, _P, _I, _P],
    "mi_conv_small_cin_fwd_dual_supported": [_I, _I, _I, _I, _I, _I],
    "mi_conv_small_cin_fwd_dual_chores": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P, C.c_size_t, _P, _P, _P, _I, _I, _P],
    "mi_conv_small_cin_fwd_dual": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P],
    "mi_conv_small_cin_fwd_io": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "mi_conv_small_cin_wgrad_io": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P, _Z, _P],
    "mi_conv_small_cin_bf16_supported": [_I, _I, _I, _I, _I, _I, _I],
    "mi_conv_small_cin_fwd": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P],
    "mi_conv_small_cin_wgrad": [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _Z, _P],
    "mi_conv_wgrad": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _P],
    "mi_conv3x3_wgrad": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _P],
    "mi_conv3x3_wgrad_supported": [C.POINTER(MiWgradDesc)],
    "mi_conv3x3_wgrad_ws": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _P, _Z, _P],
    "mi_conv3x3_wgrad_bias": [C.POINTER(MiWgradDesc), _P, _P, _P, _P, _P, _P, _Z, _P],
    "mi_colsum": [_I, _I, _P, _I, _P, _P],
    "mi_gn_mish_fwd": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "mi_gn_mish_bwd": [C.POINTER(MiGnDesc), _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P],
    "mi_chan_layernorm_fwd": [_I, _I, _P, _I, _P, _P, _F, _P, _I, _P],
    "mi_chan_layernorm_bwd": [_I, _I, _P, _I, _P, _F, _P, _I, _P, _I, _I, _P, _P, _P],
    "mi_linattn_fwd": [_I, _I, _I, _P, _P, _P, _P, _P],
    "mi_linattn_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mi_vq_partials": [_I],
    "mi_vq_nearest_fwd": [_I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P],
    "mi_vq_bwd": [_I, _I, _I, _P, _I, _P, _P, _F, _F, _P, _P, _I, _I, _P, _P],
    "mi_vq_scatter_rows": [_I, _I, _I, _P, _I, _P, _P, _P],
    "mi_small_gemm": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _P],
    "mi_small_gemm_supported": [_I, _I, _I, _I, _I, _I, _I],
    "mi_linattn_fwd_io": [_I, _I, _I, _P, _P, _P, _P, _I, _P],
    "mi_linattn_bwd_io": [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "mi_linattn_fwd_ws": [_I, _I, _I, _P, _P, _P, _P, _I, _P, _Z, _P],
    "mi_linattn_bwd_ws": [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _Z, _P],
    "mi_chan_layernorm_fwd_io": [_I, _I, _P, _I, _P, _P, _F, _P, _I, _I, _P],
    "mi_chan_layernorm_bwd_io": [_I, _I, _P, _I, _P, _F, _P, _I, _P, _I, _I, _P, _P, _I, _P],
    "mi_chan_layernorm_bwd_part": [_I, _I, _P, _I, _P, _F, _P, _I, _P, _I, _I, _P, _I, _P],
    "mi_rowsum_batch": [_I, _P, _P],
    "mi_f32_to_bf16_colsum_part": [_Z, _I, _P, _I, _P, _I, _P, _Z, C.POINTER(C.c_int), _P],
    "mi_time_embed": [_I, _I, _P, _P, _P],
    "mi_sample_norm_supported": [_I, _I, _I],
    "mi_sample_norm_fwd": [_I, _I, _I, _P, _P, _P, _P, _P, _F, _P],
    "mi_sample_norm_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mi_sample_norm_bwd2": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mi_leaky_relu_fwd": [_Z, _P, _P, _F, _P],
    "mi_leaky_relu_bwd": [_Z, _P, _P, _P, _F, _P],
    "mi_tanh_fwd": [_Z, _P, _P, _P],
    "mi_tanh_bwd": [_Z, _P, _P, _P, _P],
    "mi_lerp_rows": [_I, _Z, _P, _P, _P, _P, _P],
    "mi_gp_penalty": [_I, _Z, _P, _P, _P, _F, _P, _P],
    "mi_batchnorm_workspace": [_I],
    "mi_batchnorm_fwd": [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P],
    "mi_batchnorm_bwd": [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mi_vae_latent_fwd": [_I, _I, _P, _I, _P, _P, _P, _P],
    "mi_vae_latent_bwd": [_I, _I, _P, _I, _P, _P, _F, _P, _P, _I, _P],
    "mi_adam_tick": [_P, _P],
    "mi_adam_step_dev": [_Z, _P, _P, _P, _P, _P, C.c_double, C.c_double, _F, _F, _P],
    "mi_relu_fwd": [_Z, _P, _P, _P],
    "mi_relu_bwd": [_Z, _P, _P, _P, _I, _P],
    "mi_mish_fwd": [_Z, _P, _P, _P],
    "mi_mish_bwd": [_Z, _P, _P, _P, _P],
    "mi_nchw_to_nhwc": [_I, _I, _I, _P, _P, _I, _P],
    "mi_nhwc_to_nchw": [_I, _I, _I, _P, _I, _P, _P],
    "mi_q_sample": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "mi_eps_loss": [_I, _I, _I, _P, _I, _P, _I, _P, _P, _F, _P],
    "mi_p_sample_update": [_I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "mi_adam_step": [_Z, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _F, _P],
    "mi_axpby": [_Z, _F, _P, _I, _P, _P],
    "mi_axpby2d": [_I, _I, _F, _P, _I, _I, _P, _I, _P],
    "mi_gather_rows": [_I, _I, _P, _P, _P, _P],
    "mi_u8_gather_normalize": [_I, _I, _I, _I, _P, _P, _P, _I, _P, _P],
    "mi_scale_by_device_scalar": [_I, _I, _P, _I, _P, _P],
}
OTHER = {"mi_abi_version": ([], C.c_int), "mi_last_error": ([], C.c_char_p),
         "mi_conv3x3_wgrad_workspace": ([C.POINTER(MiWgradDesc)], C.c_size_t),
         "mi_conv3x3_wgrad_tr_workspace": ([C.POINTER(MiWgradDesc)], C.c_size_t),
         "mi_conv3x3_wgrad_tr_batch_workspace": ([_I, C.POINTER(MiWgradDesc)], C.c_size_t),
         "mi_conv1x1_wgrad_tr_batch_workspace": ([_I, C.POINTER(MiWgradDesc), C.POINTER(_I)], C.c_size_t),
         "mi_conv_s2_wgrad_tr_batch_workspace": ([_I, C.POINTER(MiWgradDesc)], C.c_size_t),
         "mi_conv_s2_wgrad_f32_workspace": ([C.POINTER(MiWgradDesc)], C.c_size_t),
         "mi_f32_to_bf16_colsum_workspace": ([_Z, _I], C.c_size_t),
         "mi_linattn_workspace": ([_I, _I, _I], C.c_size_t),
         "mi_chan_layernorm_bwd_part_rows": ([_I, _I], C.c_int),
         "mi_conv_small_wgrad_workspace": ([_I], C.c_size_t)}
ABI_VERSION = 4


def load_library():
    """dlopen libmi_ddpm.so once; raise (never fall back) when it is absent or stale."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `make -C {_PKG_ROOT}` (hipcc --offload-arch=gfx950). "
            "This package has no CPU or PyTorch fallback.")
    # PyTorch bundles its own libamdhip64; import it first so this library binds to the SAME HIP
    # runtime (same SONAME -> the loader reuses it) and stream handles are interchangeable.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError -> missing symbol, loudly
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.mi_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.mi_abi_version()} != {ABI_VERSION}; rebuild")
    if os.environ.get("MI_DEBUG_KNOBS") == "1" and os.environ.get("MI_PW_AUTO256") is not None:
        lib.mi_debug_conv_pw_auto256(int(os.environ["MI_PW_AUTO256"]))      # A/B: 0 = no 256-pixel conv tiles, n = from n workgroups up
    if os.environ.get("MI_DEBUG_KNOBS") == "1" and os.environ.get("MI_PW1_NLOOP") is not None:
        lib.mi_debug_conv1x1_pw_nloop(int(os.environ["MI_PW1_NLOOP"]))     # A/B: 0 = to_qkv on the 2-D grid
    _LIB = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load_library().mi_last_error().decode(errors="replace")
        raise RuntimeError(f"libmi_ddpm {what} failed (rc={rc}): {msg}")
