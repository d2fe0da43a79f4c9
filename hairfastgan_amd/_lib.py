"""ctypes binding of libhairfast_hip.so (the C ABI declared in include/hairfast_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``hairfastgan_amd/csrc/build.sh`` (hipcc, --offload-arch=gfx950).  There is no
fallback: if the shared object is missing or a symbol is absent, importing the
ops raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HAIRFAST_HIP_LIB: kernel-development override (experimental builds of the same ABI)
LIB_PATH = os.environ.get("HAIRFAST_HIP_LIB") or os.path.join(_HERE, "csrc", "libhairfast_hip.so")

_f = ctypes.c_void_p      # float* / const float* (device pointers)
_i = ctypes.c_int
_ll = ctypes.c_longlong
_fl = ctypes.c_float
_st = ctypes.c_void_p     # hipStream_t

# name -> argtypes; every function returns int (0 == HF_OK) unless noted.
SIGNATURES = {
    "hf_upfirdn2d_f32": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _st],
    "hf_fused_bias_act_f32": [_f, _f, _f, _ll, _i, _i, _fl, _fl, _st],
    "hf_noise_bias_act_f32": [_f, _f, _f, _f, _f, _i, _i, _i, _ll, _fl, _fl, _st],
    "hf_modconv_prepare_f32": [_f, _f, _f, _i, _i, _i, _st],
    "hf_modulation_f32": [_f, _f, _ll, _f, _f, _i, _i, _i, _st],
    "hf_style_batch_f32": [_f, _f, _ll, _ll, _f, _i, _i, _i, _i, _i, _st],
    "hf_demod_f32": [_f, _f, _f, _i, _i, _i, _st],
    "hf_style_normalize_f32": [_f, _f, _i, _i, _i, _st],
    "hf_modconv3x3_f32": [_f, _f, _f, _f, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _f, _ll, _st],
    "hf_conv_split_weights_f16": [_f, _f, _f, _i, _i, _st],
    "hf_conv_split_weights_f16_taps": [_f, _f, _f, _i, _i, _i, _st],
    "hf_modconv3x3_small_f16_f32": [_f, _f, _f, _f, _i, _f, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _i, _i, _f, _ll, _st],
    "hf_modconv3x3_small_up_blur_f16_f32": [_f, _f, _f, _f, _f, _f, _i, _f, _f, _f, _f, _f, _ll, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _f, _ll, _st],
    "hf_conv1x1_f16_f32": [_f, _f, _f, _f, _f, _f, _i, _f, _f, _f, _f, _i, _f, _fl, _f, _i, _i, _i, _i, _i, _i, _i, _ll, _f, _ll, _st],
    "hf_modconv3x3_f16_f32": [_f, _f, _f, _f, _i, _f, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _st],
    "hf_modconv3x3_f16_rgb_f32": [_f, _f, _f, _f, _i, _f, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _f, _f, _f, _st],
    "hf_modconv3x3_f16_pre_image_f32": [_f, _f, _f, _f, _f, _i, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _f, _f, _f, _f, _f, _st],
    "hf_modconv3x3_f16_pre_f32": [_f, _f, _f, _f, _f, _i, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _f, _f, _f, _f, _f, _f, _st],
    "hf_modconv3x3_up_f16_pre_f32": [_f, _f, _f, _f, _f, _i, _f, _i, _i, _i, _i, _i, _i, _st],
    "hf_modconv3x3_up_f16_f32": [_f, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i, _i, _st],
    "hf_modconv3x3_up_blur_f16_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(ctypes.c_float), _f, _f, _ll, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _st],
    "hf_modconv3x3_up_f32": [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _ll, _st],
    "hf_modconv_up_pitch": [_i],
    "hf_blur_noise_bias_act_f32": [_f, _f, _f, _f, _f, _ll, _f, _i, _i, _i, _i, _i, _fl, _fl, _st],
    "hf_blur_noise_bias_act_split_f16": [_f, _f, _f, _f, _f, _f, _ll, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _st],
    "hf_torgb_f32": [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _st],
    "hf_conv_prepare_f32": [_f, _f, _i, _i, _i, _fl, _st],
    "hf_bn_fold_f32": [_f, _f, _f, _f, _f, _f, _f, _fl, _i, _st],
    "hf_conv2d_f32": [_f, _f, _f, _f, _f, _f, _f, _i, _f, _fl, _f, _i, _i, _i, _i, _i, _i, _i, _i, _ll, _f, _ll, _st],
    "hf_conv2d_f16_f32": [_f, _f, _f, _f, _f, _f, _i, _f, _f, _f, _f, _i, _f, _fl, _f, _i, _i, _i, _i, _i, _i, _i, _ll, _f, _ll, _st],
    "hf_split_activation_f16": [_f, _f, _f, _f, _f, _ll, _i, _i, _i, _st],
    "hf_split_activation_mod_f16": [_f, _f, _f, _f, _ll, _i, _i, _i, _st],
    "hf_plane_mean_f32": [_f, _f, _i, _i, _st],
    "hf_se_gate_f32": [_f, _f, _f, _f, _i, _i, _i, _st],
    "hf_scale_shortcut_add_f32": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _st],
    "hf_upsample_bilinear_add_f32": [_f, _f, _f, _i, _i, _i, _i, _i, _st],
    "hf_adaptive_avgpool_f32": [_f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _st],
    "hf_downscale2x_f32": [_f, _f, _i, _i, _i, _st],
    "hf_linear_f32": [_f, _f, _ll, _f, _f, _i, _i, _i, _fl, _st],
    "hf_equal_linear_f32": [_f, _f, _ll, _f, _f, _i, _i, _i, _fl, _i, _fl, _fl, _st],
    "hf_pixel_norm_f32": [_f, _f, _i, _i, _st],
    "hf_bicubic_down_f32": [_f, _f, _f, _ll, _i, _i, _i, _st],
    "hf_dilate_erode_f32": [_f, _f, _f, _ll, _i, _i, _i, _st],
    "hf_maxpool3x3s2_f32": [_f, _f, _ll, _i, _i, _st],
    "hf_gate_f32": [_f, _f, _f, _f, _f, _fl, _ll, _i, _st],
    "hf_upsample_nearest_f32": [_f, _f, _ll, _i, _i, _i, _i, _st],
    "hf_parsing_mask_i64": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _st],
    "hf_layernorm_f32": [_f, _f, _f, _f, _i, _i, _fl, _i, _fl, _st],
    "hf_layernorm_grouped_f32": [_f, _f, _f, _f, _i, _i, _i, _fl, _i, _fl, _st],
    "hf_modulate_f32": [_f, _f, _f, _f, _ll, _i, _fl, _st],
    "hf_sample_layernorm_f32": [_f, _f, _f, _f, _i, _i, _i, _ll, _fl, _fl, _f, _ll, _st],
    "hf_pixel_norm_dim1_f32": [_f, _f, _i, _i, _i, _st],
    "hf_axpby_bcast_f32": [_f, _f, _fl, _f, _fl, _ll, _ll, _st],
    "hf_add_bcast_f32": [_f, _f, _f, _ll, _ll, _st],
    "hf_label_conv3x3_f32": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _st],
    "hf_ace_modulate_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _i, _st],
    "hf_conv2d_f16_split_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f, _f, _f, _f, _i, _f, _fl, _f, _i, _i, _i, _i, _i, _i, _st],
    "hf_stem7x7s2_f16_f32": [_f, _f, _f, _f, _f, _f, _f, _fl, _i, _i, _i, _i, _i, _st],
    "hf_ace_modulate_table_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _fl, _i, _i, _st],
    "hf_region_mean_f32": [_f, _f, _f, _i, _i, _i, _i, _i, _ll, _ll, _i, _i, _st],
    "hf_tanh_f32": [_f, _f, _ll, _st],
    "hf_channel_layernorm_f32": [_f, _f, _f, _f, _i, _ll, _fl, _st],
    "hf_mha_small_f32": [_f, _f, _i, _i, _i, _i, _st],
    "hf_quick_gelu_f32": [_f, _f, _ll, _st],
    "hf_debug_set_dispatch": [_i, _i],
    "hf_debug_last_path": [],
    "hf_debug_set_persistent_blocks": [_i],
    "hf_debug_set_tuning": [_i],
    "hf_set_batch_invariant": [_i],
    "hf_set_splitk_counters": [_f, _i],
    "hf_modconv3x3_f16_rgb_slabs": [_i],
    "hf_profile_marker": [_i, _st],
    "hf_conv2d_f16_split_output_ok": [_i, _i, _i, _i, _i, _i, _i, _i],
    "hf_scale_shortcut_add_split_f16": [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _st],
}


class HairfastLibError(RuntimeError):
    pass


def bind(cdll):
    """Attach argtypes/restype for every entry point of the ABI; raises if one is missing."""
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError as e:
            raise HairfastLibError(f"{cdll._name}: missing symbol {name}") from e
        fn.argtypes = args
        fn.restype = ctypes.c_int
    cdll.hf_strerror.argtypes = [ctypes.c_int]
    cdll.hf_strerror.restype = ctypes.c_char_p
    cdll.hf_modconv_workspace_floats.argtypes = [_i, _i, _i, _i, _i, _i]
    cdll.hf_modconv_workspace_floats.restype = ctypes.c_longlong
    cdll.hf_conv2d_workspace_floats.argtypes = [_i, _i, _i, _i, _i, _i, _i, _i]
    cdll.hf_conv2d_workspace_floats.restype = ctypes.c_longlong
    cdll.hf_conv2d_f16_workspace_floats.argtypes = [_i, _i, _i, _i, _i, _i, _i]
    cdll.hf_conv2d_f16_workspace_floats.restype = ctypes.c_longlong
    cdll.hf_modconv3x3_small_workspace_floats.argtypes = [_i, _i, _i, _i, _i]
    cdll.hf_modconv3x3_small_workspace_floats.restype = ctypes.c_longlong
    cdll.hf_conv1x1_f16_workspace_floats.argtypes = [_i, _i, _i, _i, _i, _i, _i]
    cdll.hf_conv1x1_f16_workspace_floats.restype = ctypes.c_longlong
    cdll.hf_sample_layernorm_workspace_floats.argtypes = [_i, _i, _i]
    cdll.hf_sample_layernorm_workspace_floats.restype = ctypes.c_longlong
    cdll.hf_f16_overflow_count.argtypes = [_i]
    cdll.hf_f16_overflow_count.restype = ctypes.c_longlong
    cdll.hf_abi_version.argtypes = []
    cdll.hf_abi_version.restype = ctypes.c_int
    return cdll


_LIB = None


def load():
    """Load (once) and return the bound HIP library.  No fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HairfastLibError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or hairfastgan_amd/csrc/build.sh (hipcc --offload-arch=gfx950). "
                "hairfastgan_amd has no CPU / PyTorch fallback.")
        _LIB = bind(ctypes.CDLL(LIB_PATH))
    return _LIB


def check(lib, code, what):
    if code != 0:
        raise RuntimeError(f"{what}: {lib.hf_strerror(code).decode()} (code {code})")
