"""ctypes binding of libunirestore_hip.so (the C ABI declared in include/unirestore_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module raises at
import time, so a GPU run can never silently execute something else.
"""
import ctypes as C
import os

# torch FIRST: it brings its own HIP runtime (torch/lib/libamdhip64.so).  If libunirestore_hip.so is dlopen'ed before torch, the
# process ends up with /opt/rocm's runtime bound to this library and torch's bound to torch - and the library's first launch fails
# with "no ROCm-capable device is detected" (seen with __graft_entry__.build() followed by smoke() in one process, round 5).
import torch  # noqa: F401  (load order only)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UR_LIB") or os.path.join(_HERE, "libunirestore_hip.so")     # UR_LIB: an A/B build of the same ABI

UR_ACT_NONE, UR_ACT_SILU, UR_ACT_GELU, UR_ACT_GEGLU, UR_ACT_GATE, UR_ACT_TANH, UR_ACT_RELU = range(7)
UR_DT_BF16, UR_DT_F16 = 0, 1
UR_E_INVALID, UR_E_UNSUPPORTED = -1, -2


class ConvDesc(C.Structure):
    """Mirror of `ur_conv_desc` (field order and types must match the header exactly)."""
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("y", C.c_void_p), ("yt", C.c_void_p), ("gn_part", C.c_void_p), ("gn_ab", C.c_void_p), ("gn_silu", C.c_int),
        ("row_stats", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_dim", C.c_int), ("ln_parts", C.c_int),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("C1", C.c_int), ("ldx", C.c_int), ("C2", C.c_int), ("ldx2", C.c_int),
        ("Cout", C.c_int), ("ldw", C.c_int), ("ldy", C.c_int), ("ldr", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad_t", C.c_int), ("pad_l", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int),
        ("upsample2x", C.c_int), ("act", C.c_int), ("out_f32", C.c_int),
        ("n_split", C.c_int), ("t_rows", C.c_int), ("t_ld", C.c_int),
        ("out_scale", C.c_float),
        ("nbatch", C.c_int),
        ("bs_x", C.c_longlong), ("bs_x2", C.c_longlong), ("bs_w", C.c_longlong), ("bs_bias", C.c_longlong),
        ("bs_y", C.c_longlong), ("bs_r", C.c_longlong), ("k_chunk_major", C.c_int), ("bias_img_stride", C.c_longlong),
        ("dtype", C.c_int), ("w_frag", C.c_void_p),
    ]


class ConvPlan(C.Structure):
    """Mirror of `ur_conv_plan`."""
    _fields_ = [("row_stat_parts", C.c_int), ("gn_parts", C.c_int), ("gn_fused", C.c_int), ("prologue_ok", C.c_int)]


_P, _I, _F, _LL, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t

# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "ur_version": (_I, []),
    "ur_last_error": (C.c_char_p, []),
    "ur_conv2d_nhwc": (_I, [C.POINTER(ConvDesc), _P]),
    "ur_conv2d_plan": (_I, [C.POINTER(ConvDesc), C.POINTER(ConvPlan)]),
    "ur_gemm_bias_act": (_I, [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _I, _P]),
    "ur_groupconv3x3_nhwc": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _I, _P]),
    "ur_groupnorm_stats_parts": (_I, [_I, _I, _I]),
    "ur_groupnorm_ws_bytes": (_SZ, [_I, _I, _I]),
    "ur_groupnorm_ab_bytes": (_SZ, [_I, _I]),
    "ur_groupnorm_stats": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ur_instnorm_stats": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ur_groupnorm_finalize": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "ur_groupnorm_apply_act": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ur_groupnorm_nhwc": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _I, _P, _I, _I, _P]),
    "ur_layernorm_rows": (_I, [_P, _P, _P, _P, _LL, _I, _F, _I, _P]),
    "ur_softmax_rows_f32": (_I, [_P, _P, _LL, _I, _I, _I, _P]),
    "ur_attention_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _LL, _LL, _LL, _F, _I, _P]),
    "ur_attention_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "ur_attention_fwd_ws": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _LL, _LL, _LL, _F, _P, _SZ, _I, _P]),
    "ur_chain_tile_bytes": (_SZ, []),
    "ur_ff_geglu_fused": (_I, [_P, _P, _SZ, _P, _LL, _I, _I, _I, _I, _F, _I, _P]),
    "ur_transformer_head_fused": (_I, [_P, _P, _P, _SZ, _P, _P, _P, _P, _LL, _I, _I, _F, _I, _P]),
    "ur_transformer_tail_fused": (_I, [_P, _P, _P, _P, _SZ, _P, _P, _LL, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "ur_csce_fused": (_I, [_P, _P, _P, _SZ, _P, _P, _LL, _I, _I, _I, _I, _P]),
    "ur_dwconv3x3_nhwc": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ur_avgpool_hw": (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    "ur_scale_channels": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ur_axpy_channels": (_I, [_P, _P, _P, _P, _LL, _I, _I, _P]),
    "ur_spade_modulate": (_I, [_P, _P, _I, _P, _P, _LL, _I, _I, _P]),
    "ur_linear_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ur_tfa_prompt_update": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ur_vec_mul_group": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ur_image_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ur_nhwc_to_nchw_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "ur_nchw_f32_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ur_image_resize_pad_nhwc": (_I, [_P, _P] + [_I] * 9 + [_F, _F, _I, _P]),
    "ur_image_unpad_resize_nchw": (_I, [_P, _I, _P] + [_I] * 9 + [_F, _F, _I, _I, _P]),
    "ur_vae_sample": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "ur_add_noise": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P]),
    "ur_ddim_step": (_I, [_P, _P, _I, _P, _LL, _I, _I, _F, _F, _I, _P]),
    "ur_f32_to_bf16_scaled": (_I, [_P, _I, _P, _LL, _I, _I, _F, _I, _P]),
    "ur_profile_enable": (_I, [_I]),
    "ur_profile_report": (_I, [C.c_char_p, _SZ]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m unirestore_amd.build` (or __graft_entry__.build()). "
            "There is no CPU / eager fallback in the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class URError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        msg = lib.ur_last_error().decode("utf-8", "replace")
        if rc == UR_E_INVALID:
            raise ValueError(f"unirestore_hip: {msg}")
        if rc == UR_E_UNSUPPORTED:
            raise NotImplementedError(f"unirestore_hip (UR_E_UNSUPPORTED): {msg}")
        raise URError(f"unirestore_hip error {rc}: {msg}")
