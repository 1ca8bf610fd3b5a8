"""ctypes binding of libsdmi.so (the C ABI in include/sdmi.h).  No CPU fallback exists: if the library is missing
or fails to load, importing this module raises, and every wrapper raises ``SdmiError`` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDMI_LIB") or os.path.join(_HERE, "lib", "libsdmi.so")      # SDMI_LIB: A/B of two builds (tools/gpu)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sdmi.h")

F16, F32 = 0, 1
EP_OUT_F32, EP_GEGLU, EP_NCHW, EP_BIAS_ROW = 1, 2, 4, 8
EP_TRANSPOSE = 64
EP_WRAP = 128


class SdmiError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("resid", C.c_void_p), ("out", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32), ("lda0", C.c_int32), ("lda1", C.c_int32),
        ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("taps", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("up", C.c_int32),
        ("N", C.c_int32), ("n_real", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32),
        ("flags", C.c_int32), ("alpha", C.c_float), ("batch", C.c_int32),
        ("a_bs", C.c_int64), ("w_bs", C.c_int64), ("o_bs", C.c_int64), ("r_bs", C.c_int64),
        ("force_generic", C.c_int32), ("reserved", C.c_int32),
        ("splitk_workspace", C.c_void_p), ("splitk_workspace_bytes", C.c_int64),
    ]


class UNetConfigC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("model_channels", C.c_int32),
        ("num_levels", C.c_int32), ("channel_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32),
        ("attn_level", C.c_int32 * 8), ("transformer_depth", C.c_int32 * 8),
        ("num_heads", C.c_int32), ("num_head_channels", C.c_int32), ("context_dim", C.c_int32),
        ("adm_in_channels", C.c_int32), ("reserved", C.c_int32 * 4),
    ]


class VAEConfigC(C.Structure):
    _fields_ = [
        ("ch", C.c_int32), ("num_levels", C.c_int32), ("ch_mult", C.c_int32 * 8),
        ("num_res_blocks", C.c_int32), ("in_channels", C.c_int32), ("out_ch", C.c_int32), ("z_channels", C.c_int32),
        ("scale_factor", C.c_float), ("reserved", C.c_int32 * 4),
    ]


class ClipConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("max_positions", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32),
                ("heads", C.c_int32), ("intermediate", C.c_int32), ("act", C.c_int32), ("eps", C.c_float)]


def declared_symbols() -> list:
    """Every function name declared in include/sdmi.h (used by the CPU-side export test)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdmi_[a-z0-9_]+)\s*\(", src)))


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback for the MI355X engine.")

lib = C.CDLL(LIB_PATH)

_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
_SIGS = {
    "sdmi_version": (C.c_int, []),
    "sdmi_last_error": (C.c_char_p, []),
    "sdmi_device_ok": (C.c_int, []),
    "sdmi_attention_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "sdmi_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i64, _vp]),
    "sdmi_attention_wide_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "sdmi_attention_wide": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i64, _vp]),
    "sdmi_attention_vt": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sdmi_conv_gemm": (_i, [C.POINTER(ConvDesc), _vp]),
    "sdmi_conv_splitk_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "sdmi_bench_conv_gemm": (_i, [C.POINTER(ConvDesc), _i, C.POINTER(C.c_float), _vp]),
    "sdmi_pack_conv_weight": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sdmi_groupnorm": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _i64, _vp]),
    "sdmi_groupnorm_workspace_bytes": (_i64, [_i, _i, _i]),
    "sdmi_layernorm": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "sdmi_rowchain_ff_pack_bytes": (_i64, [_i, _i]),
    "sdmi_rowchain_ff_pack": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sdmi_rowchain_ff": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _vp]),
    "sdmi_philox_randn": (_i, [_vp, _i64, C.c_uint64, C.c_uint32, _vp]),
    "sdmi_slerp": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp]),
    "sdmi_cfg_prepare_input": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "sdmi_cfg_prepare_concat": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _u32, _vp]),
    "sdmi_cfg_combine": (_i, [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "sdmi_cfg_combine_affine": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "sdmi_euler_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    "sdmi_dpmpp2m_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    "sdmi_ddim_step": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    "sdmi_axpby": (_i, [_vp, _vp, _f, _vp, _f, _i64, _vp]),
    "sdmi_dpm_error_partials": (_i, [_vp, _vp, _vp, _f, _f, _vp, _i64, _vp]),
    "sdmi_lincomb": (_i, [_vp, C.POINTER(_vp), C.POINTER(_f), _i, _i64, _vp]),
    "sdmi_mask_blend": (_i, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "sdmi_latent_resize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sdmi_image_to_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sdmi_engine_create": (_vp, [_i]),
    "sdmi_engine_destroy": (None, [_vp]),
    "sdmi_unet_configure": (_i, [_vp, C.POINTER(UNetConfigC)]),
    "sdmi_unet_load_tensor": (_i, [_vp, C.c_char_p, _vp, _i, _i, C.POINTER(C.c_int64), _i]),
    "sdmi_unet_finalize": (_i, [_vp]),
    "sdmi_clip_configure": (_i, [_vp, _i, C.POINTER(ClipConfigC)]),
    "sdmi_clip_load_tensor": (_i, [_vp, _i, C.c_char_p, _vp, _i, _i, C.POINTER(C.c_int64), _i]),
    "sdmi_clip_finalize": (_i, [_vp, _i]),
    "sdmi_clip_forward": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sdmi_unet_update_weight": (_i, [_vp, C.c_char_p, _vp, _i, _i, C.POINTER(C.c_int64), _i]),
    "sdmi_unet_hypernet_clear": (_i, [_vp]),
    "sdmi_unet_hypernet_begin": (_i, [_vp, _f]),
    "sdmi_unet_hypernet_linear": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i]),
    "sdmi_unet_hypernet_act": (_i, [_vp, _i, _i, _i]),
    "sdmi_unet_hypernet_layernorm": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i]),
    "sdmi_unet_update_vector": (_i, [_vp, C.c_char_p, _vp, _i, _i64, _i]),
    "sdmi_lora_merge": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp]),
    "sdmi_weight_hadamard": (_i, [_vp, _vp, _vp, _vp, _f, _i64, _vp]),
    "sdmi_weight_kron": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "sdmi_weight_ia3": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sdmi_weight_dora": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sdmi_vae_configure": (_i, [_vp, C.POINTER(VAEConfigC)]),
    "sdmi_vae_load_tensor": (_i, [_vp, C.c_char_p, _vp, _i, _i, C.POINTER(C.c_int64), _i]),
    "sdmi_vae_finalize": (_i, [_vp]),
    "sdmi_unet_set_context": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sdmi_unet_set_context_cached": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sdmi_unet_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sdmi_unet_set_control": (_i, [_vp, _vp, _vp, _i, _i]),
    "sdmi_unet_forward_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sdmi_vae_decode": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sdmi_vae_encode": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sdmi_debug_set": (_i, [C.c_char_p, _i]),
    "sdmi_debug_set_str": (_i, [C.c_char_p, C.c_char_p]),
    "sdmi_profile_begin": (_i, []),
    "sdmi_profile_end": (_i, [C.c_char_p, _i]),
    "sdmi_engine_arena_bytes": (_i64, [_vp]),
    "sdmi_engine_set_option": (_i, [_vp, C.c_char_p, _i]),
    "sdmi_engine_tap_count": (_i, [_vp]),
    "sdmi_engine_tap_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_int64)]),
    "sdmi_engine_tap_read": (_i, [_vp, _i, _vp, _vp]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)       # AttributeError here == the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return (lib.sdmi_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "sdmi call"):
    if rc != 0:
        raise SdmiError(f"{what} failed: {last_error()}")


def device_ok() -> bool:
    return bool(lib.sdmi_device_ok())


def require_device():
    if not device_ok():
        raise SdmiError("no gfx950 (MI355X) device visible to HIP: the engine has no CPU path")


def ptr(t):
    """data_ptr of a torch tensor (must be contiguous where the kernel assumes it) or None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t) -> int:
    import torch
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float32:
        return F32
    raise SdmiError(f"unsupported dtype {t.dtype}: the engine takes fp16 or fp32 tensors")
