"""ctypes binding of ``libclipmi.so`` (C ABI declared in ``include/clipmi.h``).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object is missing or a symbol
cannot be resolved this module raises at import time, and every wrapper raises ``ClipmiError`` on a non-zero
return code with the library's own error text.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLIPMI_LIBRARY selects another BUILD of the same library (the tuning build with phase stamps); never a different backend
LIB_PATH = os.environ.get("CLIPMI_LIBRARY") or os.path.join(_HERE, "csrc", "libclipmi.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "clipmi.h")

ABI_VERSION = 13

OK, ERR_ARG, ERR_SHAPE, ERR_HIP, ERR_WORKSPACE, ERR_STATE = 0, -1, -2, -3, -4, -5
F16, F32 = 0, 1
COMM_ID_BYTES = 128
EPI_NONE, EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL, EPI_BIAS_RELU, EPI_BIAS_RESIDUAL16_RELU = 0, 1, 2, 3, 4, 5
CALL_DEFAULT, CALL_STREAM_F32, CALL_STREAM_F16 = 0, 1, 2   # per-call flags of the tower calls (include/clipmi.h)


class ClipmiError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str):
        super().__init__(f"{what} failed: {detail or '?'} (code {code})")
        self.code = code


class Geometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "embed_dim", "image_resolution", "patch_size", "vision_width", "vision_layers",
        "context_length", "vocab_size", "text_width", "text_layers", "text_heads")]


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_out", "b_out", "ln2_g", "ln2_b", "w_fc", "b_fc", "w_proj", "b_proj",
        "w_qkv_f", "g_qkv", "c_qkv", "w_fc_f", "g_fc", "c_fc")]


class VisionWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "conv_w", "class_embedding", "positional_embedding", "ln_pre_g", "ln_pre_b", "ln_post_g", "ln_post_b",
        "proj_t")] + [("blocks", C.POINTER(BlockWeights))]


class TextWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "token_embedding", "positional_embedding", "ln_final_g", "ln_final_b", "proj_t")] + [
        ("blocks", C.POINTER(BlockWeights))]


class PromptHook(C.Structure):
    _fields_ = [("n_ctx", C.c_int32), ("n_deep", C.c_int32), ("shallow", C.c_void_p), ("deep", C.c_void_p)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
        f"g.build()'` (or `make -C clip_calibration_amd/csrc`). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)

_vp, _i, _i64, _f, _sz, _u = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint

_SIGNATURES = {
    "clipmi_abi_version": (C.c_int, []),
    "clipmi_strerror": (C.c_char_p, [_i]),
    "clipmi_last_error": (C.c_char_p, []),
    "clipmi_set_option": (_i, [C.c_char_p, _i]),
    "clipmi_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "clipmi_gemm_f16": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "clipmi_gemm_residual_f16": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, C.POINTER(_i), _i, _i, _i, _vp]),
    "clipmi_layernorm": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp]),
    "clipmi_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "clipmi_patchify": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "clipmi_patch_embed_scratch_bytes": (_sz, [_i, _i, _i]),
    "clipmi_patch_embed": (_i, [_vp, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "clipmi_embed_ln": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "clipmi_l2_normalize": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "clipmi_logits": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "clipmi_fused_tail_workspace_bytes": (_sz, [_i, _i]),
    "clipmi_l2_normalize_to": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "clipmi_comm_unique_id": (_i, [_vp]),
    "clipmi_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "clipmi_comm_destroy": (_i, [_vp]),
    "clipmi_comm_ranks": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "clipmi_allgather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "clipmi_fused_tail": (_i, [_vp, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "clipmi_calibrate_rows": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "clipmi_conv3x3_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "clipmi_im2col3x3_nchw": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "clipmi_im2col3x3_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "clipmi_avgpool_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "clipmi_attnpool_tokens": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "clipmi_attnpool": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "clipmi_adapter_blend": (_i, [_vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _vp]),
    "clipmi_scale_add": (_i, [_vp, _vp, C.c_float, _vp, C.c_longlong, _vp]),
    "clipmi_group_mean": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "clipmi_cocoop_ctx": (_i, [_vp] * 7 + [_i] * 5 + [_vp]),
    "clipmi_cocoop_prompts": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "clipmi_logits_per_image": (_i, [_vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "clipmi_softmax_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "clipmi_knn_dists": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "clipmi_ece_accumulate": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp]),
    "clipmi_create": (_i, [C.POINTER(Geometry), C.POINTER(_vp)]),
    "clipmi_destroy": (_i, [_vp]),
    "clipmi_set_vision_weights": (_i, [_vp, C.POINTER(VisionWeights)]),
    "clipmi_set_text_weights": (_i, [_vp, C.POINTER(TextWeights)]),
    "clipmi_vision_workspace_bytes": (_sz, [_vp, _i, _i]),
    "clipmi_text_workspace_bytes": (_sz, [_vp, _i, _i]),
    "clipmi_model_set_option": (_i, [_vp, C.c_char_p, _i]),
    "clipmi_model_get_option": (_i, [_vp, C.c_char_p, C.POINTER(_i)]),
    "clipmi_encode_image": (_i, [_vp, _vp, _i, _i, C.POINTER(PromptHook), _vp, _vp, _sz, _u, _vp]),
    "clipmi_text_blocks": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(PromptHook), _vp, _sz, _u, _vp]),
    "clipmi_text_encoder": (_i, [_vp, _vp, _i, _vp, _i, _i, C.POINTER(PromptHook), _vp, _vp, _sz, _u, _vp]),
    "clipmi_encode_text": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _u, _vp]),
    "clipmi_profile_block": (_i, [_vp, _i, _i, _i, _vp, _sz, C.POINTER(_f), _vp]),
    "clipmi_encode_image_timed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _u, C.POINTER(_f), _i, C.POINTER(_i), _vp]),
    "clipmi_probe_mfma_f16": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(_i), _vp]),
}

for _name, (_res, _args) in _SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:  # pragma: no cover
        raise ImportError(f"{LIB_PATH} does not export {_name}; rebuild the extension") from e
    _fn.restype = _res
    _fn.argtypes = _args

if lib.clipmi_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH}: ABI version {lib.clipmi_abi_version()} != {ABI_VERSION}; rebuild the extension")


def last_error() -> str:
    return (lib.clipmi_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise ClipmiError(rc, what, last_error() or (lib.clipmi_strerror(rc) or b"").decode())


def set_option(name: str, value: int) -> None:
    """Process-wide runtime switch (include/clipmi.h, clipmi_set_option): tests and A/B tools only -- product code selects
    precision per model (CLIP.set_option) or per call (flags), never through this."""
    check(lib.clipmi_set_option(name.encode(), int(value)), "clipmi_set_option")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib.clipmi_get_option(name.encode(), C.byref(v)), "clipmi_get_option")
    return v.value


class option:
    """``with option("ln_fold", 0): ...`` -- set a switch for a block and restore the previous value."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


_VARIANT_LETTERS = {"a": 10, "s": 13, "r": 16}


def gemm_variant_id(v) -> int:
    """'0' '1' / 'a' 's' 'r' (the CLIPMI_GEMM_VARIANT spellings) or an int -> option value; None -> -1."""
    if v is None:
        return -1
    if isinstance(v, int):
        return v
    return _VARIANT_LETTERS.get(v, None) if v in _VARIANT_LETTERS else int(v)


def exported_symbols():
    return sorted(_SIGNATURES)
