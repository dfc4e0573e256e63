"""ctypes binding of libvitk.so (the C ABI declared in include/vitk.h).

The library is the only compute back end of this package: if it is missing or
a call fails, the caller gets an exception, never a silent fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITK_LIB selects another build of the same ABI (the diagnostic build with timing probes, libvitk_probes.so)
LIB_PATH = os.environ.get("VITK_LIB") or os.path.join(_HERE, "libvitk.so")

VITK_F32, VITK_BF16 = 0, 1
ABI_VERSION = 10

_p = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float

# name -> argument types; every function returns int except where noted.
SIGNATURES = {
    "vitk_layernorm": [_p, _i64, _p, _p, _p, _i64, _i, _i, _f, _i, _p],
    "vitk_cast_f32_bf16": [_p, _p, _i64, _p],
    "vitk_patch_embed": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "vitk_patch_embed_cols": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "vitk_gemm_bias": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vitk_gemm_bias_gelu": [_p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_gemm_bias_tanh": [_p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_gemm_bias_residual": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_qkv_proj": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vitk_attention": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "vitk_gemm_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vitk_qkv_proj_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vitk_patch_embed_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "vitk_attention_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "vitk_pos_embed_scale": [_p, _p, _i, _i, _i, _p],
    "vitk_normalize_u8": [_p, _p, _i, _i, _i, _p, _p, _i, _p],
    "vitk_target_rank": [_p, _p, _p, _i, _i, _p],
    "vitk_gemm_bias_gelu_dual": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_attention_lse": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "vitk_attention_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "vitk_colsum_bf16": [_p, _p, _p, _i, _i, _i, _p],
    "vitk_gemm_tn": [_p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "vitk_gemm_tn_colsum": [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "vitk_layernorm_backward": [_p, _i64, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "vitk_gemm_grad_act": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vitk_init": [],
    "vitk_gemm_plan": [_i, _i, _i, _i, _p],
    "vitk_colsum_bf16_static": [_p, _p, _i, _i, _p],
    "vitk_attention_rows": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "vitk_gemm_bias_residual_strided": [_p, _i64, _p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "vitk_gemm_bias_residual_ln": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_gemm_bias_residual_planes": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_ln_finalize": [_p, _p, _i, _i, _f, _p],
    "vitk_gemm_ln_bias_gelu": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vitk_qkv_proj_ln": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
}
EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_TANH = 0, 1, 2, 3      # vitk_gemm_f32 epilogues (include/vitk.h)
ACT_GELU, ACT_TANH = 0, 1                                    # vitk_gemm_grad_act
EXPORTS = sorted(list(SIGNATURES) + ["vitk_version", "vitk_last_error", "vitk_set_debug_variant", "vitk_set_cu_limit",
                                    "vitk_gemm_tn_workspace", "vitk_gemm_tn_colsum_workspace"])

_lib = None


class VitkError(RuntimeError):
    """A libvitk entry point returned a non-zero code."""


def load():
    """Load libvitk.so once and attach argument/return types."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitkError(
            f"{LIB_PATH} is missing: build it with `python -m vit_amd.build` "
            "(there is no CPU or eager fallback for the ViT forward path)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vitk_version.restype = _i
    lib.vitk_version.argtypes = []
    lib.vitk_last_error.restype = ctypes.c_char_p
    lib.vitk_last_error.argtypes = []
    lib.vitk_set_debug_variant.restype = None
    lib.vitk_set_debug_variant.argtypes = [_i, _i]
    lib.vitk_set_cu_limit.restype = None
    lib.vitk_set_cu_limit.argtypes = [_i]
    for fn in (lib.vitk_gemm_tn_workspace, lib.vitk_gemm_tn_colsum_workspace):
        fn.restype = _i64
        fn.argtypes = [_i, _i, _i]
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _i
        fn.argtypes = args
    v = lib.vitk_version()
    if v != ABI_VERSION:
        raise VitkError(f"libvitk ABI version {v}, expected {ABI_VERSION}: rebuild with `python -m vit_amd.build --force`")
    lim = os.environ.get("VITK_CU_LIMIT", "")       # measurement runs (tools/): plan for part of the chip
    if lim.isdigit():
        lib.vitk_set_cu_limit(int(lim))
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise VitkError(f"libvitk error {rc}: {lib.vitk_last_error().decode()}")
