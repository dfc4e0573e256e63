"""The C-ABI library: loads, exports every symbol include/vitk.h declares, and rejects bad
arguments before touching the GPU (none of these calls launches anything)."""
import ctypes
import os
import re

import pytest

from vit_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "vitk.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vitk_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    names = _declared()
    assert names == lib.EXPORTS
    for n in names:
        assert getattr(L, n) is not None
    assert L.vitk_version() == lib.ABI_VERSION


def test_argument_errors_are_reported_not_launched():
    L = lib.load()
    assert L.vitk_layernorm(None, 768, None, None, None, 768, 4, 768, 1e-6, 1, None) == 1
    assert b"null" in L.vitk_last_error()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    assert L.vitk_layernorm(p, 700, p, p, p, 700, 4, 700, 1e-6, 1, None) == 2      # D not supported
    assert L.vitk_attention(p, p, p, p, 1, 1, 197, 48, 0.1, None) == 2             # head_dim 48
    assert b"hd=48" in L.vitk_last_error()
    assert L.vitk_attention(p, p, p, p, 1, 1, 20000, 64, 0.1, None) == 2           # too many tokens
    assert L.vitk_gemm_bias_gelu(p, p, p, p, 16, 16, 40, None) == 2                # K % 32
    assert L.vitk_gemm_bias(p, p, p, p, 16, 16, 64, 7, None) == 2                  # dtype code
    assert L.vitk_gemm_bias_residual(p, p, p, None, p, 16, 16, 64, None) == 1
    assert L.vitk_patch_embed(p, p, p, p, p, p, 1, 224, 224, 14, 1280, None) == 2  # patch 14
    assert L.vitk_qkv_proj(p, p, p, p, 1, 197, 12, 60, None) == 2
    assert L.vitk_cast_f32_bf16(p, p, 0, None) == 1
    assert L.vitk_gemm_f32(p, p, p, None, p, 16, 16, 64, 2, None) == 1             # residual epilogue without residual
    assert L.vitk_gemm_f32(p, p, p, None, p, 16, 16, 64, 9, None) == 1             # epilogue code
    assert L.vitk_attention_f32(p, p, p, p, 1, 1, 197, 48, 0.1, None) == 2         # head_dim
    assert L.vitk_attention_f32(p, p, p, p, 1, 1, 2000, 64, 0.1, None) == 2        # too many tokens
    assert L.vitk_gemm_bias(p + 4, p, p, p, 16, 16, 64, 1, None) == 1              # misaligned


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libvitk.so")
    with pytest.raises(lib.VitkError):
        lib.load()
