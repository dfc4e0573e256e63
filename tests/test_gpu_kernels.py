"""Kernel-level parity on the GPU, through the C ABI (vit_amd.ops -> ctypes -> libvitk.so).

Each kernel is compared with the oracle's op on the same seeded inputs.  GEMM-family inputs
are rounded to bf16 first (that is the kernels' operand type), so the only differences left
are fp32 summation order and the final bf16 store; tolerances are written per test.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vit_oracle as vo
from vit_amd import lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to16(t):
    return t.to(torch.bfloat16)


def f32(t):
    return t.detach().float().cpu()


@pytest.fixture(autouse=True)
def _reset_variants():
    lib.load().vitk_set_debug_variant(0, 0)
    yield
    lib.load().vitk_set_debug_variant(0, 0)
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("D", [256, 768, 1024, 1280])
@pytest.mark.parametrize("rows", [1, 5, 788, 20000])
def test_layernorm_bf16_out(D, rows):
    x = rnd(rows, D, seed=rows + D) * 2.0 + 0.3
    g, b = 1 + 0.1 * rnd(D, seed=1), 0.1 * rnd(D, seed=2)
    want = vo.layernorm(x, g, b, 1e-6)
    out = torch.empty(rows, D, dtype=torch.bfloat16, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, out)
    # fp32 statistics, one bf16 rounding on store: half an ulp of bf16 = 2^-9 relative
    assert torch.allclose(f32(out), want, rtol=2 ** -8, atol=1e-5)


def test_layernorm_f32_out_and_strided_rows():
    B, N, D = 6, 197, 768
    x = rnd(B, N, D, seed=3)
    g, b = 1 + 0.1 * rnd(D, seed=1), 0.1 * rnd(D, seed=2)
    want = vo.layernorm(x[:, 0], g, b, 1e-5)
    out = torch.empty(B, D, dtype=torch.float32, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, out, rows=B, x_row_stride=N * D)
    assert torch.allclose(f32(out), want, rtol=1e-5, atol=2e-6)


def test_cast():
    x = rnd(3, 3, 64, 64, seed=4)
    assert torch.equal(f32(ops.cast_bf16(x.to(DEV))), f32(x.to(torch.bfloat16)))


# ----------------------------------------------------------------------------- dense GEMMs
SHAPES = [(1576, 768, 768), (1576, 3072, 768), (1576, 768, 3072), (256, 1024, 1024), (4728, 2304, 768),
          (130, 256, 64), (1000, 1280, 5120), (8200, 2304, 768), (300, 512, 128), (300, 512, 192)]


def _dense_inputs(M, N, K, seed):
    a = to16(rnd(M, K, seed=seed))
    w = to16(rnd(N, K, seed=seed + 1, scale=0.05))
    bias = rnd(N, seed=seed + 2, scale=0.1)
    return a, w, bias


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 5, 0x43, 0x13, 0xa00, 0x45, 0xa05])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_bias_residual(M, N, K, variant):
    if (variant & 15) == 2 and M * N * K > 2e9:
        pytest.skip("small-kernel variant only on small shapes")
    lib.load().vitk_set_debug_variant(variant, 0)
    a, w, bias = _dense_inputs(M, N, K, 10)
    resid = rnd(M, N, seed=13)
    want = resid + a.float() @ w.float().t() + bias
    out = resid.to(DEV).clone()
    ops.gemm_bias_residual(a.to(DEV), w.to(DEV), bias.to(DEV), out, out)      # in place, as the model uses it
    assert torch.allclose(f32(out), want, rtol=1e-4, atol=2e-4 * (K / 768) ** 0.5)


@pytest.mark.parametrize("variant", [0, 1, 5])
@pytest.mark.parametrize("M,N,K", SHAPES[:5])
def test_gemm_bias_gelu(M, N, K, variant):
    lib.load().vitk_set_debug_variant(variant, 0)
    a, w, bias = _dense_inputs(M, N, K, 20)
    want = F.gelu(a.float() @ w.float().t() + bias)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bias_gelu(a.to(DEV), w.to(DEV), bias.to(DEV), out)
    assert torch.allclose(f32(out), want, rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(8, 768, 768), (256, 1000, 768), (3, 10, 64), (256, 768, 768)])
def test_head_gemms_small_and_irregular(M, N, K):
    a, w, bias = _dense_inputs(M, N, K, 30)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_bias(a.to(DEV), w.to(DEV), bias.to(DEV), out)
    assert torch.allclose(f32(out), a.float() @ w.float().t() + bias, rtol=1e-4, atol=2e-4)
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bias_tanh(a.to(DEV), w.to(DEV), bias.to(DEV), out16)
    assert torch.allclose(f32(out16), torch.tanh(a.float() @ w.float().t() + bias), rtol=2 ** -7, atol=2e-3)
    ops.gemm_bias(a.to(DEV), w.to(DEV), None, out16)
    assert torch.allclose(f32(out16), a.float() @ w.float().t(), rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("variant", [0, 1, 5, 0x33, 0x35])
@pytest.mark.parametrize("B,N,H,hd", [(3, 197, 12, 64), (2, 577, 12, 64), (2, 197, 16, 64), (2, 197, 8, 96)])
def test_qkv_projection_head_split(B, N, H, hd, variant):
    lib.load().vitk_set_debug_variant(variant, 0)
    D = H * hd
    a, w, bias = _dense_inputs(B * N, 3 * D, D, 40)
    want = (a.float() @ w.float().t() + bias).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    qkv = torch.empty(3, B, H, N, hd, dtype=torch.bfloat16, device=DEV)
    ops.qkv_proj(a.to(DEV), w.to(DEV), bias.to(DEV), qkv, B, N, H, hd)
    assert torch.allclose(f32(qkv), want, rtol=2 ** -7, atol=2e-3)
    ops.qkv_proj(a.to(DEV), w.to(DEV), None, qkv, B, N, H, hd)                  # attn_bias=False
    assert torch.allclose(f32(qkv), want - bias.reshape(3, 1, H, 1, hd), rtol=2 ** -7, atol=2e-3)


# ----------------------------------------------------------------------------- patch embedding
@pytest.mark.parametrize("variant", [0, 1, 5])
@pytest.mark.parametrize("B,S,P,D", [(3, 224, 16, 768), (2, 384, 16, 768), (5, 224, 32, 768), (2, 224, 16, 1024),
                                      (1, 64, 8, 256)])
def test_patch_embed(B, S, P, D, variant):
    lib.load().vitk_set_debug_variant(variant, 0)
    img = to16(rnd(B, 3, S, S, seed=50))
    w = to16(rnd(D, 3, P, P, seed=51, scale=0.03))
    bias, cls = rnd(D, seed=52, scale=0.05), rnd(1, 1, D, seed=53, scale=0.2)
    N = (S // P) ** 2 + 1
    pos = rnd(1, N, D, seed=54, scale=0.02)
    sd = {"embedding.weight": w.float(), "embedding.bias": bias, "cls_token": cls, "pos_embed": pos}
    want = vo.patch_embed(img.float(), sd, P)
    out = torch.full((B, N, D), float("nan"), dtype=torch.float32, device=DEV)
    ops.patch_embed(img.to(DEV), w.reshape(D, -1).contiguous().to(DEV), bias.to(DEV), cls.reshape(D).to(DEV),
                    pos.reshape(N, D).to(DEV), out, P)
    assert torch.allclose(f32(out), want, rtol=1e-4, atol=3e-4)


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,S,P,D", [(2, 224, 14, 1280), (3, 224, 16, 768), (2, 126, 14, 256), (1, 230, 12, 512)])
def test_patch_embed_through_explicit_patch_matrix(B, S, P, D):
    """Any patch size: im2col into a caller scratch + the tiled GEMM with the patch epilogue."""
    img = to16(rnd(B, 3, S, S, seed=31))
    w = to16(rnd(D, 3, P, P, seed=32, scale=0.03))
    bias, cls = rnd(D, seed=33, scale=0.1), rnd(D, seed=34)
    n = (S // P) ** 2
    pos = rnd(n + 1, D, seed=35, scale=0.1)
    conv = F.conv2d(img.float(), w.float(), bias, stride=P).flatten(2).transpose(1, 2)
    want = torch.cat([cls.expand(B, 1, D), conv], dim=1) + pos
    k0 = 3 * P * P
    kpad = (k0 + 63) // 64 * 64
    wp = torch.zeros(D, kpad, dtype=torch.bfloat16)
    wp[:, :k0] = w.reshape(D, -1)
    out = torch.full((B, n + 1, D), float("nan"), device=DEV)
    cols = torch.full((B * n, kpad), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.patch_embed_cols(img.to(DEV), wp.to(DEV), bias.to(DEV), cls.to(DEV), pos.to(DEV), out, cols, P)
    assert torch.allclose(out.cpu(), want, rtol=1e-4, atol=2e-4)
    assert (cols.float().cpu()[:, k0:] == 0).all()


def _attn_ref(q, k, v, scale):
    s = (q.double() @ k.double().transpose(-2, -1)) * scale
    p = s.softmax(-1)
    o = p @ v.double()                              # [B,H,N,hd]
    B, H, N, hd = q.shape
    return o.transpose(1, 2).reshape(B, N, H * hd).float()


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("B,H,N", [(2, 12, 197), (1, 12, 577), (3, 2, 1), (2, 3, 32), (2, 3, 33), (1, 4, 224),
                                   (1, 2, 225), (2, 16, 197), (1, 1, 608), (2, 2, 64), (2, 2, 65), (1, 2, 129)])
def test_attention(B, H, N, variant):
    _attention_case(B, H, N, 64, variant)


@pytest.mark.parametrize("B,H,N", [(2, 8, 197), (3, 2, 1), (2, 3, 32), (2, 3, 33), (1, 4, 224), (1, 2, 225), (1, 1, 288)])
def test_attention_head_dim_96(B, H, N):
    _attention_case(B, H, N, 96, 0)


@pytest.mark.parametrize("N,hd", [(1025, 64), (785, 96), (609, 64), (289, 96)])
def test_attention_long_sequences_stream_through_lds(N, hd):
    _attention_case(1, 2, N, hd, 0)


@pytest.mark.parametrize("B,H,N", [(2, 16, 257), (3, 2, 1), (2, 3, 33), (1, 2, 288), (1, 2, 320)])
def test_attention_head_dim_80(B, H, N):
    _attention_case(B, H, N, 80, 0)


@pytest.mark.parametrize("N,hd", [(197, 96), (577, 64), (33, 64), (257, 80)])
def test_attention_forced_streaming(N, hd):
    _attention_case(2, 3, N, hd, 2)


@pytest.mark.parametrize("B,H,N,hd", [(30, 12, 197, 64), (70, 12, 50, 64), (23, 16, 129, 96), (40, 16, 160, 80),
                                      (25, 12, 320, 64), (20, 16, 256, 64), (30, 12, 225, 64)])
def test_attention_more_heads_than_workgroups(B, H, N, hd):
    """More (image, head) pairs than CUs: the persistent workgroups walk several heads each through
    their two K/V buffers (some one head more than others)."""
    _attention_case(B, H, N, hd, 0)


def _attention_case(B, H, N, hd, variant):
    if variant in (1, 3) and N > 608:
        pytest.skip("resident K/V needs ntok <= 608")
    lib.load().vitk_set_debug_variant(0, variant)
    qkv = to16(rnd(3, B, H, N, hd, seed=N + variant))
    want = _attn_ref(qkv[0].float(), qkv[1].float(), qkv[2].float(), hd ** -0.5)
    ctx = torch.full((B, N, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.attention(qkv.to(DEV), ctx, hd ** -0.5)
    # probabilities are rounded to bf16 before P.V and the output is stored in bf16
    assert torch.allclose(f32(ctx), want, rtol=2e-2, atol=6e-3)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention_online_softmax_rescale_is_exercised(variant):
    """A key far down the sequence dominates: the running max jumps in a late key block, so every
    earlier partial sum must be rescaled.  Scores of ~+-60 also check the exp2 argument handling."""
    _rescale_case(64, variant)


def test_attention_online_softmax_rescale_head_dim_96():
    _rescale_case(96, 0)


def _rescale_case(hd, variant):
    lib.load().vitk_set_debug_variant(0, variant)
    B, H, N = 1, 2, 197
    qkv = to16(rnd(3, B, H, N, hd, seed=7))
    qkv[1, 0, 0, 150] = qkv[0, 0, 0, 10] * 6.0          # key 150 aligned with query 10 of head 0
    qkv[1, 0, 1, 190] = qkv[0, 0, 1, 100] * 8.0
    want = _attn_ref(qkv[0].float(), qkv[1].float(), qkv[2].float(), hd ** -0.5)
    ctx = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device=DEV)
    ops.attention(qkv.to(DEV), ctx, hd ** -0.5)
    assert torch.isfinite(f32(ctx)).all()
    assert torch.allclose(f32(ctx), want, rtol=2e-2, atol=6e-3)


def test_attention_custom_scale_and_row_independence():
    B, H, N, hd = 4, 12, 197, 64
    qkv = to16(rnd(3, B, H, N, hd, seed=9))
    ctx = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device=DEV)
    ops.attention(qkv.to(DEV), ctx, 0.3)
    assert torch.allclose(f32(ctx), _attn_ref(qkv[0].float(), qkv[1].float(), qkv[2].float(), 0.3),
                          rtol=2e-2, atol=6e-3)
    one = torch.empty(1, N, H * hd, dtype=torch.bfloat16, device=DEV)
    ops.attention(qkv[:, 2:3].contiguous().to(DEV), one, 0.3)
    assert torch.equal(one[0], ctx[2])               # an image's result does not depend on its batch


def test_unsupported_head_dim_raises():
    qkv = torch.zeros(3, 1, 8, 197, 48, dtype=torch.bfloat16, device=DEV)
    ctx = torch.empty(1, 197, 384, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(lib.VitkError, match="hd=48"):
        ops.attention(qkv, ctx, 0.1)


# ----------------------------------------------------------------------------- full-size launches
@pytest.mark.parametrize("M,N,K", [(50432, 768, 768), (50432 + 77, 768, 768), (50432, 1024, 1024), (73856, 768, 768),
                                   (50432, 768, 3072), (40000, 512, 256), (55065, 1280, 256), (34566, 2560, 128)])
def test_gemm_tile_plans_at_full_size(M, N, K):
    """Shapes with at least one output tile per CU and few n-tiles take the two-height tile plan (an XCD's rows cut
    into tall panels, then shorter ones): every element against a torch product on the GPU, and bit-identical to the
    equal-tile plan (debug order bit 13), which accumulates each element in the same order."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=DEV) * 0.1
    resid = torch.randn(M, N, generator=g, device=DEV)
    outs = []
    for variant in (0, 0x20000):
        lib.load().vitk_set_debug_variant(variant, 0)
        x = resid.clone()
        ops.gemm_bias_residual(a, w, bias, x, x)
        o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm_bias_gelu(a, w, bias, o16)
        outs.append((x, o16))
    acc = a.float() @ w.float().t() + bias
    assert torch.allclose(outs[0][0], resid + acc, rtol=1e-4, atol=2e-4 * max(K / 768, 1.0) ** 0.5)
    assert torch.allclose(outs[0][1].float(), F.gelu(acc), rtol=2 ** -7, atol=2e-3)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _plan_is_mixed(M, N, K):
    import ctypes
    out = (ctypes.c_int * 10)()
    assert lib.load().vitk_gemm_plan(M, N, K, 0, out) == 0
    return out[4] == 1


def test_qkv_projection_under_the_two_height_plan():
    """A projection narrow enough for the plain tile order (3 n-tiles) with at least one tile per CU takes the
    two-height plan, whose last panel of an XCD overhangs the XCD's row range: the head-major scatter epilogue must
    honour the row limit.  Against torch, and bit for bit against the equal-tile plan (debug order bit 13)."""
    B, N, H, hd = 139, 197, 4, 64
    D = H * hd
    assert _plan_is_mixed(B * N, 3 * D, D)
    g = torch.Generator(device=DEV).manual_seed(11)
    a = torch.randn(B * N, D, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(3 * D, D, generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    bias = torch.randn(3 * D, generator=g, device=DEV) * 0.1
    outs = []
    for variant in (0, 0x20000):
        lib.load().vitk_set_debug_variant(variant, 0)
        qkv = torch.full((3, B, H, N, hd), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.qkv_proj(a, w, bias, qkv, B, N, H, hd)
        outs.append(qkv)
    want = (a.float() @ w.float().t() + bias).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    assert torch.allclose(outs[0].float(), want, rtol=2 ** -7, atol=2e-3)
    assert torch.equal(outs[0], outs[1])


def test_patch_matrix_embedding_under_the_two_height_plan():
    """Patch 14 through the explicit patch matrix at a batch that gives every CU a tile (3 n-tiles, K = 640): the
    patch epilogue (row -> token 1 + patch, position embedding added) under the two-height plan, against the
    convolution and bit for bit against the equal-tile plan."""
    B, S, P, D = 107, 224, 14, 768
    n = (S // P) ** 2
    k0 = 3 * P * P
    kpad = (k0 + 63) // 64 * 64
    assert _plan_is_mixed(B * n, D, kpad)
    g = torch.Generator(device=DEV).manual_seed(12)
    img = torch.randn(B, 3, S, S, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(D, 3, P, P, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    bias, cls = torch.randn(D, generator=g, device=DEV) * 0.1, torch.randn(D, generator=g, device=DEV)
    pos = torch.randn(n + 1, D, generator=g, device=DEV) * 0.1
    wp = torch.zeros(D, kpad, dtype=torch.bfloat16, device=DEV)
    wp[:, :k0] = w.reshape(D, -1)
    outs = []
    for variant in (0, 0x20000):
        lib.load().vitk_set_debug_variant(variant, 0)
        out = torch.full((B, n + 1, D), float("nan"), device=DEV)
        cols = torch.empty(B * n, kpad, dtype=torch.bfloat16, device=DEV)
        ops.patch_embed_cols(img, wp, bias, cls, pos, out, cols, P)
        outs.append(out)
    conv = F.conv2d(img.float(), w.float(), bias, stride=P).flatten(2).transpose(1, 2)
    want = torch.cat([cls.expand(B, 1, D), conv], dim=1) + pos
    assert torch.allclose(outs[0], want, rtol=1e-4, atol=3e-4)
    assert torch.equal(outs[0], outs[1])


# ----------------------------------------------------------------------------- LayerNorm folded into the GEMMs around it
@pytest.mark.parametrize("M,D,K0", [(50432, 768, 768), (1000, 768, 3072), (197 * 5 + 3, 1024, 1024), (128, 256, 128)])
def test_residual_stream_as_two_16_bit_planes(M, D, K0):
    """vitk_gemm_bias_residual_planes: the residual GEMM of a block (models/transformer.py:68-76) with x kept as
    plane H = bf16(x) rounded half up and plane L = the lower half of x's bits.  All three forms (f32 in / planes out,
    planes in place, planes in / f32 out) against the f32-stream kernels on the same inputs: the stream's values bit
    for bit, the statistics bit for bit, plane H equal to the round-to-nearest bf16 copy except on exact ties."""
    g = torch.Generator(device=DEV).manual_seed(M + D + K0)
    a1 = torch.randn(M, K0, generator=g, device=DEV).to(torch.bfloat16)
    a2 = torch.randn(M, K0, generator=g, device=DEV).to(torch.bfloat16)
    a3 = torch.randn(M, K0, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(D, K0, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    b = torch.randn(D, generator=g, device=DEV) * 0.1
    x0 = torch.randn(M, D, generator=g, device=DEV) * 1.5 + 0.4 * torch.randn(M, 1, generator=g, device=DEV)
    x0[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.00390625, 1.01171875, 3.0e-39, -65280.0], device=DEV)   # zeros, ties, a denormal
    x0[0, 8:10] = torch.tensor([0x3FFFFFFF, -0x40000001], dtype=torch.int32, device=DEV).view(torch.float32)   # the bump carries through plane H
    # the helpers agree with each other on any bit pattern
    hh, ll = ops.f32_to_planes(x0)
    assert torch.equal(ops.planes_to_f32(hh, ll), x0)

    # reference: the f32 stream through the statistics epilogue three times
    xr = x0.clone()
    yr = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    sr = torch.empty(M, D // 64, 2, device=DEV)
    want = []
    for a in (a1, a2, a3):
        ops.gemm_bias_residual_ln(a, w, b, xr, xr, yr, sr)
        want.append((xr.clone(), yr.clone(), sr.clone()))

    ph = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    pl = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    st = torch.full((M, D // 64, 2), float("nan"), device=DEV)
    x_in = x0.clone()
    ops.gemm_bias_residual_planes(a1, w, b, ph, pl, st, x_in=x_in)                 # f32 in, planes out
    assert torch.equal(x_in, x0)                                                   # the f32 input is left alone
    for step, a in ((0, None), (1, a2)):
        if a is not None:
            ops.gemm_bias_residual_planes(a, w, b, ph, pl, st)                     # planes in place
        wx, wy, ws_ = want[step]
        assert torch.equal(ops.planes_to_f32(ph, pl), wx), step
        assert torch.equal(st, ws_), step
        hw, _ = ops.f32_to_planes(wx)
        assert torch.equal(ph.view(torch.int16), hw.view(torch.int16)), step
        differs = ph.view(torch.int16) != wy.view(torch.int16)                    # round half up vs half to even: exact ties only
        assert bool(((wx.view(torch.int32) & 0xFFFF) == 0x8000)[differs].all()) and int(differs.sum()) <= max(4, M * D // 20000)
    out = torch.full((M, D), float("nan"), device=DEV)
    ph_before, pl_before = ph.clone(), pl.clone()
    ops.gemm_bias_residual_planes(a3, w, b, ph, pl, x_out=out)                     # planes in, f32 out
    assert torch.equal(out, want[2][0])
    assert torch.equal(ph.view(torch.int16), ph_before.view(torch.int16)) and torch.equal(pl.view(torch.int16), pl_before.view(torch.int16))
    with pytest.raises(lib.VitkError):
        ops.gemm_bias_residual_planes(a3, w, b, ph, pl, st, x_out=out)             # statistics go with planes written


@pytest.mark.parametrize("M,D,N2,K0", [(50432, 768, 3072, 768), (2000, 768, 2304, 3072), (788, 1024, 4096, 1024),
                                       (131, 256, 768, 256), (50432 + 77, 768, 2304, 768), (5000, 1280, 5120, 1280)])
def test_layernorm_folded_into_the_gemms_either_side(M, D, N2, K0):
    """x = x + a W0^T + b0 with the statistics and the bf16 copy from the residual epilogue, then
    gelu(LN(x) W1^T + b1) and the q/k/v projection of LN(x) from that copy with gamma-scaled weights
    (models/transformer.py:68-76).  The producer against torch (f32 product of the bf16 operands), the row factors
    against f64 statistics of the produced rows, the consumers against the oracle's LayerNorm in f32 followed by an
    f32 product: the error left is the bf16 rounding of the operands, held to the size the unfolded kernels
    (vitk_layernorm + vitk_gemm_bias_gelu) show on the same inputs."""
    g = torch.Generator(device=DEV).manual_seed(M + D + N2)
    a = torch.randn(M, K0, generator=g, device=DEV).to(torch.bfloat16)
    w0 = (torch.randn(D, K0, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    b0 = torch.randn(D, generator=g, device=DEV) * 0.1
    x0 = torch.randn(M, D, generator=g, device=DEV) * 1.5 + 0.4 * torch.randn(M, 1, generator=g, device=DEV)
    gamma = 1 + 0.2 * torch.randn(D, generator=g, device=DEV)
    beta = 0.1 * torch.randn(D, generator=g, device=DEV)
    w1 = torch.randn(N2, D, generator=g, device=DEV) * 0.04
    b1 = torch.randn(N2, generator=g, device=DEV) * 0.1
    eps = 1e-6

    x = x0.clone()
    x16 = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    stats = torch.full((M, D // 64, 2), float("nan"), device=DEV)
    ab = torch.empty(M, 2, device=DEV)
    ops.gemm_bias_residual_ln(a, w0, b0, x, x, x16, stats)
    ops.ln_finalize(stats, ab, D, eps)
    plain = x0.clone()
    ops.gemm_bias_residual(a, w0, b0, plain, plain)
    assert torch.equal(x, plain)                                   # the same residual update, bit for bit
    assert torch.equal(x16, x.to(torch.bfloat16))
    xd = x.double()
    mean, var = xd.mean(1), xd.var(1, unbiased=False)
    rstd = (var + eps).rsqrt()
    assert torch.allclose(ab[:, 0].double(), rstd, rtol=2e-5, atol=0)
    assert torch.allclose(ab[:, 1].double(), -mean * rstd, rtol=2e-5, atol=2e-6)

    wg = (w1 * gamma[None, :]).to(torch.bfloat16)
    s = wg.float().sum(1)
    c = b1 + w1 @ beta
    y = vo.layernorm(x.cpu(), gamma.cpu(), beta.cpu(), eps).to(DEV)
    want = y.double() @ w1.double().t() + b1.double()
    h = torch.empty(M, N2, dtype=torch.bfloat16, device=DEV)
    ops.gemm_ln_bias_gelu(x16, wg, ab, s, c, h)
    # the unfolded kernels on the same inputs
    y16 = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    ops.layernorm(x, gamma, beta, eps, y16)
    h_plain = torch.empty(M, N2, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bias_gelu(y16, w1.to(torch.bfloat16), b1, h_plain)
    ref = F.gelu(want).float()
    e_fold, e_plain = (h.float() - ref).abs(), (h_plain.float() - ref).abs()
    print(f"M={M} D={D}: folded max {e_fold.max():.2e} mean {e_fold.mean():.2e}; unfolded max {e_plain.max():.2e} mean {e_plain.mean():.2e}")
    assert e_fold.max() <= 4e-2 and e_fold.mean() <= 1.5 * e_plain.mean() + 1e-4

    if N2 % 3 == 0 and (N2 // 3) == D and D % 64 == 0:             # the q/k/v projection of the same rows
        H, hd = D // 64, 64
        Bq = 1
        qkv = torch.full((3, Bq, H, M, hd), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.qkv_proj_ln(x16, wg, ab, s, c, qkv, Bq, M, H, hd)
        wantq = want.float().reshape(Bq, M, 3, H, hd).permute(2, 0, 3, 1, 4)
        eq = (qkv.float() - wantq).abs()
        assert eq.max() <= 4e-2 and eq.mean() <= 1.5 * e_plain.mean() + 1e-3
