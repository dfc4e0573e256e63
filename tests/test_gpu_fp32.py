"""fp32 compute mode on the GPU: every vitk_*_f32 entry point against the oracle's op in fp64/fp32, and
whole-model logits against the reference's fp32 CPU logits (north-star bar for fp32: max-abs <= 1e-4,
identical top-1)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_model, golden_inputs, load_golden
from vit_amd import lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_FP32 = 1e-4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("M,N,K", [(1576, 768, 768), (197, 2304, 768), (130, 1000, 768), (8, 10, 64), (300, 257, 588),
                                   (129, 129, 17)])
@pytest.mark.parametrize("epi", ["bias", "gelu", "resid", "tanh"])
def test_gemm_f32(M, N, K, epi):
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3, scale=0.1)
    resid = rnd(M, N, seed=4)
    lin = (a.double() @ w.double().t() + bias.double())
    want = {"bias": lin, "gelu": F.gelu(lin), "resid": resid.double() + lin, "tanh": torch.tanh(lin)}[epi]
    out = resid.to(DEV).clone() if epi == "resid" else torch.full((M, N), float("nan"), device=DEV)
    code = {"bias": lib.EPI_BIAS, "gelu": lib.EPI_GELU, "resid": lib.EPI_RESIDUAL, "tanh": lib.EPI_TANH}[epi]
    ops.gemm_f32(a.to(DEV), w.to(DEV), bias.to(DEV), out, code, resid=out if epi == "resid" else None)
    err = (out.cpu().double() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err


def test_gemm_f32_no_bias():
    a, w = rnd(64, 96, seed=5), rnd(40, 96, seed=6)
    out = torch.empty(64, 40, device=DEV)
    ops.gemm_f32(a.to(DEV), w.to(DEV), None, out)
    assert torch.allclose(out.cpu(), a @ w.t(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,N,H,hd", [(3, 197, 12, 64), (2, 197, 8, 96), (1, 257, 16, 80)])
def test_qkv_proj_f32_head_split(B, N, H, hd):
    D = H * hd
    a, w, bias = rnd(B * N, D, seed=7), rnd(3 * D, D, seed=8, scale=0.03), rnd(3 * D, seed=9, scale=0.1)
    want = (a.double() @ w.double().t() + bias.double()).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    qkv = torch.full((3, B, H, N, hd), float("nan"), device=DEV)
    ops.qkv_proj_f32(a.to(DEV), w.to(DEV), bias.to(DEV), qkv, B, N, H, hd)
    assert (qkv.cpu().double() - want).abs().max().item() <= 3e-5


@pytest.mark.parametrize("B,S,P,D", [(3, 224, 16, 768), (2, 224, 14, 1280), (2, 64, 32, 256), (1, 230, 16, 768)])
def test_patch_embed_f32(B, S, P, D):
    img = rnd(B, 3, S, S, seed=10)
    w, bias = rnd(D, 3, P, P, seed=11, scale=0.03), rnd(D, seed=12, scale=0.1)
    n = (S // P) ** 2
    cls, pos = rnd(D, seed=13), rnd(n + 1, D, seed=14, scale=0.1)
    conv = F.conv2d(img.double(), w.double(), bias.double(), stride=P).flatten(2).transpose(1, 2)
    want = torch.cat([cls.double().expand(B, 1, D), conv], dim=1) + pos.double()
    out = torch.full((B, n + 1, D), float("nan"), device=DEV)
    ops.patch_embed_f32(img.to(DEV), w.reshape(D, -1).contiguous().to(DEV), bias.to(DEV), cls.to(DEV), pos.to(DEV),
                        out, P)
    assert (out.cpu().double() - want).abs().max().item() <= 3e-5


@pytest.mark.parametrize("B,H,N,hd", [(2, 12, 197, 64), (1, 12, 577, 64), (2, 8, 197, 96), (1, 3, 257, 80), (3, 2, 1, 64),
                                      (1, 2, 33, 128)])
def test_attention_f32(B, H, N, hd):
    qkv = rnd(3, B, H, N, hd, seed=N + hd)
    q, k, v = qkv[0].double(), qkv[1].double(), qkv[2].double()
    p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
    want = (p @ v).transpose(1, 2).reshape(B, N, H * hd)
    ctx = torch.full((B, N, H * hd), float("nan"), device=DEV)
    ops.attention_f32(qkv.to(DEV), ctx, hd ** -0.5)
    assert (ctx.cpu().double() - want).abs().max().item() <= 2e-5


@pytest.mark.parametrize("name", ["s16_224_ref", "b16_224_ref", "b16_384_ref"])
def test_fp32_mode_logits_match_reference(name):
    """BASELINE config 1 (ViT-S/16 224 fp32, batch 8) and two B/16 sets in the fp32 compute mode."""
    meta, g = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    m = build_model(meta["arch"], meta["img_size"], sd, DEV).set_precision("fp32")
    with torch.no_grad():
        y = m(x.to(DEV)).float().cpu().numpy()
    err = np.abs(y - g["logits"]).max()
    print(f"{name} fp32 mode: max-abs logit error {err:.3e}")
    assert err <= TOL_FP32
    assert (y.argmax(-1) == g["top1"]).all()
    # the same module object goes back to the bf16 path
    m.set_precision("bf16")
    with torch.no_grad():
        y16 = m(x.to(DEV)).float().cpu().numpy()
    assert 1e-5 < np.abs(y16 - g["logits"]).max() <= 1e-2
