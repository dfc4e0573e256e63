"""Encoder blocks with the reference's module tree, executed by libvitk kernels.

Mirrors /root/reference/models/transformer.py: class names, constructor
arguments, attribute names (and therefore state_dict keys) are identical, so a
checkpoint for the reference loads unchanged.  The parameters live in ordinary
nn.Linear / nn.LayerNorm containers; `forward` never calls those modules'
own forward: every op is a HIP kernel reached through vit_amd.ops.

Data flow of one block (models/transformer.py:66-76), x = fp32 residual stream:
    y   = LN1(x)                      -> bf16        vitk_layernorm
    qkv = y W_in^T + b, split by head -> bf16        vitk_qkv_proj
    ctx = softmax(q k^T s) v          -> bf16        vitk_attention
    x   = x + ctx W_out^T + b         -> fp32        vitk_gemm_bias_residual
    y   = LN2(x)                      -> bf16        vitk_layernorm
    h   = gelu(y W_1^T + b)           -> bf16        vitk_gemm_bias_gelu
    x   = x + h W_2^T + b             -> fp32        vitk_gemm_bias_residual
In the fp32 compute mode the same flow runs on f32 buffers through vitk_*_f32.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .lib import EPI_GELU, EPI_RESIDUAL, VitkError


def _eps_of(norm_module):
    if not hasattr(norm_module, "eps") or not hasattr(norm_module, "weight"):
        raise VitkError(f"norm layer {type(norm_module).__name__} is not LayerNorm-like (needs weight, bias, eps)")
    return float(norm_module.eps)


class MLPBlock(nn.Module):
    """Parameter container + fused path for models/transformer.py:9-26."""

    def __init__(self, in_channels, hidden_channels=None, out_channels=None, dropout=0.):
        super().__init__()
        hidden_channels = hidden_channels or in_channels
        out_channels = out_channels or in_channels
        self.fc1 = nn.Linear(in_channels, hidden_channels)
        self.gelu = nn.GELU()
        self.fc2 = nn.Linear(hidden_channels, out_channels)
        self.dropout = nn.Dropout(dropout)

    def fused(self, y16, x, h16, w, p):
        """x += fc2(gelu(fc1(y16)))  (rows = tokens); h16 is the [rows, hidden] scratch."""
        if y16.dtype == torch.float32:                       # fp32 compute mode
            ops.gemm_f32(y16, w[p + "fc1.w"], w[p + "fc1.b"], h16, EPI_GELU)
            ops.gemm_f32(h16, w[p + "fc2.w"], w[p + "fc2.b"], x, EPI_RESIDUAL, resid=x)
            return
        ops.gemm_bias_gelu(y16, w[p + "fc1.w"], w[p + "fc1.b"], h16)
        ops.gemm_bias_residual(h16, w[p + "fc2.w"], w[p + "fc2.b"], x, x)


class Attention(nn.Module):
    """Parameter container + fused path for models/transformer.py:29-48."""

    def __init__(self, dim, num_heads=16, attn_bias=True, attn_scaling=None, attn_dropout=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scaling = attn_scaling or head_dim ** -0.5
        self.in_proj = nn.Linear(dim, dim * 3, bias=attn_bias)
        self.dropout = nn.Dropout(attn_dropout)
        self.out_proj = nn.Linear(dim, dim)

    def fused(self, y16, x, qkv, ctx, w, p):
        """x += out_proj(attention(in_proj(y16))); qkv/ctx are scratch buffers."""
        _, B, H, N, hd = qkv.shape
        if y16.dtype == torch.float32:                       # fp32 compute mode
            ops.qkv_proj_f32(y16, w[p + "qkv.w"], w[p + "qkv.b"], qkv, B, N, H, hd)
            ops.attention_f32(qkv, ctx, self.scaling)
            ops.gemm_f32(ctx.view(B * N, H * hd), w[p + "out.w"], w[p + "out.b"], x, EPI_RESIDUAL, resid=x)
            return
        ops.qkv_proj(y16, w[p + "qkv.w"], w[p + "qkv.b"], qkv, B, N, H, hd)
        ops.attention(qkv, ctx, self.scaling)
        ops.gemm_bias_residual(ctx.view(B * N, H * hd), w[p + "out.w"], w[p + "out.b"], x, x)


class Encoder1DBlock(nn.Module):
    """models/transformer.py:51-76."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4, dropout=0., attn_bias=True, attn_scaling=None,
                 attn_dropout=0., norm=nn.LayerNorm, drop_prob=0.):
        super().__init__()
        if drop_prob > 0.:
            raise VitkError("stochastic depth (drop_prob > 0) is not part of the ViT forward path")
        self.norm1 = norm(hidden_size)
        self.attention = Attention(hidden_size, num_heads=num_heads, attn_bias=attn_bias,
                                   attn_scaling=attn_scaling, attn_dropout=attn_dropout)
        self.norm2 = norm(hidden_size)
        mlp_dim = int(hidden_size * mlp_ratio)
        self.mlp = MLPBlock(in_channels=hidden_size, hidden_channels=mlp_dim, dropout=dropout)
        self.drop_module = nn.Identity()
        self.dropout = nn.Dropout(dropout)

    def fused(self, ws, w, p):
        """One pre-LN block, in place on ws.x (fp32 [B, N, D])."""
        B, N, D = ws.x.shape
        x2 = ws.x.view(B * N, D)
        ops.layernorm(x2, w[p + "norm1.w"], w[p + "norm1.b"], _eps_of(self.norm1), ws.y)
        self.attention.fused(ws.y, x2, ws.qkv, ws.ctx, w, p)
        ops.layernorm(x2, w[p + "norm2.w"], w[p + "norm2.b"], _eps_of(self.norm2), ws.y)
        self.mlp.fused(ws.y, x2, ws.h, w, p)


class Encoder(nn.Module):
    """models/transformer.py:79-99."""

    def __init__(self, hidden_size=1024, num_layers=12, num_heads=12, mlp_ratio=4, dropout=0., attn_bias=True,
                 attn_scaling=None, attn_dropout=0., norm=nn.LayerNorm, drop_prob=0.):
        super().__init__()
        # linear ramp 0 .. drop_prob over the layers (pure Python so that meta-device construction works)
        drop_probs = [drop_prob * i / max(num_layers - 1, 1) for i in range(num_layers)]
        self.blocks = nn.ModuleList([
            Encoder1DBlock(hidden_size=hidden_size, num_heads=num_heads, mlp_ratio=mlp_ratio, dropout=dropout,
                           attn_bias=attn_bias, attn_scaling=attn_scaling, attn_dropout=attn_dropout,
                           norm=norm, drop_prob=drop_probs[i])
            for i in range(num_layers)])
        self.norm = norm(hidden_size)
        # which LayerNorms the folded path (fold_ok) takes out of the blocks: norm1 of blocks 1.. (statistics from
        # the previous block's fc2) and norm2 (statistics from out_proj)
        mode = os.environ.get("VIT_AMD_LN_FOLD_MODE", "both")
        self.fold_norm1, self.fold_norm2 = mode in ("both", "ln1"), mode in ("both", "ln2")
        # with both folds on: the residual stream between the blocks as two 16-bit planes (see _blocks_folded)
        self.residual_planes = os.environ.get("VIT_AMD_NO_PLANES", "0") != "1"
        # Inference, bf16 mode: evaluate the last block past its q/k/v projection for the cls token only (exact: the
        # encoder returns token 0, and those ops are per token).  Off by default: the full block is what the
        # reference's forward computes and what the throughput figures count.
        self.cls_only_last_block = False

    def fold_ok(self, ws):
        """Whether the LayerNorms between the blocks' GEMMs can be folded into them (bf16 mode, shapes of the
        two-group GEMM kernel): every norm LayerNorm-like, and the library's shape rule."""
        if ws.fp32 or not len(self.blocks):
            return False
        B, N, D = ws.x.shape
        return ops.ln_fold_ok(B * N, D, self.blocks[0].mlp.fc1.out_features)

    def _last_block_cls_rows(self, ws, w, p, blk):
        """What is left of the last block after its q/k/v projection when only token 0 of its output is kept
        (models/transformer.py:98-99): attention for the cls query, then out_proj, norm2 and the MLP on the cls row of
        every image (every op past the attention is per token).  The same kernels as the full block on strided rows:
        the cls rows of ws.x end up bit-identical to the full block's."""
        B, N, D = ws.x.shape
        M = blk.mlp.fc1.out_features
        ops.attention(ws.qkv, ws.ctx, blk.attention.scaling, rows=1)
        ops.gemm_bias_residual_rows(ws.ctx, N * D, w[p + "out.w"], w[p + "out.b"], ws.x, N * D, B)
        y_cls, h_cls = ws.y[:B], ws.h[:B]
        ops.layernorm(ws.x, w[p + "norm2.w"], w[p + "norm2.b"], _eps_of(blk.norm2), y_cls, rows=B, x_row_stride=N * D)
        ops.gemm_bias_gelu(y_cls, w[p + "fc1.w"], w[p + "fc1.b"], h_cls)
        ops.gemm_bias_residual_rows(h_cls, M, w[p + "fc2.w"], w[p + "fc2.b"], ws.x, N * D, B)

    def _blocks_folded(self, ws, w, stage_hook):
        """The blocks with every LayerNorm but the first folded into its neighbours: the residual GEMM before it
        leaves bf16(x) in ws.y and the row statistics, the projection after it runs on bf16(x) with gamma-scaled
        weights and applies rstd / mean in its epilogue (same values as y = LN(x); the bf16 rounding moves from y
        to x and from W to W * gamma)."""
        B, N, D = ws.x.shape
        x2 = ws.x.view(B * N, D)
        H = ws.qkv.shape[2]
        hd = D // H
        last = len(self.blocks) - 1
        fold1, fold2 = self.fold_norm1, self.fold_norm2
        # With both folds on, the stream lives as two 16-bit planes from block 0's out_proj to the last block's: plane H
        # (ws.y) is bf16(x), the operand of the projection that follows, plane L (ws.xl) the lower half of x's f32 bits.
        # Every value of x is what the f32 stream would hold (the planes restore it exactly); a residual GEMM writes
        # 4 bytes per element instead of f32 + the bf16 copy.
        planes = fold1 and fold2 and last >= 1 and self.residual_planes
        ctx2 = ws.ctx.view(B * N, D)
        in_planes = False                    # x is in (ws.y, ws.xl), ws.x is stale
        normed = False                       # ws.y / ws.ln_ab hold bf16(x) and the row factors of the coming norm1
        for i, blk in enumerate(self.blocks):
            p = f"b{i}."
            if normed:
                ops.qkv_proj_ln(ws.y, w[p + "qkv.wg"], ws.ln_ab, w[p + "qkv.s"], w[p + "qkv.c"], ws.qkv, B, N, H, hd)
            else:
                ops.layernorm(x2, w[p + "norm1.w"], w[p + "norm1.b"], _eps_of(blk.norm1), ws.y)
                ops.qkv_proj(ws.y, w[p + "qkv.w"], w[p + "qkv.b"], ws.qkv, B, N, H, hd)
            if i == last and self.cls_only_last_block:
                if in_planes:                # only the cls rows of x are read from here on
                    ws.x[:, 0].copy_(ops.planes_to_f32(ws.y.view(B, N, D)[:, 0], ws.xl.view(B, N, D)[:, 0]))
                self._last_block_cls_rows(ws, w, p, blk)
                if i == 0 and stage_hook is not None:
                    stage_hook("block0", ws.x)
                break
            ops.attention(ws.qkv, ws.ctx, blk.attention.scaling)
            if fold2 and i < last:       # (the last block keeps its norm2 launch: the cls-only form above shares its kernels)
                if planes:
                    ops.gemm_bias_residual_planes(ctx2, w[p + "out.w"], w[p + "out.b"], ws.y, ws.xl, ws.ln_stats,
                                                  x_in=None if in_planes else x2)
                    in_planes = True
                else:
                    ops.gemm_bias_residual_ln(ctx2, w[p + "out.w"], w[p + "out.b"], x2, x2, ws.y, ws.ln_stats)
                ops.ln_finalize(ws.ln_stats, ws.ln_ab, D, _eps_of(blk.norm2))
                ops.gemm_ln_bias_gelu(ws.y, w[p + "fc1.wg"], ws.ln_ab, w[p + "fc1.s"], w[p + "fc1.c"], ws.h)
            else:
                if in_planes:
                    ops.gemm_bias_residual_planes(ctx2, w[p + "out.w"], w[p + "out.b"], ws.y, ws.xl, x_out=x2)
                    in_planes = False
                else:
                    ops.gemm_bias_residual(ctx2, w[p + "out.w"], w[p + "out.b"], x2, x2)
                ops.layernorm(x2, w[p + "norm2.w"], w[p + "norm2.b"], _eps_of(blk.norm2), ws.y)
                ops.gemm_bias_gelu(ws.y, w[p + "fc1.w"], w[p + "fc1.b"], ws.h)
            normed = fold1 and i < last
            if normed and in_planes:
                ops.gemm_bias_residual_planes(ws.h, w[p + "fc2.w"], w[p + "fc2.b"], ws.y, ws.xl, ws.ln_stats)
                ops.ln_finalize(ws.ln_stats, ws.ln_ab, D, _eps_of(self.blocks[i + 1].norm1))
            elif normed:
                ops.gemm_bias_residual_ln(ws.h, w[p + "fc2.w"], w[p + "fc2.b"], x2, x2, ws.y, ws.ln_stats)
                ops.ln_finalize(ws.ln_stats, ws.ln_ab, D, _eps_of(self.blocks[i + 1].norm1))
            else:
                ops.gemm_bias_residual(ws.h, w[p + "fc2.w"], w[p + "fc2.b"], x2, x2)
            if i == 0 and stage_hook is not None:
                stage_hook("block0", ops.planes_to_f32(ws.y, ws.xl).view(B, N, D) if in_planes else ws.x)

    def fused(self, ws, w, stage_hook=None):
        """All blocks, then the final LayerNorm on the cls row only: LayerNorm is per token
        and the reference keeps just token 0 afterwards (models/transformer.py:98-99).
        `stage_hook(name, tensor)`, if given, sees the residual stream after block 0 ("block0") and the
        normalised cls rows ("encoded"): what a forward hook on the reference's modules would see."""
        if "fold" in w:
            self._blocks_folded(ws, w, stage_hook)
        else:
            for i, blk in enumerate(self.blocks):
                if i == len(self.blocks) - 1 and self.cls_only_last_block and not ws.fp32:
                    B, N, D = ws.x.shape
                    p, H = f"b{i}.", ws.qkv.shape[2]
                    x2 = ws.x.view(B * N, D)
                    ops.layernorm(x2, w[p + "norm1.w"], w[p + "norm1.b"], _eps_of(blk.norm1), ws.y)
                    ops.qkv_proj(ws.y, w[p + "qkv.w"], w[p + "qkv.b"], ws.qkv, B, N, H, D // H)
                    self._last_block_cls_rows(ws, w, p, blk)
                else:
                    blk.fused(ws, w, f"b{i}.")
                if i == 0 and stage_hook is not None:
                    stage_hook("block0", ws.x)
        B, N, D = ws.x.shape
        ops.layernorm(ws.x, w["norm.w"], w["norm.b"], _eps_of(self.norm), ws.cls_n, rows=B, x_row_stride=N * D)
        if stage_hook is not None:
            stage_hook("encoded", ws.cls_n)
        return ws.cls_n
