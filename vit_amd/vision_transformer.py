"""VisionTransformer with the reference's Python surface, run by MI355X kernels.

Mirrors /root/reference/models/vision_transformer.py: the constructor
signature (:10-13, extra *args/**kwargs swallowed as there), the attribute
names `embedding, cls_token, pos_embed, dropout, encoder, pre_logits, tanh,
head` (hence the state_dict keys converter.py emits), the initialisation
(:34-43) and the six factories `s16, b16, b32, l16, l32, h14` (:59-97).

forward(x: [B,3,H,W]) -> [B,num_classes] runs entirely on libvitk kernels
(vit_amd.ops); there is no CPU or eager-torch fallback: a CPU tensor, a
missing library or an unsupported shape raises.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from . import ops
from .autograd import forward_train
from .engine import WeightCache, Workspace, gather_patch_ok
from .lib import EPI_TANH, VitkError
from .transformer import Encoder
from .utils import PIXEL_MEAN, PIXEL_STD


class VisionTransformer(nn.Module):
    def __init__(self, img_size=384, patches=16, num_classes=1000,
                 hidden_size=1024, num_layers=12, num_heads=12, mlp_ratio=4,
                 dropout=0., attn_bias=True, attn_scaling=None, attn_dropout=0.,
                 norm=nn.LayerNorm, pre_logits=True, *args, **kwargs):
        super().__init__()
        # The one keyword beyond the reference's: precision="bf16" (default: bf16 MFMA operands, fp32
        # accumulation and residual stream) or "fp32" (every op in f32, as the reference computes with
        # fp32 parameters).  Also settable later with set_precision().
        self.precision = "bf16"
        # Statistics applied to uint8 images (the reference's loader-side normalisation, data.py:54-55).
        self.pixel_mean, self.pixel_std = PIXEL_MEAN, PIXEL_STD
        self.set_precision(kwargs.get("precision", "bf16"))
        self.num_classes = num_classes
        self.hidden_size = hidden_size
        self.patches = patches
        self.img_size = img_size
        self.num_heads = num_heads
        self._p_dropout = max(float(dropout), float(attn_dropout))

        self.embedding = nn.Conv2d(3, hidden_size, kernel_size=patches, stride=patches)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        num_patches = (img_size // patches) ** 2
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, hidden_size))
        self.dropout = nn.Dropout(dropout)
        self.encoder = Encoder(hidden_size, num_layers, num_heads, mlp_ratio, dropout, attn_bias,
                               attn_scaling, attn_dropout, norm)
        self.pre_logits = nn.Linear(hidden_size, hidden_size) if pre_logits else nn.Identity()
        self.tanh = nn.Tanh() if pre_logits else nn.Identity()
        self.head = nn.Linear(hidden_size, num_classes)

        self.apply(self._init)
        # not part of state_dict: kernel-ready weights and activation buffers
        self._cache = WeightCache(self, fp32=self.precision == "fp32")
        self._ws = {}
        # Test hook: callable(name, tensor) that sees the inference path's intermediate buffers in place ("embed":
        # tokens after cls / pos, "block0": residual stream after the first block, "encoded": normalised cls rows),
        # the points a forward hook on the reference's embedding, blocks[0] and encoder would observe.
        self.stage_hook = None
        # Inference, bf16 mode: fold each LayerNorm that follows a residual GEMM into that GEMM and the projection
        # after it (no pass of its own over the residual stream).  False keeps one vitk_layernorm per LayerNorm.
        # The fold moves bf16 roundings (from LN(x) to x, from W to W * gamma); on the 12- and 24-layer models the
        # logit error against the reference is the same or smaller, on the 32-layer ViT-H/14, whose error sits at
        # the 1e-2 bar either way, one parity set moved over it (1.011e-2 against 9.3e-3): models deeper than 24
        # layers keep the separate launches unless the caller sets the attribute.
        self.fold_layernorm = num_layers <= 24 and os.environ.get("VIT_AMD_NO_LN_FOLD") != "1"

    def set_precision(self, precision):
        if precision not in ("bf16", "fp32"):
            raise VitkError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        if precision != self.precision or not hasattr(self, "_cache"):
            self.precision = precision
            self._cache = WeightCache(self, fp32=precision == "fp32")
            self._ws = {}
        return self

    def _init(self, module):
        # same draws as the reference (:34-43); `embedding` keeps Conv2d's default init there too
        nn.init.normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.2)
        if isinstance(module, nn.Linear):
            nn.init.trunc_normal_(module.weight, std=.02)
            if module.bias is not None:
                nn.init.constant_(module.bias, 0)
        elif isinstance(module, nn.LayerNorm):
            nn.init.constant_(module.bias, 0)
            nn.init.constant_(module.weight, 1.0)

    def _workspace(self, device, B, N):
        D, H = self.hidden_size, self.num_heads
        M = self.encoder.blocks[0].mlp.fc1.out_features if len(self.encoder.blocks) else D
        key = (str(device), B, N)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 4:      # keep at most a few shapes resident
                self._ws.clear()
            ws = self._ws[key] = Workspace(device, B, N, D, H, M, self.num_classes, fp32=self.precision == "fp32")
        return ws

    def invalidate_weight_cache(self):
        """Re-derive the kernel-ready weight copies at the next forward.  Needed only after parameters were
        changed through `.data` (`p.data.copy_()`, `p.data.mul_()`: weight-averaging helpers, some third-party
        optimizers), which the parameter's version counter does not see; `optimizer.step()`, `load_state_dict`
        and in-place ops on the parameter itself are detected automatically."""
        self._cache.invalidate()

    def forward(self, x):
        if not x.is_cuda:
            raise VitkError("VisionTransformer.forward needs a CUDA/HIP tensor: this path has no CPU fallback")
        # every kernel is launched on the current stream of the current device: make that the input's device
        with torch.cuda.device(x.device):
            return self._forward(x)

    def _forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise VitkError(f"expected images [B,3,H,W], got {tuple(x.shape)}")
        w = self._cache.get()
        if w["pos"].device != x.device:
            raise VitkError(f"model is on {w['pos'].device}, input on {x.device}")
        B, _, Hi, Wi = x.shape
        P = self.patches
        N = (Hi // P) * (Wi // P) + 1
        if N != w["pos"].shape[0]:
            raise VitkError(f"input {Hi}x{Wi} gives {N} tokens but pos_embed holds {w['pos'].shape[0]}")
        if (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())) or \
                (self.training and self._p_dropout > 0.):
            # training (main.py:207-223): the same kernels as autograd Functions, fresh tensors per op; active
            # Dropout layers (training mode, dropout or attn_dropout > 0) exist on that path only
            return forward_train(self, x)
        ws = self._workspace(x.device, B, N)
        out_dtype = self.head.weight.dtype
        if self.fold_layernorm and self.precision == "bf16" and "fold" not in w and self.encoder.fold_ok(ws):
            w = self._cache.get(fold=True)
        elif "fold" in w and not (self.fold_layernorm and self.encoder.fold_ok(ws)):
            w = {k: v for k, v in w.items() if k != "fold"}      # this shape (or the switch) takes the plain blocks

        if self.precision == "fp32":
            if x.dtype == torch.uint8:
                x = ops.normalize_u8(x.contiguous(), self.pixel_mean, self.pixel_std, ws.image_buffer(x, torch.float32))
            elif x.dtype not in (torch.float32, torch.bfloat16):
                raise VitkError(f"unsupported image dtype {x.dtype}")
            ops.patch_embed_f32(x.float().contiguous(), w["embed.w"], w["embed.b"], w["cls"], w["pos"], ws.x, P)
            if self.stage_hook is not None:
                self.stage_hook("embed", ws.x)
            feat = self.encoder.fused(ws, w, self.stage_hook)
            if "pre.w" in w:
                ops.gemm_f32(feat, w["pre.w"], w["pre.b"], ws.pre, EPI_TANH)
                feat = ws.pre
            ops.gemm_f32(feat, w["head.w"], w["head.b"], ws.logits)
            return ws.logits.to(out_dtype, copy=True)

        if x.dtype == torch.bfloat16:
            img16 = x.contiguous()
        elif x.dtype == torch.float32:
            img16 = ops.cast_bf16(x.contiguous(), ws.image_buffer(x))
        elif x.dtype == torch.uint8:
            # Raw loader output (data.py:143-154): the DataFetcher's normalisation (data.py:69-72) writes the
            # patch GEMM's bf16 operand directly, 1 byte read + 2 written per pixel instead of 4 + 4 + 4 + 2.
            img16 = ops.normalize_u8(x.contiguous(), self.pixel_mean, self.pixel_std, ws.image_buffer(x))
        else:
            raise VitkError(f"unsupported image dtype {x.dtype}")

        if gather_patch_ok(P, Wi):
            ops.patch_embed(img16, w["embed.w"], w["embed.b"], w["cls"], w["pos"], ws.x, P)
        else:
            cols = ws.patch_cols(B * (N - 1), w["embed.w"].shape[1], x.device)
            ops.patch_embed_cols(img16, w["embed.w"], w["embed.b"], w["cls"], w["pos"], ws.x, cols, P)
        if self.stage_hook is not None:
            self.stage_hook("embed", ws.x)
        feat = self.encoder.fused(ws, w, self.stage_hook)    # f32 [B, D]: from here on f32 kernels in both modes
        if "pre.w" in w:
            ops.gemm_f32(feat, w["pre.w"], w["pre.b"], ws.pre, EPI_TANH)
            feat = ws.pre
        ops.gemm_f32(feat, w["head.w"], w["head.b"], ws.logits)
        return ws.logits.to(out_dtype, copy=True)


def s16(pretrained=False, **kwargs):
    return VisionTransformer(patches=16, hidden_size=768, num_layers=8, num_heads=8, mlp_ratio=3, **kwargs)


def b16(pretrained=False, **kwargs):
    return VisionTransformer(patches=16, hidden_size=768, num_layers=12, num_heads=12,
                             norm=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def b32(pretrained=False, **kwargs):
    return VisionTransformer(patches=32, hidden_size=768, num_layers=12, num_heads=12,
                             norm=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def l16(pretrained=False, **kwargs):
    return VisionTransformer(patches=16, hidden_size=1024, num_layers=24, num_heads=16,
                             norm=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def l32(pretrained=False, **kwargs):
    return VisionTransformer(patches=32, hidden_size=1024, num_layers=24, num_heads=16,
                             norm=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def h14(pretrained=False, **kwargs):
    return VisionTransformer(patches=14, hidden_size=1280, num_layers=32, num_heads=16, **kwargs)
