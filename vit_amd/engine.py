"""Device-side state of a VisionTransformer: kernel-ready weights and workspaces.

The nn.Module keeps the reference's fp32 parameters (so `state_dict()` and
`load_state_dict()` are unchanged); the kernels want bf16 GEMM weights and
fp32 vectors at fixed layouts.  `WeightCache` derives those once and
re-derives them when a parameter is modified or replaced.  `Workspace` owns the
activation buffers of one (batch, tokens) shape so that steady-state forwards
allocate nothing.

HBM layout of one forward (B images, N tokens, D channels, M = mlp width):
  x    f32  [B*N, D]      residual stream (kept fp32: bf16 here costs > 1e-2 logit error)
  y    bf16 [B*N, D]      LayerNorm output = GEMM operand
  qkv  bf16 [3, B, H, N, hd]  head-major q, k, v
  ctx  bf16 [B*N, D]      attention output, heads concatenated
  h    bf16 [B*N, M]      GELU(fc1) output
In the fp32 compute mode (`precision="fp32"`) every buffer above is f32.
"""
import torch


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _bf16(t):
    return t.detach().to(torch.bfloat16).contiguous()


def gather_patch_ok(patch, img_w=0):
    """What the GEMM's im2col-free gather loader addresses: patch sizes that are powers of two >= 8, in images
    whose width is a multiple of 8 (16-byte runs of a row).  Everything else goes through the patch matrix."""
    return patch >= 8 and (patch & (patch - 1)) == 0 and img_w % 8 == 0


class WeightCache:
    """Kernel-ready weights.  `fp32=True` (the fp32 compute mode) keeps the GEMM weights in f32: they are
    then the parameters themselves, not copies."""

    def __init__(self, model, fp32=False):
        self._model = model
        self._key = None
        self._fp32 = fp32
        self.t = {}

    def _signature(self):
        return tuple((p.data_ptr(), p._version, p.device) for p in self._model.parameters())

    def get(self, train=False, fold=False):
        """Kernel-ready tensors; train=True adds what the backward GEMMs need (bf16 mode): `*.wt`, each GEMM
        weight transposed ([in, out]), so that the input gradient g W is the forward kernel's A W'^T with
        W' = W^T.  fold=True adds what the projections need that apply the preceding LayerNorm in their own
        epilogue (bf16 mode, inference): see _add_folded."""
        key = self._signature()
        if key != self._key:
            self._rebuild()
            self._key = key
        if train and not self._fp32 and "train" not in self.t:
            self._add_transposed()
        if fold and not self._fp32 and "fold" not in self.t:
            self._add_folded()
        return self.t

    def _add_folded(self):
        """LN(x) W^T + b = rstd (x Wg^T - mean s) + c  with  Wg = W * gamma (per input column, rounded to bf16: the
        operand the matrix pipe sees), s[n] = sum_k Wg[n, k] (of the ROUNDED values, so that the mean's share of
        x Wg^T cancels exactly), c[n] = b[n] + sum_k beta[k] W[n, k] (f32).  For in_proj after norm1 and fc1 after
        norm2 of every block (models/transformer.py:68-69,73-74)."""
        t = self.t
        for i, blk in enumerate(self._model.encoder.blocks):
            p = f"b{i}."
            for name, lin, norm in (("qkv", blk.attention.in_proj, "norm1"), ("fc1", blk.mlp.fc1, "norm2")):
                w32 = lin.weight.detach().to(torch.float32)
                gamma, beta = t[p + norm + ".w"], t[p + norm + ".b"]
                wg = (w32 * gamma[None, :]).to(torch.bfloat16).contiguous()
                t[p + name + ".wg"] = wg
                t[p + name + ".s"] = wg.to(torch.float32).sum(dim=1).contiguous()
                b = t[p + name + ".b"]
                c = w32 @ beta
                t[p + name + ".c"] = (c if b is None else c + b).contiguous()
        t["fold"] = True

    def invalidate(self):
        """Drop the derived copies: call after changing parameters through `.data` (which `_version` does
        not see), e.g. `p.data.copy_()` in weight-averaging helpers."""
        self._key = None

    def _add_transposed(self):
        t = self.t
        tr = lambda w: w.t().contiguous()                                  # noqa: E731
        for i in range(len(self._model.encoder.blocks)):
            p = f"b{i}."
            for n in ("qkv", "out", "fc1", "fc2"):
                t[p + n + ".wt"] = tr(t[p + n + ".w"])
        t["train"] = True

    def _rebuild(self):
        m = self._model
        t = {}
        D = m.hidden_size
        _bf16 = _f32 if self._fp32 else globals()["_bf16"]       # GEMM weight dtype of this mode
        t["embed.w"] = _bf16(m.embedding.weight).reshape(D, -1)          # OIHW flattened: k = c*P*P + i*P + j
        if not self._fp32 and not gather_patch_ok(m.patches):
            # explicit patch matrix: weight columns zero-padded to a multiple of 64
            k0 = t["embed.w"].shape[1]
            kpad = (k0 + 63) // 64 * 64
            wp = torch.zeros(D, kpad, dtype=torch.bfloat16, device=t["embed.w"].device)
            wp[:, :k0] = t["embed.w"]
            t["embed.w"] = wp
        t["embed.b"] = _f32(m.embedding.bias)
        t["cls"] = _f32(m.cls_token).reshape(D)
        t["pos"] = _f32(m.pos_embed).reshape(-1, D)
        for i, blk in enumerate(m.encoder.blocks):
            p = f"b{i}."
            for ln in ("norm1", "norm2"):
                mod = getattr(blk, ln)
                t[p + ln + ".w"], t[p + ln + ".b"] = _f32(mod.weight), _f32(mod.bias)
            att = blk.attention
            t[p + "qkv.w"] = _bf16(att.in_proj.weight)
            t[p + "qkv.b"] = None if att.in_proj.bias is None else _f32(att.in_proj.bias)
            t[p + "out.w"], t[p + "out.b"] = _bf16(att.out_proj.weight), _f32(att.out_proj.bias)
            t[p + "fc1.w"], t[p + "fc1.b"] = _bf16(blk.mlp.fc1.weight), _f32(blk.mlp.fc1.bias)
            t[p + "fc2.w"], t[p + "fc2.b"] = _bf16(blk.mlp.fc2.weight), _f32(blk.mlp.fc2.bias)
        t["norm.w"], t["norm.b"] = _f32(m.encoder.norm.weight), _f32(m.encoder.norm.bias)
        # The head (final LayerNorm output, pre_logits + tanh, classifier: 2.7 MFLOP per image) runs in f32 in
        # both modes: its bf16 roundings were ~13 % of the bf16 mode's logit error and it costs nothing.
        if isinstance(m.pre_logits, torch.nn.Linear):
            t["pre.w"], t["pre.b"] = _f32(m.pre_logits.weight), _f32(m.pre_logits.bias)
        t["head.w"], t["head.b"] = _f32(m.head.weight), _f32(m.head.bias)
        self.t = t


class Workspace:
    """Activation buffers for one (B, N, D, H, M, classes) shape on one device."""

    def __init__(self, device, B, N, D, H, M, ncls, fp32=False):
        bf, f32 = (torch.float32 if fp32 else torch.bfloat16), torch.float32      # bf = GEMM operand dtype
        self.fp32 = fp32
        hd = D // H
        e = lambda *s, dtype: torch.empty(*s, dtype=dtype, device=device)   # noqa: E731
        self.shape = (B, N, D, H, M, ncls)
        self.x = e(B, N, D, dtype=f32)
        self.y = e(B * N, D, dtype=bf)
        self.qkv = e(3, B, H, N, hd, dtype=bf)
        self.ctx = e(B, N, D, dtype=bf)
        self.h = e(B * N, M, dtype=bf)
        self.cls_n = e(B, D, dtype=f32)          # the head runs in f32 in both modes
        self.pre = e(B, D, dtype=f32)
        self.logits = e(B, ncls, dtype=f32)
        # LayerNorm folded into the GEMMs either side of it: strip statistics and per-row factors
        self.ln_stats = None if fp32 else e(B * N, max(D // 64, 1), 2, dtype=f32)
        self.ln_ab = None if fp32 else e(B * N, 2, dtype=f32)
        self.xl = None if fp32 else e(B * N, D, dtype=bf)    # plane L of the residual stream while it lives as two 16-bit planes (plane H is y)
        self.img16 = None
        self.cols = None

    def patch_cols(self, rows, kpad, device):
        if self.cols is None or tuple(self.cols.shape) != (rows, kpad):
            self.cols = torch.empty(rows, kpad, dtype=torch.bfloat16, device=device)
        return self.cols

    def image_buffer(self, like, dtype=torch.bfloat16):
        if self.img16 is None or self.img16.shape != like.shape or self.img16.dtype != dtype:
            self.img16 = torch.empty(like.shape, dtype=dtype, device=like.device)
        return self.img16
