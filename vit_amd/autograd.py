"""Training path: the forward kernels as torch.autograd.Functions.

The reference trains the same module it evaluates (main.py:207-223: `loss = criterion(model(images),
target); loss.backward(); clip_grad_norm_; optimizer.step()`).  Here every forward op is still a libvitk
kernel; each Function keeps what its backward needs and computes the gradients with torch operations on
the GPU (library GEMMs, aten backward ops), recomputing cheap intermediates instead of storing them.
Gradients arrive at the module's ordinary fp32 Parameters, so optimizers, `clip_grad_norm_` and
DistributedDataParallel work unchanged.  Hand-written backward kernels so far: LayerNorm, the weight-gradient
GEMM (dW = dY^T X) and attention.

cd = compute dtype of the GEMM operands: torch.bfloat16, or torch.float32 in the fp32 mode.
"""
import torch
import torch.nn.functional as F

from . import ops
from .lib import EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_TANH, VitkError


def _vjp(fn, inputs, grad_out):
    """Gradients of fn(*inputs) w.r.t. inputs (detached copies), by recomputing fn with torch."""
    with torch.enable_grad():
        ins = [t.detach().requires_grad_(True) for t in inputs]
        out = fn(*ins)
        return torch.autograd.grad(out, ins, grad_out)


_handoff = [None]      # (f32 gradient, its bf16 copy) of the most recent LayerNorm backward


def _as_operand(g, cd):
    """g in the GEMM operand dtype: the copy the LayerNorm backward kernel already made, if g is that gradient."""
    h = _handoff[0]
    _handoff[0] = None
    if h is not None and cd == torch.bfloat16 and h[0].data_ptr() == g.data_ptr() and h[0].shape == g.shape \
            and g.dtype == torch.float32 and g.is_contiguous():
        return h[1]
    return g.to(cd)


class LayerNormFn(torch.autograd.Function):
    """(y, x) with y = LayerNorm(x) in cd; x f32 [rows, D] (models/transformer.py:68,73,98).

    The second output is x again, for the residual branch that skips the block's body
    (models/transformer.py:72,76): whatever gradient reaches it is added to the LayerNorm's own input gradient
    inside the backward kernel, instead of by autograd's separate add over both tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, cd):
        out = torch.empty(x.shape, dtype=cd, device=x.device)
        ops.layernorm(x, weight.detach(), bias.detach(), eps, out)
        ctx.save_for_backward(x, weight, bias)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_skip):
        x, weight, bias = ctx.saved_tensors
        if g is None:                                                     # only the skip output was used
            return g_skip, None, None, None, None
        add = None if g_skip is None else g_skip.contiguous()
        gx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if g.dtype == torch.bfloat16 else None
        gx, gw, gb = ops.layernorm_backward(x, weight.detach(), g.contiguous(), ctx.eps, add=add, dx_bf16=gx16)
        # The Linear layer that produced x needs this gradient as a bf16 GEMM operand next: the kernel wrote that
        # copy in the same pass.  Handed over out of band (autograd carries one tensor per edge); holding gx keeps
        # its storage from being reused, so the pointer comparison at the consumer cannot match another tensor.
        _handoff[0] = (gx, gx16) if gx16 is not None else None
        return gx, gw.to(weight.dtype), gb.to(bias.dtype), None, None


def _weight_grad(g, a):
    """f32 [out, in] = g^T a for g [rows, out], a [rows, in]: the HIP split-k kernel on the rows that fill whole
    64-row K tiles (bf16, out and in multiples of 256), a small library product for the remainder."""
    g, a = g.contiguous(), a.contiguous()
    rows = g.shape[0]
    r64 = rows - rows % 64
    if g.dtype == torch.bfloat16 and r64 > 0 and ops.gemm_tn_supported(g.shape[1], a.shape[1], r64):
        gw = ops.gemm_tn(g[:r64], a[:r64])
        if r64 < rows:
            gw += (g[r64:].t() @ a[r64:]).float()
        return gw
    return (g.t() @ a).float()


def _linear_backward(g_pre, a, w_cd, weight, bias):
    """g_pre [rows, out] in cd -> (grad a [rows, in] in cd, grad weight f32, grad bias f32)."""
    ga = g_pre @ w_cd
    gw = _weight_grad(g_pre, a).to(weight.dtype)
    if bias is None:
        gb = None
    elif g_pre.dtype == torch.bfloat16 and g_pre.shape[1] % 256 == 0:
        gb = ops.colsum(g_pre.contiguous()).to(bias.dtype)
    else:
        gb = g_pre.sum(0, dtype=torch.float32).to(bias.dtype)
    return ga, gw, gb


class LinearFn(torch.autograd.Function):
    """out = epilogue(a W^T + b).  kind: 'bias' (cd out), 'gelu' (cd), 'tanh' (cd), 'logits' (f32 out),
    'resid' (f32 out = resid + a W^T + b).  w_cd is the kernel-ready copy of `weight` in cd."""

    @staticmethod
    def forward(ctx, a, weight, bias, w_cd, kind, resid):
        rows, n = a.shape[0], weight.shape[0]
        cd = a.dtype
        pre = None
        b32 = None if bias is None else bias.detach()
        fp32 = cd == torch.float32
        out_dtype = torch.float32 if kind in ("resid", "logits") else cd
        out = torch.empty(rows, n, dtype=out_dtype, device=a.device)
        if fp32:
            epi = {"bias": EPI_BIAS, "logits": EPI_BIAS, "gelu": EPI_GELU, "tanh": EPI_TANH, "resid": EPI_RESIDUAL}[kind]
            ops.gemm_f32(a, w_cd, b32, out, epi, resid=resid if kind == "resid" else None)
        elif kind == "gelu" and ops.gemm_bias_gelu_dual_supported(rows, n, a.shape[1]):
            pre = torch.empty_like(out)                       # kept for the backward instead of being recomputed
            ops.gemm_bias_gelu_dual(a, w_cd, b32, pre, out)
        elif kind == "gelu":
            ops.gemm_bias_gelu(a, w_cd, b32, out)
        elif kind == "tanh":
            ops.gemm_bias_tanh(a, w_cd, b32, out)
        elif kind == "resid":
            ops.gemm_bias_residual(a, w_cd, b32, resid, out)
        else:
            ops.gemm_bias(a, w_cd, b32, out)
        ctx.kind = kind
        ctx.save_for_backward(a, weight, bias, w_cd, out if kind == "tanh" else pre)
        return out

    @staticmethod
    def backward(ctx, g):
        a, weight, bias, w_cd, out = ctx.saved_tensors
        cd = a.dtype
        g_resid = g if ctx.kind == "resid" else None
        if ctx.kind == "gelu":
            pre = out                                             # the forward's second output, when it had one
            if pre is None and cd == torch.bfloat16:              # else recomputed with one more GEMM
                pre = torch.empty(a.shape[0], weight.shape[0], dtype=cd, device=a.device)
                ops.gemm_bias(a, w_cd, None if bias is None else bias.detach(), pre)
            elif pre is None:
                pre = F.linear(a, w_cd, bias)
            g_pre = torch.ops.aten.gelu_backward(g, pre)                      # f32 math inside, cd in and out
        elif ctx.kind == "tanh":
            o = out.float()
            g_pre = (g.float() * (1.0 - o * o)).to(cd)
        else:
            g_pre = _as_operand(g, cd)
        ga, gw, gb = _linear_backward(g_pre, a, w_cd, weight, bias)
        return ga, gw, gb, None, None, g_resid


class QkvFn(torch.autograd.Function):
    """qkv [3, B, H, N, hd] = head-major split of a W_in^T + b (models/transformer.py:42)."""

    @staticmethod
    def forward(ctx, a, weight, bias, w_cd, B, N, H, hd):
        qkv = torch.empty(3, B, H, N, hd, dtype=a.dtype, device=a.device)
        b32 = None if bias is None else bias.detach()
        if a.dtype == torch.float32:
            ops.qkv_proj_f32(a, w_cd, b32, qkv, B, N, H, hd)
        else:
            ops.qkv_proj(a, w_cd, b32, qkv, B, N, H, hd)
        ctx.save_for_backward(a, weight, bias, w_cd)
        return qkv

    @staticmethod
    def backward(ctx, g):
        a, weight, bias, w_cd = ctx.saved_tensors
        _, B, H, N, hd = g.shape
        g_pre = g.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * H * hd)        # back to [rows, which*D + h*hd + e]
        ga, gw, gb = _linear_backward(g_pre, a, w_cd, weight, bias)
        return ga, gw, gb, None, None, None, None, None


def _attention_torch(q, k, v, scale):
    """[B, H, N, hd] each -> [B, N, H*hd]."""
    if q.dtype == torch.float32:
        s = (q @ k.transpose(-2, -1)) * scale
        o = s.softmax(-1) @ v
    else:
        o = F.scaled_dot_product_attention(q, k, v, scale=scale)           # torch's fused kernel where it has one
    B, H, N, hd = o.shape
    return o.transpose(1, 2).reshape(B, N, H * hd)


class AttentionFn(torch.autograd.Function):
    """ctx [B, N, H*hd] = softmax(q k^T scale) v, heads concatenated (models/transformer.py:43-46).

    bf16, head_dim 64, N <= 576: the forward also keeps each row's log-sum-exp and the backward is the HIP
    kernel pair of attention_bwd.hip.  Otherwise the gradients come from recomputing with torch."""

    @staticmethod
    def forward(ctx, qkv, scale):
        _, B, H, N, hd = qkv.shape
        out = torch.empty(B, N, H * hd, dtype=qkv.dtype, device=qkv.device)
        lse = None
        if qkv.dtype == torch.float32:
            ops.attention_f32(qkv, out, scale)
        elif ops.attention_backward_supported(N, hd):
            lse = ops.attention_lse(qkv, out, scale)
        else:
            ops.attention(qkv, out, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, lse = ctx.saved_tensors
        _, B, H, N, hd = qkv.shape
        # gradients laid out [B, N, 3, H, hd], i.e. already the [rows, 3 D] matrix the projection's backward
        # GEMMs want, handed on as a [3, B, H, N, hd] view of it
        if lse is not None:
            return ops.attention_backward(qkv, out, g.contiguous(), lse, ctx.scale).permute(2, 0, 3, 1, 4), None
        grads = _vjp(lambda q, k, v: _attention_torch(q, k, v, ctx.scale), (qkv[0], qkv[1], qkv[2]), g)
        dqkv = torch.empty(B, N, 3, H, hd, dtype=qkv.dtype, device=qkv.device).permute(2, 0, 3, 1, 4)
        for i in range(3):
            dqkv[i].copy_(grads[i])
        return dqkv, None


class PatchEmbedFn(torch.autograd.Function):
    """x f32 [B, N, D] = cat(cls, conv(img) + bias) + pos (models/vision_transformer.py:47-50)."""

    @staticmethod
    def forward(ctx, img, weight, bias, cls, pos, w_cd, b32, cls32, pos32, patch, cols):
        B = img.shape[0]
        N, D = pos32.shape
        x = torch.empty(B, N, D, dtype=torch.float32, device=img.device)
        if img.dtype == torch.float32:
            ops.patch_embed_f32(img, w_cd, b32, cls32, pos32, x, patch)
        elif cols is None:
            ops.patch_embed(img, w_cd, b32, cls32, pos32, x, patch)
        else:
            ops.patch_embed_cols(img, w_cd, b32, cls32, pos32, x, cols, patch)
        ctx.save_for_backward(img, weight, bias, cls, pos)
        ctx.patch = patch
        return x

    @staticmethod
    def backward(ctx, g):
        img, weight, bias, cls, pos = ctx.saved_tensors
        P = ctx.patch
        B, N, D = g.shape
        gh, gw_ = img.shape[2] // P, img.shape[3] // P
        # the convolution is a GEMM over the patch matrix (rows = patches, columns = (c, i, j) as the weight is
        # flattened), so its weight gradient is the same dY^T X product as every Linear layer's
        cols = img[:, :, :gh * P, :gw_ * P].reshape(B, 3, gh, P, gw_, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw_, 3 * P * P)
        gp = g[:, 1:, :]
        gp_cd = gp.reshape(B * (N - 1), D).to(img.dtype)
        gw = _weight_grad(gp_cd, cols)
        gb = gp.sum((0, 1))
        gc = g[:, 0, :].sum(0).view(1, 1, D)
        gpos = g.sum(0, keepdim=True)
        return None, gw.view(weight.shape).to(weight.dtype), gb.to(bias.dtype), gc.to(cls.dtype), gpos.to(pos.dtype), \
            None, None, None, None, None, None


def forward_train(model, x):
    """The forward of vit_amd.VisionTransformer with autograd: same kernels, fresh tensors per op."""
    _handoff[0] = None
    w = model._cache.get()
    fp32 = model.precision == "fp32"
    cd = torch.float32 if fp32 else torch.bfloat16
    B, _, Hi, Wi = x.shape
    P = model.patches
    N = (Hi // P) * (Wi // P) + 1
    D, H = model.hidden_size, model.num_heads
    hd = D // H
    if x.dtype == torch.uint8:
        img = ops.normalize_u8(x.contiguous(), model.pixel_mean, model.pixel_std,
                               torch.empty(x.shape, dtype=cd, device=x.device))
    elif x.dtype in (torch.float32, torch.bfloat16):
        img = x.to(cd).contiguous()
    else:
        raise VitkError(f"unsupported image dtype {x.dtype}")
    cols = None
    if not fp32 and w["embed.w"].shape[1] != 3 * P * P:                    # explicit patch matrix (non-power-of-two patch)
        cols = torch.empty(B * (N - 1), w["embed.w"].shape[1], dtype=torch.bfloat16, device=x.device)
    xs = PatchEmbedFn.apply(img, model.embedding.weight, model.embedding.bias, model.cls_token, model.pos_embed,
                            w["embed.w"], w["embed.b"], w["cls"], w["pos"], P, cols)
    x2 = xs.view(B * N, D)
    for i, blk in enumerate(model.encoder.blocks):
        p = f"b{i}."
        att, mlp = blk.attention, blk.mlp
        y, x2 = LayerNormFn.apply(x2, blk.norm1.weight, blk.norm1.bias, float(blk.norm1.eps), cd)
        qkv = QkvFn.apply(y, att.in_proj.weight, att.in_proj.bias, w[p + "qkv.w"], B, N, H, hd)
        ctxt = AttentionFn.apply(qkv, att.scaling)
        x2 = LinearFn.apply(ctxt.view(B * N, D), att.out_proj.weight, att.out_proj.bias, w[p + "out.w"], "resid", x2)
        y, x2 = LayerNormFn.apply(x2, blk.norm2.weight, blk.norm2.bias, float(blk.norm2.eps), cd)
        h = LinearFn.apply(y, mlp.fc1.weight, mlp.fc1.bias, w[p + "fc1.w"], "gelu", None)
        x2 = LinearFn.apply(h, mlp.fc2.weight, mlp.fc2.bias, w[p + "fc2.w"], "resid", x2)
    cls_rows = x2.view(B, N, D)[:, 0].contiguous()                         # the reference keeps token 0 (transformer.py:99)
    norm = model.encoder.norm
    feat, _ = LayerNormFn.apply(cls_rows, norm.weight, norm.bias, float(norm.eps), cd)
    if "pre.w" in w:
        feat = LinearFn.apply(feat, model.pre_logits.weight, model.pre_logits.bias, w["pre.w"], "tanh", None)
    logits = LinearFn.apply(feat, model.head.weight, model.head.bias, w["head.w"], "logits", None)
    return logits.to(model.head.weight.dtype)
