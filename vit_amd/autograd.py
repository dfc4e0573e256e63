"""Training path: the forward kernels as torch.autograd.Functions.

The reference trains the same module it evaluates (main.py:207-223: `loss = criterion(model(images),
target); loss.backward(); clip_grad_norm_; optimizer.step()`).  Here every forward op is still a libvitk
kernel; each Function keeps what its backward needs, recomputing cheap intermediates instead of storing them.
Gradients arrive at the module's ordinary fp32 Parameters, so optimizers, `clip_grad_norm_` and
DistributedDataParallel work unchanged.  In the bf16 mode every gradient of the blocks comes from a libvitk
kernel: LayerNorm backward, the weight-gradient GEMM (dW = dY^T X, which also sums dY's columns for the bias),
attention backward, and the input gradients dX = dY W through the forward GEMM kernel on the transposed weight,
with GELU' / tanh' in its epilogue.  torch operations remain for the fp32 mode's backward, for shapes the kernels
do not take (widths that are not multiples of 256, head sizes other than 64), for casts, slices and small
reductions, and for the masks (and, with attn_dropout > 0, the attention) of the training path with dropout.

cd = compute dtype of the GEMM operands: torch.bfloat16, or torch.float32 in the fp32 mode.
"""
import torch
import torch.nn.functional as F

from . import ops
from .lib import ACT_GELU, ACT_TANH, EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_TANH, VitkError


def _vjp(fn, inputs, grad_out):
    """Gradients of fn(*inputs) w.r.t. inputs (detached copies), by recomputing fn with torch."""
    with torch.enable_grad():
        ins = [t.detach().requires_grad_(True) for t in inputs]
        out = fn(*ins)
        return torch.autograd.grad(out, ins, grad_out)


def _operand(g, cd):
    return g if g.dtype == cd else g.to(cd)


def _weight_grad(g, a, bias=None):
    """(f32 [out, in] = g^T a, bias gradient or None) for g [rows, out], a [rows, in]: the HIP split-k kernel (bf16,
    out and in multiples of 256; any row count: the kernel masks the rows past its last whole 64-row tile), which
    sums g's columns in the same pass when the layer has a bias."""
    g, a = g.contiguous(), a.contiguous()
    rows = g.shape[0]
    if g.dtype == torch.bfloat16 and ops.gemm_tn_supported(g.shape[1], a.shape[1], rows):
        gb = None if bias is None else torch.empty(g.shape[1], dtype=torch.float32, device=g.device)
        gw = ops.gemm_tn(g, a, colsum=gb)
        return gw, (None if gb is None else gb.to(bias.dtype))
    return (g.t() @ a).float(), _bias_grad(g, bias)


def _bias_grad(g, bias):
    if bias is None:
        return None
    if g.dtype == torch.bfloat16 and g.shape[1] % 256 == 0:
        return ops.colsum(g.contiguous()).to(bias.dtype)
    return g.sum(0, dtype=torch.float32).to(bias.dtype)


def _input_grad(g, w_cd, wt, aux=None, act=None):
    """Gradient of a Linear layer's input, g W, for g [rows, out] in cd; with `aux` also through the activation
    that produced that input: (g W) * f'(aux).  bf16 with the transposed weight at hand: the forward GEMM kernel
    on (g, W^T), the activation's derivative in its epilogue; else torch."""
    rows = g.shape[0]
    if g.dtype == torch.bfloat16 and wt is not None and g.shape[1] == wt.shape[1] and g.shape[1] % 32 == 0:
        out = torch.empty(rows, wt.shape[0], dtype=torch.bfloat16, device=g.device)
        if aux is None:
            return ops.gemm_bias(g.contiguous(), wt, None, out)
        return ops.gemm_grad_act(g.contiguous(), wt, aux, out, act)
    ga = g[:, :w_cd.shape[0]] @ w_cd
    if aux is None:
        return ga
    if act == ACT_GELU:
        return torch.ops.aten.gelu_backward(ga, aux)
    o = aux.float()
    return (ga.float() * (1.0 - o * o)).to(g.dtype)


def _linear_backward(g_pre, a, w_cd, wt, weight, bias, aux=None, act=None):
    """g_pre [rows, out] in cd -> (grad of a [rows, in] in cd (through `act` if given), grad weight, grad bias)."""
    ga = _input_grad(g_pre, w_cd, wt, aux, act)
    gw, gb = _weight_grad(g_pre, a, bias)
    return ga, gw.to(weight.dtype), gb


def _ln_backward(x, weight, bias, eps, g, g_skip, want16):
    """LayerNorm backward kernel: (dx f32, dx as bf16 or None, dgamma, dbeta); g_skip (the gradient that reaches
    x along the residual branch) is added inside the kernel."""
    add = None if g_skip is None else g_skip.contiguous()
    gx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want16 else None
    gx, gw, gb = ops.layernorm_backward(x, weight.detach(), g.contiguous(), eps, add=add, dx_bf16=gx16)
    return gx, gx16, gw.to(weight.dtype), gb.to(bias.dtype)


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
        gx, _, gw, gb = _ln_backward(x, weight, bias, ctx.eps, g, g_skip, False)
        return gx, gw, gb, None, None


class ResidualBranchFn(torch.autograd.Function):
    """One residual branch's tail and the LayerNorm that follows it (models/transformer.py:69-76):

        attention branch:  x' = x + ctx W_out^T + b                      (a = ctx, w1 = None)
        MLP branch:        x' = x + gelu(a W_1^T + b_1) W_2^T + b_2      (a = LN2 output)
        then               y  = LayerNorm_next(x')                       (ln_w = None: no LayerNorm, the last block)

    Returns (y, x') (or x' alone).  One Function so that the backward is one chain with explicit tensors: the
    LayerNorm backward kernel writes the gradient of x' as f32 (it continues along the residual stream) and as
    bf16 (the operand of this branch's gradient GEMMs), the input gradient of W_2 goes through GELU' in the GEMM's
    epilogue, and nothing travels outside the graph."""

    @staticmethod
    def forward(ctx, a, x, w2, b2, w2_cd, w2_t, w1, b1, w1_cd, w1_t, ln_w, ln_b, eps):
        rows = a.shape[0]
        cd = a.dtype
        fp32 = cd == torch.float32
        pre = h = None
        if w1 is not None:
            b1_32 = None if b1 is None else b1.detach()
            m = w1.shape[0]
            h = torch.empty(rows, m, dtype=cd, device=a.device)
            if fp32:
                ops.gemm_f32(a, w1_cd, b1_32, h, EPI_GELU)
            elif ops.gemm_bias_gelu_dual_supported(rows, m, a.shape[1]):
                pre = torch.empty_like(h)                       # kept for GELU' instead of being recomputed
                ops.gemm_bias_gelu_dual(a, w1_cd, b1_32, pre, h)
            else:
                ops.gemm_bias_gelu(a, w1_cd, b1_32, h)
        last_in = a if w1 is None else h
        b2_32 = None if b2 is None else b2.detach()
        xn = torch.empty(rows, w2.shape[0], dtype=torch.float32, device=a.device)
        if fp32:
            ops.gemm_f32(last_in, w2_cd, b2_32, xn, EPI_RESIDUAL, resid=x)
        else:
            ops.gemm_bias_residual(last_in, w2_cd, b2_32, x, xn)
        ctx.has_mlp, ctx.has_ln, ctx.eps = w1 is not None, ln_w is not None, eps
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a, w2, b2, w2_cd, w2_t, w1, b1, w1_cd, w1_t, ln_w, ln_b, pre, h, xn if ln_w is not None else None)
        if ln_w is None:
            return xn
        y = torch.empty(rows, xn.shape[1], dtype=cd, device=a.device)
        ops.layernorm(xn, ln_w.detach(), ln_b.detach(), eps, y)
        return y, xn

    @staticmethod
    def backward(ctx, *grads):
        a, w2, b2, w2_cd, w2_t, w1, b1, w1_cd, w1_t, ln_w, ln_b, pre, h, xn = ctx.saved_tensors
        cd = a.dtype
        bf = cd == torch.bfloat16
        g_lnw = g_lnb = None
        if ctx.has_ln:
            g_y, g_skip = grads
            if g_y is None:
                gx = g_skip
                g16 = _operand(gx, cd)
            else:
                gx, g16, g_lnw, g_lnb = _ln_backward(xn, ln_w, ln_b, ctx.eps, g_y, g_skip, bf)
                if g16 is None:
                    g16 = gx
        else:
            gx = grads[0]
            g16 = _operand(gx, cd)
        if not ctx.has_mlp:
            ga, gw2, gb2 = _linear_backward(g16, a, w2_cd, w2_t, w2, b2)
            return ga, gx, gw2, gb2, None, None, None, None, None, None, g_lnw, g_lnb, None
        if pre is None:                                           # not kept by the forward: one more GEMM
            b1_32 = None if b1 is None else b1.detach()
            if bf:
                pre = torch.empty_like(h)
                ops.gemm_bias(a, w1_cd, b1_32, pre)
            else:
                pre = F.linear(a, w1_cd, b1_32)
        g_pre, gw2, gb2 = _linear_backward(g16, h, w2_cd, w2_t, w2, b2, aux=pre, act=ACT_GELU)
        ga, gw1, gb1 = _linear_backward(g_pre, a, w1_cd, w1_t, w1, b1)
        return ga, gx, gw2, gb2, None, None, gw1, gb1, None, None, g_lnw, g_lnb, None


def _mm_f32(a, w):
    """a [M, K] @ w[N, K]^T in f32 through the library's f32 GEMM (exact f32 multiply-add on the matrix pipe)."""
    out = torch.empty(a.shape[0], w.shape[0], dtype=torch.float32, device=a.device)
    return ops.gemm_f32(a.contiguous(), w.contiguous(), None, out, EPI_BIAS)


class HeadFn(torch.autograd.Function):
    """logits f32 = tanh(feat W_pre^T + b_pre) W_head^T + b_head (models/vision_transformer.py:53-55); without
    pre_logits (w_pre = None) the head alone.  f32 in both compute modes (one row per image: 2.7 MFLOP), forward
    and backward through vitk_gemm_f32."""

    @staticmethod
    def forward(ctx, feat, w_pre, b_pre, w_head, b_head):
        rows = feat.shape[0]
        t = None
        if w_pre is not None:
            t = torch.empty(rows, w_pre.shape[0], dtype=torch.float32, device=feat.device)
            ops.gemm_f32(feat, w_pre.detach().float().contiguous(), b_pre.detach().float(), t, EPI_TANH)
        last_in = feat if t is None else t
        logits = torch.empty(rows, w_head.shape[0], dtype=torch.float32, device=feat.device)
        ops.gemm_f32(last_in, w_head.detach().float().contiguous(), b_head.detach().float(), logits, EPI_BIAS)
        ctx.save_for_backward(feat, w_pre, b_pre, w_head, b_head, t)
        return logits

    @staticmethod
    def backward(ctx, g):
        feat, w_pre, b_pre, w_head, b_head, t = ctx.saved_tensors
        g = g.float().contiguous()
        last_in = feat if t is None else t
        gb_head = g.sum(0).to(b_head.dtype)
        gw_head = _mm_f32(g.t(), last_in.t()).to(w_head.dtype)            # g^T last_in: [classes, D]
        ga = _mm_f32(g, w_head.detach().float().t())                      # g W_head: [rows, D]
        if t is None:
            return ga, None, None, gw_head, gb_head
        ga = ga * (1.0 - t * t)                                           # through tanh, from its output
        gb_pre = ga.sum(0).to(b_pre.dtype)
        gw_pre = _mm_f32(ga.t(), feat.t()).to(w_pre.dtype)
        gfeat = _mm_f32(ga, w_pre.detach().float().t())
        return gfeat, gw_pre, gb_pre, gw_head, gb_head


class QkvFn(torch.autograd.Function):
    """qkv [3, B, H, N, hd] = head-major split of a W_in^T + b (models/transformer.py:42)."""

    @staticmethod
    def forward(ctx, a, weight, bias, w_cd, w_t, B, N, H, hd):
        qkv = torch.empty(3, B, H, N, hd, dtype=a.dtype, device=a.device)
        b32 = None if bias is None else bias.detach()
        if a.dtype == torch.float32:
            ops.qkv_proj_f32(a, w_cd, b32, qkv, B, N, H, hd)
        else:
            ops.qkv_proj(a, w_cd, b32, qkv, B, N, H, hd)
        ctx.save_for_backward(a, weight, bias, w_cd, w_t)
        return qkv

    @staticmethod
    def backward(ctx, g):
        a, weight, bias, w_cd, w_t = ctx.saved_tensors
        _, B, H, N, hd = g.shape
        g_pre = g.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * H * hd)        # back to [rows, which*D + h*hd + e]
        ga, gw, gb = _linear_backward(g_pre, a, w_cd, w_t, weight, bias)
        return ga, gw, gb, None, None, None, None, None, None


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

    bf16, head_dim 64, N <= 608: the forward also keeps each row's log-sum-exp and the backward is the HIP
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
        gw, _ = _weight_grad(gp_cd, cols)
        gb = gp.sum((0, 1))
        gc = g[:, 0, :].sum(0).view(1, 1, D)
        gpos = g.sum(0, keepdim=True)
        return None, gw.view(weight.shape).to(weight.dtype), gb.to(bias.dtype), gc.to(cls.dtype), gpos.to(pos.dtype), \
            None, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """act(a W^T + b) in cd, act = GELU or none, as its own graph node (no residual, no LayerNorm fused behind it):
    the building block of the training path with dropout > 0, where a Dropout sits between every Linear layer and
    whatever follows it (models/transformer.py:21-25,47,69-70)."""

    @staticmethod
    def forward(ctx, a, weight, bias, w_cd, w_t, gelu):
        rows, cd = a.shape[0], a.dtype
        b32 = None if bias is None else bias.detach()
        out = torch.empty(rows, weight.shape[0], dtype=cd, device=a.device)
        pre = None
        if cd == torch.float32:
            ops.gemm_f32(a, w_cd, b32, out, EPI_GELU if gelu else EPI_BIAS)
        elif not gelu:
            ops.gemm_bias(a, w_cd, b32, out)
        elif ops.gemm_bias_gelu_dual_supported(rows, weight.shape[0], a.shape[1]):
            pre = torch.empty_like(out)
            ops.gemm_bias_gelu_dual(a, w_cd, b32, pre, out)
        else:
            ops.gemm_bias_gelu(a, w_cd, b32, out)
        ctx.gelu = gelu
        ctx.save_for_backward(a, weight, bias, w_cd, w_t, pre)
        return out

    @staticmethod
    def backward(ctx, g):
        a, weight, bias, w_cd, w_t, pre = ctx.saved_tensors
        g = _operand(g.contiguous(), a.dtype)
        if ctx.gelu:
            if pre is None:                                       # not kept by the forward: recomputed
                pre = F.linear(a, w_cd, None if bias is None else bias.detach().to(a.dtype))
            g = torch.ops.aten.gelu_backward(g, pre)
        ga, gw, gb = _linear_backward(g, a, w_cd, w_t, weight, bias)
        return ga, gw, gb, None, None, None


def _attention_dropout(qkv, scale, p):
    """models/transformer.py:43-46 with the Dropout on the probabilities active: the probabilities have to exist as
    a tensor, so this is torch arithmetic (f32 softmax), differentiated by autograd."""
    _, B, H, N, hd = qkv.shape
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q.float() @ k.float().transpose(-2, -1)) * scale).softmax(-1)
    att = F.dropout(att, p, True)
    return (att.to(v.dtype) @ v).transpose(1, 2).reshape(B, N, H * hd)


def _forward_train_dropout(model, xs, w, cd, B, N, p, p_attn):
    """The encoder of forward_train with the reference's Dropout layers active (training mode, dropout or
    attn_dropout > 0): embedding (models/vision_transformer.py:51), attention probabilities (transformer.py:45),
    attention branch (transformer.py:70), after GELU and after fc2 (transformer.py:23,25).  The masks come from
    torch's generator (F.dropout), so a seeded run is reproducible; every GEMM, LayerNorm and (without
    attn_dropout) the attention are the same kernels as elsewhere, as separate graph nodes."""
    D, H = model.hidden_size, model.num_heads
    hd = D // H
    x2 = F.dropout(xs, p, True).view(B * N, D)
    for i, blk in enumerate(model.encoder.blocks):
        q = f"b{i}."
        att, mlp = blk.attention, blk.mlp
        y, _ = LayerNormFn.apply(x2, blk.norm1.weight, blk.norm1.bias, float(blk.norm1.eps), cd)
        qkv = QkvFn.apply(y, att.in_proj.weight, att.in_proj.bias, w[q + "qkv.w"], w.get(q + "qkv.wt"), B, N, H, hd)
        ctxt = _attention_dropout(qkv, att.scaling, p_attn) if p_attn > 0. else AttentionFn.apply(qkv, att.scaling)
        o = LinearFn.apply(ctxt.reshape(B * N, D), att.out_proj.weight, att.out_proj.bias, w[q + "out.w"], w.get(q + "out.wt"), False)
        x2 = x2 + F.dropout(o.float(), p, True)
        y, _ = LayerNormFn.apply(x2, blk.norm2.weight, blk.norm2.bias, float(blk.norm2.eps), cd)
        h = LinearFn.apply(y, mlp.fc1.weight, mlp.fc1.bias, w[q + "fc1.w"], w.get(q + "fc1.wt"), True)
        h = F.dropout(h.float(), p, True).to(cd)                 # masks drawn on f32 tensors in both modes: one seed, one mask
        z = LinearFn.apply(h, mlp.fc2.weight, mlp.fc2.bias, w[q + "fc2.w"], w.get(q + "fc2.wt"), False)
        x2 = x2 + F.dropout(z.float(), p, True)
    return x2


def _forward_train_fused(model, x2, w, cd, B, N):
    """The encoder blocks without dropout: each residual branch and the LayerNorm after it as one graph node."""
    D, H = model.hidden_size, model.num_heads
    hd = D // H
    blocks = list(model.encoder.blocks)
    if blocks:
        n1 = blocks[0].norm1
        y, x2 = LayerNormFn.apply(x2, n1.weight, n1.bias, float(n1.eps), cd)
    for i, blk in enumerate(blocks):
        p = f"b{i}."
        att, mlp = blk.attention, blk.mlp
        qkv = QkvFn.apply(y, att.in_proj.weight, att.in_proj.bias, w[p + "qkv.w"], w.get(p + "qkv.wt"), B, N, H, hd)
        ctxt = AttentionFn.apply(qkv, att.scaling)
        y, x2 = ResidualBranchFn.apply(ctxt.view(B * N, D), x2, att.out_proj.weight, att.out_proj.bias, w[p + "out.w"],
                                       w.get(p + "out.wt"), None, None, None, None,
                                       blk.norm2.weight, blk.norm2.bias, float(blk.norm2.eps))
        nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else None        # the next block's first LayerNorm
        res = ResidualBranchFn.apply(y, x2, mlp.fc2.weight, mlp.fc2.bias, w[p + "fc2.w"], w.get(p + "fc2.wt"),
                                     mlp.fc1.weight, mlp.fc1.bias, w[p + "fc1.w"], w.get(p + "fc1.wt"),
                                     None if nxt is None else nxt.weight, None if nxt is None else nxt.bias,
                                     0.0 if nxt is None else float(nxt.eps))
        if nxt is None:
            x2 = res
        else:
            y, x2 = res
    return x2


def forward_train(model, x):
    """The forward of vit_amd.VisionTransformer with autograd: same kernels, fresh tensors per op."""
    w = model._cache.get(train=True)
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
    if not fp32 and (w["embed.w"].shape[1] != 3 * P * P or Wi % 8 != 0):   # explicit patch matrix (patch not a power of two, or odd width)
        cols = torch.empty(B * (N - 1), w["embed.w"].shape[1], dtype=torch.bfloat16, device=x.device)
    xs = PatchEmbedFn.apply(img, model.embedding.weight, model.embedding.bias, model.cls_token, model.pos_embed,
                            w["embed.w"], w["embed.b"], w["cls"], w["pos"], P, cols)
    p = float(model.dropout.p) if model.training else 0.
    p_attn = float(model.encoder.blocks[0].attention.dropout.p) if model.training and len(model.encoder.blocks) else 0.
    if p > 0. or p_attn > 0.:
        x2 = _forward_train_dropout(model, xs, w, cd, B, N, p, p_attn)
    else:
        x2 = _forward_train_fused(model, xs.view(B * N, D), w, cd, B, N)
    cls_rows = x2.view(B, N, D)[:, 0].contiguous()                         # the reference keeps token 0 (transformer.py:99)
    norm = model.encoder.norm
    feat, _ = LayerNormFn.apply(cls_rows, norm.weight, norm.bias, float(norm.eps), torch.float32)
    has_pre = "pre.w" in w
    logits = HeadFn.apply(feat, model.pre_logits.weight if has_pre else None, model.pre_logits.bias if has_pre else None,
                          model.head.weight, model.head.bias)
    return logits.to(model.head.weight.dtype)
