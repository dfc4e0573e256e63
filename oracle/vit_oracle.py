"""CPU oracle for the ViT forward path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch fp32 functional ops on the CPU, the
algorithm of the reference's hot path (`/root/reference/models/
vision_transformer.py` and `/root/reference/models/transformer.py`).  It is the
checker the HIP path is compared with; it is never part of the product path.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.

Pinning: the reference holds no tests or golden vectors of its own.  The
oracle is pinned by running the reference model itself (imported from
/root/reference in the build container by `tests/golden/make_golden.py`) on the
same synthetic weights and images and committing its logits and intermediates
under `tests/golden/`; `tests/test_oracle_golden.py` checks this file against
those vectors.

All arithmetic here is PyTorch's (the reference pins no version; the vectors
were captured with torch 2.10.0+rocm7.0 on CPU).
"""

import torch
import torch.nn.functional as F


def _heads(cfg):
    D, H = cfg["hidden_size"], cfg["num_heads"]
    return D, H, D // H


def patch_embed(images, sd, P):
    """Conv2d(3, D, k=P, s=P) -> flatten(2).transpose(1, 2); prepend cls; add pos.
    models/vision_transformer.py:18-19,47-50."""
    x = F.conv2d(images, sd["embedding.weight"], sd["embedding.bias"], stride=P)
    x = x.flatten(2).transpose(1, 2)
    cls = sd["cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    return x + sd["pos_embed"]


def layernorm(x, w, b, eps):
    """nn.LayerNorm over the last dim, biased variance.
    models/transformer.py:56,60,93; eps from models/vision_transformer.py:69."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(x, sd, prefix, H, scaling=None, stages=None):
    """Attention.forward, models/transformer.py:40-48."""
    B, N, C = x.shape
    hd = C // H
    scaling = scaling or hd ** -0.5                      # transformer.py:35
    qkv = F.linear(x, sd[prefix + "in_proj.weight"], sd.get(prefix + "in_proj.bias"))
    q, k, v = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-2, -1)) * scaling
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, C)
    if stages is not None:
        stages[prefix + "q"] = q
        stages[prefix + "k"] = k
        stages[prefix + "v"] = v
        stages[prefix + "ctx"] = o
    return F.linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def mlp(x, sd, prefix):
    """MLPBlock.forward, models/transformer.py:20-26 (nn.GELU() = exact erf)."""
    h = F.gelu(F.linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]))
    return F.linear(h, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"])


def block(x, sd, i, H, eps, scaling=None, stages=None):
    """Encoder1DBlock.forward, models/transformer.py:66-76 (dropout p=0 and
    drop_module are identities on this path)."""
    p = f"encoder.blocks.{i}."
    y = layernorm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    x = attention(y, sd, p + "attention.", H, scaling, stages) + x
    y = layernorm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    return x + mlp(y, sd, p + "mlp.")


@torch.no_grad()
def forward(sd, images, cfg, stages=None, attn_scaling=None):
    """VisionTransformer.forward (models/vision_transformer.py:45-56) with
    Encoder.forward (models/transformer.py:95-99).  `cfg` holds patches,
    hidden_size, num_layers, num_heads, eps.  `stages`, when a dict, receives
    named intermediates."""
    _, H, _ = _heads(cfg)
    eps = cfg["eps"]
    x = patch_embed(images, sd, cfg["patches"])
    if stages is not None:
        stages["embed"] = x
    for i in range(cfg["num_layers"]):
        x = block(x, sd, i, H, eps, attn_scaling, stages if i == 0 else None)
        if stages is not None and i == 0:
            stages["block0"] = x
    x = layernorm(x, sd["encoder.norm.weight"], sd["encoder.norm.bias"], eps)[:, 0]
    if stages is not None:
        stages["encoded"] = x
    if "pre_logits.weight" in sd:
        x = torch.tanh(F.linear(x, sd["pre_logits.weight"], sd["pre_logits.bias"]))
    return F.linear(x, sd["head.weight"], sd["head.bias"])


# ---------------------------------------------------------------------------
# bf16-operand emulation: rounds to bf16 exactly where the HIP path does
# (GEMM operands, Q/K/V, softmax probabilities, attention context, GELU
# output), and keeps the residual stream, LayerNorm statistics, softmax and
# accumulation in fp32.  Used by GPU tests for a tight kernel-level check next
# to the fp32 tolerance the north star states.
# ---------------------------------------------------------------------------

def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def linear16(x16, w, b):
    y = x16 @ r16(w).t()
    return y if b is None else y + b


@torch.no_grad()
def forward_bf16_emulated(sd, images, cfg, stages=None, attn_scaling=None):
    D, H, hd = _heads(cfg)
    eps, P = cfg["eps"], cfg["patches"]
    scaling = attn_scaling or hd ** -0.5
    B = images.shape[0]
    x = F.conv2d(r16(images), r16(sd["embedding.weight"]), sd["embedding.bias"], stride=P)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), x), dim=1) + sd["pos_embed"]
    if stages is not None:
        stages["embed"] = x
    N = x.shape[1]
    for i in range(cfg["num_layers"]):
        p = f"encoder.blocks.{i}."
        y = r16(layernorm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps))
        qkv = r16(linear16(y, sd[p + "attention.in_proj.weight"], sd.get(p + "attention.in_proj.bias")))
        q, k, v = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-2, -1)) * scaling
        m = s.amax(dim=-1, keepdim=True)
        e = torch.exp(s - m)
        l = e.sum(dim=-1, keepdim=True)
        o = (r16(e) @ v) / l
        o = r16(o.transpose(1, 2).reshape(B, N, D))
        x = x + linear16(o, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        y = r16(layernorm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
        h = r16(F.gelu(linear16(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])))
        x = x + linear16(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        if stages is not None and i == 0:
            stages["block0"] = x
    # the head (one row per image) runs in f32 on the HIP path too: no rounding from here on
    x = layernorm(x[:, 0], sd["encoder.norm.weight"], sd["encoder.norm.bias"], eps)
    if stages is not None:
        stages["encoded"] = x
    if "pre_logits.weight" in sd:
        x = torch.tanh(F.linear(x, sd["pre_logits.weight"], sd["pre_logits.bias"]))
    return F.linear(x, sd["head.weight"], sd["head.bias"])


def flops_per_image(cfg, img_size, num_classes=1000):
    """Algorithmic FLOPs (2 per multiply-add) of one forward."""
    D, H, hd = _heads(cfg)
    P, L = cfg["patches"], cfg["num_layers"]
    M = int(D * cfg["mlp_ratio"])
    n = (img_size // P) ** 2
    N = n + 1
    per_layer = 2 * N * D * 3 * D + 2 * (2 * H * N * N * hd) + 2 * N * D * D + 2 * (2 * N * D * M)
    return 2 * n * (3 * P * P) * D + L * per_layer + 2 * D * D + 2 * D * num_classes
