"""TEST INFRASTRUCTURE ONLY: CPU restatement of one training step of the reference.

Follows /root/reference/main.py:207-223 with the reference's defaults (main.py:128 `nn.CrossEntropyLoss()`,
run.py:224 `gradient_clip=1.0`, run.py:55-60 AdamW lr 1e-3, weight_decay 0.03): logits = model(images) with the
forward of oracle/vit_oracle.py, loss = cross entropy, gradients by torch autograd in fp32, global-norm clipping.
Pinned by tests/golden/train_small.npz (and train_dropout.npz for the forward with active Dropout layers): loss,
logits and parameter gradients captured from the reference model class itself under torch autograd
(tests/golden/make_golden.py train / train_dropout).  Only tests may import this module.
"""
import torch
import torch.nn.functional as F

from . import vit_oracle

# the configuration of the training fixture: small enough for a CPU run and for a committed fixture, with the
# shape properties the kernels need (hidden size a multiple of 256, head_dim 64)
SMALL = dict(patches=8, hidden_size=256, num_layers=2, num_heads=4, mlp_ratio=4, eps=1e-6)
SMALL_IMG, SMALL_CLASSES, SMALL_BATCH = 32, 10, 6


def forward_dropout(sd, images, cfg, p, p_attn):
    """The reference forward in training mode with its Dropout layers active: after the position embedding
    (models/vision_transformer.py:51), on the attention probabilities (models/transformer.py:45), after the
    attention branch (transformer.py:70), after GELU and after fc2 (transformer.py:23,25).  The masks come from
    torch's global generator in this order, on tensors of the reference's shapes, so a seeded run draws the masks
    a seeded run of the reference draws (pinned by tests/golden/train_dropout.npz)."""
    D, H = cfg["hidden_size"], cfg["num_heads"]
    hd, eps = D // H, cfg["eps"]
    x = F.dropout(vit_oracle.patch_embed(images, sd, cfg["patches"]), p, True)
    B, N, _ = x.shape
    for i in range(cfg["num_layers"]):
        q = f"encoder.blocks.{i}."
        y = vit_oracle.layernorm(x, sd[q + "norm1.weight"], sd[q + "norm1.bias"], eps)
        qkv = F.linear(y, sd[q + "attention.in_proj.weight"], sd.get(q + "attention.in_proj.bias"))
        qh, kh, vh = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        att = F.dropout(((qh @ kh.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1), p_attn, True)
        o = F.linear((att @ vh).transpose(1, 2).reshape(B, N, D), sd[q + "attention.out_proj.weight"],
                     sd[q + "attention.out_proj.bias"])
        x = F.dropout(o, p, True) + x
        y = vit_oracle.layernorm(x, sd[q + "norm2.weight"], sd[q + "norm2.bias"], eps)
        h = F.dropout(F.gelu(F.linear(y, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"])), p, True)
        x = x + F.dropout(F.linear(h, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"]), p, True)
    x = vit_oracle.layernorm(x, sd["encoder.norm.weight"], sd["encoder.norm.bias"], eps)[:, 0]
    if "pre_logits.weight" in sd:
        x = torch.tanh(F.linear(x, sd["pre_logits.weight"], sd["pre_logits.bias"]))
    return F.linear(x, sd["head.weight"], sd["head.bias"])


def loss_and_grads(sd, images, targets, cfg, dropout=None):
    """sd: fp32 state_dict -> (loss, logits, {key: grad}) of mean cross entropy; dropout = (p, p_attn) runs the
    training-mode forward with active Dropout layers (seed torch's generator first)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    if dropout is not None:
        logits = forward_dropout(params, images, cfg, *dropout)
    else:
        logits = vit_oracle.forward.__wrapped__(params, images, cfg)       # the forward itself, without its no_grad wrapper
    loss = F.cross_entropy(logits, targets)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), logits.detach(), dict(zip(params.keys(), grads))


def clip_coefficient(grads, max_norm=1.0):
    """The factor torch.nn.utils.clip_grad_norm_ (main.py:221-223) scales every gradient by, and the norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total
