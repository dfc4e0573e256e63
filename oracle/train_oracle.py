"""TEST INFRASTRUCTURE ONLY: CPU restatement of one training step of the reference.

Follows /root/reference/main.py:207-223 with the reference's defaults (main.py:128 `nn.CrossEntropyLoss()`,
run.py:224 `gradient_clip=1.0`, run.py:55-60 AdamW lr 1e-3, weight_decay 0.03): logits = model(images) with the
forward of oracle/vit_oracle.py, loss = cross entropy, gradients by torch autograd in fp32, global-norm clipping.
Pinned by tests/golden/train_small.npz: loss, logits and parameter gradients captured from the reference model
class itself under torch autograd (tests/golden/make_golden.py train).  Only tests may import this module.
"""
import torch
import torch.nn.functional as F

from . import vit_oracle

# the configuration of the training fixture: small enough for a CPU run and for a committed fixture, with the
# shape properties the kernels need (hidden size a multiple of 256, head_dim 64)
SMALL = dict(patches=8, hidden_size=256, num_layers=2, num_heads=4, mlp_ratio=4, eps=1e-6)
SMALL_IMG, SMALL_CLASSES, SMALL_BATCH = 32, 10, 6


def loss_and_grads(sd, images, targets, cfg):
    """sd: fp32 state_dict -> (loss, logits, {key: grad}) of mean cross entropy."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = vit_oracle.forward.__wrapped__(params, images, cfg)           # the forward itself, without its no_grad wrapper
    loss = F.cross_entropy(logits, targets)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), logits.detach(), dict(zip(params.keys(), grads))


def clip_coefficient(grads, max_norm=1.0):
    """The factor torch.nn.utils.clip_grad_norm_ (main.py:221-223) scales every gradient by, and the norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total
