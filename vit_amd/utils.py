"""The helpers of the reference's utils.py / data.py that sit either side of the forward path.

`pos_embed_scale` mirrors /root/reference/utils.py:241-261 (same name, arguments and return convention): it
lets a checkpoint trained at one resolution load into a model built for another (utils.py:234-235).  The
resampling itself runs on the GPU (`vitk_pos_embed_scale`); there is no scipy / CPU path.
"""
import math

import torch

from . import lib as _lib
from . import ops as _ops

# The reference's input statistics in uint8 pixel units (data.py:54-55).
PIXEL_MEAN = (0.485 * 255, 0.456 * 255, 0.406 * 255)
PIXEL_STD = (0.229 * 255, 0.224 * 255, 0.225 * 255)


def normalize_u8(images, mean=PIXEL_MEAN, std=PIXEL_STD, dtype=torch.float32, out=None):
    """The DataFetcher's on-GPU input step (data.py:69-72): uint8 [B,3,H,W] -> (float(x) - mean) / std.

    dtype=torch.float32 gives exactly what the reference hands to the model; dtype=torch.bfloat16 writes the
    patch GEMM's operand type directly (the model does this itself when it is called with uint8 images)."""
    if not images.is_cuda:
        raise _lib.VitkError("normalize_u8 needs a CUDA/HIP tensor: this path has no CPU fallback")
    if out is None:
        out = torch.empty(images.shape, dtype=dtype, device=images.device)
    with torch.cuda.device(images.device):
        return _ops.normalize_u8(images.contiguous(), mean, std, out)


def accuracy(output, target, topk=(1,)):
    """Precision@k in percent, one [1]-shaped tensor per k, as utils.py:182-195 returns them.

    One kernel ranks the target class of every row; no top-k index matrix is built.  Ties in the logits
    count towards the lower class index."""
    if not output.is_cuda:
        raise _lib.VitkError("accuracy needs CUDA/HIP tensors: this path has no CPU fallback")
    batch_size = target.size(0)
    with torch.cuda.device(output.device):
        rank = _ops.target_rank(output.detach().float().contiguous(),
                                target.to(device=output.device, dtype=torch.int64).contiguous())
    res = []
    for k in topk:
        correct_k = (rank < k).float().sum(0, keepdim=True)
        res.append(correct_k.mul_(100.0 / batch_size))
    return res


def pos_embed_scale(pos_embed, img_size, patch_size, mode='constant', order=1):
    """pos_embed [1, 1 + g*g, D] -> [1, 1 + G*G, D], G = img_size // patch_size; returned on the input's
    device and dtype.  Unchanged (the same tensor) when the lengths already agree, as in the reference."""
    n_ckpt = pos_embed.shape[1]
    n_model = (img_size * img_size) // (patch_size * patch_size) + 1
    if n_ckpt == n_model:
        return pos_embed
    if order != 1 or mode != 'constant':
        raise _lib.VitkError(f"pos_embed_scale: only order=1, mode='constant' (the reference's defaults) are "
                             f"implemented, got order={order}, mode={mode!r}")
    g, G = int(math.sqrt(n_ckpt - 1)), int(math.sqrt(n_model - 1))
    if g * g != n_ckpt - 1 or pos_embed.dim() != 3 or pos_embed.shape[0] != 1:
        raise _lib.VitkError(f"pos_embed_scale: expected [1, 1 + g*g, D], got {tuple(pos_embed.shape)}")
    if not torch.cuda.is_available():
        raise _lib.VitkError("pos_embed_scale needs a GPU: this path has no CPU fallback")
    L = _lib.load()
    dev = pos_embed.device if pos_embed.is_cuda else torch.device("cuda", torch.cuda.current_device())
    src = pos_embed.detach().to(device=dev, dtype=torch.float32).contiguous()
    D = src.shape[2]
    out = torch.empty(1, 1 + G * G, D, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.vitk_pos_embed_scale(src.data_ptr(), out.data_ptr(), g, G, D,
                                          torch.cuda.current_stream().cuda_stream), L)
    return out.to(device=pos_embed.device, dtype=pos_embed.dtype)


def adapt_state_dict(state_dict, img_size, num_classes=1000, pretrained=False, patch_size=16):
    """The state_dict edits of the reference's load_checkpoint (utils.py:211-235), without its file and
    optimizer handling: a checkpoint dict is unwrapped to its 'state_dict'; with `pretrained` the head is
    replaced by zeros of [num_classes, hidden] and the pre_logits layer is dropped (fine-tuning start, so the
    model must be built with pre_logits=False); pos_embed is rescaled to the model's grid.  Returns the
    (modified in place, as in the reference) state_dict, ready for a strict load_state_dict."""
    if 'state_dict' in state_dict:
        state_dict = state_dict['state_dict']
    if pretrained:
        hidden_width = state_dict['head.weight'].shape[1]
        state_dict['head.weight'] = torch.zeros(num_classes, hidden_width)
        state_dict['head.bias'] = torch.zeros(num_classes)
        state_dict.pop('pre_logits.weight', None)
        state_dict.pop('pre_logits.bias', None)
    state_dict['pos_embed'] = pos_embed_scale(state_dict['pos_embed'], img_size=img_size, patch_size=patch_size)
    return state_dict
