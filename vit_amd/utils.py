"""The one helper of the reference's utils.py that sits next to the forward path.

`pos_embed_scale` mirrors /root/reference/utils.py:241-261 (same name, arguments and return convention): it
lets a checkpoint trained at one resolution load into a model built for another (utils.py:234-235).  The
resampling itself runs on the GPU (`vitk_pos_embed_scale`); there is no scipy / CPU path.
"""
import math

import torch

from . import lib as _lib


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
