"""Bit-reproducible synthetic weights and images for the ViT forward path.

Every value is a pure function of (seed, tensor name, element index): a
splitmix64 hash in uint64 arithmetic, reduced to small integers that convert
to float32 exactly.  There is no dependence on any library RNG stream, so the
same tensors come out in the build container (where golden vectors are
captured) and on the GPU box (where the HIP path consumes them).

Magnitudes follow the reference initialisation
(/root/reference/models/vision_transformer.py:34-43): Linear weights std 0.02,
cls_token std 0.2, pos_embed std 0.02, conv weight/bias at PyTorch's default
kaiming-uniform bound 1/sqrt(fan_in) (the reference never re-initialises
`embedding`).  `style="reference"` keeps Linear biases 0 and LayerNorm at
(1, 0) as the reference does; `style="perturbed"` gives them small non-zero
values so that a dropped bias or affine term shows up in parity tests.
"""
import math
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(z):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + _GOLD) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * _M1) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * _M2) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _stream_key(seed, name):
    tag = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    s = np.array([np.uint64(seed)], dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = _splitmix64(s ^ (tag << np.uint64(32)) ^ tag)
    return k[0]


def _bits(seed, name, n, chunk=1 << 22):
    """n 64-bit hash words for tensor `name`."""
    key = _stream_key(seed, name)
    out = np.empty(n, dtype=np.uint64)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = np.arange(lo, hi, dtype=np.uint64)
        with np.errstate(over="ignore"):
            out[lo:hi] = _splitmix64((idx * _GOLD + key) & _MASK)
    return out


def uniform(seed, name, shape, bound):
    """U(-bound, bound): a 24-bit integer scaled in float32 (exact everywhere)."""
    n = int(np.prod(shape))
    u = (_bits(seed, name, n) >> np.uint64(40)).astype(np.int64)   # [0, 2^24)
    x = (u - (1 << 23)).astype(np.float32) * np.float32(bound / float(1 << 23))
    return x.reshape(shape)


def normal(seed, name, shape, std):
    """Approximately N(0, std^2): Irwin-Hall sum of four 16-bit integers.

    The sum is an exact integer < 2^18, so the float32 conversion and the
    single multiply are bit-identical on every platform.
    """
    n = int(np.prod(shape))
    b = _bits(seed, name, n)
    m16 = np.uint64(0xFFFF)
    s = ((b & m16) + ((b >> np.uint64(16)) & m16) + ((b >> np.uint64(32)) & m16)
         + (b >> np.uint64(48))).astype(np.int64)
    s = s - 2 * 65535
    sigma = math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)
    x = s.astype(np.float32) * np.float32(std / sigma)
    return x.reshape(shape)


# Architecture table: the six factories of
# /root/reference/models/vision_transformer.py:59-97.
ARCHS = {
    "s16": dict(patches=16, hidden_size=768, num_layers=8, num_heads=8, mlp_ratio=3, eps=1e-5),
    "b16": dict(patches=16, hidden_size=768, num_layers=12, num_heads=12, mlp_ratio=4, eps=1e-6),
    "b32": dict(patches=32, hidden_size=768, num_layers=12, num_heads=12, mlp_ratio=4, eps=1e-6),
    "l16": dict(patches=16, hidden_size=1024, num_layers=24, num_heads=16, mlp_ratio=4, eps=1e-6),
    "l32": dict(patches=32, hidden_size=1024, num_layers=24, num_heads=16, mlp_ratio=4, eps=1e-6),
    "h14": dict(patches=14, hidden_size=1280, num_layers=32, num_heads=16, mlp_ratio=4, eps=1e-5),
}


def state_dict_spec(patches, hidden_size, num_layers, mlp_ratio, img_size,
                    num_classes=1000, pre_logits=True, **_):
    """Ordered (key, shape) list: the 4 + 12*L + 6 entries the reference's
    `state_dict()` holds (models/vision_transformer.py:18-30,
    models/transformer.py:15-17,36-38,56-60,93)."""
    D, P = hidden_size, patches
    M = int(D * mlp_ratio)
    N = (img_size // P) ** 2 + 1
    spec = [("cls_token", (1, 1, D)), ("pos_embed", (1, N, D)),
            ("embedding.weight", (D, 3, P, P)), ("embedding.bias", (D,))]
    for i in range(num_layers):
        p = f"encoder.blocks.{i}."
        spec += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                 (p + "attention.in_proj.weight", (3 * D, D)), (p + "attention.in_proj.bias", (3 * D,)),
                 (p + "attention.out_proj.weight", (D, D)), (p + "attention.out_proj.bias", (D,)),
                 (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
                 (p + "mlp.fc1.weight", (M, D)), (p + "mlp.fc1.bias", (M,)),
                 (p + "mlp.fc2.weight", (D, M)), (p + "mlp.fc2.bias", (D,))]
    spec += [("encoder.norm.weight", (D,)), ("encoder.norm.bias", (D,))]
    if pre_logits:
        spec += [("pre_logits.weight", (D, D)), ("pre_logits.bias", (D,))]
    spec += [("head.weight", (num_classes, D)), ("head.bias", (num_classes,))]
    return spec


def synth_state_dict(cfg, img_size, seed=0, style="reference", num_classes=1000,
                     pre_logits=True):
    """fp32 state_dict for architecture dict `cfg` (an ARCHS entry or alike)."""
    spec = state_dict_spec(img_size=img_size, num_classes=num_classes,
                           pre_logits=pre_logits, **cfg)
    P = cfg["patches"]
    conv_bound = 1.0 / math.sqrt(3 * P * P)
    pert = style == "perturbed"
    if style not in ("reference", "perturbed"):
        raise ValueError(f"unknown style {style!r}")
    sd = {}
    for key, shape in spec:
        if key == "cls_token":
            v = normal(seed, key, shape, 0.2)
        elif key == "pos_embed":
            v = normal(seed, key, shape, 0.02)
        elif key.startswith("embedding."):
            v = uniform(seed, key, shape, conv_bound)
        elif "norm" in key and key.endswith(".weight"):
            v = np.ones(shape, np.float32)
            if pert:
                v = v + normal(seed, key, shape, 0.05)
        elif key.endswith(".bias"):
            v = normal(seed, key, shape, 0.02) if pert else np.zeros(shape, np.float32)
        else:
            v = normal(seed, key, shape, 0.02)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synth_images(batch, img_size, seed=1, first=0, indices=None):
    """Images ~N(0,1), [batch,3,S,S] fp32: the statistics of ImageNet pixels
    after the reference's on-device normalisation (data.py:72).  Image i of the
    stream is independent of the batch it is drawn in, so shards of one logical
    batch can be produced on different ranks: `first` is the stream index of
    the first image, or `indices` lists the stream indices explicitly."""
    if indices is None:
        indices = range(first, first + batch)
    indices = list(indices)
    out = np.empty((len(indices), 3, img_size, img_size), np.float32)
    for b, i in enumerate(indices):
        out[b] = normal(seed, f"image{i}", (3, img_size, img_size), 1.0)
    return torch.from_numpy(out)
