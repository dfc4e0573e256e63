"""MI355X-native ViT forward path with the Python surface of ZhiyuanChen/vit's
`models` package: `getattr(vit_amd, arch)(**kwargs)` for arch in
s16/b16/b32/l16/l32/h14 builds a module whose state_dict keys, constructor and
forward signature match the reference's `models/vision_transformer.py`."""
from .lib import VitkError  # noqa: F401
from .graph import GraphedForward  # noqa: F401
from .utils import pos_embed_scale, normalize_u8, accuracy, adapt_state_dict  # noqa: F401
from .vision_transformer import VisionTransformer, s16, b16, b32, l16, l32, h14  # noqa: F401

__all__ = ["VisionTransformer", "s16", "b16", "b32", "l16", "l32", "h14", "VitkError", "pos_embed_scale",
           "normalize_u8", "accuracy", "adapt_state_dict", "GraphedForward"]
