"""Flax/JAX ViT `.npz` -> state_dict with the keys the reference's converter emits.

Contract (from /root/reference/converter.py:11-58, checked against its output
on a synthetic checkpoint in tests/test_converter.py): 4 + 12*L + 6 tensors
named as `VisionTransformer.state_dict()` names them, so the result loads with
`load_state_dict(strict=True)`.

Layout changes per tensor family:
  cls                               -> cls_token, kept [1,1,D]
  embedding/kernel [P,P,3,D]        -> embedding.weight [D,3,P,P]      (HWIO -> OIHW)
  .../{query,key,value}/kernel [D,H,hd], bias [H,hd]
                                    -> attention.in_proj.weight [3D,D], .bias [3D]  (q;k;v stacked)
  .../out/kernel [H,hd,D]           -> attention.out_proj.weight [D,D]
  MlpBlock_3/Dense_{0,1}/kernel     -> mlp.fc{1,2}.weight, transposed to [out,in]
  head/kernel [D,classes]           -> head.weight [classes,D]
  pre_logits/kernel [D,D]           -> pre_logits.weight, NOT transposed by the reference
                                       (converter.py:22-37 transposes only keys containing
                                       'fc', 'head' or 'att'); `fix_pre_logits=True` transposes
                                       it to nn.Linear's [out,in] convention instead.
Everything except pos_embed and cls_token is squeezed, as the reference does.
"""
import re
from collections import OrderedDict

import numpy as np
import torch

_BLOCK = re.compile(r"^Transformer/encoderblock_(\d+)/(.+)$")
_ATT = "MultiHeadDotProductAttention_1/"

_BLOCK_LEAVES = {
    "LayerNorm_0/scale": "norm1.weight", "LayerNorm_0/bias": "norm1.bias",
    "LayerNorm_2/scale": "norm2.weight", "LayerNorm_2/bias": "norm2.bias",
    "MlpBlock_3/Dense_0/kernel": "mlp.fc1.weight", "MlpBlock_3/Dense_0/bias": "mlp.fc1.bias",
    "MlpBlock_3/Dense_1/kernel": "mlp.fc2.weight", "MlpBlock_3/Dense_1/bias": "mlp.fc2.bias",
}
_TOP = {
    "embedding/kernel": "embedding.weight", "embedding/bias": "embedding.bias",
    "Transformer/posembed_input/pos_embedding": "pos_embed",
    "Transformer/encoder_norm/scale": "encoder.norm.weight",
    "Transformer/encoder_norm/bias": "encoder.norm.bias",
    "pre_logits/kernel": "pre_logits.weight", "pre_logits/bias": "pre_logits.bias",
    "head/kernel": "head.weight", "head/bias": "head.bias",
}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def convert(npz, fix_pre_logits=False):
    """`npz`: a mapping name -> array (np.load result).  Returns an OrderedDict in the
    reference's emission order (cls_token first, then the archive's key order)."""
    sd = OrderedDict()
    sd["cls_token"] = _t(npz["cls"])
    for name in npz:
        if name == "cls" or "/key/" in name or "/value/" in name:
            continue
        m = _BLOCK.match(name)
        if m is None:
            if name not in _TOP:
                raise KeyError(f"unrecognised checkpoint entry {name!r}")
            key, val = _TOP[name], _t(npz[name])
            if key == "embedding.weight":
                val = val.permute(3, 2, 0, 1)
            elif key == "head.weight" or (key == "pre_logits.weight" and fix_pre_logits):
                val = val.T
            sd[key] = val if key == "pos_embed" else val.squeeze()
            continue
        prefix, leaf = f"encoder.blocks.{m.group(1)}.", m.group(2)
        if leaf in _BLOCK_LEAVES:
            key, val = _BLOCK_LEAVES[leaf], _t(npz[name])
            sd[prefix + key] = (val.T if key.endswith("weight") and "fc" in key else val).squeeze()
        elif leaf.startswith(_ATT):
            part, kind = leaf[len(_ATT):].split("/")
            if part == "out":
                val = _t(npz[name])
                if kind == "kernel":                       # [H,hd,D] -> [D_out, D_in]
                    val = val.reshape(-1, val.shape[-1]).T
                sd[prefix + "attention.out_proj." + ("weight" if kind == "kernel" else "bias")] = val.squeeze()
            elif part == "query":
                trio = [_t(npz[name.replace("/query/", f"/{p}/")]) for p in ("query", "key", "value")]
                if kind == "kernel":                       # [D,H,hd] each -> [3D, D]
                    val = torch.cat([w.reshape(w.shape[0], -1).T for w in trio])
                    sd[prefix + "attention.in_proj.weight"] = val.squeeze()
                else:                                      # [H,hd] each -> [3D]
                    sd[prefix + "attention.in_proj.bias"] = torch.cat([b.reshape(-1) for b in trio]).squeeze()
            else:
                raise KeyError(f"unrecognised attention entry {name!r}")
        else:
            raise KeyError(f"unrecognised block entry {name!r}")
    return sd


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="Convert a JAX/Flax ViT .npz to a PyTorch state_dict (.pth)")
    ap.add_argument("-s", "--source", required=True)
    ap.add_argument("-t", "--target", default=None)
    ap.add_argument("--fix-pre-logits", action="store_true")
    a = ap.parse_args()
    if not a.source.endswith(".npz"):
        raise ValueError("Invalid source file provided, it must end with .npz")
    torch.save(convert(np.load(a.source), a.fix_pre_logits), a.target or a.source.replace(".npz", ".pth"))
