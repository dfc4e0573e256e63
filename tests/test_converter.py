"""converter contract: same keys, shapes and values as the reference's convert() on a Flax-style npz."""
import json
import os

import numpy as np
import torch

import vit_amd
from vit_amd import converter
from tests.helpers import GOLDEN


def _fixture():
    src = np.load(os.path.join(GOLDEN, "flax_tiny.npz"))
    want = np.load(os.path.join(GOLDEN, "flax_tiny_converted.npz"))
    with open(os.path.join(GOLDEN, "flax_tiny_converted.json")) as f:
        order = json.load(f)
    return src, want, order


def test_convert_matches_reference_output():
    src, want, order = _fixture()
    got = converter.convert(src)
    assert [[k, list(v.shape)] for k, v in got.items()] == order
    for k, v in got.items():
        assert np.array_equal(v.numpy(), want[k]), k


def test_converted_checkpoint_loads_strict():
    src, _, _ = _fixture()
    sd = converter.convert(src)
    m = vit_amd.VisionTransformer(img_size=32, patches=8, hidden_size=32, num_layers=2, num_heads=2,
                                  num_classes=10)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # q;k;v stacking: rows [0,D) of in_proj are the query projection, output index = h*hd + e
    q = src["Transformer/encoderblock_0/MultiHeadDotProductAttention_1/query/kernel"]   # [D,H,hd]
    w = sd["encoder.blocks.0.attention.in_proj.weight"].numpy()
    assert np.array_equal(w[:32], q.reshape(32, 32).T)
    conv = src["embedding/kernel"]                                                      # [P,P,3,D]
    assert np.array_equal(sd["embedding.weight"].numpy(), conv.transpose(3, 2, 0, 1))


def test_pre_logits_quirk_and_fix():
    src, _, _ = _fixture()
    raw = converter.convert(src)["pre_logits.weight"].numpy()
    fixed = converter.convert(src, fix_pre_logits=True)["pre_logits.weight"].numpy()
    assert np.array_equal(raw, src["pre_logits/kernel"])          # reference copies it untransposed
    assert np.array_equal(fixed, src["pre_logits/kernel"].T)
