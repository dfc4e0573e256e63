"""The CPU oracle against vectors captured from the reference itself (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo
from tests.helpers import GOLDEN, golden_inputs, load_golden

# fp32 CPU, same torch ops as the reference: differences are summation-order noise only
TOL = 2e-5


def test_tiny_model_every_stage():
    with open(os.path.join(GOLDEN, "tiny.json")) as f:
        kw = json.load(f)
    z = np.load(os.path.join(GOLDEN, "tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    cfg = dict(patches=kw["patches"], hidden_size=kw["hidden_size"], num_layers=kw["num_layers"],
               num_heads=kw["num_heads"], mlp_ratio=kw["mlp_ratio"], eps=kw["eps"])
    st = {}
    y = vo.forward(sd, torch.from_numpy(z["input"]), cfg, stages=st)
    assert np.abs(y.numpy() - z["logits"]).max() < TOL
    assert np.abs(st["embed"].numpy() - z["stage.embed"]).max() < TOL
    assert np.abs(st["block0"].numpy() - z["stage.block0"]).max() < TOL
    assert np.abs(st["encoded"].numpy() - z["stage.encoded"]).max() < TOL
    ctx = st["encoder.blocks.0.attention.ctx"]
    assert np.abs(ctx.numpy() - z["stage.ctx0"]).max() < TOL


@pytest.mark.parametrize("name", ["s16_224_ref", "b16_224_ref", "b16_224_pert", "b16_384_ref", "l16_224_ref"])
def test_full_size_logits(name):
    meta, g = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    st = {}
    y = vo.forward(sd, x, cfg, stages=st).numpy()
    assert y.shape == g["logits"].shape
    assert np.abs(y - g["logits"]).max() < TOL
    assert (y.argmax(-1) == g["top1"]).all()
    assert np.abs(st["embed"][:, :5, :32].numpy() - g["embed_slice"]).max() < TOL
    assert np.abs(st["block0"][:, -1, :].numpy() - g["block0_last"]).max() < TOL
    assert np.abs(st["encoded"].numpy() - g["encoded"]).max() < TOL


def test_bf16_emulation_within_north_star_tolerance():
    """bf16 operands + fp32 residual stream stays within 1e-2 of the reference logits and keeps top-1."""
    meta, g = load_golden("b16_224_ref")
    cfg, sd, x = golden_inputs(meta)
    y = vo.forward_bf16_emulated(sd, x[:4], cfg).numpy()
    assert np.abs(y - g["logits"][:4]).max() < 1e-2
    assert (y.argmax(-1) == g["top1"][:4]).all()


def test_flops_formula():
    from vit_amd import synth
    assert vo.flops_per_image(synth.ARCHS["b16"], 224) == 35_128_836_096
    assert vo.flops_per_image(synth.ARCHS["l16"], 224) == 123_111_522_304
    assert vo.flops_per_image(synth.ARCHS["b16"], 384) == 110_969_880_576
    assert vo.flops_per_image(synth.ARCHS["s16"], 224) == 19_778_949_120


def test_oracle_matches_reference_on_unpicked_pool_images():
    """The whole-pool fixtures (every image of the stream's head, no margin selection): the oracle reproduces
    the reference's logits on the first images of each, b32 / l32 / h14 included."""
    import json
    import os

    import numpy as np
    from tests.helpers import GOLDEN
    from vit_amd import synth
    for name, n in (("s16_224_pool", 3), ("b16_224_pool", 2), ("b32_224_pool", 2), ("l32_224_pool", 1), ("h14_224_pool", 1)):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            meta = json.load(f)
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        cfg = synth.ARCHS[meta["arch"]]
        sd = synth.synth_state_dict(cfg, meta["img_size"], seed=meta["seed"], style=meta["style"])
        x = synth.synth_images(n, meta["img_size"], seed=meta["image_seed"])
        y = vo.forward(sd, x, cfg).numpy()
        assert np.abs(y - g["logits"][:n]).max() <= 2e-5, name
