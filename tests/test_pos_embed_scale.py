"""pos_embed_scale (utils.py:241-261 of the reference): oracle against the vectors captured from the reference
function, and the GPU kernel against the same vectors."""
import numpy as np
import pytest
import torch

from oracle import pos_embed_oracle as po
from tests.helpers import load_golden


def _cases():
    meta, g = load_golden("pos_embed_scale")
    for i, c in enumerate(meta["cases"]):
        yield c, g[f"in{i}"], g[f"out{i}"]


def test_oracle_matches_reference_vectors():
    n = 0
    for c, x, want in _cases():
        got = po.pos_embed_scale(x, c["img_size"], c["patch_size"])
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-7
        n += 1
    assert n == 5


@pytest.mark.gpu
def test_gpu_matches_reference_vectors():
    import vit_amd
    for c, x, want in _cases():
        src = torch.from_numpy(x)
        got = vit_amd.pos_embed_scale(src, c["img_size"], c["patch_size"])            # CPU tensor in, CPU tensor out
        assert got.device == src.device and got.dtype == src.dtype
        assert np.abs(got.numpy() - want).max() <= 2e-7
        got_gpu = vit_amd.pos_embed_scale(src.cuda(), c["img_size"], c["patch_size"])
        assert got_gpu.is_cuda and torch.equal(got_gpu.cpu(), got)
    same = torch.zeros(1, 197, 8)
    assert vit_amd.pos_embed_scale(same, 224, 16) is same                              # lengths agree: untouched
    with pytest.raises(vit_amd.VitkError):
        vit_amd.pos_embed_scale(torch.zeros(1, 197, 8), 384, 16, order=3)


@pytest.mark.gpu
def test_checkpoint_at_224_loads_into_a_384_model():
    """The use the reference makes of it (utils.py:234-236): rescale, then a strict load."""
    import vit_amd
    from vit_amd import synth
    cfg = synth.ARCHS["b16"]
    sd = synth.synth_state_dict(cfg, 224, seed=0)
    with torch.device("meta"):
        m = vit_amd.b16(img_size=384)
    m = m.to_empty(device="cuda:0")
    sd = dict(sd)
    sd["pos_embed"] = vit_amd.pos_embed_scale(sd["pos_embed"], img_size=384, patch_size=16)
    m.load_state_dict(sd, strict=True)
    x = synth.synth_images(1, 384, seed=1).cuda()
    with torch.no_grad():
        y = m.eval()(x)
    assert y.shape == (1, 1000) and torch.isfinite(y).all()


def test_adapt_state_dict_head_reset_on_cpu():
    """load_checkpoint's --pretrained edits (utils.py:224-233); same resolution, so no resampling (and no GPU)."""
    import vit_amd
    sd = {"pos_embed": torch.zeros(1, 197, 8), "head.weight": torch.ones(21843, 8), "head.bias": torch.ones(21843),
          "pre_logits.weight": torch.ones(8, 8), "pre_logits.bias": torch.ones(8), "cls_token": torch.zeros(1, 1, 8)}
    pe = sd["pos_embed"]
    out = vit_amd.adapt_state_dict({"state_dict": sd, "epoch": 3}, img_size=224, num_classes=10, pretrained=True)
    assert out is sd and out["pos_embed"] is pe
    assert out["head.weight"].shape == (10, 8) and not out["head.weight"].any() and out["head.bias"].shape == (10,)
    assert "pre_logits.weight" not in out and "pre_logits.bias" not in out
    sd2 = {"pos_embed": pe, "head.weight": torch.ones(5, 8), "head.bias": torch.ones(5), "pre_logits.weight": torch.ones(8, 8)}
    out2 = vit_amd.adapt_state_dict(sd2, img_size=224, num_classes=10)
    assert out2["head.weight"].shape == (5, 8) and "pre_logits.weight" in out2          # untouched without pretrained


@pytest.mark.gpu
def test_adapt_state_dict_fine_tune_start_at_384():
    import vit_amd
    from vit_amd import synth
    sd = dict(synth.synth_state_dict(synth.ARCHS["s16"], 224, seed=0))
    sd = vit_amd.adapt_state_dict({"state_dict": sd}, img_size=384, num_classes=10, pretrained=True)
    m = vit_amd.s16(img_size=384, num_classes=10, pre_logits=False).cuda().eval()
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        y = m(synth.synth_images(2, 384, seed=1).cuda())
    assert y.shape == (2, 10) and not y.any()                                             # zero head: zero logits
