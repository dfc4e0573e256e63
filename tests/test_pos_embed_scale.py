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
