"""The steps either side of the forward: uint8 input normalisation (data.py:54-55,69-72 of the reference) and
accuracy (utils.py:182-195).  CPU tests pin the oracle (accuracy: vectors captured from the reference function;
normalisation: the same three torch operations, no reference capture exists); GPU tests compare the kernels with
the oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import io_oracle as io
from tests.helpers import load_golden


def _acc_cases():
    meta, g = load_golden("accuracy")
    for i, c in enumerate(meta["cases"]):
        yield c, g[f"logits{i}"], g[f"target{i}"], g[f"res{i}"]


def _u8(shape, seed):
    return np.random.RandomState(seed).randint(0, 256, size=shape, dtype=np.uint8)


def test_accuracy_oracle_matches_reference_vectors():
    n = 0
    for c, logits, target, want in _acc_cases():
        got = io.accuracy(logits, target, tuple(c["topk"]))
        assert np.allclose(got, want, rtol=0, atol=1e-4), (got, want)
        rank = io.target_rank(logits, target)
        for k, w in zip(c["topk"], want):
            assert abs((rank < k).sum() * 100.0 / len(target) - w) <= 1e-4
        n += 1
    assert n == 4


def test_accuracy_oracle_edge_cases():
    logits = np.array([[1., 3., 3., 0.], [2., 2., 2., 2.], [0., 1., 2., 3.]], dtype=np.float32)
    assert io.target_rank(logits, [2, 3, 7]).tolist() == [1, 3, 4]           # ties towards the lower index; 7 is no class
    assert io.accuracy(logits, np.array([1, 0, 3]), (1, 2)) == [100.0, 100.0]
    assert io.accuracy(logits, np.array([2, 1, 0]), (1, 2)) == [0.0, pytest.approx(200.0 / 3)]


def test_normalize_oracle_is_the_reference_operations():
    x = _u8((2, 3, 8, 16), 0)
    mean = torch.tensor([0.485 * 255, 0.456 * 255, 0.406 * 255]).view(1, 3, 1, 1)      # data.py:54-55
    std = torch.tensor([0.229 * 255, 0.224 * 255, 0.225 * 255]).view(1, 3, 1, 1)
    want = torch.from_numpy(x).float().sub_(mean).div_(std)                              # data.py:69-72
    assert np.array_equal(io.normalize_u8(x), want.numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 4, 4), (3, 3, 32, 32), (5, 3, 224, 224), (2, 3, 384, 384), (2, 3, 20, 12)])
def test_gpu_normalize_matches_oracle(shape):
    import vit_amd
    x = _u8(shape, sum(shape))
    want = io.normalize_u8(x)
    got = vit_amd.normalize_u8(torch.from_numpy(x).cuda())
    assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), want)         # bit-exact in f32
    got16 = vit_amd.normalize_u8(torch.from_numpy(x).cuda(), dtype=torch.bfloat16)
    assert torch.equal(got16.cpu(), torch.from_numpy(want).to(torch.bfloat16))            # and after RNE rounding
    mean, std = (10.0, 20.5, 127.5), (1.0, 64.0, 127.5)
    got = vit_amd.normalize_u8(torch.from_numpy(x).cuda(), mean=mean, std=std)
    assert np.array_equal(got.cpu().numpy(), io.normalize_u8(x, mean, std))


@pytest.mark.gpu
def test_gpu_normalize_any_image_size():
    """Planes that are not a multiple of 16 pixels take the per-pixel kernel: same values."""
    import vit_amd
    from oracle import io_oracle as io
    rng = np.random.default_rng(5)
    for shape in ((1, 3, 5, 5), (2, 3, 87, 87), (3, 3, 13, 22)):
        x = rng.integers(0, 256, shape, dtype=np.uint8)
        got = vit_amd.normalize_u8(torch.from_numpy(x).cuda())
        assert np.array_equal(got.cpu().numpy(), io.normalize_u8(x))


@pytest.mark.gpu
def test_gpu_normalize_rejects_bad_input():
    import vit_amd
    with pytest.raises(vit_amd.VitkError):
        vit_amd.normalize_u8(torch.zeros(1, 3, 8, 8).cuda())                                # not uint8
    with pytest.raises(vit_amd.VitkError):
        vit_amd.normalize_u8(torch.zeros(1, 3, 8, 8, dtype=torch.uint8).cuda(), std=(1., 0., 1.))
    with pytest.raises(vit_amd.VitkError):
        vit_amd.normalize_u8(torch.zeros(1, 3, 8, 8, dtype=torch.uint8))                   # CPU tensor


@pytest.mark.gpu
def test_gpu_accuracy_matches_reference_vectors():
    import vit_amd
    from vit_amd import ops
    for c, logits, target, want in _acc_cases():
        lg, tg = torch.from_numpy(logits).cuda(), torch.from_numpy(target).cuda()
        rank = ops.target_rank(lg, tg)
        assert np.array_equal(rank.cpu().numpy(), io.target_rank(logits, target))
        res = vit_amd.accuracy(lg, tg, topk=tuple(c["topk"]))
        assert all(r.shape == (1,) and r.is_cuda for r in res)
        assert np.allclose([float(r) for r in res], want, rtol=0, atol=1e-4)


@pytest.mark.gpu
def test_gpu_accuracy_ties_and_bad_targets():
    from vit_amd import ops
    logits = torch.tensor([[1., 3., 3., 0.], [2., 2., 2., 2.], [0., 1., 2., 3.], [0., 1., 2., 3.]]).cuda()
    target = torch.tensor([2, 3, 7, -1]).cuda()
    assert ops.target_rank(logits, target).tolist() == [1, 3, 4, 4]
    big = torch.randn(1000, 1000, generator=torch.Generator().manual_seed(0))
    tgt = torch.randint(0, 1000, (1000,), generator=torch.Generator().manual_seed(1))
    assert np.array_equal(ops.target_rank(big.cuda(), tgt.cuda()).cpu().numpy(), io.target_rank(big.numpy(), tgt.numpy()))


@pytest.mark.gpu
def test_model_takes_uint8_images():
    """uint8 images through the model = the DataFetcher's normalisation followed by the forward."""
    import vit_amd
    from vit_amd import synth
    m = vit_amd.s16(img_size=224).cuda().eval()
    m.load_state_dict(synth.synth_state_dict(synth.ARCHS["s16"], 224, seed=0))
    x8 = torch.from_numpy(_u8((3, 3, 224, 224), 5)).cuda()
    with torch.no_grad():
        got = m(x8)
        want = m(vit_amd.normalize_u8(x8))                      # f32 normalised, then the model's own bf16 cast
        assert torch.equal(got, want)
        m.set_precision("fp32")
        got32 = m(x8)
        want32 = m(vit_amd.normalize_u8(x8))
        assert torch.equal(got32, want32)
        assert (got32 - got).abs().max() < 2e-2
