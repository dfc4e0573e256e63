"""vitk_colsum_bf16_static, the entry point that allocates nothing (its partial sums live in a device buffer
built into the library), against vitk_colsum_bf16 with a caller workspace: same kernels, same summation
order, so the results must be bit-identical.  Two calls in a row, the second on a smaller input, since the
built-in buffer keeps the first call's contents."""
import ctypes

import pytest
import torch

from vit_amd import lib, ops

pytestmark = pytest.mark.gpu


def _g(rows, cols, seed):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(rows, cols, generator=gen).to(torch.bfloat16).cuda()


def test_static_scratch_matches_workspace_entry_point_two_calls_in_a_row():
    L = lib.load()
    assert L.vitk_init() == 0
    big, small = _g(50432, 3072, 1), _g(1000, 768, 2)
    out_big = torch.empty(3072, dtype=torch.float32, device="cuda")
    out_small = torch.empty(768, dtype=torch.float32, device="cuda")
    want_big, want_small = ops.colsum(big), ops.colsum(small)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    ops.colsum_static(big, out_big)
    ops.colsum_static(small, out_small)          # smaller input right after: stale partial rows must not leak in
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.equal(out_big, want_big)
    assert torch.equal(out_small, want_small)
    ref = small.double().sum(0)
    assert torch.allclose(out_small.double(), ref, rtol=1e-6, atol=1e-4)
    ops.colsum_static(big, out_big)              # and back to the large shape
    assert torch.equal(out_big, want_big)


def test_static_scratch_rejects_sizes_beyond_its_maximum():
    L = lib.load()
    g = _g(64, 4352, 3)
    out = torch.empty(4352, dtype=torch.float32, device="cuda")
    rc = L.vitk_colsum_bf16_static(ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(out.data_ptr()), 64, 4352, None)
    assert rc == 2 and b"exceeds" in L.vitk_last_error()
    with pytest.raises(lib.VitkError):
        ops.colsum_static(g, out)
