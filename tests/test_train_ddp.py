"""Data-parallel training step on more than one rank (BASELINE configs[4]; the reference wraps its model in
DistributedDataParallel, main.py:125-126), and one full-size step."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_two_rank_ddp_gradients_equal_one_rank_on_the_whole_batch(precision, tol):
    """Two fresh processes share the GPU over gloo: the averaged gradients of 2 ranks x 4 images equal the
    gradients of 1 rank x 8 images.  fp32 mode: to summation order; bf16 mode: to bf16 rounding (the per-rank
    GEMMs see different row counts, so roundings differ)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "ddp_worker.py"), precision], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["world"] == 2
    assert abs(res["loss_ddp"] - res["loss_one"]) <= max(tol, 1e-5)
    assert res["worst_rel"] <= tol, res


@pytest.mark.gpu
def test_full_size_b16_224_batch_256_step_is_finite_and_close_to_the_fp32_mode():
    """One ViT-B/16 224px batch-256 training step (forward, CrossEntropyLoss, backward, clip_grad_norm_ 1.0:
    main.py:207-223) in the bf16 mode: finite loss and gradients, and the total gradient norm within 5 % of the
    same step in the fp32 mode (size-independent property; the per-gradient vectors are pinned at small size)."""
    import vit_amd
    from vit_amd import synth
    cfg = synth.ARCHS["b16"]
    sd = synth.synth_state_dict(cfg, 224, seed=0)
    x = synth.synth_images(32, 224, seed=1).repeat(8, 1, 1, 1).cuda()
    target = torch.randint(0, 1000, (256,), generator=torch.Generator().manual_seed(0)).cuda()
    crit = torch.nn.CrossEntropyLoss()
    res = {}
    for precision in ("bf16", "fp32"):
        with torch.device("meta"):
            m = vit_amd.b16(img_size=224, precision=precision)
        m = m.to_empty(device="cuda")
        m.load_state_dict(sd, strict=True)
        m.train()
        loss = crit(m(x), target)
        loss.backward()
        grads = [p.grad for p in m.parameters()]
        assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads)
        total = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0))
        res[precision] = (float(loss.detach()), total)
        del m, loss, grads
        torch.cuda.empty_cache()
    (l16, n16), (l32, n32) = res["bf16"], res["fp32"]
    print(f"loss bf16 {l16:.5f} fp32 {l32:.5f}; grad norm bf16 {n16:.5f} fp32 {n32:.5f}")
    assert abs(l16 - l32) <= 1e-2
    assert abs(n16 - n32) <= 0.05 * n32
