"""One training step (main.py:207-223 of the reference: forward, CrossEntropyLoss, backward, clip_grad_norm_,
optimizer step).  CPU: the oracle against loss / logits / gradients captured from the reference model class
under torch autograd.  GPU: the model's autograd path (forward kernels + torch backward) against the same
vectors, near-exact in the fp32 mode and to bf16 accuracy in the default mode."""
import numpy as np
import pytest
import torch

from oracle import train_oracle as to
from tests.helpers import load_golden
from vit_amd import synth


def _setup():
    meta, g = load_golden("train_small")
    sd = synth.synth_state_dict(to.SMALL, to.SMALL_IMG, seed=meta["seed"], style="perturbed", num_classes=to.SMALL_CLASSES)
    x = synth.synth_images(to.SMALL_BATCH, to.SMALL_IMG, seed=meta["image_seed"])
    return meta, g, sd, x, torch.from_numpy(g["target"])


def _check_grads(grads, g, stride, rel_tol, cos_min):
    worst = 0.0
    for k, want_sub in ((k[2:], v) for k, v in g.items() if k.startswith("g.")):
        got = grads[k].detach().float().cpu()
        want_norm = float(g["gnorm." + k])
        got_sub = got.reshape(-1)[::stride].numpy()
        err = np.linalg.norm(got_sub - want_sub) / max(np.linalg.norm(want_sub), 1e-12)
        cos = float(np.dot(got_sub, want_sub) / max(np.linalg.norm(got_sub) * np.linalg.norm(want_sub), 1e-20))
        assert err <= rel_tol and cos >= cos_min, f"{k}: relative error {err:.3e}, cosine {cos:.6f}"
        assert abs(float(got.norm()) - want_norm) <= rel_tol * want_norm + 1e-12, f"{k}: norm"
        worst = max(worst, err)
    return worst


def test_oracle_training_step_matches_reference_vectors():
    meta, g, sd, x, target = _setup()
    loss, logits, grads = to.loss_and_grads(sd, x, target, to.SMALL)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert np.abs(logits.numpy() - g["logits"]).max() <= 2e-5
    assert len(grads) == meta["n_params"]
    _check_grads(grads, g, meta["stride"], 1e-4, 0.999999)
    coef, total = to.clip_coefficient(grads, 1.0)
    assert abs(float(total) - float(g["total_norm"])) <= 1e-4 * float(g["total_norm"])
    assert 0.0 < float(coef) < 1.0                                        # the fixture's step is clipped


def _model(precision, **extra):
    import vit_amd
    cfg = to.SMALL
    m = vit_amd.VisionTransformer(img_size=to.SMALL_IMG, patches=cfg["patches"], num_classes=to.SMALL_CLASSES,
                                  hidden_size=cfg["hidden_size"], num_layers=cfg["num_layers"],
                                  num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"],
                                  norm=lambda d: torch.nn.LayerNorm(d, eps=cfg["eps"]), precision=precision, **extra)
    return m.cuda().train()


def test_oracle_dropout_forward_matches_reference_vectors():
    """Where the reference's Dropout layers sit and the order they draw their masks in: the oracle's training-mode
    forward, seeded like the reference run that wrote the fixture, reproduces its loss, logits and gradients."""
    meta, g = load_golden("train_dropout")
    _, g0, sd, x, target = _setup()
    torch.manual_seed(meta["torch_seed"])
    loss, logits, grads = to.loss_and_grads(sd, x, target, to.SMALL, dropout=(meta["dropout"], meta["attn_dropout"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert np.abs(logits.numpy() - g["logits"]).max() <= 2e-5
    assert np.abs(logits.numpy() - g0["logits"]).max() > 1e-3             # and the masks did something
    _check_grads(grads, g, meta["stride"], 1e-4, 0.999999)


@pytest.mark.gpu
@pytest.mark.parametrize("attn_on", [True, False])
@pytest.mark.parametrize("precision,logit_tol,rel_tol,cos_min", [("fp32", 5e-5, 5e-4, 0.99999), ("bf16", 2e-2, 6e-2, 0.998)])
def test_gpu_training_with_dropout_matches_the_oracle_under_the_same_seed(precision, logit_tol, rel_tol, cos_min, attn_on):
    """dropout / attn_dropout > 0 in training mode (models/transformer.py:23,25,45,70, vision_transformer.py:51):
    the model draws its masks from torch's generator in the reference's order, so under one seed on one device it
    and the oracle's training-mode forward (itself pinned by the reference's vectors on the CPU) see the same masks."""
    meta, _ = load_golden("train_dropout")
    _, g0, sd, x, target = _setup()
    p, pa = meta["dropout"], meta["attn_dropout"] if attn_on else 0.0     # without attn_dropout: the attention kernels
    m = _model(precision, dropout=p, attn_dropout=pa)
    m.load_state_dict(sd, strict=True)
    xs, ts = x.cuda(), target.cuda()
    sd_gpu = {k: v.cuda() for k, v in sd.items()}
    torch.manual_seed(123)
    want_loss, want_logits, want_grads = to.loss_and_grads(sd_gpu, xs, ts, to.SMALL, dropout=(p, pa))
    torch.manual_seed(123)
    logits = m(xs)
    loss = torch.nn.CrossEntropyLoss()(logits, ts)
    loss.backward()
    err = (logits.detach() - want_logits).abs().max().item()
    assert err <= logit_tol, f"logits {err:.3e}"
    assert abs(float(loss) - float(want_loss)) <= logit_tol
    for k, prm in m.named_parameters():
        got, want = prm.grad.float().reshape(-1), want_grads[k].reshape(-1)
        rel = float((got - want).norm() / want.norm().clamp_min(1e-12))
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()).clamp_min(1e-20))
        assert rel <= rel_tol and cos >= cos_min, f"{k}: relative error {rel:.3e}, cosine {cos:.6f}"
    torch.manual_seed(123)
    with torch.no_grad():
        assert torch.equal(m(xs), logits.detach())                        # same seed, same masks, with or without a graph
        assert not torch.equal(m(xs), logits.detach())                    # the generator has moved on
    m.eval()
    with torch.no_grad():                                                  # eval mode: no masks, the inference path
        e = m(xs).cpu().numpy()
    assert np.abs(e - g0["logits"]).max() <= (2e-5 if precision == "fp32" else 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,logit_tol,rel_tol,cos_min", [("fp32", 2e-5, 2e-4, 0.99999), ("bf16", 1e-2, 4e-2, 0.999)])
def test_gpu_gradients_match_reference_vectors(precision, logit_tol, rel_tol, cos_min):
    meta, g, sd, x, target = _setup()
    m = _model(precision)
    m.load_state_dict(sd, strict=True)
    logits = m(x.cuda())
    assert logits.requires_grad and logits.dtype == torch.float32
    loss = torch.nn.CrossEntropyLoss()(logits, target.cuda())
    loss.backward()
    err = np.abs(logits.detach().cpu().numpy() - g["logits"]).max()
    assert err <= logit_tol, f"logits {err:.3e}"
    assert abs(float(loss) - float(g["loss"])) <= logit_tol
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert all(v is not None for v in grads.values())
    worst = _check_grads(grads, g, meta["stride"], rel_tol, cos_min)
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    assert abs(float(total) - float(g["total_norm"])) <= rel_tol * float(g["total_norm"])
    print(f"{precision}: logits {err:.3e}, worst relative gradient error {worst:.3e}")
    with torch.no_grad():     # the inference path with one launch per LayerNorm gives the same logits, bit for bit;
        m.fold_layernorm = False      # the default one (LayerNorms folded into the GEMMs around them) within the bar
        assert torch.equal(m(x.cuda()), logits.detach())
        m.fold_layernorm = True
        assert (m(x.cuda()) - logits.detach()).abs().max().item() <= logit_tol


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,heads", [(1280, 16), (768, 8)])
def test_gpu_gradients_at_head_dim_80_and_96_use_the_attention_backward_kernel(hidden, heads, monkeypatch):
    """ViT-H/14 has 80 columns per head (1280 / 16): its training step takes the same attention kernels, forward with
    the row log-sum-exp and the two-pass backward, and the gradients agree with the oracle's autograd on the same
    weights and images to the bf16 bar of the fixture test above."""
    import vit_amd
    from vit_amd import ops
    cfg = dict(patches=8, hidden_size=hidden, num_layers=2, num_heads=heads, mlp_ratio=4, eps=1e-6)
    sd = synth.synth_state_dict(cfg, 32, seed=5, style="perturbed", num_classes=10)
    x = synth.synth_images(6, 32, seed=9).cuda()
    target = torch.tensor([1, 7, 3, 0, 9, 4], device="cuda")
    calls = []
    real = ops.attention_backward
    monkeypatch.setattr(ops, "attention_backward", lambda *a: (calls.append(a[0].shape), real(*a))[1])
    m = vit_amd.VisionTransformer(img_size=32, patches=8, num_classes=10, hidden_size=hidden, num_layers=2, num_heads=heads,
                                  mlp_ratio=4, norm=lambda d: torch.nn.LayerNorm(d, eps=1e-6), precision="bf16").cuda().train()
    m.load_state_dict(sd, strict=True)
    logits = m(x)
    loss = torch.nn.CrossEntropyLoss()(logits, target)
    loss.backward()
    assert len(calls) == 2 and calls[0][-1] == hidden // heads, calls     # one launch per block, through the kernel
    want_loss, want_logits, want_grads = to.loss_and_grads({k: v.cuda() for k, v in sd.items()}, x, target, cfg)
    assert (logits.detach() - want_logits).abs().max().item() <= 1e-2
    assert abs(float(loss) - float(want_loss)) <= 1e-2
    for k, prm in m.named_parameters():
        got, want = prm.grad.float().reshape(-1), want_grads[k].reshape(-1)
        rel = float((got - want).norm() / want.norm().clamp_min(1e-12))
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()).clamp_min(1e-20))
        assert rel <= 4e-2 and cos >= 0.999, f"{k}: relative error {rel:.3e}, cosine {cos:.6f}"


@pytest.mark.gpu
def test_gpu_training_steps_reduce_the_loss_and_refresh_the_weight_cache():
    meta, g, sd, x, target = _setup()
    m = _model("bf16")
    m.load_state_dict(sd, strict=True)
    opt = torch.optim.AdamW(m.parameters(), 1e-3, weight_decay=0.03)      # run.py:55-60
    crit = torch.nn.CrossEntropyLoss()
    xs, ts = x.cuda(), target.cuda()
    losses = []
    for _ in range(6):
        loss = crit(m(xs), ts)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert abs(losses[0] - float(g["loss"])) <= 1e-2
    assert losses[-1] < losses[0] - 0.5, losses                            # six steps on one batch: it is being fitted
    m.eval()
    with torch.no_grad():
        assert abs(float(crit(m(xs), ts)) - losses[-1]) < 1.0               # kernels see the updated weights


@pytest.mark.gpu
def test_gpu_uint8_images():
    m = _model("bf16")
    x8 = torch.randint(0, 256, (2, 3, to.SMALL_IMG, to.SMALL_IMG), dtype=torch.uint8).cuda()
    y = m(x8)
    y.sum().backward()
    assert m.head.weight.grad is not None and torch.isfinite(m.embedding.weight.grad).all()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,D", [(1, 256), (7, 768), (1000, 768), (4099, 1024), (50432, 768), (33, 1280)])
@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16])
def test_gpu_layernorm_backward_kernel(rows, D, gdt):
    """vitk_layernorm_backward against torch autograd of F.layer_norm in fp64 on the same inputs."""
    from vit_amd import ops
    gen = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=gen) * 1.7 + 0.3
    gamma = 1.0 + 0.1 * torch.randn(D, generator=gen)
    beta = 0.1 * torch.randn(D, generator=gen)
    g = torch.randn(rows, D, generator=gen).to(gdt)
    xd, gd_, bd = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    y = torch.nn.functional.layer_norm(xd, (D,), gd_, bd, 1e-6)
    wx, wg, wb = torch.autograd.grad(y, (xd, gd_, bd), g.double())
    dx, dg, db = ops.layernorm_backward(x.cuda(), gamma.cuda(), g.cuda(), 1e-6)
    assert torch.allclose(dx.cpu().double(), wx, rtol=1e-4, atol=2e-5)
    scale = max(1.0, rows ** 0.5)
    assert (dg.cpu().double() - wg).abs().max() <= 2e-5 * scale
    assert (db.cpu().double() - wb).abs().max() <= 2e-5 * scale
    dx2, dg2, db2 = ops.layernorm_backward(x.cuda(), gamma.cuda(), g.cuda(), 1e-6)
    assert torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(dx, dx2)      # fixed summation order
    skip = torch.randn(rows, D, generator=gen)
    dx3, _, _ = ops.layernorm_backward(x.cuda(), gamma.cuda(), g.cuda(), 1e-6, add=skip.cuda())
    assert torch.allclose(dx3.cpu().double(), wx + skip.double(), rtol=1e-4, atol=2e-5)  # residual-branch gradient added
    d16 = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    dx4, _, _ = ops.layernorm_backward(x.cuda(), gamma.cuda(), g.cuda(), 1e-6, add=skip.cuda(), dx_bf16=d16)
    assert torch.equal(dx4, dx3) and torch.equal(d16, dx3.to(torch.bfloat16))             # the bf16 copy of the same values


@pytest.mark.gpu
@pytest.mark.parametrize("R,M,N", [(64, 256, 256), (256, 256, 512), (4096, 768, 256), (12608, 768, 768),
                                   (50432, 3072, 768), (50432, 768, 3072), (50432, 2304, 768), (50432, 768, 768),
                                   # row counts that do not fill the last 64-row tile (batches of 5 and 3 images at 197
                                   # tokens, single rows, one row more / fewer than a tile)
                                   (985, 768, 768), (591, 2304, 768), (1, 256, 256), (63, 256, 256), (65, 256, 512),
                                   (100, 256, 256), (50432 + 37, 768, 768), (12607, 3072, 768)])
def test_gpu_weight_gradient_gemm(R, M, N):
    """vitk_gemm_tn (dW = dY^T X with the k range split over workgroups) against an fp64 product of the same
    bf16 operands; reproducible run to run.  The operands are views of larger buffers whose other rows hold NaN:
    a kernel that read a row past the end of either operand would show it."""
    from vit_amd import ops
    gen = torch.Generator().manual_seed(R + M + N)
    g = (torch.randn(R, M, generator=gen) * 0.5).to(torch.bfloat16)
    x = torch.randn(R, N, generator=gen).to(torch.bfloat16)
    gbig = torch.full((R + 70, M), float("nan"), dtype=torch.bfloat16, device="cuda")
    xbig = torch.full((R + 70, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    gbig[:R], xbig[:R] = g.cuda(), x.cuda()
    gd, xd = gbig[:R], xbig[:R]
    want = (gd.double().t() @ xd.double()).cpu()
    got = ops.gemm_tn(gd, xd)
    err = (got.cpu().double() - want).abs().max()
    assert err <= 1e-5 * (R ** 0.5) * 4 + 1e-6, f"max abs error {err:.3e}"
    assert torch.equal(got, ops.gemm_tn(gd, xd))
    assert not ops.gemm_tn_supported(1000, 768, R) and ops.gemm_tn_supported(768, 768, 100)
    # the same launch with the bias gradient (column sums of g) fused in: identical product, sums vs fp64
    cs = torch.full((M,), float("nan"), device="cuda")
    got2 = ops.gemm_tn(gd, xd, colsum=cs)
    assert torch.equal(got2, got)
    cerr = (cs.cpu().double() - g.double().sum(0)).abs().max()
    assert cerr <= 1e-5 * (R ** 0.5) * 4 + 1e-6, f"column sums: max abs error {cerr:.3e}"
    cs2 = torch.empty_like(cs)
    ops.gemm_tn(gd, xd, colsum=cs2)
    assert torch.equal(cs, cs2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,N,hd", [(2, 3, 197, 64), (1, 2, 32, 64), (1, 2, 33, 64), (2, 2, 64, 64), (1, 4, 257, 64), (1, 1, 576, 64),
                                      (3, 12, 197, 64), (1, 2, 577, 64), (2, 1, 608, 64),
                                      # head_dim 80 (ViT-H/14: 257 tokens) and 96: the two-pass kernels, up to 288 tokens
                                      (2, 3, 257, 80), (1, 2, 33, 80), (2, 2, 288, 80), (1, 5, 1, 80), (300, 1, 50, 80),
                                      (2, 3, 197, 96), (1, 2, 64, 96), (1, 2, 288, 96), (1, 3, 17, 96)])
def test_gpu_attention_backward_kernel(B, H, N, hd):
    """vitk_attention_lse + vitk_attention_backward against torch autograd of softmax(q k^T s) v in fp64 on the
    same bf16 q, k, v and output gradient."""
    from vit_amd import ops
    scale = hd ** -0.5
    gen = torch.Generator().manual_seed(B * 1000 + N)
    qkv = torch.randn(3, B, H, N, hd, generator=gen).to(torch.bfloat16)
    dctx = torch.randn(B, N, H * hd, generator=gen).to(torch.bfloat16)
    q, k, v = (qkv[i].double().requires_grad_(True) for i in range(3))
    o = ((q @ k.transpose(-2, -1)) * scale).softmax(-1) @ v
    o2 = o.transpose(1, 2).reshape(B, N, H * hd)
    gq, gk, gv = torch.autograd.grad(o2, (q, k, v), dctx.double())
    want = torch.stack((gq, gk, gv)).permute(1, 3, 0, 2, 4)                 # [B, N, 3, H, hd]
    qd = qkv.cuda()
    ctx = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device="cuda")
    lse = ops.attention_lse(qd, ctx, scale)
    s2 = (q.detach() @ k.detach().transpose(-2, -1)) * scale * 1.4426950408889634
    want_lse = torch.logsumexp(s2 * 0.6931471805599453, -1) * 1.4426950408889634
    # the row sums add the bf16-rounded probabilities P.V uses (relative error up to 2^-9 each): 2^-9 log2(e) = 2.8e-3
    assert torch.allclose(lse.cpu().double(), want_lse, atol=4e-3)
    assert torch.allclose(ctx.float().cpu().double(), o2.detach(), rtol=2e-2, atol=6e-3)
    got = ops.attention_backward(qd, ctx, dctx.cuda(), lse, scale)
    assert got.shape == (B, N, 3, H, hd)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, ops.attention_backward(qd, ctx, dctx.cuda(), lse, scale))
    err = (got.float().cpu().double() - want).abs().max()
    ref = want.abs().max()
    assert err <= 2e-2 * ref + 1e-3, f"max abs error {err:.3e} (gradient magnitude {ref:.3e})"
    for w in range(3):
        a, bb = got[:, :, w].float().cpu().double().reshape(-1), want[:, :, w].reshape(-1)
        if float(bb.norm()) < 1e-9:                          # one token: dQ and dK vanish identically
            assert float(a.abs().max()) <= 1e-3, w
            continue
        cos = float(a @ bb / (a.norm() * bb.norm()))
        assert cos > 0.9995, (w, cos)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,N", [(60, 12, 197), (70, 8, 33), (40, 8, 288), (36, 8, 257), (20, 16, 289), (300, 1, 1)])
def test_gpu_attention_backward_more_heads_than_compute_units(B, H, N):
    """More (image, head) pairs than CUs (several rounds of workgroups), token counts either side of a block edge:
    against fp64 autograd on the GPU, repeatable, and every image's result equal to the same image computed in a
    launch of its own."""
    from vit_amd import ops
    hd, scale = 64, 64 ** -0.5
    gen = torch.Generator(device="cuda").manual_seed(B * 1000 + N)
    qkv = torch.randn(3, B, H, N, hd, generator=gen, device="cuda").to(torch.bfloat16)
    dctx = torch.randn(B, N, H * hd, generator=gen, device="cuda").to(torch.bfloat16)
    q, k, v = (qkv[i].double().requires_grad_(True) for i in range(3))
    o2 = (((q @ k.transpose(-2, -1)) * scale).softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * hd)
    gq, gk, gv = torch.autograd.grad(o2, (q, k, v), dctx.double())
    want = torch.stack((gq, gk, gv)).permute(1, 3, 0, 2, 4)                 # [B, N, 3, H, hd]
    ctx = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device="cuda")
    lse = ops.attention_lse(qkv, ctx, scale)
    got = ops.attention_backward(qkv, ctx, dctx, lse, scale)
    err = (got.double() - want).abs().amax(dim=(1, 2, 4))                    # per (image, head)
    ref = want.abs().max()
    assert float(err.max()) <= 2e-2 * float(ref) + 1e-3, f"worst head {err.max():.3e} (gradient magnitude {ref:.3e})"
    assert torch.equal(got, ops.attention_backward(qkv, ctx, dctx, lse, scale))
    # up to 288 tokens one kernel does both passes from one LDS image of Q, K, V, dO; the two-kernel form (larger token
    # counts; here forced by the debug variant) computes the same sums in the same order
    from vit_amd import lib
    try:
        lib.load().vitk_set_debug_variant(0, 0x200)
        assert torch.equal(got, ops.attention_backward(qkv, ctx, dctx, lse, scale))
    finally:
        lib.load().vitk_set_debug_variant(0, 0)
    for b in (0, B // 2, B - 1):                                            # a launch with few heads: one head per workgroup
        alone = ops.attention_backward(qkv[:, b:b + 1].contiguous(), ctx[b:b + 1].contiguous(), dctx[b:b + 1].contiguous(),
                                       lse[b:b + 1].contiguous(), scale)
        assert torch.equal(alone[0], got[b]), b


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(300, 512, 128), (50432, 3072, 768), (8200, 1024, 192)])
def test_gpu_gelu_gemm_with_pre_activation_output(M, N, K):
    """vitk_gemm_bias_gelu_dual: out equals vitk_gemm_bias_gelu's bit for bit, pre equals vitk_gemm_bias's."""
    from vit_amd import ops
    gen = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=gen).cuda()
    want_out, want_pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_bias_gelu(a, w, b, want_out)
    ops.gemm_bias(a, w, b, want_pre)
    out, pre = torch.full_like(want_out, float("nan")), torch.full_like(want_out, float("nan"))
    ops.gemm_bias_gelu_dual(a, w, b, pre, out)
    assert torch.equal(out, want_out) and torch.equal(pre, want_pre)


@pytest.mark.gpu
@pytest.mark.parametrize("R,M", [(1, 256), (102, 768), (5000, 1024), (50432, 2304), (50432, 3072)])
def test_gpu_bias_gradient_column_sum(R, M):
    from vit_amd import ops
    g = torch.randn(R, M, generator=torch.Generator().manual_seed(R + M)).to(torch.bfloat16).cuda()
    got = ops.colsum(g)
    want = g.double().sum(0)
    assert (got.double() - want).abs().max() <= 1e-5 * max(1.0, R ** 0.5) * 4
    assert torch.equal(got, ops.colsum(g))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 768, 1024), (50432, 3072, 768), (50432, 768, 2304), (300, 512, 192), (100, 320, 96)])
@pytest.mark.parametrize("act", ["gelu", "tanh"])
def test_gpu_input_gradient_gemm_with_activation_derivative(M, N, K, act):
    """vitk_gemm_grad_act: bf16(g Wt^T) * f'(aux) against fp64 on the same bf16 operands (GELU' by exact erf,
    tanh' from the saved output), over the two-group kernel (K/64 even), the one-group kernel (K/64 odd) and
    the untiled kernel (M < 128 or N not a multiple of 256)."""
    from vit_amd import lib, ops
    gen = torch.Generator().manual_seed(M + N + K)
    g = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
    wt = (torch.randn(N, K, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    aux = (torch.randn(M, N, generator=gen) * (1.5 if act == "gelu" else 0.6)).clamp(-0.99 if act == "tanh" else -9, 0.99 if act == "tanh" else 9)
    aux = aux.to(torch.bfloat16).cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm_grad_act(g, wt, aux, out, lib.ACT_GELU if act == "gelu" else lib.ACT_TANH)
    prod = (g.double() @ wt.double().t()).to(torch.bfloat16).double()      # a bf16 matmul's output
    a = aux.double()
    if act == "gelu":
        d = 0.5 * (1 + torch.erf(a / 2 ** 0.5)) + a * torch.exp(-a * a / 2) / (2 * torch.pi) ** 0.5
    else:
        d = 1 - a * a
    want = prod * d
    err = (out.double() - want).abs()
    # one bf16 rounding of the result, the product's own rounding boundary, and the derivative's approximation
    # (GELU' uses the polynomial Phi of the forward epilogue: 3.3e-4 absolute)
    tol = want.abs() * 2 ** -7 + prod.abs() * 2 ** -7 * d.abs() + prod.abs() * (4e-4 if act == "gelu" else 0) + 1e-5
    assert bool((err <= tol).all()), f"max excess {(err - tol).max():.3e}"
    assert torch.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [64, 5])
def test_gpu_no_library_gemm_or_aten_gelu_backward_on_the_bf16_training_path(batch):
    """One bf16 step under the profiler: every GEMM of the backward is a libvitk kernel (no hipBLASLt / rocBLAS
    `Cijk_` kernel, no aten gelu_backward, no fused-attention backward), with 64 images x 65 tokens (whole 64-row
    tiles) and with 5 images (325 rows: the weight-gradient GEMM masks its last K tile)."""
    import vit_amd
    from torch.profiler import ProfilerActivity, profile
    m = vit_amd.VisionTransformer(img_size=128, patches=16, num_classes=1000, hidden_size=256, num_layers=2, num_heads=4,
                                  norm=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).cuda().train()
    x = torch.randn(batch, 3, 128, 128, device="cuda")
    t = torch.randint(0, 1000, (batch,), device="cuda")
    crit = torch.nn.CrossEntropyLoss()
    crit(m(x), t).backward()
    m.zero_grad()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        crit(m(x), t).backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    bad = [n for n in names if "Cijk" in n or "gelu_backward" in n.lower() or "GeluBackward" in n
           or "attention_backward" in n.lower() and "vitk" not in n.lower() or n.startswith("aten::mm") or n.startswith("aten::matmul")]
    assert not bad, bad
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
