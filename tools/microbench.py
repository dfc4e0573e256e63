"""Per-kernel timings at the headline shapes (ViT-B/16 224px, batch 256 by default).

    python tools/microbench.py [--batch 256] [--arch b16] [--img-size 224] [--iters 20]

Prints one line per kernel: average launch time (HIP events around `iters` back-to-back launches),
achieved TFLOP/s for the MFMA kernels and achieved GB/s (algorithmic bytes) for the HBM-bound ones.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops, synth   # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arch", default="b16")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gemm-variant", type=lambda v: int(v, 0), default=0)
    ap.add_argument("--attn-variant", type=int, default=0)
    a = ap.parse_args()
    cfg = synth.ARCHS[a.arch]
    lib.load().vitk_set_debug_variant(a.gemm_variant, a.attn_variant)
    dev = "cuda:0"
    B, D, H, P = a.batch, cfg["hidden_size"], cfg["num_heads"], cfg["patches"]
    hd, M = D // H, int(D * cfg["mlp_ratio"])
    N = (a.img_size // P) ** 2 + 1
    R = B * N
    bf, f32 = torch.bfloat16, torch.float32
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, dt=bf, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)   # noqa: E731
    x = rn(R, D, dt=f32)
    y = rn(R, D)
    h = rn(R, M)
    qkv = rn(3, B, H, N, hd)
    ctx = torch.empty(B, N, D, dtype=bf, device=dev)
    img = rn(B, 3, a.img_size, a.img_size)
    w_qkv, w_out, w_fc1, w_fc2 = rn(3 * D, D, sc=.02), rn(D, D, sc=.02), rn(M, D, sc=.02), rn(D, M, sc=.02)
    w_emb = rn(D, 3 * P * P, sc=.02)
    b_qkv, b_d, b_m = rn(3 * D, dt=f32), rn(D, dt=f32), rn(M, dt=f32)
    gam, bet = rn(D, dt=f32), rn(D, dt=f32)
    pos, cls = rn(N, D, dt=f32), rn(D, dt=f32)
    x3 = x.view(B, N, D)

    rows = []

    def rec(name, t, flops=None, bytes_=None):
        s = f"{name:34s} {t * 1e6:9.1f} us"
        if flops:
            s += f"  {flops / t / 1e12:8.1f} TFLOP/s ({flops / t / 2.5166e15 * 100:5.1f}% of bf16 MFMA peak)"
        if bytes_:
            s += f"  {bytes_ / t / 1e9:8.1f} GB/s ({bytes_ / t / 8e12 * 100:5.1f}% of 8 TB/s)"
        print(s, flush=True)
        rows.append((name, t))

    it = a.iters
    print(f"# {a.arch} {a.img_size}px batch {B}: rows={R} D={D} H={H} hd={hd} M={M} N={N} "
          f"gemm_variant={a.gemm_variant} attn_variant={a.attn_variant}", flush=True)
    rec("layernorm f32->bf16", timeit(lambda: ops.layernorm(x, gam, bet, 1e-6, y), it), bytes_=R * D * 6)
    rec("patch_embed (+cls,pos)", timeit(lambda: ops.patch_embed(img, w_emb, b_d, cls, pos, x3, P), it),
        flops=2.0 * B * (N - 1) * 3 * P * P * D)
    rec("qkv_proj", timeit(lambda: ops.qkv_proj(y, w_qkv, b_qkv, qkv, B, N, H, hd), it), flops=2.0 * R * 3 * D * D)
    rec("attention", timeit(lambda: ops.attention(qkv, ctx, hd ** -0.5), it), flops=4.0 * B * H * N * N * hd)
    rec("out_proj + residual", timeit(lambda: ops.gemm_bias_residual(ctx.view(R, D), w_out, b_d, x, x), it),
        flops=2.0 * R * D * D)
    rec("fc1 + bias + GELU", timeit(lambda: ops.gemm_bias_gelu(y, w_fc1, b_m, h), it), flops=2.0 * R * M * D)
    rec("fc2 + residual", timeit(lambda: ops.gemm_bias_residual(h, w_fc2, b_d, x, x), it), flops=2.0 * R * M * D)
    # the same GEMMs with the LayerNorm between them folded in (what the inference path of models up to 24 layers runs)
    stats = torch.empty(R, D // 64, 2, dtype=f32, device=dev)
    ab = torch.empty(R, 2, dtype=f32, device=dev)
    cs_q, c_q = rn(3 * D, dt=f32), rn(3 * D, dt=f32)
    cs_m, c_m = rn(M, dt=f32), rn(M, dt=f32)
    if ops.ln_fold_ok(R, D, M):
        rec("out_proj + residual + LN stats", timeit(lambda: ops.gemm_bias_residual_ln(ctx.view(R, D), w_out, b_d, x, x, y, stats), it),
            flops=2.0 * R * D * D)
        rec("ln_finalize", timeit(lambda: ops.ln_finalize(stats, ab, D, 1e-6), it), bytes_=R * (D // 64) * 8 + R * 8)
        rec("fc1 (LN folded) + GELU", timeit(lambda: ops.gemm_ln_bias_gelu(y, w_fc1, ab, cs_m, c_m, h), it),
            flops=2.0 * R * M * D)
        rec("fc2 + residual + LN stats", timeit(lambda: ops.gemm_bias_residual_ln(h, w_fc2, b_d, x, x, y, stats), it),
            flops=2.0 * R * M * D)
        ph, pl = torch.empty(R, D, dtype=bf, device=dev), torch.empty(R, D, dtype=bf, device=dev)
        ops.gemm_bias_residual_planes(ctx.view(R, D), w_out, b_d, ph, pl, stats, x_in=x)
        rec("out_proj + residual planes + LN stats", timeit(lambda: ops.gemm_bias_residual_planes(ctx.view(R, D), w_out, b_d, ph, pl, stats), it),
            flops=2.0 * R * D * D)
        rec("fc2 + residual planes + LN stats", timeit(lambda: ops.gemm_bias_residual_planes(h, w_fc2, b_d, ph, pl, stats), it),
            flops=2.0 * R * M * D)
        rec("qkv_proj (LN folded)", timeit(lambda: ops.qkv_proj_ln(y, w_qkv, ab, cs_q, c_q, qkv, B, N, H, hd), it),
            flops=2.0 * R * 3 * D * D)
    o16 = torch.empty(R, D, dtype=bf, device=dev)
    rec("gemm_bias bf16 (DxD, no epilogue math)", timeit(lambda: ops.gemm_bias(y, w_out, b_d, o16), it),
        flops=2.0 * R * D * D)
    npx = B * 3 * a.img_size * a.img_size
    img8 = torch.randint(0, 256, (B, 3, a.img_size, a.img_size), device=dev, dtype=torch.uint8)
    mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
    rec("normalize u8->bf16 (input step)", timeit(lambda: ops.normalize_u8(img8, mean, std, img), it), bytes_=npx * 3)
    img32 = torch.empty(B, 3, a.img_size, a.img_size, dtype=f32, device=dev)
    rec("normalize u8->f32", timeit(lambda: ops.normalize_u8(img8, mean, std, img32), it), bytes_=npx * 5)
    logits = rn(B, 1000, dt=f32)
    tgt = torch.randint(0, 1000, (B,), device=dev)
    rk = torch.empty(B, dtype=torch.int32, device=dev)
    rec("target_rank (accuracy)", timeit(lambda: ops.target_rank(logits, tgt, rk), it), bytes_=B * 1000 * 4)
    L = cfg["num_layers"]
    per_layer = sum(t for n, t in rows if n in ("qkv_proj", "attention", "out_proj + residual",
                                               "fc1 + bias + GELU", "fc2 + residual")) + 2 * rows[0][1]
    total = L * per_layer + rows[1][1]
    print(f"# sum of kernels for one forward ~ {total * 1e3:.3f} ms -> {B / total:.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
