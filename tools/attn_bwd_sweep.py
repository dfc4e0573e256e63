"""Time the attention backward (vitk_attention_backward: the dQ pass and the dK/dV pass) on the training shapes.

    python tools/attn_bwd_sweep.py [--hd=64] [--torch] [B H N ...]

--torch also times torch's scaled_dot_product_attention backward on the same tensors (what head sizes without a kernel
fall back to).
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops   # noqa: E402


def timeit(fn, launches=50, rounds=5):
    ts = []
    for r in range(rounds + 1):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / launches * 1e3)
    return statistics.median(ts)


def main():
    hd = next((int(v.split("=")[1]) for v in sys.argv[1:] if v.startswith("--hd=")), 64)
    with_torch = "--torch" in sys.argv[1:]
    a = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [256, 12, 197, 128, 12, 576, 256, 16, 197]
    for B, H, N in zip(a[0::3], a[1::3], a[2::3]):
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = torch.randn(3, B, H, N, hd, generator=g, device="cuda").to(torch.bfloat16)
        out = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device="cuda")
        scale = hd ** -0.5
        lse = ops.attention_lse(qkv, out, scale)
        dctx = torch.randn(B, N, H * hd, generator=g, device="cuda").to(torch.bfloat16)
        t = timeit(lambda: ops.attention_backward(qkv, out, dctx, lse, scale))
        fl = 7 * 2.0 * B * H * N * N * hd
        print(f"B={B} H={H} N={N} hd={hd}: attention backward {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s (7 products)", flush=True)
        if with_torch:
            q, k, v = (qkv[i].detach().clone().requires_grad_(True) for i in range(3))
            g4 = dctx.view(B, N, H, hd).transpose(1, 2)

            def torch_bwd():
                o = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=scale)
                torch.autograd.grad(o, (q, k, v), g4)

            tt = timeit(torch_bwd, launches=10, rounds=3)
            print(f"    torch scaled_dot_product_attention forward + backward {tt:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
