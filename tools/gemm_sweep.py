"""Time the tiled GEMM (bf16 out, bias) over a list of shapes: us per launch, TFLOP/s, and us per
K tile per round of tiles (256 CUs, one 256x256 tile each) -- a stride/latency diagnostic.

    python tools/gemm_sweep.py M N K [M N K ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402


def main():
    argv = sys.argv[1:]
    variant, epi = 0, "bias"
    while argv and argv[0].startswith("--"):
        k, v = argv.pop(0)[2:].split("=")
        if k == "variant":
            variant = int(v, 0)
        elif k == "epi":
            epi = v
    a = [int(v) for v in argv]
    dev = "cuda:0"
    lib.load().vitk_set_debug_variant(variant, 0)
    print(f"# gemm variant {variant:#x} epilogue {epi}")
    for M, N, K in zip(a[0::3], a[1::3], a[2::3]):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device=dev)
        if epi == "resid":
            out = torch.zeros(M, N, dtype=torch.float32, device=dev)
            run = lambda: ops.gemm_bias_residual(A, W, b, out, out)   # noqa: E731
        elif epi == "gelu":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            run = lambda: ops.gemm_bias_gelu(A, W, b, out)            # noqa: E731
        else:
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            run = lambda: ops.gemm_bias(A, W, b, out)                 # noqa: E731
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        tiles = ((M + 255) // 256) * (N // 256)
        rounds = (tiles + 255) // 256
        print(f"M={M} N={N} K={K}: {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s  tiles={tiles} "
              f"rounds={rounds}  us/round={t * 1e6 / rounds:6.1f}  us/(round*ktile)={t * 1e6 / rounds / (K // 64):5.2f}",
              flush=True)


if __name__ == "__main__":
    main()
