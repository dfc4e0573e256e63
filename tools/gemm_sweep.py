"""Time the tiled GEMM over a list of shapes and debug variants.  Variants are timed interleaved, in
rounds, inside one process (median and min over the rounds are printed), because launch-to-launch
clock and box-to-box differences are larger than most kernel changes.

    python tools/gemm_sweep.py [--variant=0,4,...] [--epi=bias|resid|residln|residplanes|residpf|gelu|lngelu] [--rounds=5] [--launches=10] M N K [M N K ...]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402


def main():
    argv = sys.argv[1:]
    variants, epi, rounds, launches = [0], "bias", 5, 10
    while argv and argv[0].startswith("--"):
        k, v = argv.pop(0)[2:].split("=")
        if k == "variant":
            variants = [int(x, 0) for x in v.split(",")]
        elif k == "epi":
            epi = v
        elif k == "rounds":
            rounds = int(v)
        elif k == "launches":
            launches = int(v)
    a = [int(v) for v in argv]
    dev = "cuda:0"
    print(f"# epilogue {epi}, {rounds} interleaved rounds of {launches} launches: median us (min us) median TFLOP/s")
    for M, N, K in zip(a[0::3], a[1::3], a[2::3]):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device=dev)
        if epi == "resid":
            out = torch.zeros(M, N, dtype=torch.float32, device=dev)
            run = lambda: ops.gemm_bias_residual(A, W, b, out, out)   # noqa: E731
        elif epi == "residln":      # the residual epilogue that also writes the bf16 copy and the LayerNorm statistics
            out = torch.zeros(M, N, dtype=torch.float32, device=dev)
            out16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            stats = torch.empty(M, N // 64, 2, device=dev)
            run = lambda: ops.gemm_bias_residual_ln(A, W, b, out, out, out16, stats)   # noqa: E731
        elif epi == "residplanes":  # the same with the stream as two 16-bit planes, in place
            ph = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            pl = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            stats = torch.empty(M, N // 64, 2, device="cuda")
            run = lambda: ops.gemm_bias_residual_planes(A, W, b, ph, pl, stats)   # noqa: E731
        elif epi == "residpf":      # planes in, f32 out, no statistics (the last block's out_proj)
            ph = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            pl = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            out = torch.zeros(M, N, dtype=torch.float32, device=dev)
            run = lambda: ops.gemm_bias_residual_planes(A, W, b, ph, pl, x_out=out)   # noqa: E731
        elif epi == "lngelu":       # the GELU epilogue with the LayerNorm row factors (here 1 and 0: same values as gelu)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ab = torch.zeros(M, 2, device=dev)
            ab[:, 0] = 1.0
            cs = torch.zeros(N, device=dev)
            run = lambda: ops.gemm_ln_bias_gelu(A, W, ab, cs, b, out)   # noqa: E731
        elif epi == "gelu":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            run = lambda: ops.gemm_bias_gelu(A, W, b, out)            # noqa: E731
        else:
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            run = lambda: ops.gemm_bias(A, W, b, out)                 # noqa: E731
        times = {v: [] for v in variants}
        for r in range(rounds + 1):                 # round 0 is warm-up
            for v in variants:
                lib.load().vitk_set_debug_variant(v, 0)
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(launches):
                    run()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[v].append(e0.elapsed_time(e1) / launches * 1e3)
        print(f"M={M} N={N} K={K}")
        for v in variants:
            med, lo = statistics.median(times[v]), min(times[v])
            print(f"   variant {v:#9x}: {med:8.1f} us ({lo:8.1f})  {2.0 * M * N * K / med / 1e6:7.1f} TFLOP/s", flush=True)
    lib.load().vitk_set_debug_variant(0, 0)


if __name__ == "__main__":
    main()
