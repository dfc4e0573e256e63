"""Time the weight-gradient GEMM (vitk_gemm_tn) beside torch's g.t() @ x on the ViT-B/16 shapes.

    python tools/tn_sweep.py [R M N ...]
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
    a = [int(v) for v in sys.argv[1:]] or [50432, 3072, 768, 50432, 768, 3072, 50432, 2304, 768, 50432, 768, 768]
    for R, M, N in zip(a[0::3], a[1::3], a[2::3]):
        # TN_SCALE=0 multiplies zeros: the same instructions at a fraction of the switching power
        scale = float(os.environ.get("TN_SCALE", "1"))
        g = (torch.randn(R, M, device="cuda") * scale).to(torch.bfloat16)
        x = (torch.randn(R, N, device="cuda") * scale).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.float32, device="cuda")
        fl = 2.0 * R * M * N
        cs = torch.empty(M, device="cuda")
        t_own = timeit(lambda: ops.gemm_tn(g, x, out))
        t_cs = timeit(lambda: ops.gemm_tn(g, x, out, colsum=cs))
        t_lib = timeit(lambda: g.t() @ x)
        print(f"R={R} M={M} N={N}: vitk_gemm_tn {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TFLOP/s, with column sums {t_cs:8.1f} us "
              f"| torch {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
