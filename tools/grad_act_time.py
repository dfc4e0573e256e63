"""Time vitk_gemm_grad_act (dX through GELU') next to the forward fc1 GEMM of the same shape, one process.
    python tools/grad_act_time.py [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50432, 3072, 768)
dev = "cuda:0"
g = torch.randn(M, K, device=dev).to(torch.bfloat16)
wt = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
b = torch.zeros(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
runs = {"grad_act(gelu)": lambda: ops.gemm_grad_act(g, wt, aux, out, lib.ACT_GELU),
        "bias_gelu (forward)": lambda: ops.gemm_bias_gelu(g, wt, b, out),
        "bias": lambda: ops.gemm_bias(g, wt, b, out)}
for r in range(4):
    for name, fn in runs.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            print(f"{name:22s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
