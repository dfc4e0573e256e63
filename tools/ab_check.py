"""Race screen + equality check of a GEMM variant against variant 0 at the headline shapes: outputs must be
bit-identical (same accumulation order) over repeated launches.    python tools/ab_check.py VARIANT [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402

v = int(sys.argv[1], 0)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
L = lib.load()
bad = 0
for M, N, K in [(50432, 3072, 768), (50432, 768, 3072), (50432, 768, 768), (50432, 2304, 768), (25216 + 77, 768, 1024), (73856, 768, 768), (4728, 2304, 768)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r0 = torch.randn(M, N, device=dev)
    for epi in ("resid", "gelu", "bias"):
        def run(variant):
            L.vitk_set_debug_variant(variant, 0)
            if epi == "resid":
                out = r0.clone()
                ops.gemm_bias_residual(A, W, b, out, out)
            elif epi == "gelu":
                out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                ops.gemm_bias_gelu(A, W, b, out)
            else:
                out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                ops.gemm_bias(A, W, b, out)
            return out
        want = run(0)
        n_bad = 0
        for _ in range(reps):
            got = run(v)
            if not torch.equal(got, want):
                n_bad += 1
                d = (got.float() - want.float()).abs()
                print(f"  MISMATCH {M}x{N}x{K} {epi}: max diff {d.max().item():.3e}, {int((d > 0).sum())} elements, rows {d.amax(1).nonzero().flatten()[:8].tolist()}")
        print(f"{M}x{N}x{K} {epi}: {reps - n_bad}/{reps} identical", flush=True)
        bad += n_bad
L.vitk_set_debug_variant(0, 0)
sys.exit(1 if bad else 0)
