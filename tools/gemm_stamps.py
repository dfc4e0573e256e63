"""Where one K tile of the two-group GEMM kernel spends its cycles: launches the stamped probe build
(debug variant 0xe00004) on a large GEMM and prints, per phase, the average cycles between the
s_memtime stamps of one wave of each group of workgroup 0 (summed over all its K tiles).

    python tools/gemm_stamps.py [M N K]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402


def main():
    M, N, K = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (8192, 8192, 8192)
    dev = "cuda:0"
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib.load().vitk_set_debug_variant(4, 0)
    for _ in range(20):
        ops.gemm_bias(A, W, b, out)
    lib.load().vitk_set_debug_variant(0xe00004, 0)
    for _ in range(3):
        ops.gemm_bias(A, W, b, out)
    torch.cuda.synchronize()
    lib.load().vitk_set_debug_variant(0, 0)
    raw = out[0, :128].contiguous().view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
    names = ["bar after M", "L issue", "vmcnt wait", "bar after L", "lgkm wait", "MFMA issue"]
    for g in range(2):
        r = raw[g * 32:g * 32 + 25]
        n = max(int(r[24]), 1)
        print(f"group {g} (wave {4 * g}): mean cycles per segment over {n} K tiles")
        print("  phase " + " ".join(f"{x:>12s}" for x in names) + "        total")
        tot = 0.0
        for ph in range(4):
            v = [r[ph * 6 + i] / n for i in range(6)]
            tot += sum(v)
            print(f"  {ph}     " + " ".join(f"{x:12.0f}" for x in v) + f" {sum(v):12.0f}")
        print(f"  K tile total {tot:.0f} cycles")


if __name__ == "__main__":
    main()
