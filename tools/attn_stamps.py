"""Cycle stamps of the persistent attention kernel (diagnostic build, attention variant bit 10): where workgroup 0's
waves spend a head's time.  VITK_LIB=vit_amd/libvitk_probes.so python tools/attn_stamps.py [B H N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402


def main():
    B, H, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 12, 197)
    hd, NW = 64, 7
    qkv = torch.randn(3, B, H, N, hd, device="cuda").to(torch.bfloat16)
    ctx = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device="cuda")
    L = lib.load()
    for _ in range(3):
        ops.attention(qkv, ctx, 0.125)
    L.vitk_set_debug_variant(0, 1 << 10)
    ops.attention(qkv, ctx, 0.125)
    torch.cuda.synchronize()
    L.vitk_set_debug_variant(0, 0)
    raw = ctx.view(-1)[: 12 * NW * 8 * 2].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    st = raw.reshape(12, NW, 8)
    names = ["wait+barrier", "issue next head", "step 0", "step 1", "step 2", "step 3 (1 block)", "finish"]
    d = np.diff(st, axis=2)[2:11] & 0xFFFFFFFF        # heads 2..10: steady state
    print("cycles per segment, mean over heads 2..10; rows = waves 0..6 (waves w and w + 4 share SIMD w)")
    print("wave " + " ".join(f"{n:>17s}" for n in names) + "   head total")
    for w in range(NW):
        head = (st[3:12, w, 0] - st[2:11, w, 0]) & 0xFFFFFFFF
        print(f"{w:4d} " + " ".join(f"{d[:, w, i].mean():17.0f}" for i in range(7)) + f"   {head.mean():10.0f}")


if __name__ == "__main__":
    main()
