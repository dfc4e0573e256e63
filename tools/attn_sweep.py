"""Time the attention kernel over debug variants, interleaved in rounds inside one process.

    python tools/attn_sweep.py [--variant=0,3,...] [--rounds=5] [--launches=100] B H N hd [B H N hd ...]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import lib, ops   # noqa: E402


def main():
    argv = sys.argv[1:]
    variants, rounds, launches = [0], 5, 100
    while argv and argv[0].startswith("--"):
        k, v = argv.pop(0)[2:].split("=")
        if k == "variant":
            variants = [int(x, 0) for x in v.split(",")]
        elif k == "rounds":
            rounds = int(v)
        elif k == "launches":
            launches = int(v)
    a = [int(v) for v in argv]
    dev = "cuda:0"
    for B, H, N, hd in zip(a[0::4], a[1::4], a[2::4], a[3::4]):
        qkv = torch.randn(3, B, H, N, hd, device=dev).to(torch.bfloat16)
        ctx = torch.empty(B, N, H * hd, dtype=torch.bfloat16, device=dev)
        times = {v: [] for v in variants}
        for r in range(rounds + 1):
            for v in variants:
                lib.load().vitk_set_debug_variant(0, v)
                ops.attention(qkv, ctx, hd ** -0.5)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(launches):
                    ops.attention(qkv, ctx, hd ** -0.5)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[v].append(e0.elapsed_time(e1) / launches * 1e3)
        print(f"B={B} H={H} N={N} hd={hd}")
        fl = 4.0 * B * H * N * N * hd
        for v in variants:
            med = statistics.median(times[v])
            print(f"   variant {v:#6x}: {med:8.1f} us ({min(times[v]):8.1f})  {fl / med / 1e6:7.1f} TFLOP/s", flush=True)
    lib.load().vitk_set_debug_variant(0, 0)


if __name__ == "__main__":
    main()
