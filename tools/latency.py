"""Small-batch latency of the forward: eager launches vs HIP-graph replay (vit_amd.GraphedForward).

    python tools/latency.py [--arch b16] [--batches 1,8,32]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd                      # noqa: E402
from vit_amd import synth           # noqa: E402


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="b16")
    ap.add_argument("--batches", default="1,8,32")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    m = getattr(vit_amd, a.arch)(img_size=224).to(dev).eval()
    m.load_state_dict(synth.synth_state_dict(synth.ARCHS[a.arch], 224, seed=0))
    for B in (int(b) for b in a.batches.split(",")):
        x = synth.synth_images(B, 224, seed=1).to(dev).to(torch.bfloat16)
        with torch.no_grad():
            eager = timed(lambda: m(x), 100)
            g = vit_amd.GraphedForward(m, x)
            same = torch.equal(g(x), m(x))
            graph = timed(lambda: g(x), 100)
        print(f"{a.arch} batch {B:3d}: eager {eager:7.3f} ms  graph {graph:7.3f} ms  ({B / graph * 1e3:8.0f} images/s)  identical={same}", flush=True)


if __name__ == "__main__":
    main()
