"""A/B of kernel debug variants on the whole forward (the number that counts): the model is built once,
variants are timed interleaved in rounds, each round = `--steps` back-to-back forwards after 2 warm-ups.

    python tools/ab_forward.py --gemm-variants=0,5 [--attn-variants=0] [--rounds=4] [--steps=10] [--arch b16] ...
"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd                          # noqa: E402
from vit_amd import lib, synth          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gemm-variants", default="0")
    ap.add_argument("--attn-variants", default="0")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--arch", default="b16")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.ARCHS[a.arch]
    sd = synth.synth_state_dict(cfg, a.img_size, seed=0)
    with torch.device("meta"):
        model = getattr(vit_amd, a.arch)(img_size=a.img_size)
    model = model.to_empty(device=dev)
    model.load_state_dict(sd, strict=True)
    model.eval()
    uniq = min(a.batch, 32)
    x = synth.synth_images(uniq, a.img_size, seed=1)
    x = x.repeat((a.batch + uniq - 1) // uniq, 1, 1, 1)[: a.batch].to(dev).to(torch.bfloat16).contiguous()
    combos = [(int(g, 0), int(t, 0)) for g in a.gemm_variants.split(",") for t in a.attn_variants.split(",")]
    times = {c: [] for c in combos}
    with torch.no_grad():
        for r in range(a.rounds + 1):
            for c in combos:
                lib.load().vitk_set_debug_variant(*c)
                for _ in range(2):
                    model(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    model(x)
                torch.cuda.synchronize()
                if r:
                    times[c].append((time.perf_counter() - t0) / a.steps * 1e3)
    lib.load().vitk_set_debug_variant(0, 0)
    print(f"# {a.arch} {a.img_size}px batch {a.batch}: ms per forward, median (min) over {a.rounds} rounds of {a.steps}")
    for c in combos:
        med = statistics.median(times[c])
        print(f"  gemm {c[0]:#9x} attn {c[1]:#4x}: {med:8.3f} ms ({min(times[c]):8.3f})  {a.batch / med * 1e3:9.0f} img/s", flush=True)


if __name__ == "__main__":
    main()
