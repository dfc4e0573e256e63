"""Experiment: the batch as S micro-batches on S streams, each kernel's persistent grid sized for 1/S of the
chip (VITK_CU_LIMIT), the streams started a fraction of a layer apart, so that one micro-batch's
bandwidth-bound phases (LayerNorm, residual epilogues, attention) run beside the other's matrix-bound ones.

    VITK_CU_LIMIT=128 python tools/two_stream.py --streams 2 --batch 256
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd                                   # noqa: E402
from vit_amd import synth                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="b16")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--delay-us", type=float, default=0.0, help="start stream i this many microseconds x i late")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.ARCHS[a.arch]
    sd = synth.synth_state_dict(cfg, a.img_size, seed=0)
    S = a.streams
    models = []
    for i in range(S):
        with torch.device("meta"):
            m = getattr(vit_amd, a.arch)(img_size=a.img_size)
        m = m.to_empty(device=dev)
        m.load_state_dict(sd, strict=True)
        m.eval()
        models.append(m)
    for m in models[1:]:                         # one set of kernel-ready weights for all micro-batches
        m._cache = models[0]._cache
    x = synth.synth_images(32, a.img_size, seed=1).repeat((a.batch + 31) // 32, 1, 1, 1)[: a.batch]
    x = x.to(dev).to(torch.bfloat16).contiguous()
    per = a.batch // S
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]

    def step(first=False):
        outs = []
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                if first and i and a.delay_us > 0:
                    torch.cuda._sleep(int(a.delay_us * i * 2100))
                outs.append(models[i](xs[i]))
        return outs

    with torch.no_grad():
        ref = models[0](x[:8]).float().cpu()
        for k in range(a.warmup):
            step(first=(k == 0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            outs = step(first=(k == 0))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    y = torch.cat([o.float().cpu() for o in outs])
    err = (y[:8] - ref).abs().max().item()
    print(f"streams {S} cu_limit {os.environ.get('VITK_CU_LIMIT', '-')} delay {a.delay_us} us: "
          f"{a.batch * a.steps / dt:9.1f} images/s  {dt / a.steps * 1e3:7.3f} ms/step  max|first 8 - single| {err:.2e}",
          flush=True)


if __name__ == "__main__":
    main()
