"""Where one training step's GPU time goes: torch.profiler over a few steps of bench.py's train mode
(ViT-B/16 224px, batch 256), kernels grouped by name.

    python tools/train_profile.py [--batch 256] [--steps 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd                      # noqa: E402
from vit_amd import synth           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--arch", default="b16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.ARCHS[a.arch]
    model = getattr(vit_amd, a.arch)(img_size=224).to(dev).train()
    model.load_state_dict(synth.synth_state_dict(cfg, 224, seed=0))
    x = synth.synth_images(32, 224, seed=1).repeat(a.batch // 32, 1, 1, 1).to(dev).to(torch.bfloat16)
    t = torch.randint(0, 1000, (a.batch,), device=dev)
    opt = torch.optim.AdamW(model.parameters(), 1e-3, weight_decay=0.03, fused=True)
    crit = torch.nn.CrossEntropyLoss()

    def step():
        loss = crit(model(x), t)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        if dt > 0 and e.device_type == torch.autograd.DeviceType.CUDA:
            rows.append((dt / a.steps / 1e3, e.count // a.steps, e.key[:110]))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"# GPU kernel time per step: {total:.2f} ms")
    for ms, n, name in rows[:40]:
        print(f"{ms:9.3f} ms  x{n:<5d} {name}")


if __name__ == "__main__":
    main()
