"""Shader clock while one kernel runs back to back (rocm-smi sampled from a second thread).

    python tools/clock_probe.py

The matrix kernels of this library run power-limited: what the clock settles at under each of them tells how much
of a kernel's time is the frequency the power manager grants it rather than cycles it could save.
"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops   # noqa: E402


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showgpuclocks", "--json"], capture_output=True, text=True,
                             timeout=20).stdout
        cards = json.loads(out)
    except Exception as e:   # noqa: BLE001
        return f"rocm-smi failed: {e}"
    res = []
    for name, c in sorted(cards.items()):
        sclk = next((v for k, v in c.items() if "sclk" in k.lower()), "?")
        res.append(f"{name}: sclk {sclk}")          # (the socket power reading is not available to an unprivileged user)
    return "; ".join(res)


def run(label, fn, seconds=2.5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 20 * 1e-3
    n = max(20, int(seconds / per))
    samples = []

    def sampler():
        time.sleep(0.8)
        for _ in range(3):
            samples.append(sample())

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = time.time() - t0
    th.join()
    print(f"{label}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us per launch over {wall:.1f} s", flush=True)
    for s in samples:
        print("    " + s, flush=True)


def main():
    print("idle: " + sample(), flush=True)
    print(subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True).stdout.strip())
    R, M, N = 50432, 3072, 768
    g = torch.randn(R, M, device="cuda").to(torch.bfloat16)
    x = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    run("gemm_tn 50432 x 3072 x 768", lambda: ops.gemm_tn(g, x, out))
    w = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    b = torch.zeros(M, device="cuda")
    h = torch.empty(R, M, dtype=torch.bfloat16, device="cuda")
    run("fc1 forward GEMM 50432 x 3072 x 768", lambda: ops.gemm_bias_gelu(x, w, b, out=h))
    run("torch g.t() @ x", lambda: torch.matmul(g.t(), x))
    y = torch.empty_like(x)
    run("copy 77 MB", lambda: y.copy_(x))


if __name__ == "__main__":
    main()
