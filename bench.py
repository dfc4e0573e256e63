"""Headline benchmark: ViT-B/16 224px bf16 forward, images/sec for the whole node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  Every rank holds the full weights and runs its own batch of 256 synthetic
images (weak scaling, no data-path collective); a "step" is one forward over that batch with the
images already resident in HBM as bf16.  Rank 0 prints ONE JSON line.  Next to the throughput it
reports the dominant kernel (the fc1 GEMM+bias+GELU launch) against the bf16 MFMA roofline, timed
with HIP events inside the timed region, and the CPU oracle timed on this box's host cores.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import vit_amd                                  # noqa: E402
from vit_amd import dist as vdist               # noqa: E402
from vit_amd import ops, synth                  # noqa: E402

MFMA_BF16_PEAK = 2.5166e15     # dense bf16 FLOP/s per MI355X: 256 CU x 2.4 GHz x 4096 FLOP/clk/CU


def profiled_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` in `workload` from the newest profiles/rNN/traffic.json (written
    by tools/pmc_summary.py from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes), or None when no profile of
    this workload is committed.  Returns (bytes, file) so the JSON line names its source."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            with open(path) as f:
                table = json.load(f)
        except (OSError, ValueError):
            continue
        entry = table.get(workload, {}).get(kernel)
        if entry is not None:
            return float(entry["hbm_bytes"]), os.path.relpath(path, ROOT)
    return None, None


def flops_per_image(cfg, img_size, num_classes=1000):
    """Algorithmic FLOPs of one forward (2 per multiply-add, full sequence in every layer)."""
    D, H, P, L = cfg["hidden_size"], cfg["num_heads"], cfg["patches"], cfg["num_layers"]
    hd, M = D // H, int(D * cfg["mlp_ratio"])
    n = (img_size // P) ** 2
    N = n + 1
    per_layer = 2 * N * D * 3 * D + 2 * (2 * H * N * N * hd) + 2 * N * D * D + 2 * (2 * N * D * M)
    return 2 * n * (3 * P * P) * D + L * per_layer + 2 * D * D + 2 * D * num_classes


class KernelTimer:
    """HIP-event pairs around every launch of one named kernel, on the stream it is launched on."""

    def __init__(self, name):
        self.name, self.pairs, self.enabled = name, [], False

    def __call__(self, name):
        return self if (self.enabled and name == self.name) else ops._NULL

    def __enter__(self):
        self._a = torch.cuda.Event(enable_timing=True)
        self._a.record()

    def __exit__(self, *exc):
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        self.pairs.append((self._a, b))
        return False

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(len(self.pairs), 1)


def cpu_baseline(arch, img_size, sd_cpu, batch):
    """The CPU oracle (a port of the reference's forward) on this host's cores: warm-up + 3 runs."""
    from oracle import vit_oracle as vo
    cfg = synth.ARCHS[arch]
    x = synth.synth_images(batch, img_size, seed=1)
    vo.forward(sd_cpu, x[:2], cfg)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        vo.forward(sd_cpu, x, cfg)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    # cores = the threads torch actually ran on (the box grants this job a CPU share smaller than the machine
    # and sets OMP_NUM_THREADS to it); host_cpus = what the machine reports
    return {"value": round(batch / ts[1], 3), "unit": "images/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{arch} {img_size}px fp32 forward, batch {batch}, median of 3 after warm-up"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` as ONE command: this parent starts N fresh child processes (one per GPU,
    the reference's process model: run.py:121-124, utils.py:95-96) with the torch.distributed environment
    set, forwards rank 0's JSON line and returns non-zero if any rank fails.  The parent never touches a
    GPU (no torch.cuda call happens before this point), and nothing is exec'ed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout.read().decode()
    rc = 0
    deadline = time.time() + 600
    for p in procs:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode
    for line in out0.splitlines():          # the JSON line to stdout, library chatter (gloo / RCCL banners) to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    if rc != 0:
        print(f"bench.py: a rank exited with status {rc}", file=sys.stderr)
    return 1 if rc != 0 else 0


def rehearse(a):
    """Launcher / barrier / max-over-ranks / JSON plumbing with NO GPU work (CPU-only test boxes): the step is
    a fixed host-side matmul.  The line it prints is labelled as a rehearsal and carries no throughput claim."""
    rank, world, _ = vdist.init_from_env(backend="gloo")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    m = torch.ones(64, 64)
    for _ in range(a.warmup):
        m @ m
    vdist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m @ m
    vdist.barrier()
    dt = vdist.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "rehearsal: launcher and reduction plumbing only, no GPU work", "value": None,
                          "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(dt / a.steps * 1e3, 4), "data": "none"}), flush=True)
    if world > 1:
        vdist.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="b16")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="compute mode; the headline metric is bf16, fp32 is the reference-accuracy mode")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"],
                    help="fwd: the headline forward benchmark; train: one training step per timed step "
                         "(forward, CrossEntropyLoss, backward, clip_grad_norm_ 1.0, AdamW; main.py:207-223 of the reference)")
    ap.add_argument("--rehearse", action="store_true",
                    help="exercise the multi-process launch and the timing reduction on CPU (gloo), no GPU work")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    if a.rehearse:
        return rehearse(a)

    # Rehearsal switches for a box with fewer GPUs than ranks (never set by the real launch): the ranks
    # then share GPU (local % count) and talk over gloo, since RCCL refuses two ranks on one device.
    share = os.environ.get("VIT_AMD_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count()
    local_env = int(os.environ.get("LOCAL_RANK", "0"))
    if local_env >= ndev and not share:
        raise SystemExit(f"LOCAL_RANK {local_env} but only {ndev} GPU(s) visible")
    if share:
        torch.cuda.set_device(local_env % ndev)
    rank, world, local = vdist.init_from_env(backend="gloo" if share else None)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    local = local % ndev if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = synth.ARCHS[a.arch]
    sd = synth.synth_state_dict(cfg, a.img_size, seed=0)
    with torch.device("meta"):
        model = getattr(vit_amd, a.arch)(img_size=a.img_size)
    model = model.to_empty(device=dev)
    model.load_state_dict(sd, strict=True)
    model.eval()
    model.set_precision(a.precision)
    # rank r owns images [r*B, (r+1)*B) of the synthetic stream; generate a few and tile them
    lo = rank * a.batch
    uniq = min(a.batch, 32)
    x = synth.synth_images(uniq, a.img_size, seed=1, first=lo)
    x = x.repeat((a.batch + uniq - 1) // uniq, 1, 1, 1)[: a.batch].to(dev).to(torch.bfloat16).contiguous()

    ntok = (a.img_size // cfg["patches"]) ** 2 + 1
    timer = KernelTimer("gemm_bias_gelu" if a.precision == "bf16" else "attention_f32")
    ops.set_event_hook(timer)

    if a.mode == "train":
        # BASELINE configs[4]: the reference's step with its defaults (run.py:55-60, 224: AdamW lr 1e-3,
        # weight_decay 0.03, gradient_clip 1.0); gradients all-reduced over RCCL by DistributedDataParallel
        # as in main.py:125-126
        model.train()
        net = model
        if world > 1:       # over RCCL in the real launch; over gloo when rehearsing several ranks on one GPU
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], gradient_as_bucket_view=True)
        opt = torch.optim.AdamW(model.parameters(), 1e-3, weight_decay=0.03, fused=True)
        crit = torch.nn.CrossEntropyLoss()
        target = torch.randint(0, 1000, (a.batch,), device=dev, generator=torch.Generator(device=dev).manual_seed(rank))

        def step():
            out = net(x)
            loss = crit(out, target)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            return out.detach()
        ctx = torch.enable_grad()
    else:
        def step():
            return model(x)
        ctx = torch.no_grad()

    with ctx:
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        vdist.barrier()
        torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(a.steps):
            y = step()
        torch.cuda.synchronize()
        vdist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
    if not torch.isfinite(y).all():
        raise SystemExit("non-finite logits")
    dt = vdist.max_over_ranks(dt, device=dev if (torch.distributed.is_initialized() and not share) else "cpu")

    if rank == 0:
        F = flops_per_image(cfg, a.img_size)
        value = world * a.batch * a.steps / dt
        D, Mh = cfg["hidden_size"], int(cfg["hidden_size"] * cfg["mlp_ratio"])
        k_flops = 2.0 * a.batch * ntok * Mh * D                  # one fc1 launch: [B*N, D] x [M, D]^T
        k_ms = timer.mean_ms()
        traffic, traffic_src = profiled_traffic(f"{a.arch}_{a.img_size}_bs{a.batch}", "fc1")
        achieved = k_flops / (k_ms * 1e-3) / 1e12
        out = {
            "metric": f"images/sec (whole node) ViT-{a.arch.upper()[0]}/{cfg['patches']} {a.img_size}px {a.precision} "
                      + ("fwd" if a.mode == "fwd" else "train step"),
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"ViT-{a.arch.upper()[0]}/{cfg['patches']} {a.img_size}px "
                                   + ("forward, " if a.mode == "fwd" else "training step (forward and backward kernels of the library; CE, clip 1.0, AdamW from torch), ")
                                   + f"batch {a.batch} per GPU, random-init weights"
                                   + (" (BASELINE configs[1])" if (a.mode, a.arch, a.img_size, a.batch) == ("fwd", "b16", 224, 256) else ""),
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world, "tokens": ntok,
                       "parallelism": f"dp{world}: batch sharded, weights replicated, "
                                      + ("no collectives" if a.mode == "fwd" else "gradient all-reduce over RCCL (DDP)")},
            "gflop_per_image": round(F / 1e9, 3),
            "mfma_peak_frac": round(value * F * (1 if a.mode == "fwd" else 3) / (world * MFMA_BF16_PEAK), 4),
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_pp<bias+GELU> (fc1)",
                         "achieved": round(achieved, 1), "peak": round(MFMA_BF16_PEAK / 1e12, 1),
                         "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_BF16_PEAK, 4),
                         "avg_launch_ms": round(k_ms, 4), "launches": len(timer.pairs),
                         # HBM-side bytes of one fc1 launch from the committed rocprofv3 PMC passes of this
                         # workload (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE); not collected live
                         "traffic": traffic, "traffic_source": traffic_src},
        }
        if a.precision != "bf16":
            out.pop("roofline")          # the roofline entry describes the bf16 fc1 kernel only
        if a.mode == "train":
            out["train_flop_factor"] = 3          # forward + 2 x forward for the backward GEMMs
        if world == 1 and not a.no_cpu_baseline and a.mode == "fwd":
            out["cpu_baseline"] = cpu_baseline(a.arch, a.img_size, sd, 32 if a.img_size <= 224 else 8)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        vdist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
