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


def host_threads():
    """The CPU threads this job may use: the scheduler affinity, capped by OMP_NUM_THREADS where the box sets
    one (a GPU box reports the whole machine in os.cpu_count() but grants the job a share of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = os.environ.get("OMP_NUM_THREADS", "")
    if omp.isdigit() and int(omp) > 0:
        n = min(n, int(omp))
    return max(n, 1)


def _time_oracle(arch, img_size, sd_cpu, batch):
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
    return round(batch / ts[1], 3)


def cpu_baseline(arch, img_size, sd_cpu, batch):
    """The CPU oracle (a port of the reference's forward) on every host core this job may use: the benchmarked
    model (warm-up + median of 3) and, beside it, ViT-S/16 224 fp32 batch 8, the reference's own CPU-runnable
    configuration (BASELINE configs[0])."""
    threads = host_threads()
    torch.set_num_threads(threads)
    main = _time_oracle(arch, img_size, sd_cpu, batch)
    samples = [{"workload": f"{arch} {img_size}px fp32 forward, batch {batch}", "value": main, "unit": "images/s"}]
    if (arch, img_size) != ("s16", 224):
        sd_s = synth.synth_state_dict(synth.ARCHS["s16"], 224, seed=0)
        samples.append({"workload": "s16 224px fp32 forward, batch 8 (BASELINE configs[0])",
                        "value": _time_oracle("s16", 224, sd_s, 8), "unit": "images/s"})
    # cores = the threads torch ran on (set explicitly above); host_cpus = what the machine reports
    return {"value": main, "unit": "images/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{arch} {img_size}px fp32 forward, batch {batch}, median of 3 after warm-up, "
                      f"torch.set_num_threads({threads})",
            "samples": samples}


def launch_ranks(n, argv, timeout_s=600.0):
    """`python bench.py --gpus N` as ONE command: this parent starts N fresh child processes (one per GPU,
    the reference's process model: run.py:121-124, utils.py:95-96) with the torch.distributed environment
    set, forwards rank 0's JSON line and returns non-zero if any rank fails or the launch outlives
    `timeout_s`.  The parent never touches a GPU (no torch.cuda call happens before this point), nothing is
    exec'ed, and no child outlives the parent: every rank is polled, the first non-zero exit (or the
    deadline) ends the others, and the first failing rank's last stderr lines are printed."""
    import tempfile
    import threading

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    out0 = []
    first_bad = None                       # (rank, status) of the first rank seen to fail
    timed_out = False
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            errs.append(tempfile.TemporaryFile())
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r]))
        # rank 0's stdout is drained by a thread, so that neither a full pipe nor a silent rank 0 blocks the poll
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = time.time() + timeout_s
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and first_bad is None:
                first_bad = bad[0]
            if first_bad is not None or all(c is not None for c in codes):
                break
            if time.time() > deadline:
                timed_out = True
                break
            time.sleep(0.05)
    finally:
        for p in procs:                    # whoever is still running goes now: SIGTERM, then SIGKILL
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    text = out0[0].decode(errors="replace") if out0 else ""
    ok = first_bad is None and not timed_out
    for line in text.splitlines():          # the JSON line to stdout, library chatter (gloo / RCCL banners) to stderr
        is_json = line.startswith("{")
        if is_json and not ok:
            continue                        # no throughput claim from a launch that failed
        print(line, file=sys.stdout if is_json else sys.stderr, flush=True)

    def tail(r, lines=15):
        errs[r].seek(0)
        return "\n".join(errs[r].read().decode(errors="replace").splitlines()[-lines:])

    if timed_out:
        alive = [r for r, p in enumerate(procs) if p.returncode not in (0,)]
        print(f"bench.py: launch exceeded {timeout_s:.0f} s; ranks ended by the parent: {alive}", file=sys.stderr)
        for r in alive[:2]:
            print(f"--- rank {r} stderr (tail)\n{tail(r)}", file=sys.stderr)
    elif first_bad is not None:
        r, c = first_bad
        print(f"bench.py: rank {r} exited first with status {c}; the other ranks were ended\n"
              f"--- rank {r} stderr (tail)\n{tail(r)}", file=sys.stderr)
    else:
        for r in range(n):                  # a clean launch still passes on what the ranks wrote to stderr
            t = tail(r, 5)
            if t:
                print(t, file=sys.stderr)
    for f in errs:
        f.close()
    return 0 if ok else 1


def rehearse(a):
    """Launcher / barrier / max-over-ranks / JSON plumbing with NO GPU work (CPU-only test boxes): the step is
    a fixed host-side matmul.  The line it prints is labelled as a rehearsal and carries no throughput claim."""
    if a.rehearse_hang_rank == int(os.environ.get("RANK", "0")):
        time.sleep(3600)                    # a rank that never reaches the rendezvous (launcher tests)
    if a.rehearse_die_rank == int(os.environ.get("RANK", "0")):
        raise SystemExit(3)                 # a rank that dies before the rendezvous (launcher tests)
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
    ap.add_argument("--launch-timeout", type=float, default=600.0,
                    help="seconds after which the self-launching parent (--gpus N without WORLD_SIZE) ends every rank")
    ap.add_argument("--rehearse-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--rehearse-die-rank", type=int, default=-1, help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:], a.launch_timeout))
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
        if torch.distributed.is_initialized():   # over RCCL in the real launch (also a lone rank with the group forced on); over gloo when rehearsing several ranks on one GPU
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

    # Beside the headline (never instead of it): the same forward with the last block evaluated past its q/k/v
    # projection for the cls token only.  The encoder returns token 0 (models/transformer.py:98-99 of the reference), so
    # the logits are bit-identical; fewer FLOPs are executed, and the figure is reported against those.
    pruned = None
    if a.mode == "fwd" and a.precision == "bf16" and world == 1:
        model.encoder.cls_only_last_block = True
        with torch.no_grad():
            for _ in range(2):
                yp = model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                yp = model(x)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t0
        model.encoder.cls_only_last_block = False
        pruned = (dtp, bool(torch.equal(yp, y)))

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
        if pruned is not None:
            hd_, H_ = D // cfg["num_heads"], cfg["num_heads"]
            # per image, the last block does its q/k/v projection on all tokens and everything else on one row
            saved = (ntok - 1) * (2 * (2 * H_ * ntok * hd_) + 2 * D * D + 2 * (2 * D * Mh))
            Fp = F - saved
            vp = a.batch * a.steps / pruned[0]
            out["cls_only_last_block"] = {
                "value": round(vp, 2), "unit": "images/s", "ms_per_step": round(pruned[0] / a.steps * 1e3, 4),
                "flops_executed_per_image": Fp, "mfma_peak_frac_executed": round(vp * Fp / MFMA_BF16_PEAK, 4),
                "logits_bit_identical_to_full": pruned[1],
                "note": "optional mode (model.encoder.cls_only_last_block): not the headline value; the headline counts and executes the full work"}
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
