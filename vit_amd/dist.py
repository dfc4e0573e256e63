"""One process per GPU; the image batch shards across ranks with no data-path collective.

The reference's only parallelism is data parallelism (main.py:125-126,
utils.py:93-99: one process per GPU, NCCL).  Images are independent on the
forward path (no BatchNorm, no cross-sample op), so inference needs no
exchange at all: rank r runs images [r*B/n, (r+1)*B/n) with replicated
weights.  The only collectives here are the barrier and the max-reduction
that bracket a timed region, and an optional gather of logits for checks.
On ROCm the "nccl" backend is RCCL; "gloo" serves CPU-only test runs.
"""
import os

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` images for `rank` of `world` (sizes differ by at most 1)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.
    Returns (rank, world, local_rank); a lone process (no WORLD_SIZE) returns (0, 1, 0)
    without creating a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VIT_AMD_FORCE_GROUP=1 creates the process group even for a lone rank (a one-GPU box can then exercise the
    # RCCL initialisation, barrier and reduction that a multi-GPU launch goes through)
    force = os.environ.get("VIT_AMD_FORCE_GROUP") == "1" and "WORLD_SIZE" in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # bind the communicator to this rank's GPU: barriers and reductions then never have to guess a device
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device="cpu"):
    """Largest `value` (a float) over all ranks; the identity without a process group."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_rows(local, device=None):
    """Concatenate per-rank [n_r, C] tensors in rank order on every rank (shards may differ in n_r)."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: int(s.item())] for b, s in zip(bufs, sizes)])
