"""`python bench.py --gpus N` as one command: the parent starts one child per rank and forwards rank 0's line.
Runs on CPU through the rehearsal switch (gloo, no GPU work)."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_self_launches_two_ranks_and_prints_one_line():
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1


def test_gpus_1_stays_a_single_process():
    r = _run("--gpus", "1", "--steps", "2", "--warmup", "1", "--rehearse")
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_failed_rank_gives_non_zero_exit():
    # without the rehearsal switch every rank whose LOCAL_RANK has no GPU stops before any timing (all of them on a
    # CPU box, rank 1 on a one-GPU box, where rank 0 would otherwise sit in the rendezvous): the parent must end
    # the survivors and report the failure, within seconds
    t0 = time.time()
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "exited first with status" in r.stderr
    assert time.time() - t0 < 120


def test_rank_that_never_arrives_is_ended_at_the_deadline():
    # rank 0 sleeps instead of joining the rendezvous: its stdout stays silent and open.  The parent's deadline
    # covers rank 0 as well; it ends every child and returns non-zero shortly after the deadline.
    t0 = time.time()
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse", "--rehearse-hang-rank", "0",
             "--launch-timeout", "20")
    took = time.time() - t0
    assert r.returncode != 0
    assert "launch exceeded 20 s" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert took < 90, took


def test_rank_that_dies_ends_a_waiting_rank_zero():
    # rank 1 never arrives because it is dead: rank 0 waits in the gloo rendezvous until the parent ends it
    t0 = time.time()
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse", "--rehearse-die-rank", "1")
    assert r.returncode != 0
    assert "rank 1 exited first with status 3" in r.stderr
    assert time.time() - t0 < 90



@pytest.mark.gpu
def test_rccl_group_initialises_and_reduces_on_the_gpu_with_one_rank():
    """The multi-GPU launch's RCCL path (process group bound to the rank's GPU, barrier, max-reduction on the
    device) on a one-GPU box: a lone rank with the group forced on.  torch.distributed's "nccl" backend is RCCL."""
    r = _run("--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533",
                  "VIT_AMD_FORCE_GROUP": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0


@pytest.mark.gpu
def test_training_step_under_ddp_over_rccl_on_the_gpu_with_one_rank():
    """`--mode train` through DistributedDataParallel on the RCCL backend (main.py:125-126 of the reference): a lone
    rank with the group forced on, so that DDP's bucket all-reduce runs over RCCL on real hardware."""
    r = _run("--gpus", "1", "--mode", "train", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534",
                  "VIT_AMD_FORCE_GROUP": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "train" in out["metric"]
    assert "RCCL" in out["config"]["parallelism"]
