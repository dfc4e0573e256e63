"""`python bench.py --gpus N` as one command: the parent starts one child per rank and forwards rank 0's line.
Runs on CPU through the rehearsal switch (gloo, no GPU work)."""
import json
import os
import subprocess
import sys

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
    # without the rehearsal switch and without a GPU every rank stops before any timing: the parent must report it
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]



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
