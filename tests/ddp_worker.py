"""One rank of the two-rank data-parallel training test (started by tests/test_train_ddp.py, never collected).

Both ranks share the box's one GPU and talk over gloo (RCCL refuses two ranks on one device); the model is
wrapped in DistributedDataParallel exactly as the reference wraps it (main.py:125-126).  Rank r takes images
[r*b, (r+1)*b) of a 2b-image batch; after backward every rank holds the average of the two ranks' gradients,
which must equal the gradient of one process on all 2b images (CrossEntropyLoss averages over the batch).
Rank 0 prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import vit_amd                                       # noqa: E402
from vit_amd import dist as vdist                    # noqa: E402
from vit_amd import synth                            # noqa: E402

CFG = dict(patches=16, hidden_size=256, num_layers=2, num_heads=4, mlp_ratio=4, eps=1e-6)
IMG, CLASSES, PER_RANK = 64, 10, 4


def build(precision, sd):
    m = vit_amd.VisionTransformer(img_size=IMG, patches=CFG["patches"], num_classes=CLASSES, hidden_size=CFG["hidden_size"],
                                  num_layers=CFG["num_layers"], num_heads=CFG["num_heads"], mlp_ratio=CFG["mlp_ratio"],
                                  norm=lambda d: torch.nn.LayerNorm(d, eps=CFG["eps"]), precision=precision)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def main():
    precision = sys.argv[1]
    torch.cuda.set_device(0)
    rank, world, _ = vdist.init_from_env(backend="gloo")
    sd = synth.synth_state_dict(CFG, IMG, seed=3, style="perturbed", num_classes=CLASSES)
    total = world * PER_RANK
    x = synth.synth_images(total, IMG, seed=4).cuda()
    target = (torch.arange(total) * 7 % CLASSES).cuda()
    crit = torch.nn.CrossEntropyLoss()
    lo = rank * PER_RANK
    net = torch.nn.parallel.DistributedDataParallel(build(precision, sd), device_ids=[0])
    loss = crit(net(x[lo:lo + PER_RANK]), target[lo:lo + PER_RANK])
    loss.backward()
    torch.cuda.synchronize()
    mean_loss = vdist.sum_over_ranks(float(loss)) / world
    if rank == 0:
        one = build(precision, sd)
        loss1 = crit(one(x), target)
        loss1.backward()
        worst, worst_name = 0.0, ""
        for (n, p), q in zip(one.named_parameters(), net.module.parameters()):
            err = float((p.grad - q.grad).norm() / p.grad.norm().clamp_min(1e-12))
            if err > worst:
                worst, worst_name = err, n
        print(json.dumps({"world": world, "loss_ddp": mean_loss, "loss_one": float(loss1), "worst_rel": worst,
                          "worst_name": worst_name}), flush=True)
    vdist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
