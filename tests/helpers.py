"""Shared test helpers: golden fixtures, synthetic models, oracle access."""
import json
import os

import numpy as np
import torch

from vit_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    data = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return meta, data


def golden_inputs(meta):
    """(cfg, state_dict, images) the golden `meta` was captured on."""
    cfg = synth.ARCHS[meta["arch"]]
    sd = synth.synth_state_dict(cfg, meta["img_size"], seed=meta["seed"], style=meta["style"])
    x = synth.synth_images(meta["batch"], meta["img_size"], seed=meta["image_seed"],
                           indices=meta["image_indices"])
    return cfg, sd, x


def build_model(arch, img_size, sd, device, precision="bf16"):
    """A vit_amd model holding `sd`, built on the meta device to skip the random init."""
    import vit_amd
    with torch.device("meta"):
        m = getattr(vit_amd, arch)(img_size=img_size)
    m = m.to_empty(device=device)
    m.load_state_dict(sd, strict=True)
    if precision != "bf16":
        m.set_precision(precision)
    return m.eval()
