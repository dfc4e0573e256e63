"""TEST INFRASTRUCTURE ONLY: CPU restatements of the two steps either side of the reference's forward.

`normalize_u8` follows /root/reference/data.py:54-55,69-72 (DataFetcher): `x.float().sub_(mean).div_(std)` with
mean/std in uint8 pixel units, all in float32.  Parity unpinned by reference outputs: the DataFetcher needs a CUDA
device and the `mc` module at import, so no fixture could be captured from it; the restatement is its three
float32 operations in the same order.

`accuracy` follows /root/reference/utils.py:182-195: top-maxk indices per row, compared with the target, summed over
the first k ranks and scaled to percent.  Pinned by tests/golden/accuracy.npz, captured from the reference function
itself (tests/golden/make_golden.py accuracy).  `target_rank` is the quantity the GPU kernel returns; the
reference's result is `(rank < k).sum() * 100 / batch`.

Only tests may import this module.
"""
import numpy as np

PIXEL_MEAN = np.array([0.485 * 255, 0.456 * 255, 0.406 * 255], dtype=np.float32)
PIXEL_STD = np.array([0.229 * 255, 0.224 * 255, 0.225 * 255], dtype=np.float32)


def normalize_u8(images, mean=PIXEL_MEAN, std=PIXEL_STD):
    """uint8 [B,3,H,W] -> float32 [B,3,H,W]."""
    x = np.asarray(images).astype(np.float32)
    m = np.asarray(mean, dtype=np.float32).reshape(1, 3, 1, 1)
    s = np.asarray(std, dtype=np.float32).reshape(1, 3, 1, 1)
    return ((x - m) / s).astype(np.float32)


def target_rank(output, target):
    """Position of the target class in the descending order of each row, ties towards the lower index."""
    out = np.asarray(output, dtype=np.float32)
    tgt = np.asarray(target, dtype=np.int64)
    rank = np.empty(len(tgt), dtype=np.int32)
    for r, t in enumerate(tgt):
        if t < 0 or t >= out.shape[1]:
            rank[r] = out.shape[1]
            continue
        row = out[r]
        rank[r] = int((row > row[t]).sum() + (row[:t] == row[t]).sum())
    return rank


def accuracy(output, target, topk=(1,)):
    """Precision@k in percent, one float per k."""
    out = np.asarray(output, dtype=np.float32)
    tgt = np.asarray(target, dtype=np.int64)
    maxk = max(topk)
    pred = np.argsort(-out, axis=1, kind="stable")[:, :maxk]          # descending, ties towards the lower index
    correct = pred == tgt[:, None]
    return [float(correct[:, :k].sum()) * 100.0 / len(tgt) for k in topk]
