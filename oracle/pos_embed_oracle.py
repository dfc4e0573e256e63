"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's position-embedding rescale.

Follows /root/reference/utils.py:241-261 (`pos_embed_scale`): the cls row is kept, the grid rows are
reshaped to [g, g, D] and resampled to [G, G, D] with `scipy.ndimage.zoom(order=1, mode='constant')`, whose
order-1 resampling places output index o at input coordinate o * (g - 1) / (G - 1) on each zoomed axis and
interpolates linearly (scipy 1.x `zoom` with grid_mode=False; the channel axis has zoom 1 and stays as is).
Pinned by tests/golden/pos_embed_scale.npz, captured from the reference function itself
(tests/golden/make_golden.py pos_embed).  Only tests may import this module.
"""
import numpy as np


def pos_embed_scale(pos_embed, img_size, patch_size):
    """pos_embed: float array [1, 1 + g*g, D] -> [1, 1 + G*G, D] with G = img_size // patch_size."""
    pe = np.asarray(pos_embed, dtype=np.float64)
    n_model = (img_size * img_size) // (patch_size * patch_size) + 1
    if pe.shape[1] == n_model:
        return np.asarray(pos_embed, dtype=np.float32)
    tok, grid = pe[:, :1], pe[0, 1:]
    g = int(np.sqrt(len(grid)))
    G = int(np.sqrt(n_model - 1))
    grid = grid.reshape(g, g, -1)
    out = np.empty((G, G, grid.shape[2]))
    for oi in range(G):
        ci = oi * (g - 1) / (G - 1) if G > 1 else 0.0
        i0 = min(int(np.floor(ci)), g - 1)
        i1 = min(i0 + 1, g - 1)
        fi = ci - i0
        for oj in range(G):
            cj = oj * (g - 1) / (G - 1) if G > 1 else 0.0
            j0 = min(int(np.floor(cj)), g - 1)
            j1 = min(j0 + 1, g - 1)
            fj = cj - j0
            out[oi, oj] = ((1 - fi) * (1 - fj) * grid[i0, j0] + (1 - fi) * fj * grid[i0, j1]
                           + fi * (1 - fj) * grid[i1, j0] + fi * fj * grid[i1, j1])
    return np.concatenate([tok, out.reshape(1, G * G, -1)], axis=1).astype(np.float32)
