"""torch-CPU `F.grid_sample` formulation of the deformable pyramid sample.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This is the "reference CPU
F.grid_sample path" that BASELINE.json asks to be timed beside the GPU numbers.  The
reference itself has no CPU implementation and no grid_sample call; the
formulation below is this build's own and is checked against the explicit-gather oracle
(oracle/lgu_oracle.c) in tests/test_oracle.py: bilinear, zeros padding,
align_corners=True, times the reference's "top-left corner in bounds" tap mask
(offersample_LGS/defCorrSample_kernel.cu:67).  grid_sample's coordinate normalisation
costs ~5e-6 absolute, so this path is a timing baseline and a loose (1e-4) cross-check,
not the parity oracle.
"""
import torch
import torch.nn.functional as F


def defcorr_level(volume, coords, offset, radius):
    """One level: volume (E,H1,W1,H2,W2), coords (E,2,H1,W1) already divided by 2^l,
    offset (E,H1,W1,rd,rd,2) or None.  Returns (E,rd,rd,H1,W1)."""
    E, H1, W1, H2, W2 = volume.shape
    rd = 2 * radius + 1
    x0 = coords[:, 0].reshape(E * H1 * W1, 1, 1)
    y0 = coords[:, 1].reshape(E * H1 * W1, 1, 1)
    d = torch.arange(-radius, radius + 1, dtype=volume.dtype)
    di = d.view(1, rd, 1)  # tap index i moves in x
    dj = d.view(1, 1, rd)  # tap index j moves in y
    if offset is not None:
        off = offset.reshape(E * H1 * W1, rd, rd, 2).clone()
        off[:, radius, radius, :] = 0
        ox, oy = off[..., 0] + x0, off[..., 1] + y0
    else:
        ox, oy = x0.expand(-1, rd, rd), y0.expand(-1, rd, rd)
    px = torch.floor(ox) + di + (ox - torch.floor(ox))
    py = torch.floor(oy) + dj + (oy - torch.floor(oy))
    gx = 2.0 * px / max(W2 - 1, 1) - 1.0
    gy = 2.0 * py / max(H2 - 1, 1) - 1.0
    grid = torch.stack([gx, gy], dim=-1)  # (N, rd(i), rd(j), 2)
    img = volume.reshape(E * H1 * W1, 1, H2, W2)
    s = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)[:, 0]
    x1 = torch.floor(ox) + di
    y1 = torch.floor(oy) + dj
    mask = (x1 >= 0) & (x1 < W2) & (y1 >= 0) & (y1 < H2)
    s = s * mask.to(s.dtype)
    return s.view(E, H1, W1, rd, rd).permute(0, 3, 4, 1, 2).contiguous()


def defcorr_pyramid(volumes, coords, offsets, radius):
    """4× defcorr_level with coords / 2^l, concatenated like corr.py:109 → (E, L*rd*rd, H1, W1)."""
    outs = []
    E, _, H1, W1 = coords.shape
    for l, v in enumerate(volumes):
        o = defcorr_level(v, coords / 2 ** l, offsets[l], radius)
        outs.append(o.view(E, -1, H1, W1))
    return torch.cat(outs, dim=1)
