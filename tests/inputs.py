"""Seeded synthetic inputs shared by the oracle tests, the GPU parity tests and bench.py.

Distributions follow SURVEY.md §8(d): coords = pixel grid + N(0, sigma^2),
offset_0 = 4*tanh(N(0,1)), offset_1 = (4*tanh(N(0,1)) + offset_0)/2, deeper levels zero
(reference corr.py:126-135), volumes ~ N(0,1) for kernel-level tests.
"""
import numpy as np


def grid_coords(rng, E, H1, W1, sigma):
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    c = np.stack([xs, ys])[None].repeat(E, 0)
    return (c + rng.standard_normal((E, 2, H1, W1)).astype(np.float32) * sigma).astype(np.float32)


def canonical_offsets(rng, E, H1, W1, L, radius=3, scale=4.0):
    rd = 2 * radius + 1
    o0 = (scale * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2)))).astype(np.float32)
    o1 = ((scale * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2))) + o0) / 2).astype(np.float32)
    offs = [o0, o1] + [None] * max(0, L - 2)
    return offs[:L]


def volume_pyramid(rng, E, H1, W1, L, H2=None, W2=None):
    H2 = H1 if H2 is None else H2
    W2 = W1 if W2 is None else W2
    return [rng.standard_normal((E, H1, W1, H2 >> l, W2 >> l)).astype(np.float32) for l in range(L)]


def pyramid_case(seed, E, H1, W1, L, radius=3, sigma=3.0, off_scale=4.0, dense_offsets=False):
    """Returns dict(volumes, coords, offsets) of numpy arrays; offsets[l] may be None."""
    rng = np.random.default_rng(seed)
    vols = volume_pyramid(rng, E, H1, W1, L)
    coords = grid_coords(rng, E, H1, W1, sigma)
    offs = canonical_offsets(rng, E, H1, W1, L, radius, off_scale)
    if dense_offsets:  # every level carries a real offset tensor
        rd = 2 * radius + 1
        offs = [o if o is not None else (off_scale * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2)))).astype(np.float32)
                for o in offs]
    return dict(volumes=vols, coords=coords, offsets=offs)


def fmap_case(seed, B, S, H1, W1, H2, W2, C, radius, sigma=3.0, scale_down=1.0, n_offset=None, off_scale=4.0):
    rng = np.random.default_rng(seed)
    rd = 2 * radius + 1
    f1 = (rng.standard_normal((B, H1, W1, C)) * 0.5 / 4).astype(np.float32)
    f2 = (rng.standard_normal((B, H2, W2, C)) * 0.5 / 4).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    c = np.stack([xs, ys], -1)[None, None].repeat(B, 0).repeat(S, 1)  # (B,S,H1,W1,2)
    c = ((c + rng.standard_normal(c.shape) * sigma) * scale_down).astype(np.float32)
    no = B if n_offset is None else n_offset
    off = (off_scale * np.tanh(rng.standard_normal((no, H1, W1, rd, rd, 2)))).astype(np.float32)
    return dict(fmap1=f1, fmap2=f2, coords=c, offset=off)
