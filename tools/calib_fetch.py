#!/usr/bin/env python3
"""PMC calibration (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`): launches whose
HBM read bytes are known exactly, in this kernel's own access pattern, so the gfx950
FETCH_SIZE correction factor can be read off (MI355X_MICROARCH.md §HBM).
  A. pyramid kernel on two SMALL levels only (12x16 and 6x8 slices, no offsets): every
     slice is staged whole by LDS-DMA, once -> reads = E*3072*(768+192) B + coords.
  B. torch elementwise copy of the 755 MB level-0 volume (wide coalesced streaming read).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd  # noqa: E402

dev = torch.device("cuda:0")
E, H1, W1 = 20, 48, 64
v2 = torch.randn(E, H1, W1, 12, 16, device=dev)
v3 = torch.randn(E, H1, W1, 6, 8, device=dev)
big = torch.randn(E, H1, W1, 48, 64, device=dev)
ys, xs = torch.meshgrid(torch.arange(H1, device=dev).float(), torch.arange(W1, device=dev).float(), indexing="ij")
coords = (torch.stack([xs, ys])[None] + 3 * torch.randn(E, 2, H1, W1, device=dev)).contiguous() / 4
for _ in range(3):
    flush = big * 1.0001           # B: streaming read (and evicts v2/v3 from the 256 MB L3)
    out = lgu_slam_amd.ops.defcorr_pyramid_forward([v2, v3], coords, [None, None], 3)   # A
torch.cuda.synchronize()
print("expected_read_bytes_A", E * H1 * W1 * (768 + 192) + coords.numel() * 4)
print("expected_write_bytes_A", out.numel() * 4)
print("expected_read_bytes_B", big.numel() * 4)
