#!/usr/bin/env python3
"""DIAGNOSTIC ONLY (never the product library, never timed for a result): phase stamps of the
low-memory tile kernel — where one workgroup's lifetime goes, per pyramid level."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "liblgu_diag2.so"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, C = 16, 60, 80, 128
f1 = (torch.randn(B, H, W, C, device=dev) * 0.125).contiguous()
ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
base = torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device=dev)
off0 = (4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev))).contiguous()
vp = ctypes.c_void_p
for l in range(4):
    f2 = (torch.randn(B, H >> l, W >> l, C, device=dev) * 0.125).contiguous()
    cl = (base / 2 ** l).contiguous()
    corr = torch.empty(B, 1, 7, 7, H, W, device=dev)
    nwg = B * ((W + 15) // 16) * ((H + 3) // 4)
    stamps = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    lib.lgu_diag_set_stamps(vp(stamps.data_ptr()))
    for it in range(2):
        rc = lib.lgu_diag_lowmem(vp(f1.data_ptr()), vp(f2.data_ptr()), vp(cl.data_ptr()), vp(off0.data_ptr()), vp(corr.data_ptr()),
                                 B, 1, H, W, H >> l, W >> l, C, 3, None)
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64) * 10e-3  # us
    print("level %d: per-WG sums over the chunk loop (us, wave 0): wait-at-top-barrier %.1f | LDS writes (incl. waiting for prefetched data) %.1f | barrier-2 %.1f | issue next prefetch %.1f | compute %.1f"
          % (l, *[np.median(s[:, i]) for i in range(5)]))
