#!/usr/bin/env python3
"""DIAGNOSTIC ONLY (never the product library, never timed for a result): phase stamps of the
low-memory tile kernel — where one workgroup's lifetime goes, per pyramid level."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "liblgu_diag.so"))
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
    s = stamps.cpu().numpy().astype(np.float64) * 10e-3  # 100 MHz ticks -> microseconds
    d = np.diff(s[:, :6], axis=1)
    tot = s[:, 5] - s[:, 0]
    span = s[:, 5].max() - s[:, 0].min()
    print("level %d: kernel span %.1f us, WG lifetime median %.1f us | phase0+boxes %.1f | window+lpos+prefetch0 %.1f | chunk loop %.1f | sample %.1f | write-out %.1f"
          % (l, span, np.median(tot), *[np.median(d[:, i]) for i in range(5)]))
