#!/usr/bin/env python3
"""Kernel timeline of one AltCorrBlock.__call__ (16 edges over 8 half frames, 60x80; autocast off as in
factor_graph.update_lowmem).  Run under rocprofv3 --kernel-trace; the last call lies between the last two launches of
the fused low-memory kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd as lgu  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
with torch.no_grad():
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    N, H, W = 8, 60, 80
    fm = (torch.randn(1, N, 128, H, W, device=dev) * 0.5).half()
    ii = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7], device=dev)
    jj = torch.tensor([1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 0, 0, 1], device=dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
    cb = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, 16, H, W, 2, device=dev)
    alt = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
    for it in range(12):
        if it == 2:
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        out = alt(cb, ii, jj)
    e1.record()
    torch.cuda.synchronize()
    print("AltCorrBlock.__call__ 16 edges 60x80: %.3f ms per call (device)" % (e0.elapsed_time(e1) / 10))
