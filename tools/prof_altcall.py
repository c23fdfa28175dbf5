#!/usr/bin/env python3
"""Where AltCorrBlock.__call__ spends its time (16 edges over 8 frames of 60x80x128 half maps, the chunk of update_lowmem):
host wall per call with the queue kept full, device time per call, and a cProfile of the host side.
    python tools/prof_altcall.py            (under rocprofv3 --kernel-trace --stats for the kernel list)"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd as lgu  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(7)
N, H, W = 8, 60, 80
with torch.no_grad():
    fm = (torch.randn(1, N, 128, H, W, device=dev, generator=g) * 0.5).half()
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    ii = torch.arange(16, device=dev) // 2
    jj = (ii + 1 + torch.arange(16, device=dev) % 2) % N
    ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, 16, H, W, 2, device=dev, generator=g)).contiguous()
    blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
    for _ in range(20):
        blk(coords, ii, jj)
    torch.cuda.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        blk(coords, ii, jj)
    e1.record()
    t_issue = time.perf_counter() - t0
    e1.synchronize()
    print("per call: host issue %.1f us, device span %.1f us" % (t_issue / n * 1e6, e0.elapsed_time(e1) / n * 1e3))
    if len(sys.argv) > 1 and sys.argv[1] == "cprofile":
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            blk(coords, ii, jj)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(35)
