#!/usr/bin/env python3
"""Kernel breakdown of CorrBlock.__init__ (run under rocprofv3 --kernel-trace --stats): the all-pairs matmul, the two
offset convolutions, the Gaussian parameters and the fused pyramid builder, for E edges at 48x64.
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_init.py [E] [half]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd as lgu  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 20
half = len(sys.argv) > 2 and sys.argv[2] == "half"
dev = "cuda"
torch.manual_seed(0)
h, w = 48, 64
with torch.no_grad():
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    GA = lgu.GaussianMask(h, w).to(dev)
    f1 = torch.randn(1, E, 128, h, w, device=dev) * 0.5
    f2 = torch.randn(1, E, 128, h, w, device=dev) * 0.5
    if half:
        f1, f2 = f1.half(), f2.half()
    for it in range(12):
        if it == 2:
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        with torch.autocast("cuda", dtype=torch.float16, enabled=half):
            blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
    e1.record()
    torch.cuda.synchronize()
    print("CorrBlock.__init__ E=%d half=%s: %.3f ms per construction (device)" % (E, half, e0.elapsed_time(e1) / 10))
