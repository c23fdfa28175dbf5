#!/usr/bin/env python3
"""Driver for rocprofv3 runs of the low-memory path (BASELINE config 4 shapes: 60x80, C=128, B=16)."""
import os

os.environ.setdefault("LGU_DEBUG_KNOBS", "1")   # this tool switches kernel variants through the library's debug variables
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd  # noqa: E402

ops = lgu_slam_amd.ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, C = 16, 60, 80, 128
f1 = (torch.randn(B, H, W, C, device=dev) * 0.125).contiguous()
ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
base = torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device=dev)
off0 = (4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev))).contiguous()
zero = torch.zeros_like(off0)
for l in range(4):
    f2 = (torch.randn(B, H >> l, W >> l, C, device=dev) * 0.125).contiguous()
    cl = (base / 2 ** l).contiguous()
    f1h, f2h = f1.half(), f2.half()
    for _ in range(3):
        ops.lowMem_defSample(f1, f2, cl, off0 if l < 2 else zero, 3)
    os.environ["LGU_LOWMEM_VARIANT"] = "2"
    for _ in range(3):
        ops.lowMem_defSample(f1, f2, cl, off0 if l < 2 else zero, 3)
    os.environ.pop("LGU_LOWMEM_VARIANT")
    for _ in range(5):
        ops.lowMem_defSample_mixed(f1h, f2h, cl, off0 if l < 2 else zero, 3)
torch.cuda.synchronize()
# all four levels in one launch (what AltCorrBlock issues for half feature buffers)
f1h = f1.half()
f2hs = [(torch.randn(B, H >> l, W >> l, C, device=dev) * 0.125).half().contiguous() for l in range(4)]
off1 = ((4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev)) + off0) / 2).contiguous()
plan = ops.LowmemPyramidPlan(f1h, f2hs, [off0, off1, None, None], 3)
for _ in range(5):
    plan(base.contiguous())
torch.cuda.synchronize()
