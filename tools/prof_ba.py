#!/usr/bin/env python3
"""Driver for rocprofv3 runs of the dense BA: backend scale by default (200 keyframes of 60x80, 1970 edges within 5
frames, window [1, 200)), `frontend` as first argument for 12 keyframes of 48x64, edges within 3 frames, window [2, 12);
2 iterations either way."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd as lgu  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
FRONT = len(sys.argv) > 1 and sys.argv[1] == "frontend"
N, h, w, span, T0 = (12, 48, 64, 3, 2) if FRONT else (200, 60, 80, 5, 1)
ii_l = [i for i in range(N) for j in range(N) if i != j and abs(i - j) <= span]
jj_l = [j for i in range(N) for j in range(N) if i != j and abs(i - j) <= span]
ii, jj = torch.tensor(ii_l, device=dev), torch.tensor(jj_l, device=dev)
poses = torch.zeros(N, 7, device=dev)
poses[:, 6] = 1
poses[:, 0] = torch.arange(N, device=dev) * 0.05
disps = 0.3 + 0.7 * torch.rand(N, h, w, device=dev)
intr = torch.tensor([60.0, 60.0, 40.0, 30.0], device=dev)
ys, xs = torch.meshgrid(torch.arange(h, device=dev).float(), torch.arange(w, device=dev).float(), indexing="ij")
tgt = torch.stack([xs, ys])[None].repeat(len(ii_l), 1, 1, 1) + torch.randn(len(ii_l), 2, h, w, device=dev)
wgt = torch.rand(len(ii_l), 2, h, w, device=dev)
eta = torch.full((N, h, w), 1e-3, device=dev)
sens = torch.zeros_like(disps)
for rep in range(12 if FRONT else 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lgu.ba.ba(poses.clone(), disps.clone(), intr, sens, tgt, wgt, eta, ii, jj, T0, N, 2, 1e-4, 0.1, False)
    torch.cuda.synchronize()
    print("ba call %d: %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3))
