#!/usr/bin/env python3
"""End-to-end cost of the glue classes per call (wall time incl. Python, one MI355X): CorrBlock lookup (20 edges,
48x64), CorrBlock construction, cat / boolean-mask pruning, AltCorrBlock lookup (16 edges over 8 frames, 60x80)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd as lgu  # noqa: E402

dev = "cuda"
torch.manual_seed(0)


def wall(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


with torch.no_grad():
    E, h, w = 20, 48, 64
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
    GA = lgu.GaussianMask(h, w).to(dev)
    f1 = torch.randn(1, E, 128, h, w, device=dev) * 0.5
    f2 = torch.randn(1, E, 128, h, w, device=dev) * 0.5
    ys, xs = torch.meshgrid(torch.arange(h, device=dev).float(), torch.arange(w, device=dev).float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, h, w, 2, device=dev)
    blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
    res = {"CorrBlock.__call__ 20 edges 48x64 (ms, wall incl. Python)": wall(lambda: blk(coords)),
           "CorrBlock.__init__ 20 edges (matmul + offsets convs + fused pyramid)": wall(lambda: lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2), iters=10, warm=2)}
    new = lgu.CorrBlock(ofsMap, ofsRes, GA, f1[:, :4], f2[:, :4])
    mask = torch.ones(24, dtype=torch.bool, device=dev)
    mask[:4] = False

    def grow_and_prune():
        blk.cat(new)      # 4 new edges appended into free slots
        blk[mask]         # the 4 oldest dropped: slot list edit only

    res["CorrBlock.cat(4 edges) + [bool mask] on a 20-edge block (slot store)"] = wall(grow_and_prune, iters=20, warm=3)
    lgu.CorrBlock.TILED_PYRAMID = False
    ref_like = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
    ref_like.corr_pyramid = [v.clone() for v in ref_like.corr_pyramid]   # plain tensors: torch.cat / indexing as the reference does
    new2 = lgu.CorrBlock(ofsMap, ofsRes, GA, f1[:, :4], f2[:, :4])
    new2.corr_pyramid = [v.clone() for v in new2.corr_pyramid]

    def grow_and_prune_ref():
        ref_like.cat(new2)
        ref_like[mask]

    res["the same with plain tensors (torch.cat + boolean indexing, as the reference)"] = wall(grow_and_prune_ref, iters=20, warm=3)
    lgu.CorrBlock.TILED_PYRAMID = True

    N, H, W = 8, 60, 80
    fm = (torch.randn(1, N, 128, H, W, device=dev) * 0.5).half()
    ii = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7], device=dev)
    jj = torch.tensor([1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 0, 0, 1], device=dev)
    ysb, xsb = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
    cb = torch.stack([xsb, ysb], -1)[None, None] + 2 * torch.randn(1, 16, H, W, 2, device=dev)
    alt = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
    res["AltCorrBlock.__call__ 16 edges 60x80 half (offset convs + probe + fused launch)"] = wall(lambda: alt(cb, ii, jj), iters=20, warm=3)
    # dense BA, frontend-sized window: 12 keyframes of 48x64, edges within 3 frames (66 edges), window [2, 12), 2 iterations
    import numpy as np
    N, h, w = 12, 48, 64
    ii_l = [i for i in range(N) for j in range(N) if i != j and abs(i - j) <= 3]
    jj_l = [j for i in range(N) for j in range(N) if i != j and abs(i - j) <= 3]
    iib, jjb = torch.tensor(ii_l, device=dev), torch.tensor(jj_l, device=dev)
    poses = torch.zeros(N, 7, device=dev)
    poses[:, 6] = 1
    poses[:, 0] = torch.arange(N, device=dev) * 0.05
    disps = 0.3 + 0.7 * torch.rand(N, h, w, device=dev)
    intr = torch.tensor([50.0, 50.0, 32.0, 24.0], device=dev)
    ys2, xs2 = torch.meshgrid(torch.arange(h, device=dev).float(), torch.arange(w, device=dev).float(), indexing="ij")
    tgt = torch.stack([xs2, ys2])[None].repeat(len(ii_l), 1, 1, 1) + torch.randn(len(ii_l), 2, h, w, device=dev)
    wgt = torch.rand(len(ii_l), 2, h, w, device=dev)
    eta = torch.full((N, h, w), 1e-3, device=dev)
    sens = torch.zeros_like(disps)

    pbuf, dbuf = poses.clone(), disps.clone()   # persistent state buffers, as depth_video holds them

    def run_ba():
        pbuf.copy_(poses); dbuf.copy_(disps)
        lgu.ba.ba(pbuf, dbuf, intr, sens, tgt, wgt, eta, iib, jjb, 2, N, 2, 1e-4, 0.1, False)

    res["ba: 12 keyframes 48x64, %d edges, 2 iterations (ms, wall incl. host bookkeeping)" % len(ii_l)] = wall(run_ba, iters=10, warm=2)
    # the demo's frontend window (25 keyframes): the reduced system (150 x 150) still fits the single-workgroup LDS solver
    N2 = 26
    ii2 = [i for i in range(N2) for j in range(N2) if i != j and abs(i - j) <= 3]
    jj2 = [j for i in range(N2) for j in range(N2) if i != j and abs(i - j) <= 3]
    ii2t, jj2t = torch.tensor(ii2, device=dev), torch.tensor(jj2, device=dev)
    poses2 = torch.zeros(N2, 7, device=dev)
    poses2[:, 6] = 1
    poses2[:, 0] = torch.arange(N2, device=dev) * 0.05
    disps2 = 0.3 + 0.7 * torch.rand(N2, h, w, device=dev)
    tgt2 = torch.stack([xs2, ys2])[None].repeat(len(ii2), 1, 1, 1) + torch.randn(len(ii2), 2, h, w, device=dev)
    wgt2 = torch.rand(len(ii2), 2, h, w, device=dev)
    eta2 = torch.full((N2, h, w), 1e-3, device=dev)
    sens2 = torch.zeros_like(disps2)
    pb2, db2 = poses2.clone(), disps2.clone()

    def run_ba25():
        pb2.copy_(poses2); db2.copy_(disps2)
        lgu.ba.ba(pb2, db2, intr, sens2, tgt2, wgt2, eta2, ii2t, jj2t, 1, N2, 2, 1e-4, 0.1, False)

    res["ba: 26 keyframes 48x64 (window of 25), %d edges, 2 iterations (ms, wall)" % len(ii2)] = wall(run_ba25, iters=10, warm=2)
    # backend-sized BA: 200 keyframes of 60x80, edges within 5 frames (1970 edges), window [1, 200), 2 iterations
    N, h, w = 200, 60, 80
    ii_l = [i for i in range(N) for j in range(N) if i != j and abs(i - j) <= 5]
    jj_l = [j for i in range(N) for j in range(N) if i != j and abs(i - j) <= 5]
    iib, jjb = torch.tensor(ii_l, device=dev), torch.tensor(jj_l, device=dev)
    poses = torch.zeros(N, 7, device=dev)
    poses[:, 6] = 1
    poses[:, 0] = torch.arange(N, device=dev) * 0.05
    disps = 0.3 + 0.7 * torch.rand(N, h, w, device=dev)
    intr = torch.tensor([60.0, 60.0, 40.0, 30.0], device=dev)
    ys2, xs2 = torch.meshgrid(torch.arange(h, device=dev).float(), torch.arange(w, device=dev).float(), indexing="ij")
    tgt = torch.stack([xs2, ys2])[None].repeat(len(ii_l), 1, 1, 1) + torch.randn(len(ii_l), 2, h, w, device=dev)
    wgt = torch.rand(len(ii_l), 2, h, w, device=dev)
    eta = torch.full((N, h, w), 1e-3, device=dev)
    sens = torch.zeros_like(disps)

    def run_ba_backend():
        lgu.ba.ba(poses.clone(), disps.clone(), intr, sens, tgt, wgt, eta, iib, jjb, 1, N, 2, 1e-4, 0.1, False)

    res["ba: 200 keyframes 60x80, %d edges, 2 iterations (ms, wall incl. host bookkeeping; 1194x1194 system, library Cholesky)" % len(ii_l)] = wall(run_ba_backend, iters=3, warm=1)
print(json.dumps(res, indent=1))
