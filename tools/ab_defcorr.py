#!/usr/bin/env python3
"""Interleaved A/B of the fused-sampler kernel variants in ONE process on one device
(cdna_hip_programming.md rule 24): N configs x M rounds, HIP-event time per launch, median/min."""
import json
import os

os.environ.setdefault("LGU_DEBUG_KNOBS", "1")   # this tool switches kernel variants through the library's debug variables
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lgu_slam_amd  # noqa: E402

ops = lgu_slam_amd.ops
dev = torch.device("cuda:0")
E, H1, W1, L, R = 20, 48, 64, 4, 3
vols, coords, offs = bench.make_inputs(E, H1, W1, L, R, 1234, dev)
out = torch.empty(E, 196, H1, W1, device=dev)
tvols = [ops.volume_retile(v) for v in vols]
hw = [(H1 >> l, W1 >> l) for l in range(L)]
configs = [dict(variant=v, tiled=t, probe=pr) for t in (True, False) for v in ((0, 5) if t else (0, 5, 3, 1)) for pr in (False, True)]
times = {i: [] for i in range(len(configs))}


def run(c, iters):
    os.environ["LGU_DEFCORR_VARIANT"] = str(c["variant"])
    v = tvols if c["tiled"] else vols
    for _ in range(3):
        ops.defcorr_pyramid_forward(v, coords, offs, R, probe=c["probe"], out=out, tiled=c["tiled"], level_hw=hw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.defcorr_pyramid_forward(v, coords, offs, R, probe=c["probe"], out=out, tiled=c["tiled"], level_hw=hw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for rnd in range(8):
    for i, c in enumerate(configs):
        times[i].append(run(c, 50))
for i, c in enumerate(configs):
    t = np.array(times[i])
    print(json.dumps(dict(c, us_median=float(np.median(t)), us_min=float(t.min()),
                          Mpix_edges_per_s=E * H1 * W1 / float(np.median(t)))))
