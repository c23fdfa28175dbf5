#!/usr/bin/env python3
"""DIAGNOSTIC: does the time per launch of the metric kernel depend on how many launches run back to back?
(bench.py times blocks of 200, tools/ab_cold.py interleaved blocks of 100: they disagree by 3-7 % tiled, more row-major.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lgu_slam_amd  # noqa: E402

ops = lgu_slam_amd.ops
dev = torch.device("cuda:0")
sets = bench.DefcorrSets(ops, 20, 4, 1234, dev, keep_rowmajor=True)
for layout in ("tiled", "rowmajor"):
    plans = sets.plans(layout, False, "planar")
    step = sets.stepper(plans, sets.out, True)
    for j in range(20):
        step(j)
    torch.cuda.synchronize()
    for n in (25, 50, 100, 200, 400, 800, 1600, 100, 25):
        time.sleep(0.2)
        ms = bench.time_blocks(step, n, 1)[0]
        print("%-9s block of %4d launches: %.2f us per launch" % (layout, n, ms * 1e3), flush=True)
