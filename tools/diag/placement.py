#!/usr/bin/env python3
"""DIAGNOSTIC: does the metric kernel's time depend on where its buffers sit?  (config 3 moved 97 <-> 109 us between
two runs.)  Builds the E = 20 and E = 40 cold sets behind dummy allocations of different sizes and times each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lgu_slam_amd  # noqa: E402

ops = lgu_slam_amd.ops
dev = torch.device("cuda:0")
for E, nsets in ((20, 4), (40, 3)):
    for pad_mb in (0, 1, 3, 64, 129, 517):
        torch.cuda.empty_cache()
        pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
        sets = bench.DefcorrSets(ops, E, nsets, 1234, dev)
        plans = sets.plans("tiled", False, "planar")
        step = sets.stepper(plans, sets.out, True)
        for j in range(16):
            step(j)
        ms = bench.time_blocks(step, 100, 3)
        ptr = sets.tiled[0][0].data_ptr()
        print("E=%d pad %4d MB: %.2f us per launch (blocks %s), level-0 buffer of set 0 at 0x%x (mod 2 MiB = %d KiB)" %
              (E, pad_mb, sorted(ms)[1] * 1e3, [round(x * 1e3, 1) for x in ms], ptr, (ptr % (2 << 20)) >> 10), flush=True)
        del sets, plans, step, pad
