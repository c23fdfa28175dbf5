#!/usr/bin/env python3
"""Interleaved A/B of the low-memory path (BASELINE config 4: 16 edges, 60x80x128 half maps, 4 levels, chunk-planar target
maps) over environment settings given as "NAME=V,NAME=V;NAME=V;..." (";" separates variants; "" = defaults).
Device time per call, median of 5 rounds of 20 calls each."""
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
variants = (sys.argv[1] if len(sys.argv) > 1 else ";LGU_LOWMEM_COOP=0").split(";")
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = bench.lowmem_setup(ops, dev, edges, 4321)
res = {v: [] for v in variants}
keys = set(k.split("=")[0] for v in variants for k in v.split(",") if k)
ref = None
for rnd in range(5):
    for v in variants:
        for k in keys:
            os.environ.pop(k, None)
        for kv in v.split(","):
            if kv:
                k, val = kv.split("=")
                os.environ[k] = val
        for _ in range(3):
            S["plan"](S["coords"], out=S["out"])
        res[v] += bench.time_blocks(lambda i: S["plan"](S["coords"], out=S["out"]), 20, 1)
        if rnd == 0:
            o = S["out"].clone()
            if ref is None:
                ref = o
            else:
                assert float((o - ref).abs().max()) <= 2e-5, (v, float((o - ref).abs().max()))
for v in variants:
    print(json.dumps({"env": v or "(defaults)", "edges": edges, "us_per_call": round(float(np.median(res[v])) * 1e3, 1),
                      "Mpix_edges_per_s": round(S["units"] / float(np.median(res[v])) / 1e3, 1)}))
