#!/usr/bin/env python3
"""Per-level and fused timings of the half-feature low-memory path (BASELINE config 4 shapes: 16 edges, 60x80x128)
for settings "MT[:c]" given on the command line (default "1,2"; MT = LGU_LOWMEM_MT, ":c" = fmap2 in the chunk-planar
layout), interleaved in one process."""
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
os.environ["LGU_LOWMEM_COOP"] = "0"   # this tool compares variants of the one-wave-per-block kernel (lowmem_mfma.hip)
dev = torch.device("cuda:0")
mts = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
S = bench.lowmem_setup(ops, dev, 16, 4321)
L = S["L"]
offs = [S["o0"], S["o1"], None, None]
f2c = [ops.lowmem_chunked(f) for f in S["f2s"]]
plans, plans_c = {"all": ops.LowmemPyramidPlan(S["f1"], S["f2s"], offs, 3)}, {"all": ops.LowmemPyramidPlan(S["f1"], f2c, offs, 3, chunked=True)}
for l in range(L):
    plans[l] = ops.LowmemPyramidPlan(S["f1"], [S["f2s"][l]], [offs[l]], 3, lbase=l)
    plans_c[l] = ops.LowmemPyramidPlan(S["f1"], [f2c[l]], [offs[l]], 3, lbase=l, chunked=True)
outs = {"all": S["out"]}
for l in range(L):
    outs[l] = torch.empty(16, 1, 49, S["H1"], S["W1"], device=dev)
res = {(m, k): [] for m in mts for k in plans}
for rnd in range(5):
    for m in mts:
        os.environ["LGU_LOWMEM_MT"] = m.split(":")[0]
        for k, pl in (plans_c if m.endswith(":c") else plans).items():
            for _ in range(3):
                pl(S["coords"], out=outs[k])
            res[(m, k)] += bench.time_blocks(lambda i: pl(S["coords"], out=outs[k]), 20, 1)
for m in mts:
    print(json.dumps({"LGU_LOWMEM_MT": m, **{"us_" + str(k): round(float(np.median(res[(m, k)])) * 1e3, 1) for k in plans}}))
