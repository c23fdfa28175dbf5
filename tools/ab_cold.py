#!/usr/bin/env python3
"""Interleaved A/B of metric-kernel variants in ONE process on one device, cold cache (rotating input sets, as
bench.py's headline): configs x rounds, HIP-event time per launch, median / min.
Usage: ab_cold.py [variants, default "0,6"] [rounds] [layouts, default "tiled,rowmajor"] [probe values, default "0,1"]"""
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
# variant[:prefetch distance], e.g. "0:0,0:48,6"
# fourth field: LGU_LDS_PAD (extra LDS bytes per workgroup: limits workgroups per CU); the third field is unused (it
# selected the launch bound of the lean kernel while both bounds were built)
def _spec(v):
    f = [int(x) for x in v.split(":")]
    return tuple(f + [-1, -1, 0][len(f) - 1:])


variants = [_spec(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,6").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
layouts = (sys.argv[3] if len(sys.argv) > 3 else "tiled,rowmajor").split(",")
probes = [bool(int(v)) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,1").split(",")]
sets = bench.DefcorrSets(ops, 20, 4, 1234, dev, keep_rowmajor="rowmajor" in layouts)
configs = [dict(variant=v, pf=pf, lw=lw, pad=pad, layout=lay, probe=pr) for lay in layouts for pr in probes for v, pf, lw, pad in variants]


def select(c):
    os.environ["LGU_DEFCORR_VARIANT"] = str(c["variant"])
    if c["pf"] >= 0:
        os.environ["LGU_DEFCORR_PF"] = str(c["pf"])
    else:
        os.environ.pop("LGU_DEFCORR_PF", None)
    os.environ["LGU_LDS_PAD"] = str(c["pad"])


plans = {(lay, pr): sets.plans(lay, pr, "planar") for lay in layouts for pr in probes}
times = {i: [] for i in range(len(configs))}
ref = {}
for i, c in enumerate(configs):  # results of every variant agree bit for bit (fresh offsets each)
    select(c)
    sets.restore_offsets()
    out = torch.empty_like(sets.out)
    plans[(c["layout"], c["probe"])][0](sets.coords[0], out=out)
    torch.cuda.synchronize()
    key = c["probe"]
    if key not in ref:
        ref[key] = out
    else:
        assert torch.equal(ref[key], out), "variant %r differs" % (c,)
sets.restore_offsets()
for rnd in range(rounds):
    for i, c in enumerate(configs):
        select(c)
        step = sets.stepper(plans[(c["layout"], c["probe"])], sets.out, True)
        sets.restore_offsets()
        for j in range(8):
            step(j)
        # EVERY block starts from the original offsets: a probe-on launch scales offset[1] in place (corr.py:99), and
        # probe-off configurations timed after it would otherwise sample a shrunken level-1 footprint (fewer lines, up to
        # 15 % faster: the first A/B files of round 2 carry that bias in their probe-off rows, see profiles/README.md)
        n = 16 if c["probe"] else 100
        times[i] += [x * 1e3 for x in bench.time_blocks(step, n, 6 if c["probe"] else 1, sets.restore_offsets)]
for i, c in enumerate(configs):
    t = np.array(times[i])
    print(json.dumps(dict(c, us_median=round(float(np.median(t)), 2), us_min=round(float(t.min()), 2),
                          Mpix_edges_per_s=round(sets.units / float(np.median(t)), 1))))
