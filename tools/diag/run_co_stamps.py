#!/usr/bin/env python3
"""DIAGNOSTIC ONLY (never the product library, never timed for a result): per-wave, per-level phase stamps of the
cooperative low-memory kernel (lgu-slam_amd/csrc/lowmem_coop.hip built with -DLGU_MM_STAMPS), BASELINE config 4 shapes."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "liblgu_codiag.so")
CSRC = os.path.join(ROOT, "lgu-slam_amd", "csrc")
EXTRA = [a for a in sys.argv[1:] if a.startswith("-D")]   # e.g. -DLGU_CO_DIAG_SAMELOAD, -DLGU_CO_DIAG_NOSCATTER (timing experiments, wrong results)
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
subprocess.check_call(["hipcc", "-Wno-unused-value", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                       "-DLGU_MM_STAMPS", "-o", SO, os.path.join(CSRC, "lowmem_coop.hip"), os.path.join(CSRC, "capi.hip")] + EXTRA)
print("build flags:", EXTRA)
lib = ctypes.CDLL(SO)
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, C, L = 16, 60, 80, 128, 4
levels = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(L))
f1 = (torch.randn(B, H, W, C, device=dev) * 0.125).half().contiguous()
ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device=dev)).contiguous()
o0 = (4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev))).contiguous()
o1 = ((4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev)) + o0) / 2).contiguous()
f2 = []
for l in range(L):
    f = (torch.randn(B, H >> l, W >> l, C, device=dev) * 0.125).half()
    f2.append(f.view(B, H >> l, W >> l, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous())
offs = [o0, o1, None, None]
vp = ctypes.c_void_p
nl = len(levels)
assert levels == list(range(levels[0], levels[0] + nl)), "consecutive levels only (lbase = 0 is assumed for the first)"
if levels[0] != 0:
    coords = (coords / 2 ** levels[0]).contiguous()
F2 = (vp * nl)(*[f2[l].data_ptr() for l in levels])
OF = (vp * nl)(*[offs[l].data_ptr() if offs[l] is not None else None for l in levels])
H2 = (ctypes.c_int * nl)(*[H >> l for l in levels])
W2 = (ctypes.c_int * nl)(*[W >> l for l in levels])
corr = torch.empty(B, 1, nl * 49, H, W, device=dev)
nwg = 4 * ((B + 7) // 8) * 8 * ((W + 7) // 8) * ((H + 3) // 4)  # up to one work unit per level
stamps = torch.zeros(nwg, 4, 32, dtype=torch.int64, device=dev)
lib.lgu_co_diag_set_stamps(vp(stamps.data_ptr()))
for it in range(3):
    stamps.zero_()
    rc = lib.lgu_co_diag_pyramid(vp(f1.data_ptr()), F2, OF, vp(coords.data_ptr()), vp(corr.data_ptr()), H2, W2, nl, B, 1, H, W, C, 3, 1)
    assert rc == 0, rc
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64).reshape(nwg * 4, 4, 8)[:, :nl, :6] * 10e-3  # 100 MHz ticks -> microseconds
live = s[:, :, 5] > 0   # [wave][level]: levels that wave served
t0 = s[:, :, 0][live].min()
print("levels %s: %d waves, kernel span %.1f us" % (levels, int(live.any(1).sum()), s[:, :, 5].max() - t0))
names = ["boxes (+ wait for the offsets)", "barrier + window + first loads", "sweep", "barrier + sampling", "write-out"]
for k in range(nl):
    sk = s[live[:, k], k, :]
    d = np.diff(sk, axis=1)
    print("  level %d: %.1f us (p90 %.1f) | " % (levels[k], np.median(sk[:, 5] - sk[:, 0]), np.percentile(sk[:, 5] - sk[:, 0], 90))
          + " | ".join("%s %.1f" % (n, np.median(d[:, i])) for i, n in enumerate(names))
          + " | starts p10 %.1f p50 %.1f p90 %.1f, last end %.1f" % (*[np.percentile(sk[:, 0] - t0, q) for q in (10, 50, 90)], sk[:, 5].max() - t0))
