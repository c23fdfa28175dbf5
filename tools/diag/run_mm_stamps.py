#!/usr/bin/env python3
"""DIAGNOSTIC ONLY (never the product library, never timed for a result): per-wave phase stamps of the
matrix-core low-memory kernel (lgu-slam_amd/csrc/lowmem_mfma.hip built with -DLGU_MM_STAMPS)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "liblgu_mmdiag.so")
CSRC = os.path.join(ROOT, "lgu-slam_amd", "csrc")
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                       "-DLGU_MM_STAMPS", "-o", SO, os.path.join(CSRC, "lowmem_mfma.hip"), os.path.join(CSRC, "capi.hip")])
lib = ctypes.CDLL(SO)
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, C = 16, 60, 80, 128
f1 = (torch.randn(B, H, W, C, device=dev) * 0.125).half().contiguous()
ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
base = torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device=dev)
off0 = (4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev))).contiguous()
vp = ctypes.c_void_p
names = ["boxes", "lane setup + A + first loads", "sweep", "sampling", "write-out"]
import sys
CHUNKED = 0 if (len(sys.argv) > 1 and sys.argv[1] == "channel_last") else 1   # fmap2 layout (production: chunk-planar)
print("fmap2 layout:", "chunk-planar" if CHUNKED else "channel-last", "| levels 2, 3 with null (zero) offsets, as AltCorrBlock launches them")
for l in range(4):
    f2 = (torch.randn(B, H >> l, W >> l, C, device=dev) * 0.125).half().contiguous()
    if CHUNKED:
        f2 = f2.view(B, H >> l, W >> l, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
    cl = (base / 2 ** l).contiguous()
    corr = torch.empty(B, 1, 7, 7, H, W, device=dev)
    nwg = ((B + 7) // 8) * 8 * ((W + 3) // 4) * ((H + 3) // 4)
    stamps = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    lib.lgu_mm_diag_set_stamps(vp(stamps.data_ptr()))
    for it in range(2):
        stamps.zero_()
        rc = lib.lgu_mm_diag_lowmem(vp(f1.data_ptr()), vp(f2.data_ptr()), vp(cl.data_ptr()), vp(off0.data_ptr()) if l < 2 else None,
                                    vp(corr.data_ptr()), B, 1, H, W, H >> l, W >> l, C, 3, CHUNKED, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64) * 10e-3  # 100 MHz ticks -> microseconds
    s = s[s[:, 5] > 0]
    d = np.diff(s[:, :6], axis=1)
    tot = s[:, 5] - s[:, 0]
    span = s[:, 5].max() - s[:, 0].min()
    print("level %d: %d waves, kernel span %.1f us, wave lifetime median %.1f us (p90 %.1f) | " % (l, len(s), span, np.median(tot), np.percentile(tot, 90))
          + " | ".join("%s %.1f" % (n, np.median(d[:, i])) for i, n in enumerate(names)))
    st = np.sort(s[:, 0] - s[:, 0].min())
    print("   wave start times: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (np.percentile(st, 10), np.percentile(st, 50), np.percentile(st, 90), st.max()))
