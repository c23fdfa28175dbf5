#!/usr/bin/env python3
"""A/B of CorrBlock.__init__'s volume build at E edges of 48x64x128 fp32 maps: the matrix-core build straight into the tiled
pyramid (ops.volume_build_pyramid) against the library GEMM + fused post-processing (torch.matmul + ops.volume_pyramid)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd as lgu  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C, H, W = 128, 48, 64
dev = "cuda"
torch.manual_seed(0)
f1 = torch.randn(E, C, H, W, device=dev) * 0.5
f2 = torch.randn(E, C, H, W, device=dev) * 0.5
ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
means = (torch.stack([xs, ys], -1)[None] + torch.randn(E, H, W, 2, device=dev)).contiguous()
covs = (torch.rand(E, H, W, 2, device=dev) * 5 + 0.05).contiguous()


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def matmul_only():
    return torch.matmul((f1.reshape(E, C, H * W) / 4.0).transpose(1, 2), f2.reshape(E, C, H * W) / 4.0)


def library_path():
    raw = matmul_only().view(E, H, W, H, W)
    return lgu.ops.volume_pyramid(means, covs, raw, 4, 4, inplace=True, tiled=True)


res = {"edges": E, "matmul_ms": timed(matmul_only), "matmul_plus_fused_postprocessing_ms": timed(library_path),
       "matrix_core_build_ms": timed(lambda: lgu.ops.volume_build_pyramid(f1, f2, means, covs))}
out_bytes = sum(int(np.prod(lgu.ops.tiled_shape(E, H, W, H >> l, W >> l))) for l in range(4)) * 4
res["pyramid_bytes"] = out_bytes
res["matrix_core_build_write_GBps"] = out_bytes / (res["matrix_core_build_ms"] * 1e-3) / 1e9
res["gemm_TFLOPs_in_build"] = 2.0 * E * (H * W) ** 2 * C / (res["matrix_core_build_ms"] * 1e-3) / 1e12
# half maps (autocast): library half GEMM + fused post-processing against the matrix-core build with in-kernel half rounding
h1, h2 = f1.half(), f2.half()
th = torch.cat((h1, h2), 1).permute(0, 2, 3, 1).contiguous()


def half_library():
    raw = torch.matmul((h1.reshape(E, C, H * W) / 4.0).transpose(1, 2), h2.reshape(E, C, H * W) / 4.0).view(E, H, W, H, W)
    return lgu.ops.volume_pyramid(means, covs, raw, 4, 4, inplace=True, tiled=True)


res["half_matmul_plus_fused_postprocessing_ms"] = timed(half_library)
res["half_matrix_core_build_ms"] = timed(lambda: lgu.ops.volume_build_pyramid(th, None, means, covs))
res["half_matrix_core_build_write_GBps"] = out_bytes / (res["half_matrix_core_build_ms"] * 1e-3) / 1e9
print(json.dumps(res))
