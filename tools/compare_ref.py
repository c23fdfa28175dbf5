#!/usr/bin/env python3
"""GPU-box tool (test infrastructure): full-size parity and timing of this library against
the reference's own kernels (oracle/_ref, built by oracle/build_ref.py) on the same MI355X.

BASELINE config 2 (E=20, 48x64, L=4, r=3): the reference path is what CorrBlock.__call__
issues — 4x defCorr_index_forward + torch.cat (corr.py:101-109); ours is one fused launch.
Also times the other operators at representative shapes.  Prints JSON lines.
"""
import importlib.util
import json
import os

os.environ.setdefault("LGU_DEBUG_KNOBS", "1")   # this tool switches kernel variants through the library's debug variables
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lgu_slam_amd  # noqa: E402


def load_ext(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "_ref", name + ".so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ref = load_ext("ref_defCorrSample")
    alt = load_ext("ref_altcorr")
    ops = lgu_slam_amd.ops
    dev = torch.device("cuda:0")
    E, H1, W1, L, R = 20, 48, 64, 4, 3
    vols, coords, offs = bench.make_inputs(E, H1, W1, L, R, 1234, dev)
    zeros = torch.zeros(E, H1, W1, 7, 7, 2, device=dev)
    ref_offs = [o if o is not None else zeros for o in offs]
    lvl_coords = [(coords / 2 ** l).contiguous() for l in range(L)]

    def ref_call():
        return torch.cat([ref.defCorr_index_forward(vols[l], lvl_coords[l], ref_offs[l], R)[0].view(E, 49, H1, W1)
                          for l in range(L)], dim=1)

    def our_call():
        return ops.defcorr_pyramid_forward(vols, coords, offs, R)

    a, b = ref_call(), our_call()
    res = {"op": "defcorr_pyramid cfg2 E=20", "max_abs_diff_vs_reference": float((a - b).abs().max()),
           "ref_abs_max": float(a.abs().max()), "ref_ms": timeit(ref_call), "ours_ms": timeit(our_call)}
    res["speedup"] = res["ref_ms"] / res["ours_ms"]
    print(json.dumps(res))

    # probe r=1 on level 1
    c2 = lvl_coords[1]
    a, = ref.corr_index_forward(vols[1], c2, 1)
    b, = ops.corr_index_forward(vols[1], c2, 1)
    print(json.dumps({"op": "corr_index r=1 level1 E=20", "max_abs_diff_vs_reference": float((a - b).abs().max()),
                      "ref_ms": timeit(lambda: ref.corr_index_forward(vols[1], c2, 1)),
                      "ours_ms": timeit(lambda: ops.corr_index_forward(vols[1], c2, 1))}))

    # gaussianMask on the level-0 volume of 8 edges
    Eg = 8
    ys, xs = torch.meshgrid(torch.arange(H1, device=dev).float(), torch.arange(W1, device=dev).float(), indexing="ij")
    means = (torch.stack([xs, ys], -1)[None] + 2 * torch.randn(Eg, H1, W1, 2, device=dev)).contiguous()
    covs = (torch.rand(Eg, H1, W1, 2, device=dev) * 5 + 0.05).contiguous()
    vg = vols[0][:Eg].contiguous()
    a, = ref.gaussianMask(means, covs, vg, 4)
    b, = ops.gaussianMask(means, covs, vg, 4)
    print(json.dumps({"op": "gaussianMask E=8", "max_abs_diff_vs_reference": float((a - b).abs().max()),
                      "ref_ms": timeit(lambda: ref.gaussianMask(means, covs, vg, 4)),
                      "ours_ms": timeit(lambda: ops.gaussianMask(means, covs, vg, 4))}))
    # CorrBlock.__init__ volume post-processing: reference op + torch passes vs the fused kernel
    def ref_post():
        c1, = ref.gaussianMask(means, covs, vg, 4)
        lvl = c1 / (6.28 * torch.sqrt(covs[..., 0] * covs[..., 1]))[..., None, None] + vg
        outs = [lvl]
        x = lvl.view(-1, 1, H1, W1)
        for _ in range(3):
            x = torch.nn.functional.avg_pool2d(x, 2, stride=2)
            outs.append(x)
        return outs
    r_ = ref_post()
    o_ = ops.volume_pyramid(means, covs, vg, 4, 4)
    print(json.dumps({"op": "volume post-processing (mask + /den + corr + 3 pools) E=8",
                      "max_abs_diff_vs_reference": max(float((a_.reshape(-1) - b_.reshape(-1)).abs().max()) for a_, b_ in zip(r_, o_)),
                      "ref_ms": timeit(ref_post), "ours_ms": timeit(lambda: ops.volume_pyramid(means, covs, vg, 4, 4))}))
    # backward operators (training path), 4 edges of BASELINE config 2
    Eb = 4
    vb = [v[:Eb].contiguous() for v in vols[:2]]
    cb0, cb1 = lvl_coords[0][:Eb].contiguous(), lvl_coords[1][:Eb].contiguous()
    ob = ref_offs[0][:Eb].contiguous()
    g49 = torch.randn(Eb, 7, 7, H1, W1, device=dev)
    g9 = torch.randn(Eb, 3, 3, H1, W1, device=dev)
    ra = ref.defCorr_index_backward(vb[0], cb0, ob.clone(), g49, R)
    oa = ops.defCorr_index_backward(vb[0], cb0, ob.clone(), g49, R)
    print(json.dumps({"op": "defCorr_index_backward level0 E=4", "max_abs_diff_vs_reference": max(float((x - y).abs().max()) for x, y in zip(ra, oa)),
                      "ref_ms": timeit(lambda: ref.defCorr_index_backward(vb[0], cb0, ob, g49, R), iters=10, warm=2),
                      "ours_ms": timeit(lambda: ops.defCorr_index_backward(vb[0], cb0, ob, g49, R), iters=10, warm=2)}))
    ra = ref.corr_index_backward(vb[1], cb1, g9, 1)
    oa = ops.corr_index_backward(vb[1], cb1, g9, 1)
    print(json.dumps({"op": "corr_index_backward r=1 level1 E=4", "max_abs_diff_vs_reference": float((ra[0] - oa[0]).abs().max()),
                      "ref_ms": timeit(lambda: ref.corr_index_backward(vb[1], cb1, g9, 1), iters=10, warm=2),
                      "ours_ms": timeit(lambda: ops.corr_index_backward(vb[1], cb1, g9, 1), iters=10, warm=2)}))
    gv = torch.randn_like(vg[:Eb])
    ra = ref.gaussianMask_backward(means[:Eb].contiguous(), covs[:Eb].contiguous(), vg[:Eb].contiguous(), gv, 4)
    oa = ops.gaussianMask_backward(means[:Eb].contiguous(), covs[:Eb].contiguous(), vg[:Eb].contiguous(), gv, 4)
    print(json.dumps({"op": "gaussianMask_backward E=4", "max_rel_diff_vs_reference": max(float(((x - y).abs() / (y.abs() + 1e-3)).max()) for x, y in zip(oa, ra)),
                      "ref_ms": timeit(lambda: ref.gaussianMask_backward(means[:Eb].contiguous(), covs[:Eb].contiguous(), vg[:Eb].contiguous(), gv, 4), iters=10, warm=2),
                      "ours_ms": timeit(lambda: ops.gaussianMask_backward(means[:Eb].contiguous(), covs[:Eb].contiguous(), vg[:Eb].contiguous(), gv, 4), iters=10, warm=2)}))
    del vols, vg, a, b, r_, o_, vb, gv, ra, oa
    torch.cuda.empty_cache()

    # low-memory path, BASELINE config 4 shapes: 60x80 fmaps (needs H%4==0, W%8==0), C=128, B=16 edges
    B, H, W, C = 16, 60, 80, 128
    f1 = (torch.randn(B, H, W, C, device=dev) * 0.125).contiguous()
    ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
    base = torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device=dev)
    off0 = (4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device=dev))).contiguous()
    tot_ref = tot_our = tot_old = tot_valu = tot_mixed = tot_mixed_valu = worst_mixed = 0.0
    per_level = []
    per_level_mixed = []
    f1h = f1.half()
    worst = 0.0
    for l in range(4):
        f2 = (torch.randn(B, H >> l, W >> l, C, device=dev) * 0.125).contiguous()
        cl = (base / 2 ** l).contiguous()
        a, = ref.lowMem_defSample(f1, f2, cl, off0.clone(), 3)
        b, = ops.lowMem_defSample(f1, f2, cl, off0.clone(), 3)
        worst = max(worst, float((a - b).abs().max()))
        tot_ref += timeit(lambda: ref.lowMem_defSample(f1, f2, cl, off0, 3), iters=5, warm=1)
        t_l = timeit(lambda: ops.lowMem_defSample(f1, f2, cl, off0, 3), iters=5, warm=1)
        per_level.append(t_l)
        tot_our += t_l
        f2h = f2.half()
        t_m = timeit(lambda: ops.lowMem_defSample_mixed(f1h, f2h, cl, off0, 3), iters=5, warm=1)
        per_level_mixed.append(t_m)
        tot_mixed += t_m
        os.environ["LGU_LOWMEM_H16_VARIANT"] = "1"
        tot_mixed_valu += timeit(lambda: ops.lowMem_defSample_mixed(f1h, f2h, cl, off0, 3), iters=5, warm=1)
        os.environ.pop("LGU_LOWMEM_H16_VARIANT")
        m_, = ops.lowMem_defSample_mixed(f1h, f2h, cl, off0.clone(), 3)
        r_, = ref.lowMem_defSample(f1h.float(), f2h.float(), cl, off0.clone(), 3)
        worst_mixed = max(worst_mixed, float((m_ - r_).abs().max()))
        os.environ["LGU_LOWMEM_VARIANT"] = "1"
        tot_old += timeit(lambda: ops.lowMem_defSample(f1, f2, cl, off0, 3), iters=5, warm=1)
        os.environ["LGU_LOWMEM_VARIANT"] = "2"
        tot_valu += timeit(lambda: ops.lowMem_defSample(f1, f2, cl, off0, 3), iters=5, warm=1)
        os.environ.pop("LGU_LOWMEM_VARIANT")
        if l == 1:
            ga = torch.randn(B, 1, 9, H, W, device=dev)
            rb = alt.altcorr_backward(f1, f2, cl, ga, 1)
            ob_ = ops.altcorr_backward(f1, f2, cl, ga, 1)
            print(json.dumps({"op": "altcorr_backward r=1 level1 B=16 60x80", "max_abs_diff_vs_reference": max(float((x - y).abs().max()) for x, y in zip(rb[:2], ob_[:2])),
                              "ref_ms": timeit(lambda: alt.altcorr_backward(f1, f2, cl, ga, 1), iters=5, warm=1),
                              "ours_ms": timeit(lambda: ops.altcorr_backward(f1, f2, cl, ga, 1), iters=5, warm=1)}))
            a, = alt.altcorr_forward(f1, f2, cl, 1)
            b, = ops.altcorr_forward(f1, f2, cl, 1)
            print(json.dumps({"op": "altcorr r=1 level1 B=16 60x80", "max_abs_diff_vs_reference": float((a - b).abs().max()),
                              "ref_ms": timeit(lambda: alt.altcorr_forward(f1, f2, cl, 1), iters=5, warm=1),
                              "ours_ms": timeit(lambda: ops.altcorr_forward(f1, f2, cl, 1), iters=5, warm=1)}))
    print(json.dumps({"op": "lowMem_defSample 4 levels B=16 60x80 C=128", "max_abs_diff_vs_reference": worst,
                      "ref_ms": tot_ref, "ours_ms": tot_our, "ours_wave_per_pixel_ms": tot_old, "ours_valu_tile_kernel_ms": tot_valu, "ours_ms_per_level": per_level, "ours_half_features_ms": tot_mixed, "ours_half_features_valu_kernel_ms": tot_mixed_valu, "half_features_max_abs_diff_vs_reference_on_float_copies": worst_mixed, "ours_half_features_Mpix_edges_per_s": B * H * W / tot_mixed / 1e3, "ours_half_features_ms_per_level": per_level_mixed, "speedup": tot_ref / tot_our,
                      "ours_Mpix_edges_per_s": B * H * W / tot_our / 1e3}))


if __name__ == "__main__":
    main()
