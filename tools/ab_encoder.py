"""A/B of the lookup's output format against its consumer: the fused 4-level lookup followed by
UpdateModule.corr_encoder (reference droid_slam/droid_net.py:76-80: Conv2d(196,128,1) + ReLU + Conv2d(128,128,3) +
ReLU) run under autocast as factor_graph.py runs the update operator.  Random-init weights of that architecture.

For each output format of the lookup (planar fp32 = the reference tensor; channel-last fp32; channel-last half) prints
one JSON line with the device time of the lookup alone, of lookup + first convolution, of lookup + whole encoder, and
the largest difference of the encoder output against the planar/fp32-input run.

    python tools/ab_encoder.py [--edges 20] [--reps 200]
"""
import argparse
import json
import os

os.environ.setdefault("LGU_DEBUG_KNOBS", "1")   # this tool switches kernel variants through the library's debug variables
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dev_time(fn, reps, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=20)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import bench
    import lgu_slam_amd as lgu
    from lgu_slam_amd import ops

    dev = torch.device("cuda:0")
    E, H1, W1, L, R = args.edges, 48, 64, 4, 3
    vols, coords, offs = bench.make_inputs(E, H1, W1, L, R, 0, dev)
    level_hw = [(H1 >> l, W1 >> l) for l in range(L)]
    tv = [ops.volume_retile(v) for v in vols]
    torch.manual_seed(0)
    enc = torch.nn.Sequential(torch.nn.Conv2d(L * 49, 128, 1), torch.nn.ReLU(inplace=True),
                              torch.nn.Conv2d(128, 128, 3, padding=1), torch.nn.ReLU(inplace=True)).to(dev).eval()
    enc_cl = torch.nn.Sequential(torch.nn.Conv2d(L * 49, 128, 1), torch.nn.ReLU(inplace=True),
                                 torch.nn.Conv2d(128, 128, 3, padding=1), torch.nn.ReLU(inplace=True)).to(dev).eval()
    enc_cl.load_state_dict(enc.state_dict())
    enc_cl = enc_cl.to(memory_format=torch.channels_last)

    base = None
    for fmt in ("planar", "nhwc", "nhwc_f16"):
        plan = ops.DefcorrPyramidPlan(tv, [o.clone() if o is not None else None for o in offs], R, tiled=True,
                                      level_hw=level_hw, out_format=fmt)
        out = ops._pyr_out(fmt, E, L * 49, H1, W1, dev, None)
        net = enc if fmt == "planar" else enc_cl

        def lookup():
            return plan(coords, out=out)

        def first():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                return net[1](net[0](lookup()))

        def whole():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                return net(lookup())

        y = whole().float()
        if base is None:
            base = y
        rec = {"out_format": fmt, "edges": E,
               "lookup_ms": dev_time(lookup, args.reps),
               "lookup_plus_conv1x1_ms": dev_time(first, args.reps),
               "lookup_plus_corr_encoder_ms": dev_time(whole, args.reps),
               "encoder_out_max_abs_diff_vs_planar": float((y - base).abs().max()),
               "encoder_out_abs_max": float(base.abs().max())}
        print(json.dumps(rec), flush=True)

    # the 1x1 convolution over a channel-last tensor IS a plain GEMM (M = pixels, K = 196, N = 128): hand the
    # lookup's half rows to the library GEMM with bias + ReLU in its epilogue instead of MIOpen's convolution
    plan = ops.DefcorrPyramidPlan(tv, [o.clone() if o is not None else None for o in offs], R, tiled=True,
                                  level_hw=level_hw, out_format="nhwc_f16")
    out = ops._pyr_out("nhwc_f16", E, L * 49, H1, W1, dev, None)
    w1 = enc[0].weight.detach().view(128, L * 49).half().contiguous()
    b1 = enc[0].bias.detach().half().contiguous()
    w1t = w1.t()

    def gemm_first(mode):
        x = plan(coords, out=out).permute(0, 2, 3, 1).reshape(-1, L * 49)  # a view: the rows are already contiguous
        if mode == "addmm_relu":
            return torch._addmm_activation(b1, x, w1t, use_gelu=False)
        return torch.relu_(torch.nn.functional.linear(x, w1, b1))

    for mode in ("linear+relu", "addmm_relu"):
        def first():
            with torch.no_grad():
                return gemm_first(mode)

        def whole():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                h = gemm_first(mode).view(E, H1, W1, 128).permute(0, 3, 1, 2)
                return enc_cl[3](enc_cl[2](h))

        y = whole().float()
        rec = {"out_format": "nhwc_f16 + library GEMM (%s) for the 1x1 convolution" % mode, "edges": E,
               "lookup_plus_conv1x1_ms": dev_time(first, args.reps),
               "lookup_plus_corr_encoder_ms": dev_time(whole, args.reps),
               "encoder_out_max_abs_diff_vs_planar": float((y - base).abs().max()),
               "encoder_out_abs_max": float(base.abs().max())}
        print(json.dumps(rec), flush=True)

    # first layer inside the lookup launch (lgu_defcorr_pyramid_enc_fwd_f32): the 196 samples never reach HBM
    fplan = ops.DefcorrPyramidPlan(tv, [o.clone() if o is not None else None for o in offs], R, tiled=True,
                                   level_hw=level_hw, encoder=ops.pack_encoder_layer(enc[0].weight, enc[0].bias))
    fout = ops._pyr_out("nhwc_f16", E, 128, H1, W1, dev, None)

    def ffirst():
        return fplan(coords, out=fout)

    def fwhole():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return enc_cl[3](enc_cl[2](ffirst()))

    y = fwhole().float()
    print(json.dumps({"out_format": "first encoder layer fused into the lookup launch (matrix cores)", "edges": E,
                      "lookup_plus_conv1x1_ms": dev_time(ffirst, args.reps),
                      "lookup_plus_corr_encoder_ms": dev_time(fwhole, args.reps),
                      "encoder_out_max_abs_diff_vs_planar": float((y - base).abs().max()),
                      "encoder_out_abs_max": float(base.abs().max())}), flush=True)


if __name__ == "__main__":
    main()

