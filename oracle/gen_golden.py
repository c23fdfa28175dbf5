#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own kernels on the GPU box.

TEST INFRASTRUCTURE.  Loads oracle/_ref/ref_defCorrSample.so and ref_altcorr.so — the
reference extension sources compiled unmodified for gfx950 by oracle/build_ref.py — feeds
them small seeded inputs and stores inputs + outputs.  Run on an MI355X:
    python oracle/gen_golden.py gpurun_out/golden
then copy the .npz files into tests/golden/ (they are committed; the .so files are not).
Every array is float32.  Shapes respect the reference's own constraints (H1 % 4 == 0 and
W1 % 8 == 0 for the (4,8)-block altcorr/lowMem kernels, C % 32 == 0).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import inputs  # noqa: E402


def load_ext(name):
    path = os.path.join(ROOT, "oracle", "_ref", name + ".so")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    ref = load_ext("ref_defCorrSample")
    alt = load_ext("ref_altcorr")
    rng = np.random.default_rng(2024)

    def save(name, **arrs):
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})
        print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})

    # --- defCorr_index forward/backward + corr_index forward/backward ---
    for tag, (E, H1, W1, H2, W2, r, sigma, cscale) in {
        "defcorr_r3_interior": (1, 8, 16, 8, 16, 3, 1.5, 1.0),
        "defcorr_r3_border": (2, 8, 16, 4, 8, 3, 5.0, 0.5),   # level-1-like: coords/2, many OOB taps
        "defcorr_r1": (1, 8, 16, 8, 16, 1, 3.0, 1.0),
    }.items():
        rd = 2 * r + 1
        vol = rng.standard_normal((E, H1, W1, H2, W2)).astype(np.float32)
        coords = (inputs.grid_coords(rng, E, H1, W1, sigma) * cscale).astype(np.float32)
        off = (4 * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2)))).astype(np.float32)
        g = rng.standard_normal((E, rd, rd, H1, W1)).astype(np.float32)
        o1 = dev(off)
        corr, = ref.defCorr_index_forward(dev(vol), dev(coords), o1, r)
        o2 = dev(off)
        vg, og = ref.defCorr_index_backward(dev(vol), dev(coords), o2, dev(g), r)
        pc, = ref.corr_index_forward(dev(vol), dev(coords), r)
        pvg, = ref.corr_index_backward(dev(vol), dev(coords), dev(g), r)
        torch.cuda.synchronize()
        save(tag, volume=vol, coords=coords, offset=off, corr_grad=g, radius=r, corr=host(corr), offset_after=host(o1),
             volume_grad=host(vg), offset_grad=host(og), plain_corr=host(pc), plain_volume_grad=host(pvg))

    # --- gaussianMask forward/backward ---
    E, H1, W1 = 1, 8, 16
    vol = rng.standard_normal((E, H1, W1, H1, W1)).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    means = (np.stack([xs, ys], -1)[None].repeat(E, 0) + 2 * rng.standard_normal((E, H1, W1, 2))).astype(np.float32)
    covs = rng.uniform(0.05, 5.05, (E, H1, W1, 2)).astype(np.float32)
    g = rng.standard_normal(vol.shape).astype(np.float32)
    v1, = ref.gaussianMask(dev(means), dev(covs), dev(vol), 4)
    mg, cg = ref.gaussianMask_backward(dev(means), dev(covs), dev(vol), dev(g), 4)
    save("gaussmask_r4", means=means, covs=covs, volume=vol, volume1_grad=g, radius=4, volume1=host(v1),
         means_grad=host(mg), covs_grad=host(cg))

    # --- lowMem_defSample ---
    for tag, (B, S, H1, W1, H2, W2, C, r, sigma, cscale) in {
        "lowmem_l0": (2, 1, 8, 16, 8, 16, 128, 3, 2.0, 1.0),
        "lowmem_l1": (2, 1, 8, 16, 4, 8, 64, 3, 4.0, 0.5),
    }.items():
        case = inputs.fmap_case(int(rng.integers(1 << 30)), B, S, H1, W1, H2, W2, C, r, sigma, cscale)
        o = dev(case["offset"])
        corr, = ref.lowMem_defSample(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), o, r)
        save(tag, radius=r, corr=host(corr), offset_after=host(o), **case)

    # --- altcorr forward/backward ---
    for tag, (B, S, H1, W1, H2, W2, C, r, sigma, cscale) in {
        "altcorr_r1": (2, 1, 8, 16, 4, 8, 64, 1, 4.0, 0.5),
        "altcorr_r3": (1, 2, 8, 16, 8, 16, 64, 3, 3.0, 1.0),
    }.items():
        case = inputs.fmap_case(int(rng.integers(1 << 30)), B, S, H1, W1, H2, W2, C, r, sigma, cscale)
        corr, = alt.altcorr_forward(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), r)
        g = rng.standard_normal(tuple(corr.shape)).astype(np.float32)
        f1g, f2g, cg = alt.altcorr_backward(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), dev(g), r)
        save(tag, radius=r, corr=host(corr), corr_grad=g, fmap1_grad=host(f1g), fmap2_grad=host(f2g),
             coords_grad=host(cg), fmap1=case["fmap1"], fmap2=case["fmap2"], coords=case["coords"])

    # --- pyramid composition exactly as CorrBlock.__call__ drives the reference ops
    # (corr.py:94-109): probe r=1 on level 1, var, sigmoid, offset[1] *= mask, 4 levels, cat
    case = inputs.pyramid_case(77, 1, 16, 16, 3, 3, 3.0, 4.0, False)
    E, H1, W1 = 1, 16, 16
    vols = [dev(v) for v in case["volumes"]]
    coords = dev(case["coords"])
    offs = [dev(o) if o is not None else torch.zeros(E, H1, W1, 7, 7, 2, device="cuda") for o in case["offsets"]]
    probe, = ref.corr_index_forward(vols[1], coords / 2, 1)
    mask = torch.sigmoid(torch.var(probe.permute(0, 3, 4, 1, 2), dim=[3, 4])).view(E, H1, W1, 1, 1, 1)
    offs[1] = (offs[1] * mask).contiguous()
    outs = [ref.defCorr_index_forward(vols[l], (coords / 2 ** l).contiguous(), offs[l], 3)[0].view(E, 49, H1, W1)
            for l in range(3)]
    save("pyramid_corrblock_call", volume0=case["volumes"][0], volume1=case["volumes"][1], volume2=case["volumes"][2],
         coords=case["coords"], offset0=case["offsets"][0], offset1=case["offsets"][1], radius=3,
         probe=host(probe), offset1_after=host(offs[1]), out=host(torch.cat(outs, dim=1)))
    print("GOLDEN_DONE")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
