"""GPU parity: every C-ABI entry point (through the Python operator layer, which is a thin
ctypes pass-through) against the CPU oracle on identical seeded inputs.

Tolerances: north_star states 1e-5 fp32.  The forward samplers share the oracle's
evaluation order and are held to 1e-6 (they are expected to be bit-exact); ops whose
reduction order differs (tree sums over channels / taps, atomics) are held to 1e-5
absolute at the magnitudes of these inputs, or 1e-5 relative to the output scale where
the outputs are large (gradients).
"""
import os

import numpy as np
import pytest

from tests import inputs

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native(lgu):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    # the HIP library must be the thing under test: fail loudly if it is absent
    assert os.path.exists(lgu._lib.so_path()), "liblgu_corr.so missing — run __graft_entry__.build()"
    lgu._lib.load()


def set_variant(v):
    os.environ["LGU_DEFCORR_VARIANT"] = str(v)


@pytest.fixture(autouse=True)
def _reset_variant():
    yield
    os.environ.pop("LGU_DEFCORR_VARIANT", None)


def run_pyramid(lgu, case, radius, variant=0):
    set_variant(variant)
    vols = [dev(v) for v in case["volumes"]]
    offs = [dev(o) if o is not None else None for o in case["offsets"]]
    out = lgu.ops.defcorr_pyramid_forward(vols, dev(case["coords"]), offs, radius)
    torch.cuda.synchronize()
    return host(out), [host(o) if o is not None else None for o in offs]


def oracle_pyramid(O, case, radius):
    offs = [o.copy() if o is not None else None for o in case["offsets"]]
    out = O.defcorr_pyramid_forward(case["volumes"], case["coords"], offs, radius)
    return out, offs


PYR_CASES = {
    # name: (seed, E, H1, W1, L, radius, sigma, off_scale, dense)
    "tiny": (1, 2, 12, 16, 3, 3, 3.0, 4.0, False),
    "cfg2_shape": (2, 2, 48, 64, 4, 3, 3.0, 4.0, False),
    "border_stress": (3, 2, 24, 32, 3, 3, 20.0, 4.0, False),
    "dense_offsets": (4, 1, 24, 32, 3, 3, 3.0, 4.0, True),
    "huge_offsets_direct_path": (5, 1, 48, 64, 2, 3, 3.0, 14.0, True),
    "ragged_width": (6, 2, 30, 40, 2, 3, 3.0, 4.0, False),   # W1 = 40 = 2.5 tiles; level 1 is 15x20
    "radius1": (7, 2, 24, 32, 2, 1, 3.0, 2.0, True),
    "radius2": (8, 1, 24, 32, 2, 2, 3.0, 3.0, True),
    "generic_w2_not_mult4": (9, 1, 20, 24, 3, 3, 3.0, 4.0, False),  # level 2 is 5x6
}


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("name", list(PYR_CASES))
def test_defcorr_pyramid_matches_oracle(lgu, oracle, name, variant):
    seed, E, H1, W1, L, radius, sigma, osc, dense = PYR_CASES[name]
    case = inputs.pyramid_case(seed, E, H1, W1, L, radius, sigma, osc, dense)
    want, want_offs = oracle_pyramid(oracle, case, radius)
    got, got_offs = run_pyramid(lgu, case, radius, variant)
    err = np.abs(got - want).max()
    assert err <= 1e-6, "%s variant %d: max abs err %g" % (name, variant, err)
    # in-place side effect: centre offsets zeroed, everything else untouched
    for a, b in zip(got_offs, want_offs):
        if a is not None:
            assert np.array_equal(a, b)



PROBE_CASES = {
    # name: (seed, E, H1, W1, L, radius, sigma, off_scale, dense)
    "cfg2_shape": (21, 2, 48, 64, 4, 3, 3.0, 4.0, False),
    "border_stress": (22, 2, 24, 32, 3, 3, 14.0, 4.0, False),
    "dense_offsets": (23, 1, 24, 32, 3, 3, 3.0, 4.0, True),
    "two_levels": (24, 1, 24, 32, 2, 3, 3.0, 4.0, True),
    "radius1": (25, 1, 24, 32, 3, 1, 3.0, 2.0, False),
}


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 5])
@pytest.mark.parametrize("name", list(PROBE_CASES))
def test_fused_probe_matches_oracle(lgu, oracle, name, variant):
    """LGU_PYR_PROBE: in-kernel 3x3 probe of level 1 -> var -> sigmoid -> offset[1] *= mask
    (written back) -> sample.  Two consecutive calls check the persistent compounding."""
    seed, E, H1, W1, L, radius, sigma, osc, dense = PROBE_CASES[name]
    set_variant(variant)
    case = inputs.pyramid_case(seed, E, H1, W1, L, radius, sigma, osc, dense)
    ref_offs = [o.copy() if o is not None else None for o in case["offsets"]]
    vols = [dev(v) for v in case["volumes"]]
    offs = [dev(o) if o is not None else None for o in case["offsets"]]
    coords = dev(case["coords"])
    for call in range(2):
        want = oracle.defcorr_pyramid_forward(case["volumes"], case["coords"], ref_offs, radius, probe=True)
        got = lgu.ops.defcorr_pyramid_forward(vols, coords, offs, radius, probe=True)
        torch.cuda.synchronize()
        # the mask goes through expf and a different (equally valid) summation order: ~1e-7
        # relative on the offsets, which the bilinear gradient turns into <= ~1e-6 on outputs
        assert np.abs(host(offs[1]) - ref_offs[1]).max() <= 2e-6, "call %d offsets" % call
        assert np.abs(host(got) - want).max() <= 1e-5, "call %d" % call
        assert np.array_equal(host(offs[0]), ref_offs[0])


def test_probe_unsupported_patterns_raise(lgu):
    v = [torch.randn(1, 8, 16, 8 >> l, 16 >> l, device="cuda") for l in range(2)]
    c = torch.rand(1, 2, 8, 16, device="cuda") * 8
    with pytest.raises(RuntimeError):
        lgu.ops.defcorr_pyramid_forward(v, c, [None, None], 3, probe=True)
    o = torch.zeros(1, 8, 16, 7, 7, 2, device="cuda")
    with pytest.raises(lgu._lib.UnsupportedShape):  # level 0 null, level 1 not: no single fused launch
        lgu.ops.defcorr_pyramid_forward(v, c, [None, o], 3, probe=True)  # level 1 has no offsets to mask


def test_mixed_null_patterns(lgu, oracle):
    """Offset patterns other than none / all / levels>=2 are split into several launches."""
    case = inputs.pyramid_case(31, 1, 24, 32, 3, 3, 3.0, 4.0, True)
    for pattern in ([True, False, True], [False, True, True], [False, False, True], [True, True, False]):
        offs_np = [o.copy() if keep else None for o, keep in zip(case["offsets"], pattern)]
        want = oracle.defcorr_pyramid_forward(case["volumes"], case["coords"], [o.copy() if o is not None else None for o in offs_np], 3)
        got = lgu.ops.defcorr_pyramid_forward([dev(v) for v in case["volumes"]], dev(case["coords"]),
                                              [dev(o) if o is not None else None for o in offs_np], 3)
        assert np.abs(host(got) - want).max() <= 1e-6, pattern


def test_defcorr_single_level_and_plain_identity(lgu, oracle):
    case = inputs.pyramid_case(11, 2, 24, 32, 1, 3, 3.0, 4.0, True)
    v, c, off = case["volumes"][0], case["coords"], case["offsets"][0]
    o_ref = off.copy()
    want, = oracle.defCorr_index_forward(v, c, o_ref, 3)
    o_dev = dev(off)
    got, = lgu.ops.defCorr_index_forward(dev(v), dev(c), o_dev, 3)
    assert isinstance(got, torch.Tensor) and got.shape == (2, 7, 7, 24, 32)
    assert np.abs(host(got) - want).max() <= 1e-6
    assert np.array_equal(host(o_dev), o_ref)
    # defCorr(offset = 0) == corr_index (reference identity): bit-exact
    z = torch.zeros_like(o_dev)
    a, = lgu.ops.defCorr_index_forward(dev(v), dev(c), z, 3)
    b, = lgu.ops.corr_index_forward(dev(v), dev(c), 3)
    assert torch.equal(a, b)
    want_b, = oracle.corr_index_forward(v, c, 3)
    assert np.abs(host(b) - want_b).max() <= 1e-6


def test_probe_shape_r1_level1(lgu, oracle):
    # the call CorrBlock makes for the uncertainty probe (corr.py:94): level-1 volume, coords/2, r=1
    rng = np.random.default_rng(12)
    v = rng.standard_normal((3, 48, 64, 24, 32)).astype(np.float32)
    c = inputs.grid_coords(rng, 3, 48, 64, 3.0) / 2
    want, = oracle.corr_index_forward(v, c, 1)
    got, = lgu.ops.corr_index_forward(dev(v), dev(c), 1)
    assert got.shape == (3, 3, 3, 48, 64)
    assert np.abs(host(got) - want).max() <= 1e-6


@pytest.mark.parametrize("radius,sigma", [(3, 3.0), (3, 15.0), (1, 3.0)])
def test_defcorr_backward(lgu, oracle, radius, sigma):
    rng = np.random.default_rng(20 + radius)
    E, H1, W1, H2, W2 = 2, 12, 16, 12, 16
    rd = 2 * radius + 1
    v = rng.standard_normal((E, H1, W1, H2, W2)).astype(np.float32)
    c = inputs.grid_coords(rng, E, H1, W1, sigma)
    off = (4 * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2)))).astype(np.float32)
    g = rng.standard_normal((E, rd, rd, H1, W1)).astype(np.float32)
    o_ref = off.copy()
    vg_w, og_w = oracle.defCorr_index_backward(v, c, o_ref, g, radius)
    o_dev = dev(off)
    vg, og = lgu.ops.defCorr_index_backward(dev(v), dev(c), o_dev, dev(g), radius)
    assert np.abs(host(og) - og_w).max() <= 1e-5 * max(1.0, np.abs(og_w).max())
    assert np.abs(host(vg) - vg_w).max() <= 1e-5 * max(1.0, np.abs(vg_w).max())
    assert np.array_equal(host(o_dev), o_ref)
    # plain sampler backward
    vg2_w, = oracle.corr_index_backward(v, c, g, radius)
    vg2, = lgu.ops.corr_index_backward(dev(v), dev(c), dev(g), radius)
    assert np.abs(host(vg2) - vg2_w).max() <= 1e-5 * max(1.0, np.abs(vg2_w).max())


@pytest.mark.parametrize("shape", [(2, 12, 16, 12, 16), (1, 48, 64, 48, 64), (1, 10, 10, 9, 10)])
def test_gaussian_mask_forward_backward(lgu, oracle, shape):
    E, H1, W1, H2, W2 = shape
    rng = np.random.default_rng(30 + H1)
    v = rng.standard_normal(shape).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    means = (np.stack([xs, ys], -1)[None].repeat(E, 0) + rng.standard_normal((E, H1, W1, 2)) * 2).astype(np.float32)
    covs = (rng.uniform(0.05, 5.05, (E, H1, W1, 2))).astype(np.float32)
    want, = oracle.gaussianMask(means, covs, v, 4)
    got, = lgu.ops.gaussianMask(dev(means), dev(covs), dev(v), 4)
    # expf differs by <= 2 ulp between libm and the device: relative 2.4e-7 of values <= ~15
    assert np.abs(host(got) - want).max() <= 1e-5
    # identical support: exactly zero outside the window (inside it, exp() underflow may
    # flush differently on the two sides, so only compare where the oracle is not tiny)
    assert not host(got)[want == 0].any() or np.abs(host(got)[want == 0]).max() < 1e-30
    assert (host(got)[np.abs(want) > 1e-30] != 0).all()
    g = rng.standard_normal(shape).astype(np.float32)
    mg_w, cg_w = oracle.gaussianMask_backward(means, covs, v, g, 4)
    mg, cg = lgu.ops.gaussianMask_backward(dev(means), dev(covs), dev(v), dev(g), 4)
    assert np.abs(host(mg) - mg_w).max() <= 1e-5 * max(1.0, np.abs(mg_w).max())
    assert np.abs(host(cg) - cg_w).max() <= 1e-5 * max(1.0, np.abs(cg_w).max())


LOWMEM_CASES = [
    # B, S, H1, W1, H2, W2, C, radius, sigma, scale, off_scale
    (3, 1, 12, 16, 12, 16, 128, 3, 3.0, 1.0, 4.0),
    (2, 1, 12, 16, 6, 8, 128, 3, 3.0, 0.5, 4.0),
    (2, 1, 15, 20, 7, 10, 64, 3, 6.0, 0.5, 4.0),     # ragged: W1 = 20, odd H2/W2 as in 60x80 level 3
    (1, 2, 8, 16, 8, 16, 32, 1, 3.0, 1.0, 4.0),      # S = 2 uses offset[b*n]
    (2, 1, 60, 80, 60, 80, 128, 3, 3.0, 1.0, 4.0),   # BASELINE config 4 level-0 shape
    (1, 1, 60, 80, 30, 40, 128, 3, 3.0, 0.5, 4.0),   # level 1
    (1, 1, 24, 32, 24, 32, 64, 3, 3.0, 1.0, 14.0),   # boxes > 256 positions: per-pixel fallback
    (1, 1, 60, 80, 60, 80, 32, 3, 40.0, 1.0, 4.0),   # scattered coords: tile window > LDS budget, per-tap fallback
    (1, 1, 12, 16, 12, 16, 40, 2, 3.0, 1.0, 3.0),    # radius 2, C = 40 (C % 8 == 0, C % 32 != 0 is rejected by the ABI)
    (3, 2, 10, 13, 10, 13, 64, 3, 3.0, 1.0, 4.0),    # B = 3, S = 2: offset[b*s] takes rows 0, 1, 2 (reference indexing), ragged block grid
]


@pytest.mark.parametrize("variant", [0, 1, 2])   # 0 = fp32 matrix-core kernel, 2 = VALU tile kernel, 1 = wave-per-pixel kernel
@pytest.mark.parametrize("cfg", LOWMEM_CASES)
def test_lowmem_defsample(lgu, oracle, cfg, variant):
    B, S, H1, W1, H2, W2, C, radius, sigma, scale, osc = cfg
    if C % 32 != 0:
        with pytest.raises(lgu._lib.UnsupportedShape):
            lgu.ops.lowMem_defSample(torch.zeros(B, H1, W1, C, device="cuda"), torch.zeros(B, H2, W2, C, device="cuda"),
                                     torch.zeros(B, S, H1, W1, 2, device="cuda"),
                                     torch.zeros(B, H1, W1, 2 * radius + 1, 2 * radius + 1, 2, device="cuda"), radius)
        return
    os.environ["LGU_LOWMEM_VARIANT"] = str(variant)
    try:
        case = inputs.fmap_case(40 + H2, B, S, H1, W1, H2, W2, C, radius, sigma, scale, off_scale=osc)
        o_ref = case["offset"].copy()
        want, = oracle.lowMem_defSample(case["fmap1"], case["fmap2"], case["coords"], o_ref, radius)
        o_dev = dev(case["offset"])
        got, = lgu.ops.lowMem_defSample(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), o_dev, radius)
        assert got.shape == want.shape
        assert np.abs(host(got) - want).max() <= 1e-5
        assert np.array_equal(host(o_dev), o_ref)  # edge-0 centre zeroed (b*n indexing), rest untouched
    finally:
        os.environ.pop("LGU_LOWMEM_VARIANT", None)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("cfg", [(3, 1, 12, 16, 6, 8, 128, 1, 3.0, 0.5), (2, 2, 8, 16, 8, 16, 64, 3, 5.0, 1.0),
                                 (1, 1, 15, 20, 7, 10, 128, 1, 8.0, 0.5), (2, 1, 60, 80, 30, 40, 128, 1, 3.0, 0.5)])
def test_altcorr_forward_backward(lgu, oracle, cfg, variant):
    B, S, H1, W1, H2, W2, C, radius, sigma, scale = cfg
    os.environ["LGU_LOWMEM_VARIANT"] = str(variant)
    try:
        case = inputs.fmap_case(50 + H2, B, S, H1, W1, H2, W2, C, radius, sigma, scale)
        want, = oracle.altcorr_forward(case["fmap1"], case["fmap2"], case["coords"], radius)
        got, = lgu.ops.altcorr_forward(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), radius)
        assert got.shape == want.shape
        assert np.abs(host(got) - want).max() <= 1e-5
        if variant == 1 or H1 > 20:
            return
        rng = np.random.default_rng(60)
        g = rng.standard_normal(want.shape).astype(np.float32)
        f1g_w, f2g_w, cg_w = oracle.altcorr_backward(case["fmap1"], case["fmap2"], case["coords"], g, radius)
        f1g, f2g, cg = lgu.ops.altcorr_backward(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), dev(g), radius)
        assert np.abs(host(f1g) - f1g_w).max() <= 1e-5 * max(1.0, np.abs(f1g_w).max())
        assert np.abs(host(f2g) - f2g_w).max() <= 1e-5 * max(1.0, np.abs(f2g_w).max())
        assert not host(cg).any()
    finally:
        os.environ.pop("LGU_LOWMEM_VARIANT", None)


@pytest.mark.parametrize("shape,L", [((2, 12, 16, 12, 16), 3), ((1, 48, 64, 48, 64), 4), ((2, 6, 8, 20, 24), 2)])
def test_volume_pyramid_fused(lgu, oracle, shape, L):
    """lgu_volume_pyramid_f32 == gaussianMask / (6.28 sqrt(det)) + corr, then avg-pool pyramid
    (reference gaussianMask_cuda.py:84-86, corr.py:79-86): vs the oracle and vs the same
    composition done with the separate op + torch on the GPU; in-place level 0."""
    E, H1, W1, H2, W2 = shape
    rng = np.random.default_rng(70 + H2)
    v = rng.standard_normal(shape).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    means = (np.stack([xs, ys], -1)[None].repeat(E, 0) * (W2 / W1) + rng.standard_normal((E, H1, W1, 2)) * 2).astype(np.float32)
    covs = rng.uniform(0.05, 5.05, (E, H1, W1, 2)).astype(np.float32)
    want = oracle.volume_pyramid(means, covs, v, L, 4)
    got = lgu.ops.volume_pyramid(dev(means), dev(covs), dev(v), L, 4)
    assert [tuple(t.shape) for t in got] == [w.shape for w in want]
    for l in range(L):
        assert np.abs(host(got[l]) - want[l]).max() <= 1e-5, l
    # the composition of separate ops on the device
    vd, md, cd = dev(v), dev(means), dev(covs)
    c1, = lgu.ops.gaussianMask(md, cd, vd, 4)
    lvl = c1 / (6.28 * torch.sqrt(cd[..., 0] * cd[..., 1]))[..., None, None] + vd
    for l in range(L):
        assert (got[l] - lvl).abs().max() <= 1e-6, l
        lvl = torch.nn.functional.avg_pool2d(lvl.view(-1, 1, H2 >> l, W2 >> l), 2, stride=2).view(E, H1, W1, H2 >> (l + 1), W2 >> (l + 1))
    # in place
    vd2 = dev(v)
    lev = lgu.ops.volume_pyramid(md, cd, vd2, L, 4, inplace=True)
    assert lev[0].data_ptr() == vd2.data_ptr() and torch.equal(lev[0], got[0]) and torch.equal(lev[-1], got[-1])


def test_altcorrblock_matches_oracle_composition(lgu, oracle):
    """Low-memory glue (reference corr.py:155-249): AltCorrBlock.__call__ == per level
    [altcorr probe r=1 on level 1 -> var -> sigmoid -> offset[1] *= mask] + lowMem_defSample with
    coords / 2^l, concatenated; every edge of the chunk samples with edge 0's offsets
    (the reference's offset[b*n] indexing)."""
    torch.manual_seed(5)
    N, C, H, W = 4, 128, 24, 32
    dev_ = "cuda"
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    fmaps = torch.randn(1, N, C, H, W, device=dev_) * 0.5
    ii = torch.tensor([0, 0, 1, 2, 3], device=dev_)
    jj = torch.tensor([1, 2, 3, 0, 2], device=dev_)
    E = ii.numel()
    ys, xs = torch.meshgrid(torch.arange(H, device=dev_).float(), torch.arange(W, device=dev_).float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, H, W, 2, device=dev_)
    with torch.no_grad():
        blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        got = blk(coords, ii, jj)
        assert got.shape == (1, E, 196, H, W)
        # oracle composition from the same pyramid / offsets
        f1 = host(blk.pyramid[0][0, ii].float().contiguous())
        feats = torch.cat(((blk.pyramid[0][0, ii] * 4.0).permute(0, 3, 1, 2), (blk.pyramid[0][0, jj] * 4.0).permute(0, 3, 1, 2)), 1).float()
        offs, _ = lgu.corr.generate_offsets(ofsMap, ofsRes, feats, 4)
        offs = [host(o.contiguous()).reshape(E, H, W, 7, 7, 2).copy() for o in offs]
        c = host(coords[0])  # (E,H,W,2)
        outs = []
        for l in range(4):
            f2 = host(blk.pyramid[l][0, jj].float().contiguous())
            cl = (c / 2 ** l).astype(np.float32).reshape(E, 1, H, W, 2)
            if l == 1:
                probe, = oracle.altcorr_forward(f1, f2, cl, 1)
                pr = torch.from_numpy(probe).permute(0, 1, 3, 4, 2).contiguous().view(E, H, W, 3, 3)
                mask = torch.sigmoid(torch.var(pr, dim=[3, 4])).numpy().reshape(E, H, W, 1, 1, 1)
                offs[1] = (offs[1] * mask).astype(np.float32)
            corr, = oracle.lowMem_defSample(f1, f2, cl, offs[l], 3)
            outs.append(corr.reshape(E, 49, H, W))
        want = np.concatenate(outs, 1)
    assert np.abs(host(got)[0] - want).max() <= 2e-5


MIXED_CASES = [(2, 1, 60, 80, 60, 80, 128, 3, 3.0, 1.0, 4.0), (2, 1, 60, 80, 30, 40, 128, 3, 3.0, 0.5, 4.0),
               (1, 1, 24, 32, 6, 8, 64, 3, 3.0, 0.25, 4.0), (1, 1, 24, 32, 24, 32, 32, 3, 3.0, 1.0, 14.0),
               (1, 2, 8, 16, 8, 16, 64, 1, 3.0, 1.0, 2.0),
               # ragged sizes (H1, W1 not multiples of the 4x4 block / 8x8 tile), 9 edges (XCD dealing with a tail),
               # C = 256, radius 2, border stress
               (9, 1, 10, 13, 7, 9, 128, 3, 6.0, 0.5, 4.0), (2, 1, 12, 16, 12, 16, 256, 2, 3.0, 1.0, 3.0),
               (3, 2, 10, 13, 10, 13, 64, 3, 3.0, 1.0, 4.0),
               (1, 1, 17, 23, 17, 23, 128, 3, 20.0, 1.0, 4.0)]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("cfg", MIXED_CASES)
def test_mixed_precision_lowmem_equals_the_float_call_site(lgu, oracle, cfg, variant):
    """lgu_*_h16: half feature maps, fp32 products and sums = the reference call sites corr.py:202,209
    (`fmap.float()` first).  Variant 0 (production) runs the correlation on the matrix cores: exact half
    products, fp32 accumulation, so it differs from the f32 operator by summation order only (<= 1e-5,
    the north-star bound).  Variant 1 (VALU tile kernel) sums in the f32 operator's order: BIT FOR BIT."""
    B, S, H1, W1, H2, W2, C, radius, sigma, scale, osc = cfg
    case = inputs.fmap_case(400 + H2 + C, B, S, H1, W1, H2, W2, C, radius, sigma, scale, off_scale=osc)
    f1h, f2h = dev(case["fmap1"]).half(), dev(case["fmap2"]).half()
    coords = dev(case["coords"])
    o_m, o_f = dev(case["offset"]), dev(case["offset"])
    os.environ["LGU_LOWMEM_H16_VARIANT"] = str(variant)
    try:
        got, = lgu.ops.lowMem_defSample_mixed(f1h, f2h, coords, o_m, radius)
        a, = lgu.ops.altcorr_forward_mixed(f1h, f2h, coords, 1)
    finally:
        os.environ.pop("LGU_LOWMEM_H16_VARIANT")
    os.environ["LGU_LOWMEM_VARIANT"] = "2"   # the VALU tile kernel over the float copies: the summation order variant 1 shares
    try:
        want, = lgu.ops.lowMem_defSample(f1h.float(), f2h.float(), coords, o_f, radius)
        b, = lgu.ops.altcorr_forward(f1h.float(), f2h.float(), coords, 1)
    finally:
        os.environ.pop("LGU_LOWMEM_VARIANT")
    assert got.dtype == torch.float32 and torch.equal(o_m, o_f)
    if variant == 1:
        assert torch.equal(got, want) and torch.equal(a, b)
    else:
        assert float((got - want).abs().max()) <= 1e-5 and float((a - b).abs().max()) <= 1e-5
    ref_np, = oracle.lowMem_defSample(host(f1h.float()), host(f2h.float()), case["coords"], case["offset"].copy(), radius)
    assert np.abs(host(got) - ref_np).max() <= 1e-5
    ref_a, = oracle.altcorr_forward(host(f1h.float()), host(f2h.float()), case["coords"], 1)
    assert np.abs(host(a) - ref_a).max() <= 1e-5


def test_lowmem_backend_scale_tile_kernel_equals_wave_kernel(lgu):
    """BASELINE config 5 scale per GPU (250 edges x 60x80, C=128; here B=192 to bound the test's
    memory): the production (matrix-core) kernel and the independent wave-per-pixel kernel agree on the LAST edges
    (largest addresses) — guards the 32-bit in-edge offsets and the grid decomposition."""
    torch.manual_seed(13)
    B, H, W, C = 192, 60, 80, 128
    f1 = torch.randn(B, H, W, C, device="cuda") * 0.125
    f2 = torch.randn(B, H // 2, W // 2, C, device="cuda") * 0.125
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = ((torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device="cuda")) / 2).contiguous()
    off = (4 * torch.tanh(torch.randn(1, H, W, 7, 7, 2, device="cuda"))).contiguous()
    got, = lgu.ops.lowMem_defSample(f1, f2, coords, off, 3)
    tail = slice(B - 2, B)
    os.environ["LGU_LOWMEM_VARIANT"] = "1"
    try:
        want, = lgu.ops.lowMem_defSample(f1[tail].contiguous(), f2[tail].contiguous(), coords[tail].contiguous(), off, 3)
    finally:
        os.environ.pop("LGU_LOWMEM_VARIANT")
    assert float((got[tail] - want).abs().max()) <= 1e-5
    assert torch.isfinite(got).all()


def test_altcorrblock_half_features_equal_float_features(lgu):
    """AltCorrBlock over a half-precision feature buffer (video.fmaps is torch.half) takes the mixed
    operators and returns exactly what the float-converted path returns."""
    torch.manual_seed(6)
    N, C, H, W = 4, 128, 24, 32
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    fm = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
    ii = torch.tensor([0, 0, 1, 2], device="cuda")
    jj = torch.tensor([1, 2, 3, 0], device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, 4, H, W, 2, device="cuda")
    with torch.no_grad():
        a = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
        got = a(coords, ii, jj)
        assert a.pyramid[0].dtype == torch.float16
        b = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
        b.pyramid = [p.float() for p in b.pyramid]   # float copies of the SAME half values
        want = b(coords, ii, jj)
    assert got.shape == (1, 4, 196, H, W)
    assert float((got - want).abs().max()) <= 1e-5   # offsets come from convs that are not bitwise reproducible


def test_sharded_altcorr_world1_equals_unsharded(lgu):
    """the sharded lookup driver on one rank visits the reference's own source-frame
    chunks (factor_graph.py:272-276) and reproduces AltCorrBlock on each of them."""
    torch.manual_seed(11)
    N, C, H, W = 20, 128, 24, 32
    dev_ = "cuda"
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    fmaps = torch.randn(1, N, C, H, W, device=dev_) * 0.5
    ii = torch.tensor([0, 0, 1, 7, 8, 9, 15, 16, 16, 19], device=dev_)
    jj = torch.tensor([1, 2, 0, 8, 7, 10, 16, 15, 17, 18], device=dev_)
    E = ii.numel()
    ys, xs = torch.meshgrid(torch.arange(H, device=dev_).float(), torch.arange(W, device=dev_).float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, H, W, 2, device=dev_)
    with torch.no_grad():
        sh = lgu.sharded.ShardedAltCorr(ofsMap, ofsRes, None, fmaps, ii, jj, rank=0, world=1)
        ref_blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        seen = []
        feats = torch.zeros(E, 196, H, W, device=dev_)
        for idx, corr in sh.lookup(coords):
            want = ref_blk(coords[:, idx], ii[idx], jj[idx] + (ii[idx] == jj[idx]).long())
            # (the offset convolutions are not bitwise reproducible call to call on every backend)
            assert float((corr - want).abs().max()) <= 1e-5
            assert int(ii[idx].max()) // 8 == int(ii[idx].min()) // 8
            feats[idx] = corr[0]
            seen += idx.tolist()
        assert sorted(seen) == list(range(E))
        local = feats[sh.edges.my_edges]
        assert torch.equal(sh.edges.gather(local), feats)


def test_autograd_functions_route_gradients(lgu, oracle):
    """a10: CorrSampler / DefCorrSampler / GaussianMaskCuda (reference corr.py:10-42,
    gaussianMask_cuda.py:7-23): grads for volume and offset (never coords), mean and cov (never corr)."""
    rng = np.random.default_rng(90)
    E, H1, W1 = 1, 8, 16
    v = rng.standard_normal((E, H1, W1, H1, W1)).astype(np.float32)
    c = inputs.grid_coords(rng, E, H1, W1, 2.0)
    off = (3 * np.tanh(rng.standard_normal((E, H1, W1, 7, 7, 2)))).astype(np.float32)
    g = rng.standard_normal((E, 7, 7, H1, W1)).astype(np.float32)
    vd = dev(v).requires_grad_(True)
    od = dev(off).requires_grad_(True)
    cd = dev(c).requires_grad_(True)
    out = lgu.DefCorrSampler.apply(vd, cd, od, 3)
    out.backward(dev(g))
    vg_w, og_w = oracle.defCorr_index_backward(v, c, off.copy(), g, 3)
    assert cd.grad is None
    assert np.abs(host(vd.grad) - vg_w).max() <= 1e-5 * max(1.0, np.abs(vg_w).max())
    assert np.abs(host(od.grad) - og_w).max() <= 1e-5 * max(1.0, np.abs(og_w).max())
    vd2 = dev(v).requires_grad_(True)
    out2 = lgu.CorrSampler.apply(vd2, dev(c), 3)
    out2.backward(dev(g))
    pvg_w, = oracle.corr_index_backward(v, c, g, 3)
    assert np.abs(host(vd2.grad) - pvg_w).max() <= 1e-5 * max(1.0, np.abs(pvg_w).max())
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    means = (np.stack([xs, ys], -1)[None] + rng.standard_normal((E, H1, W1, 2))).astype(np.float32)
    covs = rng.uniform(0.5, 4.0, (E, H1, W1, 2)).astype(np.float32)
    md, kd, vd3 = dev(means).requires_grad_(True), dev(covs).requires_grad_(True), dev(v).requires_grad_(True)
    gv = rng.standard_normal(v.shape).astype(np.float32)
    lgu.GaussianMaskCuda.apply(md, kd, vd3, 4).backward(dev(gv))
    mg_w, cg_w = oracle.gaussianMask_backward(means, covs, v, gv, 4)
    assert vd3.grad is None
    assert np.abs(host(md.grad) - mg_w).max() <= 1e-5 * max(1.0, np.abs(mg_w).max())
    assert np.abs(host(kd.grad) - cg_w).max() <= 1e-5 * max(1.0, np.abs(cg_w).max())


def test_empty_edge_list(lgu):
    """E = 0 (a factor graph with no edges yet): every op returns correctly shaped empty tensors."""
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    assert lgu.ops.corr_index_forward(z(0, 8, 16, 8, 16), z(0, 2, 8, 16), 1)[0].shape == (0, 3, 3, 8, 16)
    assert lgu.ops.defCorr_index_forward(z(0, 8, 16, 8, 16), z(0, 2, 8, 16), z(0, 8, 16, 7, 7, 2), 3)[0].shape == (0, 7, 7, 8, 16)
    out = lgu.ops.defcorr_pyramid_forward([z(0, 8, 16, 8, 16), z(0, 8, 16, 4, 8)], z(0, 2, 8, 16), [z(0, 8, 16, 7, 7, 2)] * 2, 3)
    assert out.shape == (0, 98, 8, 16)
    assert lgu.ops.gaussianMask(z(0, 8, 16, 2), z(0, 8, 16, 2), z(0, 8, 16, 8, 16), 4)[0].numel() == 0
    assert lgu.ops.lowMem_defSample(z(0, 8, 16, 32), z(0, 8, 16, 32), z(0, 1, 8, 16, 2), z(1, 8, 16, 7, 7, 2), 3)[0].shape == (0, 1, 7, 7, 8, 16)
    assert lgu.ops.altcorr_forward(z(0, 8, 16, 32), z(0, 8, 16, 32), z(0, 1, 8, 16, 2), 1)[0].shape == (0, 1, 9, 8, 16)


def test_streams_and_graph_capture(lgu):
    """Kernels are enqueued on torch's CURRENT stream (reference: legacy default stream only) and the
    launch path makes no allocation / synchronisation, so a lookup can be captured in a HIP graph
    and replayed (the update() hot loop is launch-bound around this kernel)."""
    case = inputs.pyramid_case(121, 2, 24, 32, 3, 3, 3.0, 4.0, False)
    vols = [dev(v) for v in case["volumes"]]
    offs = [dev(o) if o is not None else None for o in case["offsets"]]
    coords = dev(case["coords"])
    want = lgu.ops.defcorr_pyramid_forward(vols, coords, offs, 3)
    plan = lgu.ops.DefcorrPyramidPlan(vols, offs, 3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out_side = plan(coords)
    side.synchronize()
    assert torch.equal(out_side, want)
    static_out = torch.zeros_like(want)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan(coords, out=static_out)
    static_out.zero_()
    coords.copy_(coords + 0.25)  # new inputs in the captured buffers
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, lgu.ops.defcorr_pyramid_forward(vols, coords, offs, 3))


def test_torch_ops_namespace(lgu):
    from lgu_slam_amd import torch_ops  # noqa: F401  (registers torch.ops.lgu.*)
    case = inputs.pyramid_case(81, 1, 12, 16, 1, 3, 3.0, 4.0, True)
    v, c, o = dev(case["volumes"][0]), dev(case["coords"]), dev(case["offsets"][0])
    a = torch.ops.lgu.defCorr_index_forward(v, c, o.clone(), 3)
    b = lgu.ops.defCorr_index_forward(v, c, o.clone(), 3)
    assert isinstance(a, (list, tuple)) and torch.equal(a[0], b[0])


def test_baseline_config1_plumbing_case(lgu, oracle):
    """BASELINE config 1: single frame pair, 30x40 fmap, 1 pyramid level, r=3 — full oracle check
    (W1 = 40 is 2.5 tiles of 16; the dispatcher picks 16-pixel tiles here)."""
    case = inputs.pyramid_case(101, 1, 30, 40, 1, 3, 3.0, 4.0, True)
    want, _ = oracle_pyramid(oracle, case, 3)
    got, _ = run_pyramid(lgu, case, 3, 0)
    assert got.shape == (1, 49, 30, 40) and np.abs(got - want).max() <= 1e-6


def test_radius_above_three_uses_generic_kernel(lgu, oracle):
    for radius in (4, 5):
        case = inputs.pyramid_case(110 + radius, 1, 12, 16, 2, radius, 3.0, 4.0, True)
        want, _ = oracle_pyramid(oracle, case, radius)
        got, _ = run_pyramid(lgu, case, radius, 0)
        assert np.abs(got - want).max() <= 1e-6
        v, c, off = case["volumes"][0], case["coords"], case["offsets"][0]
        rd = 2 * radius + 1
        g = np.random.default_rng(radius).standard_normal((1, rd, rd, 12, 16)).astype(np.float32)
        vg_w, og_w = oracle.defCorr_index_backward(v, c, off.copy(), g, radius)
        vg, og = lgu.ops.defCorr_index_backward(dev(v), dev(c), dev(off), dev(g), radius)
        assert np.abs(host(vg) - vg_w).max() <= 1e-5 * max(1.0, np.abs(vg_w).max())
        assert np.abs(host(og) - og_w).max() <= 1e-5 * max(1.0, np.abs(og_w).max())


@pytest.mark.parametrize("E", [40, 48])
def test_large_edge_counts_stereo_and_frontend_cap(lgu, E):
    """BASELINE config 3 (~40 edges, stereo + temporal) and the frontend cap of 48 edges
    (droid_frontend.py:13): 2.0 / 2.4 GB of pyramid — 64-bit slice addressing.  Checked by
    agreement of the fused launch with per-level launches and with the generic kernel on the
    LAST edges (highest addresses)."""
    torch.manual_seed(E)
    H1, W1, L, r = 48, 64, 4, 3
    vols = [torch.randn(E, H1, W1, H1 >> l, W1 >> l, device="cuda") for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H1, device="cuda").float(), torch.arange(W1, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys])[None] + 3 * torch.randn(E, 2, H1, W1, device="cuda")).contiguous()
    o0 = 4 * torch.tanh(torch.randn(E, H1, W1, 7, 7, 2, device="cuda"))
    o1 = (4 * torch.tanh(torch.randn(E, H1, W1, 7, 7, 2, device="cuda")) + o0) / 2
    out = lgu.ops.defcorr_pyramid_forward(vols, coords, [o0, o1, None, None], r)
    tail = slice(E - 2, E)
    os.environ["LGU_DEFCORR_VARIANT"] = "2"
    gen = lgu.ops.defcorr_pyramid_forward([v[tail].contiguous() for v in vols], coords[tail].contiguous(),
                                          [o0[tail].contiguous(), o1[tail].contiguous(), None, None], r)
    os.environ.pop("LGU_DEFCORR_VARIANT")
    assert torch.equal(out[tail], gen)
    c0, = lgu.ops.defCorr_index_forward(vols[0], coords, o0, r)
    assert torch.equal(c0.view(E, 49, H1, W1), out[:, :49])


@pytest.mark.parametrize("probe", [False, True])
def test_config3_forty_edges_against_the_oracle(lgu, oracle, probe):
    """BASELINE config 3 at full size (EuRoC stereo: E = 40 edges of 48x64, L = 4, r = 3, 2 GB of pyramid) through the
    production path (tiled pyramid, lean kernel, probe fused or not): the first, a middle and the LAST edge (highest
    addresses) against the C oracle — the reference's per-level kernels composed as corr.py:88-109 does."""
    torch.manual_seed(40)
    E, H1, W1, L, r = 40, 48, 64, 4, 3
    vols = [torch.randn(E, H1, W1, H1 >> l, W1 >> l, device="cuda") for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H1, device="cuda").float(), torch.arange(W1, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys])[None] + 3 * torch.randn(E, 2, H1, W1, device="cuda")).contiguous()
    o0 = 4 * torch.tanh(torch.randn(E, H1, W1, 7, 7, 2, device="cuda"))
    o1 = (4 * torch.tanh(torch.randn(E, H1, W1, 7, 7, 2, device="cuda")) + o0) / 2
    sel = [0, 17, E - 1]
    host_in = ([host(v[sel]) for v in vols], host(coords[sel]), [host(o0[sel]).copy(), host(o1[sel]).copy(), None, None])
    tv = [lgu.ops.volume_retile(v) for v in vols]
    hw = [(H1 >> l, W1 >> l) for l in range(L)]
    got = lgu.ops.defcorr_pyramid_forward(tv, coords, [o0, o1, None, None], r, probe=probe, tiled=True, level_hw=hw)
    pv, pc, po = host_in
    if probe:   # corr.py:94-99: 3x3 plain sample of level 1, unbiased variance, sigmoid, offset[1] *= mask
        pr, = oracle.corr_index_forward(pv[1], (pc / 2).astype(np.float32), 1)
        var = torch.var(torch.from_numpy(pr).permute(0, 3, 4, 1, 2), dim=[3, 4])
        po[1] = (po[1] * torch.sigmoid(var).numpy().reshape(len(sel), H1, W1, 1, 1, 1)).astype(np.float32)
    want = oracle.defcorr_pyramid_forward(pv, pc, po, r)
    assert np.abs(host(got[sel]) - want).max() <= (2e-5 if probe else 1e-5)
    if probe:   # the persistent offset[1] *= mask landed in the caller's tensor
        assert np.abs(host(o1[sel]) - po[1]).max() <= 1e-5


def test_config5_shard_of_250_edges_against_the_oracle(lgu, oracle):
    """BASELINE config 5's per-GPU shard (250 edges over 40 frames of 60x80x128 half feature maps, all levels, frame
    buffers read in place, chunk-planar target maps = what ShardedAltCorr launches): the first and the LAST edge against
    the C oracle of lowMem_defSample on the float copies, level by level (incl. the reference's offset[b*n] indexing:
    every edge reads edge 0's offsets)."""
    torch.manual_seed(18)
    F_, H, W, C, L, E = 40, 60, 80, 128, 4, 250
    frames = [(torch.randn(F_, H >> l, W >> l, C, device="cuda") * 0.125).half() for l in range(L)]
    ii = torch.randint(0, F_, (E,), device="cuda")
    jj = torch.randint(0, F_, (E,), device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(E, 1, H, W, 2, device="cuda")).contiguous()
    o0 = (4 * torch.tanh(torch.randn(1, H, W, 7, 7, 2, device="cuda"))).contiguous()
    o1 = ((4 * torch.tanh(torch.randn(1, H, W, 7, 7, 2, device="cuda")) + o0) / 2).contiguous()
    o0_h, o1_h = host(o0).copy(), host(o1).copy()
    out = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], [lgu.ops.lowmem_chunked(f) for f in frames], coords,
                                               [o0, o1, None, None], 3, ii=ii, jj=jj, chunked=True)
    assert out.shape == (E, 1, 196, H, W) and torch.isfinite(out).all()
    zero = np.zeros_like(o0_h)
    for e in (0, E - 1):
        f1 = host(frames[0][ii[e]].float())[None]
        for l, off in enumerate((o0_h, o1_h, zero, zero)):
            f2 = host(frames[l][jj[e]].float())[None]
            want, = oracle.lowMem_defSample(f1, f2, (host(coords[e:e + 1]) / 2 ** l).astype(np.float32), off.copy(), 3)
            got = host(out[e, 0, 49 * l:49 * (l + 1)]).reshape(want.shape[2:])
            assert np.abs(got - want[0, 0]).max() <= 1e-5, (e, l)


LEAN_CASES = {
    # name: (E, H1, W1, H2, W2, sigma): source size H1 x W1 (W1 % 16 != 0: partial last tile), target slices H2 x W2
    "partial_tile": (2, 20, 27, 48, 64, 3.0),
    "one_tile_wide": (3, 5, 9, 24, 32, 3.0),          # W1 < 16: a single, partial tile per row; magic divisor 1
    "border_heavy": (2, 12, 40, 24, 32, 15.0),        # most taps out of bounds
    "one_row": (1, 1, 33, 48, 64, 3.0),               # H1 = 1: magic divisor 1 for the row -> edge split
}


@pytest.mark.parametrize("mode", ["plain", "coords_last", "slots"])
@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("probe", [False, True])
@pytest.mark.parametrize("name", list(LEAN_CASES))
def test_lean_production_kernel_against_the_oracle(lgu, oracle, name, probe, tiled, mode):
    """csrc/defcorr_lean.hip (radius 3, four levels, offsets on levels 0-1) on shapes its fast paths do not see at the
    BASELINE sizes: partial tiles (pixels beyond the row end are computed on a duplicate of the last pixel and must
    neither be stored nor — with the probe — scale that pixel's offsets twice), single-tile rows, H1 = 1, border-heavy
    coords; with planar / interleaved coords and slot-indirected volumes.  Against the C oracle's composition
    (probe + mask + four levels), offsets' in-place side effects included."""
    E, H1, W1, H2, W2, sigma = LEAN_CASES[name]
    rng = np.random.default_rng(700 + list(LEAN_CASES).index(name))
    vols_np = inputs.volume_pyramid(rng, E, H1, W1, 4, H2, W2)
    coords_np = ((inputs.grid_coords(rng, E, H1, W1, sigma)) * np.array([W2 / W1, H2 / H1], np.float32).reshape(1, 2, 1, 1)).astype(np.float32)
    offs_np = inputs.canonical_offsets(rng, E, H1, W1, 4)
    hw = [(H2 >> l, W2 >> l) for l in range(4)]
    vols = [dev(v) for v in vols_np]
    slots = None
    if mode == "slots":    # volumes live in a larger buffer at permuted slots
        perm = torch.randperm(E + 2, device="cuda")[:E].to(torch.int32)
        big = []
        for v in vols:
            b = torch.randn((E + 2,) + tuple(v.shape[1:]), device="cuda")
            b[perm.long()] = v
            big.append(b)
        vols, slots = big, perm.contiguous()
    if tiled:
        vols = [lgu.ops.volume_retile(v.contiguous()) for v in vols]
    coords = dev(coords_np)
    offs = [dev(o) if o is not None else None for o in offs_np]
    plan = lgu.ops.DefcorrPyramidPlan(vols, offs, 3, probe=probe, tiled=tiled, level_hw=hw if tiled else None,
                                      coords_last=(mode == "coords_last"), slots=slots)
    got = plan(coords.permute(0, 2, 3, 1).contiguous() if mode == "coords_last" else coords)
    po = [o.copy() if o is not None else None for o in offs_np]
    if probe:
        pr, = oracle.corr_index_forward(vols_np[1], (coords_np / 2).astype(np.float32), 1)
        var = torch.var(torch.from_numpy(pr).permute(0, 3, 4, 1, 2), dim=[3, 4])
        po[1] = (po[1] * torch.sigmoid(var).numpy().reshape(E, H1, W1, 1, 1, 1)).astype(np.float32)
    want = oracle.defcorr_pyramid_forward(vols_np, coords_np, po, 3)
    assert np.abs(host(got) - want).max() <= (2e-5 if probe else 1e-5)
    # in-place side effects: centres zeroed; level-1 offsets scaled ONCE by the mask
    assert float(offs[0][:, :, :, 3, 3].abs().max()) == 0 and float(offs[1][:, :, :, 3, 3].abs().max()) == 0
    assert np.abs(host(offs[1]) - po[1]).max() <= 1e-5
    o0_want = offs_np[0].copy()
    o0_want[:, :, :, 3, 3] = 0
    assert np.array_equal(host(offs[0]), o0_want)


def test_full_size_properties(lgu):
    """BASELINE cfg2 size (E=20, 48x64, L=4, r=3): size-independent properties instead of
    a full oracle run — linearity in the volume, zero-offset == plain sampler, variant
    agreement, idempotence of the in-place centre zeroing."""
    torch.manual_seed(0)
    E, H1, W1, L, r = 20, 48, 64, 4, 3
    vols = [torch.randn(E, H1, W1, H1 >> l, W1 >> l, device="cuda") for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H1, device="cuda").float(), torch.arange(W1, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys])[None] + 3 * torch.randn(E, 2, H1, W1, device="cuda")).contiguous()
    o0 = 4 * torch.tanh(torch.randn(E, H1, W1, 7, 7, 2, device="cuda"))
    o1 = (4 * torch.tanh(torch.randn(E, H1, W1, 7, 7, 2, device="cuda")) + o0) / 2
    offs = [o0, o1, None, None]
    out = lgu.ops.defcorr_pyramid_forward(vols, coords, offs, r)
    assert out.shape == (E, 196, H1, W1) and torch.isfinite(out).all()
    again = lgu.ops.defcorr_pyramid_forward(vols, coords, offs, r)
    assert torch.equal(out, again)  # deterministic + centre zeroing idempotent
    assert (o0[:, :, :, 3, 3] == 0).all() and (o1[:, :, :, 3, 3] == 0).all()
    # per-level launches through the reference-shaped op agree bit-for-bit with the fused launch
    for l in range(L):
        off_l = offs[l] if offs[l] is not None else torch.zeros_like(o0)
        c, = lgu.ops.defCorr_index_forward(vols[l], (coords / 2 ** l).contiguous(), off_l, r)
        assert torch.equal(c.view(E, 49, H1, W1), out[:, 49 * l:49 * (l + 1)])
    # linearity in the volume: f(2v + w) == 2 f(v) + f(w) to rounding
    w = [torch.randn_like(v) for v in vols]
    lhs = lgu.ops.defcorr_pyramid_forward([2 * v + x for v, x in zip(vols, w)], coords, offs, r)
    rhs = 2 * out + lgu.ops.defcorr_pyramid_forward(w, coords, offs, r)
    assert (lhs - rhs).abs().max() <= 1e-4
    # variants agree exactly
    os.environ["LGU_DEFCORR_VARIANT"] = "2"
    gen = lgu.ops.defcorr_pyramid_forward(vols, coords, offs, r)
    assert torch.equal(gen, out)


@pytest.mark.parametrize("tiled", [True, False])
def test_corrblock_matches_reference_shaped_composition(lgu, oracle, tiled, monkeypatch):
    """Host glue: CorrBlock.__call__ (fused launch) == probe + mask + 4 per-level oracle
    calls + cat, including the persistent offset[1] *= mask state across two calls.  Run over both
    storage layouts of the block's own pyramid (tiled = production, row-major = reference)."""
    monkeypatch.setattr(lgu.CorrBlock, "TILED_PYRAMID", tiled)
    torch.manual_seed(3)
    E, h, w = 2, 48, 64
    dev_ = "cuda"
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    GA = lgu.GaussianMask(h, w).to(dev_)
    torch.nn.init.normal_(GA.meanMap.weight, 0, 0.3)
    f1 = torch.randn(1, E, 128, h, w, device=dev_) * 0.5
    f2 = torch.randn(1, E, 128, h, w, device=dev_) * 0.5
    ys, xs = torch.meshgrid(torch.arange(h, device=dev_).float(), torch.arange(w, device=dev_).float(), indexing="ij")
    with torch.no_grad():
        blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        assert blk._tiled == tiled
        rowmajor = [lgu.ops.volume_retile(v.contiguous(), to_tiled=False, hw=blk._level_hw[i]) if tiled else v
                    for i, v in enumerate(blk.corr_pyramid)]
        pyr = [host(v) for v in rowmajor]
        # fused __init__ (volume_pyramid) vs the reference-shaped composition GA.forward + avg_pool2d
        vol = lgu.CorrBlock.corr(f1, f2).view(E, h, w, h, w).float()
        ref0, _, _ = GA(blk.t, vol)
        assert (rowmajor[0] - ref0).abs().max() <= 1e-6
        ref1 = torch.nn.functional.avg_pool2d(ref0.reshape(E * h * w, 1, h, w), 2, stride=2).view(E, h, w, h // 2, w // 2)
        assert (rowmajor[1] - ref1).abs().max() <= 1e-6
        offs = [host(o.contiguous()).reshape(E, h, w, 7, 7, 2).copy() for o in blk.offset]
        for it in range(2):
            coords1 = (torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, h, w, 2, device=dev_))
            got, mean_n, theta = blk(coords1)
            assert got.shape == (1, E, 196, h, w) and mean_n.shape == (1, E, h, w, 2) and theta.shape == (1, E, h, w)
            c = host(coords1.permute(0, 1, 4, 2, 3).contiguous().view(E, 2, h, w))
            probe, = oracle.corr_index_forward(pyr[1], (c / 2).astype(np.float32), 1)
            var = torch.var(torch.from_numpy(probe).permute(0, 3, 4, 1, 2), dim=[3, 4])
            mask = torch.sigmoid(var).numpy().reshape(E, h, w, 1, 1, 1)
            offs[1] = (offs[1] * mask).astype(np.float32)
            want = oracle.defcorr_pyramid_forward(pyr, c, [offs[0], offs[1], None, None], 3)
            assert np.abs(host(got)[0] - want).max() <= 2e-5, "call %d" % it


@pytest.mark.gpu
@pytest.mark.parametrize("tiled", [True, False])
def test_corrblock_under_autocast_matches_the_torch_composition_under_autocast(lgu, tiled, monkeypatch):
    """The reference builds CorrBlock inside autocast (factor_graph.py:90 `add_factors`): GA's heads run in half, `det`
    is a half tensor and `6.28 * torch.sqrt(det)` (gaussianMask_cuda.py:79-86) is rounded to half after the square root
    and after the product.  The fused builder (ops.volume_pyramid with the det handed over) must give what the torch
    composition GA.forward + avg_pool2d gives under the SAME autocast — not the fp32 evaluation of the denominator, which
    is ~1e-3 relative off in the Gaussian term."""
    monkeypatch.setattr(lgu.CorrBlock, "TILED_PYRAMID", tiled)
    torch.manual_seed(5)
    E, h, w = 2, 48, 64
    dev_ = "cuda"
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    GA = lgu.GaussianMask(h, w).to(dev_)
    torch.nn.init.normal_(GA.meanMap.weight, 0, 0.3)
    f1 = (torch.randn(1, E, 128, h, w, device=dev_) * 0.5).half()
    f2 = (torch.randn(1, E, 128, h, w, device=dev_) * 0.5).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        assert blk._store is not None and blk._tiled == tiled          # the fused builder ran
        got = [lgu.ops.volume_retile(v.contiguous(), to_tiled=False, hw=blk._level_hw[i]) if tiled else v
               for i, v in enumerate(blk.corr_pyramid)]
        # the reference-shaped composition under the same autocast (torch ops + the gaussianMask operator, which the
        # reference-build tests hold to the reference's kernel)
        monkeypatch.setattr(lgu.gaussian_mask, "FUSED_PARAMS", False)
        vol = lgu.CorrBlock.corr(f1, f2).view(E, h, w, h, w).float()
        ref0, mean_ref, det_ref = GA(blk.t, vol)
        assert det_ref.dtype == torch.float16                           # the premise: det is half in this context
        lv = ref0.reshape(E * h * w, 1, h, w)
        for i in range(4):
            want = lv.view(E, h, w, h >> i, w >> i)
            scale = float(want.abs().max())
            assert float((got[i] - want).abs().max()) <= 1e-6 * max(scale, 1.0), "level %d" % i   # measured: 0.0
            lv = torch.nn.functional.avg_pool2d(lv, 2, stride=2)
        # and the fp32 evaluation of the denominator is measurably different here (what round 1 shipped)
        fp32_den = lgu.ops.volume_pyramid(mean_ref.float().contiguous(),
                                          (torch.sigmoid(lgu.gaussian_mask.per_Corr_Normalization(
                                              GA.covMap(GA.mapA(blk.t)).view(E, h * w, 2), [1, 2])) * 5 + 0.05).view(E, h, w, 2).float().contiguous(),
                                          vol.clone(), 1)[0]
        assert float((fp32_den - ref0).abs().max()) > float((got[0] - ref0).abs().max())


@pytest.mark.gpu
def test_corrblock_routes_gradients_to_trainable_offset_heads(lgu):
    """Fine-tuning only ofsMap / ofs_residual (frozen feature maps and GA): the lookup must stay on the autograd path —
    the fused launch would return a tensor without grad_fn and scale the tracked offsets through raw pointers."""
    torch.manual_seed(9)
    E, h, w = 1, 24, 32
    dev_ = "cuda"
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev_)
    GA = lgu.GaussianMask(h, w).to(dev_)
    for q in GA.parameters():
        q.requires_grad_(False)
    f1 = torch.randn(1, E, 128, h, w, device=dev_) * 0.5
    f2 = torch.randn(1, E, 128, h, w, device=dev_) * 0.5
    ys, xs = torch.meshgrid(torch.arange(h, device=dev_).float(), torch.arange(w, device=dev_).float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + torch.randn(1, E, h, w, 2, device=dev_)
    blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
    assert blk._store is None and blk.offset[0].requires_grad      # not the inference store
    out, _, _ = blk(coords)
    assert out.requires_grad and out.shape == (1, E, 196, h, w)
    out.square().mean().backward()
    assert ofsMap.weight.grad is not None and float(ofsMap.weight.grad.abs().max()) > 0
    assert ofsRes.weight.grad is not None and float(ofsRes.weight.grad.abs().max()) > 0
    # a block built for inference whose offsets are later made trainable leaves the store on its first tracked lookup
    with torch.no_grad():
        blk2 = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
    assert blk2._store is not None
    blk2.offset[0] = blk2.offset[0].clone().requires_grad_(True)
    out2, _, _ = blk2(coords)
    assert out2.requires_grad and blk2._store is None
    with torch.no_grad():                                           # same numbers as the inference lookup
        blk3 = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        out3, _, _ = blk3(coords)
    assert float((out2.detach() - out3).abs().max()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.float64])
def test_volume_operators_accept_half_and_double(lgu, dt):
    """The reference dispatches the volume-path kernels for half and double too (AT_DISPATCH_FLOATING_TYPES_AND_HALF);
    here those dtypes run the fp32 kernels on fp32 copies: outputs = the float operator's outputs in the operands'
    dtype, the centre-zeroing side effect lands in the caller's offset tensor."""
    ops = lgu.ops
    case = inputs.pyramid_case(77, 2, 12, 16, 1, 3)
    vol = dev(case["volumes"][0]).to(dt)
    coords = dev(case["coords"])
    off = dev(case["offsets"][0]).to(dt)
    assert float(off[:, :, :, 3, 3].abs().max()) > 0
    off32 = off.float()
    want, = ops.defCorr_index_forward(vol.float(), coords, off32, 3)
    got, = ops.defCorr_index_forward(vol, coords, off, 3)
    assert got.dtype == dt and torch.equal(got, want.to(dt))
    assert float(off[:, :, :, 3, 3].abs().max()) == 0 and torch.equal(off, off32.to(dt))
    g = torch.randn_like(want)
    vg, og = ops.defCorr_index_backward(vol, coords, off, g.to(dt), 3)
    vg32, og32 = ops.defCorr_index_backward(vol.float(), coords, off.float(), g.to(dt).float(), 3)
    assert vg.dtype == dt and og.dtype == dt and torch.equal(vg, vg32.to(dt)) and torch.equal(og, og32.to(dt))
    p1, = ops.corr_index_forward(vol, coords, 1)
    assert torch.equal(p1, ops.corr_index_forward(vol.float(), coords, 1)[0].to(dt))
    gv, = ops.corr_index_backward(vol, coords, torch.ones_like(p1), 1)
    assert gv.dtype == dt and gv.shape == vol.shape
    means = (coords.permute(0, 2, 3, 1).contiguous()).to(dt)
    covs = (torch.rand_like(means.float()) * 3 + 0.5).to(dt)
    m1, = ops.gaussianMask(means, covs, vol, 2)
    assert m1.dtype == dt and torch.equal(m1, ops.gaussianMask(means.float(), covs.float(), vol.float(), 2)[0].to(dt))
    mg, cg = ops.gaussianMask_backward(means, covs, vol, torch.ones_like(vol), 2)
    assert mg.dtype == dt and cg.dtype == dt and mg.shape == means.shape


TILED_CASES = {
    # name: (seed, E, H1, W1, L, sigma, off_scale, dense)
    "cfg2_shape": (31, 2, 48, 64, 4, 3.0, 4.0, False),          # level 3 is 6x8: padded to 8x8 in the tiled form
    "border_stress": (32, 2, 24, 32, 3, 20.0, 4.0, False),
    "dense_offsets": (33, 1, 24, 32, 3, 3.0, 4.0, True),
    "ragged": (34, 2, 30, 40, 2, 3.0, 4.0, False),              # 30x40 and 15x20 slices: both dims padded
    "huge_offsets": (35, 1, 48, 64, 2, 3.0, 14.0, True),
    "w2_not_mult4": (36, 1, 20, 24, 3, 3.0, 4.0, False),        # level 2 is 5x6: row-major needs the generic kernel, tiled does not
}


@pytest.mark.parametrize("variant", [0, 4, 5])
@pytest.mark.parametrize("probe", [False, True])
@pytest.mark.parametrize("name", list(TILED_CASES))
def test_tiled_pyramid_layout_is_bitwise_the_reference_layout(lgu, oracle, name, probe, variant):
    """LGU_PYR_TILED: the same lookup over the 4x8-tiled slice layout.  volume_retile round-trips exactly,
    the tiled lookup equals the row-major lookup BIT FOR BIT (same loads, same arithmetic, other addresses),
    offsets get the same side effects, and the result matches the oracle."""
    seed, E, H1, W1, L, sigma, osc, dense = TILED_CASES[name]
    case = inputs.pyramid_case(seed, E, H1, W1, L, 3, sigma, osc, dense)
    vols = [dev(v) for v in case["volumes"]]
    hw = [tuple(v.shape[3:]) for v in vols]
    tv = [lgu.ops.volume_retile(v) for v in vols]
    for v, t, (h2, w2) in zip(vols, tv, hw):
        assert tuple(t.shape) == lgu.ops.tiled_shape(E, H1, W1, h2, w2)
        assert torch.equal(lgu.ops.volume_retile(t, to_tiled=False, hw=(h2, w2)), v)
    coords = dev(case["coords"])
    offs_a = [dev(o) if o is not None else None for o in case["offsets"]]
    offs_b = [dev(o) if o is not None else None for o in case["offsets"]]
    set_variant(variant)
    got = lgu.ops.defcorr_pyramid_forward(tv, coords, offs_a, 3, probe=probe, tiled=True, level_hw=hw)
    set_variant(2 if (name == "w2_not_mult4" and not probe) else variant)
    if name == "w2_not_mult4" and probe:
        want_np = oracle.defcorr_pyramid_forward(case["volumes"], case["coords"],
                                                 [o.copy() if o is not None else None for o in case["offsets"]], 3, probe=True)
        assert np.abs(host(got) - want_np).max() <= 1e-5
        return
    want = lgu.ops.defcorr_pyramid_forward(vols, coords, offs_b, 3, probe=probe)
    assert torch.equal(got, want)
    for a, b in zip(offs_a, offs_b):
        assert a is None or torch.equal(a, b)
    ref_offs = [o.copy() if o is not None else None for o in case["offsets"]]
    want_np = oracle.defcorr_pyramid_forward(case["volumes"], case["coords"], ref_offs, 3, probe=probe)
    assert np.abs(host(got) - want_np).max() <= (1e-5 if probe else 1e-6)


@pytest.mark.parametrize("fmt", ["nhwc", "nhwc_f16"])
@pytest.mark.parametrize("probe", [False, True])
@pytest.mark.parametrize("name", list(TILED_CASES))
def test_channel_last_output_is_the_planar_output(lgu, name, probe, fmt):
    """LGU_PYR_OUT_NHWC / LGU_PYR_OUT_F16: the lookup emitted in the consumer convolution's format.  Same values as
    the reference's planar tensor BIT FOR BIT (fp32), or exactly its Tensor.half() (round-to-nearest-even), with the
    same offset side effects; the returned tensor has the reference's logical shape and channels-last strides."""
    seed, E, H1, W1, L, sigma, osc, dense = TILED_CASES[name]
    case = inputs.pyramid_case(seed, E, H1, W1, L, 3, sigma, osc, dense)
    hw = [tuple(v.shape[3:]) for v in case["volumes"]]
    tv = [lgu.ops.volume_retile(dev(v)) for v in case["volumes"]]
    coords = dev(case["coords"])
    offs_a = [dev(o) if o is not None else None for o in case["offsets"]]
    offs_b = [dev(o) if o is not None else None for o in case["offsets"]]
    want = lgu.ops.defcorr_pyramid_forward(tv, coords, offs_a, 3, probe=probe, tiled=True, level_hw=hw)
    got = lgu.ops.defcorr_pyramid_forward(tv, coords, offs_b, 3, probe=probe, tiled=True, level_hw=hw, out_format=fmt)
    assert tuple(got.shape) == tuple(want.shape)
    assert got.is_contiguous(memory_format=torch.channels_last) or got.permute(0, 2, 3, 1).is_contiguous()
    if fmt == "nhwc":
        assert got.dtype == torch.float32 and torch.equal(got, want)
    else:
        assert got.dtype == torch.float16 and torch.equal(got, want.half())
    for a, b in zip(offs_a, offs_b):
        assert a is None or torch.equal(a, b)
    # a caller-provided buffer; the slot-indirected entry
    plan = lgu.ops.DefcorrPyramidPlan(tv, [dev(o) if o is not None else None for o in case["offsets"]], 3, probe=probe,
                                      tiled=True, level_hw=hw, out_format=fmt,
                                      slots=torch.arange(E, dtype=torch.int32, device="cuda"))
    buf = torch.full((E, H1, W1, want.shape[1]), 7.0, dtype=got.dtype, device="cuda").permute(0, 3, 1, 2)
    assert plan(coords, out=buf) is buf and torch.equal(buf, got)


def test_channel_last_output_argument_checks(lgu):
    v = [torch.randn(1, 8, 16, 8, 16, device="cuda"), torch.randn(1, 8, 16, 4, 8, device="cuda")]
    tv = [lgu.ops.volume_retile(x) for x in v]
    c = torch.zeros(1, 2, 8, 16, device="cuda")
    with pytest.raises(lgu._lib.UnsupportedShape):   # row-major pyramid: channel-last output is the tiled kernel's
        lgu.ops.defcorr_pyramid_forward(v, c, [None, None], 3, out_format="nhwc")
    with pytest.raises(RuntimeError):
        lgu.ops.defcorr_pyramid_forward(tv, c, [None, None], 3, tiled=True, out_format="nchw_f16")
    with pytest.raises(RuntimeError):                # planar buffer handed to a channel-last plan
        lgu.ops.defcorr_pyramid_forward(tv, c, [None, None], 3, tiled=True, out_format="nhwc",
                                        out=torch.empty(1, 98, 8, 16, device="cuda"))
    with pytest.raises(RuntimeError):                # fp32 buffer handed to the half plan
        lgu.ops.defcorr_pyramid_forward(tv, c, [None, None], 3, tiled=True, out_format="nhwc_f16",
                                        out=torch.empty(1, 8, 16, 98, device="cuda").permute(0, 3, 1, 2))
    lib = lgu._lib.load()                            # F16 without NHWC is a bad argument at the C ABI
    import ctypes
    vp = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in tv])
    op = (ctypes.c_void_p * 2)(None, None)
    hs, ws = (ctypes.c_int * 2)(8, 4), (ctypes.c_int * 2)(16, 8)
    o = torch.empty(1, 98, 8, 16, device="cuda")
    rc = lib.lgu_defcorr_pyramid_fwd_f32(vp, ctypes.c_void_p(c.data_ptr()), op, ctypes.c_void_p(o.data_ptr()), 2, 1, 8, 16,
                                         hs, ws, 3, lgu.ops.PYR_TILED | lgu.ops.PYR_OUT_F16, None)
    assert rc == lgu._lib.LGU_E_BADARG


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("shape,L", [((2, 12, 16, 12, 16), 3), ((1, 24, 32, 24, 32), 4), ((2, 6, 8, 20, 24), 2)])
def test_volume_pyramid_half_input_is_the_float_input(lgu, shape, L, tiled):
    """lgu_volume_pyramid_h16: the raw volume handed over in half (the matmul of half feature maps) gives BIT FOR BIT
    the levels of volume.float() (the conversion is exact and happens in the kernel's load); never in place."""
    torch.manual_seed(sum(shape) + L)
    E, H1, W1, H2, W2 = shape
    vol16 = (torch.randn(*shape, device="cuda") * 0.3).half()
    means = torch.stack([torch.rand(E, H1, W1, device="cuda") * W2, torch.rand(E, H1, W1, device="cuda") * H2], -1).contiguous()
    covs = (torch.rand(E, H1, W1, 2, device="cuda") * 5 + 0.05).contiguous()
    keep = vol16.clone()
    got = lgu.ops.volume_pyramid(means, covs, vol16, L, 4, inplace=True, tiled=tiled)
    want = lgu.ops.volume_pyramid(means, covs, vol16.float(), L, 4, tiled=tiled)
    assert torch.equal(vol16, keep)
    for g, w_ in zip(got, want):
        assert g.dtype == torch.float32 and torch.equal(g, w_)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("shape", [(3, 24, 32), (2, 25, 37), (1, 48, 64)])
def test_fused_offset_post_processing_equals_the_torch_composition(lgu, shape, half, monkeypatch):
    """lgu_offsets_finalize (standardise + 4 tanh + residual mix + nearest upsampling + channel-last transposition in
    one pass) against the reference-shaped torch composition of corr.py:117-135 on the SAME convolution outputs (two
    runs of a half convolution may pick different algorithms and differ by an ulp, which the 4 tanh(x / std) amplifies).
    fp32: the same fp32 operations, the statistics differ in summation order only (<= 1e-5 on values in [-4, 4]).
    Half (autocast, as add_factors runs it): level 0 rounded to half at every step like the framework's half kernels,
    level 1 in fp32 (autocast promotes the nearest upsampling) — equal bit for bit unless a statistic lands on the
    other side of a half rounding boundary (max <= 2^-7, mean <= 1e-4 then).
    Odd sizes exercise the nearest-neighbour index map."""
    import lgu_slam_amd.corr as corr_mod
    E, h, w = shape
    torch.manual_seed(h * w + E)
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    feats = torch.randn(E, 256, h, w, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=half):
        o0 = ofsMap(feats)
        o1_low = ofsRes(torch.nn.functional.avg_pool2d(feats, kernel_size=2, stride=2))
        assert o0.dtype == (torch.float16 if half else torch.float32)
        monkeypatch.setattr(corr_mod, "FUSED_OFFSETS", True)
        got, zg = corr_mod.finish_offsets(o0, o1_low, 4)
        monkeypatch.setattr(corr_mod, "FUSED_OFFSETS", False)
        want, zw = corr_mod.finish_offsets(o0, o1_low, 4)
        full, _ = corr_mod.generate_offsets(ofsMap, ofsRes, feats, 4)   # the public entry takes the same route
        assert tuple(full[1].shape) == (E, h, w, 98)
    assert zg == zw == [False, False, True, True] and len(got) == 4
    for l in range(4):
        assert tuple(got[l].shape) == (E, h, w, 98) == tuple(want[l].shape)
    assert got[0].dtype == torch.float32 and got[0].is_contiguous() and got[1].is_contiguous()
    assert float(got[2].abs().max()) == 0.0 and float(got[3].abs().max()) == 0.0
    for l in range(2):
        d = (got[l] - want[l].float()).abs()
        if half:
            assert float(d.max()) <= 2.0 ** -7 and float(d.mean()) <= 1e-4, (l, float(d.max()), float(d.mean()))
        else:
            assert float(d.max()) <= 1e-5, (l, float(d.max()))
        assert float(want[l].float().abs().max()) > 1.0


@pytest.mark.parametrize("cfg", [(16, 8, 60, 80), (3, 4, 13, 21), (5, 3, 24, 32), (1, 2, 7, 9)])
def test_offset_head_on_the_matrix_cores_equals_the_fp32_convolution(lgu, cfg):
    """lgu_offset_conv_frames_h16: ofsMap(cat(frames[ii] * 4, frames[jj] * 4).float()) of AltCorrBlock.corr_fn
    (reference corr.py:174-189, :220) read straight from the half frame buffers, weights split into two half parts.
    Against the module's fp32 convolution on the materialised input: <= 1e-5 of the output range (the split truncates
    the weights at 2^-22; the library convolution itself is 1e-6 away from an fp64 evaluation).  Odd sizes exercise the
    image border, partial pixel tiles and the unaligned store path."""
    E, NF, H, W = cfg
    torch.manual_seed(E * 100 + H)
    conv = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    frames = (torch.randn(NF, H, W, 128, device="cuda") * 0.5 / 4).half()
    ii = torch.randint(0, NF, (E,), device="cuda")
    jj = torch.randint(0, NF, (E,), device="cuda")
    with torch.no_grad():
        feats = torch.cat(((frames[ii] * 4.0).permute(0, 3, 1, 2), (frames[jj] * 4.0).permute(0, 3, 1, 2)), dim=1).float().contiguous()
        want = conv(feats)
        want64 = torch.nn.functional.conv2d(feats.double(), conv.weight.double(), conv.bias.double(), padding=1)
        packed = lgu.ops.pack_offset_conv(conv.weight, conv.bias)
        got = lgu.ops.offset_conv_frames(frames, ii, jj, packed)
    assert tuple(got.shape) == (E, 98, H, W) and got.dtype == torch.float32
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-5 * scale
    assert float((got.double() - want64).abs().max()) <= 4 * max(float((want.double() - want64).abs().max()), 1e-7 * scale)
    if H < 2 or W < 2:
        return
    # two-part input (the residual head, corr.py:219-220): 2 x 2 averages of the frames need up to 24 bits
    with torch.no_grad():
        pooled4 = torch.nn.functional.avg_pool2d(feats, kernel_size=2, stride=2)     # what the reference feeds ofs_residual
        want2 = conv(pooled4)
        pl_ = torch.nn.functional.avg_pool2d(frames.permute(0, 3, 1, 2).float(), kernel_size=2, stride=2).permute(0, 2, 3, 1).contiguous()
        hi = pl_.half()
        lo = (pl_ - hi.float()).half()
        got2 = lgu.ops.offset_conv_frames(hi, ii, jj, packed, frames_lo=lo)
    assert tuple(got2.shape) == tuple(want2.shape)
    assert float((got2 - want2).abs().max()) <= 1e-5 * float(want2.abs().max())


@pytest.mark.parametrize("cfg", [(1, 3, 60, 80), (2, 4, 13, 21), (3, 3, 30, 40), (1, 2, 7, 9)])
def test_small_grid_offset_convolution_is_the_full_grid_kernel_bit_for_bit(lgu, cfg, monkeypatch):
    """One or a few edges per launch (AltCorrBlock's first-edge offsets) go to offconv_small_kernel: output channels split over
    blockIdx.z, a ring of one tap, every operand by LDS-DMA.  Every output element sees the same MFMA sequence as in the
    full-grid kernel (selected here with the debug switch LGU_OFFCONV_NOSMALL), so the results are EQUAL — for the plain
    head and for the two-part input of the residual head, image borders, partial pixel tiles and the odd channel-tile count
    (98 channels = 7 tiles in groups of 2) included."""
    E, NF, H, W = cfg
    torch.manual_seed(7 * E + W)
    conv = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    frames = (torch.randn(NF, H, W, 128, device="cuda") * 0.5 / 4).half()
    lo = (torch.randn(NF, H, W, 128, device="cuda") * 1e-4).half()
    ii = torch.randint(0, NF, (E,), device="cuda")
    jj = torch.randint(0, NF, (E,), device="cuda")
    packed = lgu.ops.pack_offset_conv(conv.weight, conv.bias)
    outs = []
    for full in ("0", "1"):
        monkeypatch.setenv("LGU_OFFCONV_NOSMALL", full)
        outs.append((_banded_call(lgu, frames, ii, jj, packed, None), _banded_call(lgu, frames, ii, jj, packed, lo)))
    monkeypatch.delenv("LGU_OFFCONV_NOSMALL")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 0.1 and not torch.equal(outs[0][0], outs[0][1])


def _banded_call(lgu, frames, ii, jj, packed, lo):
    return lgu.ops.offset_conv_frames(frames, ii, jj, packed, frames_lo=lo)


def test_altcorrblock_offsets_from_frames_equal_the_general_composition(lgu, monkeypatch):
    """AltCorrBlock's inference fast path for a half pyramid (level-0 head on the matrix cores from the stored frames,
    residual head on frames pooled once per block) against the reference-shaped composition (gather, x 4, cat, float,
    two fp32 convolutions): offsets equal to 2e-5 (values in [-4, 4]; the split-half weights move the convolution
    output by ~2e-6, which 4 tanh(x / std) amplifies), the lookup to 1e-4 of its range; repeated calls reuse the
    pooled frames."""
    import lgu_slam_amd.corr as corr_mod
    torch.manual_seed(41)
    N, H, W, E = 5, 24, 32, 7
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    fm = (torch.randn(1, N, 128, H, W, device="cuda") * 0.5).half()
    ii = torch.randint(0, N, (E,), device="cuda")
    jj = torch.randint(0, N, (E,), device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, H, W, 2, device="cuda")).unsqueeze(-2)
    with torch.no_grad():
        alt = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
        assert alt._offsets_from_frames(1, ii, jj)
        fast_off = [o.clone() for o in alt.offset[:2]]
        out_fast = alt(coords, ii, jj).clone()
        out_fast2 = alt(coords, ii, jj)   # pooled frames and packed weights reused (the library's residual convolution may
        assert float((out_fast - out_fast2).abs().max()) <= 1e-5 * float(out_fast.abs().max())   # switch algorithm after its first run)
        monkeypatch.setattr(corr_mod, "FUSED_OFFSETS", False)
        ref = lgu.AltCorrBlock(ofsMap, ofsRes, None, fm)
        assert not ref._offsets_from_frames(1, ii, jj)
        out_ref = ref(coords, ii, jj)
        f1 = ref.pyramid[0][0][ii]
        f2 = ref.pyramid[0][0][jj]
        feats = torch.cat(((f1 * 4.0).permute(0, 3, 1, 2), (f2 * 4.0).permute(0, 3, 1, 2)), dim=1).float().contiguous()
        ref_off, _ = corr_mod.generate_offsets(ofsMap, ofsRes, feats, 4)
    for a_, b_ in zip(fast_off, ref_off[:2]):
        assert float((a_ - b_.float()).abs().max()) <= 2e-5
    assert tuple(out_fast.shape) == tuple(out_ref.shape)
    assert float((out_fast - out_ref).abs().max()) <= 1e-4 * float(out_ref.abs().max())


@pytest.mark.parametrize("shape", [(3, 24, 32), (2, 7, 9), (1, 60, 80)])
def test_probe_mask_scale_equals_the_torch_composition(lgu, shape):
    """lgu_probe_mask_scale_f32 == offset * sigmoid(var(probe, unbiased)) of corr.py:203-207, to fp32 rounding."""
    E, H, W = shape
    torch.manual_seed(E + H)
    probe = torch.randn(E, 1, 9, H, W, device="cuda") * 0.7
    off = torch.randn(E, H, W, 98, device="cuda") * 2
    want = off * torch.sigmoid(torch.var(probe.permute(0, 1, 3, 4, 2).contiguous().view(E, H, W, 3, 3), dim=[3, 4])).view(E, H, W, 1)
    got = lgu.ops.probe_mask_scale_(probe, off.clone())
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("shape", [(3, 24, 32), (2, 7, 9), (20, 48, 64)])
def test_fused_gaussian_parameters_equal_the_torch_composition(lgu, shape, half):
    """lgu_gaussian_params (the tail of GaussianMask.gaussian_parameters, gaussianMask_cuda.py:69-83, in one launch)
    against the torch composition on the same head outputs: fp32 <= 2e-6 relative (statistics differ in summation
    order); half (autocast) bit for bit unless a statistic lands across a half rounding boundary (then one half ulp of
    the covariance, <= 2^-8 at 5)."""
    from lgu_slam_amd.gaussian_mask import per_Corr_Normalization
    E, h, w = shape
    torch.manual_seed(E * 7 + h)
    dt = torch.float16 if half else torch.float32
    mean_ofs = (torch.randn(E, h, w, 2, device="cuda") * 0.7).to(dt)
    cov_raw = (torch.randn(E, h * w, 2, device="cuda") * 1.3 + 0.2).to(dt)
    cov = per_Corr_Normalization(cov_raw, [1, 2])
    cov = torch.sigmoid(cov) * 5 + 0.05
    det_w = cov[:, :, 0] * cov[:, :, 1]
    cov_w = cov.view(E, h, w, 2).float()
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
    mean_w = torch.stack([xs, ys], dim=-1).expand(E, h, w, 2) + mean_ofs
    mean, cov_g, det = lgu.ops.gaussian_params(mean_ofs, cov_raw, h, w)
    assert mean.dtype == torch.float32 and cov_g.dtype == torch.float32 and det.dtype == dt and tuple(det.shape) == (E, h * w)
    assert torch.equal(mean, mean_w.float())
    if half:
        assert float((cov_g - cov_w).abs().max()) <= 2.0 ** -8
        assert float((det.float() - det_w.float()).abs().max()) <= 2.0 ** -5
        assert float((cov_g != cov_w).float().mean()) <= 0.02   # bit-identical but for boundary cases
    else:
        assert float((cov_g - cov_w).abs().max()) <= 2e-6 * 5.05
        assert float((det - det_w).abs().max()) <= 4e-6 * 25.5
    # the module takes the fused route in inference and the composition under grad
    GA = lgu.GaussianMask(h, w).cuda()
    x = torch.randn(E, h, w, 256, device="cuda")
    with torch.no_grad():
        m1, c1, d1 = GA.gaussian_parameters(x)
    import lgu_slam_amd.gaussian_mask as gm
    gm.FUSED_PARAMS = False
    try:
        with torch.no_grad():
            m2, c2, d2 = GA.gaussian_parameters(x)
    finally:
        gm.FUSED_PARAMS = True
    assert float((m1 - m2).abs().max()) <= 1e-5 and float((c1 - c2).abs().max()) <= 2e-5 and tuple(d1.shape) == tuple(d2.shape)


def test_tiled_layout_rejects_what_it_does_not_serve(lgu):
    v = [torch.randn(1, 8, 16, 8, 16, device="cuda"), torch.randn(1, 8, 16, 4, 8, device="cuda")]
    tv = [lgu.ops.volume_retile(x) for x in v]
    c = torch.zeros(1, 2, 8, 16, device="cuda")
    with pytest.raises(lgu._lib.UnsupportedShape):   # radius 1: only the production radius is tiled
        lgu.ops.defcorr_pyramid_forward(tv, c, [None, None], 1, tiled=True)
    with pytest.raises(RuntimeError):                # row-major tensors passed as tiled
        lgu.ops.defcorr_pyramid_forward(v, c, [None, None], 3, tiled=True)


@pytest.mark.parametrize("shape,L", [((2, 12, 16, 12, 16), 3), ((1, 48, 64, 48, 64), 4), ((2, 6, 8, 20, 24), 2)])
def test_volume_pyramid_fused_tiled(lgu, shape, L):
    """lgu_volume_pyramid_tiled_f32 == retile(lgu_volume_pyramid_f32) bit for bit, also converting level 0 in place."""
    E, H1, W1, H2, W2 = shape
    torch.manual_seed(40 + H2)
    v = torch.randn(E, H1, W1, H2, W2, device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H1, device="cuda").float(), torch.arange(W1, device="cuda").float(), indexing="ij")
    means = (torch.stack([xs * W2 / W1, ys * H2 / H1], -1)[None] + 2 * torch.randn(E, H1, W1, 2, device="cuda")).contiguous()
    covs = (torch.rand(E, H1, W1, 2, device="cuda") * 5 + 0.05).contiguous()
    ref = lgu.ops.volume_pyramid(means, covs, v, L, 4)
    got = lgu.ops.volume_pyramid(means, covs, v, L, 4, tiled=True)
    for l in range(L):
        assert torch.equal(got[l], lgu.ops.volume_retile(ref[l]))
    if H2 % 4 == 0 and W2 % 8 == 0:
        v2 = v.clone()
        inp = lgu.ops.volume_pyramid(means, covs, v2, L, 4, inplace=True, tiled=True)
        assert inp[0].data_ptr() == v2.data_ptr() and torch.equal(inp[0], got[0]) and torch.equal(inp[-1], got[-1])


def test_corrblock_layouts_agree_and_cat(lgu, monkeypatch):
    """CorrBlock built with the tiled pyramid == built with the row-major pyramid (to 2e-5: the offset convs and
    the volume matmul of two constructions are not bitwise reproducible; the lookup itself is, see
    test_tiled_pyramid_layout_is_bitwise_the_reference_layout), and cat / __getitem__ keep working on the
    tiled form (they only touch the edge dimension)."""
    torch.manual_seed(8)
    E, h, w = 3, 24, 32
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    GA = lgu.GaussianMask(h, w).cuda()
    f1 = torch.randn(1, E, 128, h, w, device="cuda") * 0.5
    f2 = torch.randn(1, E, 128, h, w, device="cuda") * 0.5
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, h, w, 2, device="cuda")
    outs = {}
    with torch.no_grad():
        for tiled in (True, False):
            monkeypatch.setattr(lgu.CorrBlock, "TILED_PYRAMID", tiled)
            blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
            assert blk._tiled == tiled
            outs[tiled] = [blk(coords)[0].clone(), blk(coords)[0].clone()]   # second call: persistent mask state
        assert float((outs[True][0] - outs[False][0]).abs().max()) <= 2e-5
        assert float((outs[True][1] - outs[False][1]).abs().max()) <= 2e-5
        monkeypatch.setattr(lgu.CorrBlock, "TILED_PYRAMID", True)
        a = lgu.CorrBlock(ofsMap, ofsRes, GA, f1[:, :2], f2[:, :2])
        b = lgu.CorrBlock(ofsMap, ofsRes, GA, f1[:, 2:], f2[:, 2:])
        whole = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        got = a.cat(b)(coords)[0]
        assert a._tiled and float((got - whole(coords)[0]).abs().max()) <= 2e-5
        sub = whole[torch.tensor([0, 2], device="cuda")]
        assert sub(coords[:, [0, 2]])[0].shape == (1, 2, 196, h, w)


@pytest.mark.parametrize("probe", [False, True])
@pytest.mark.parametrize("name", list(TILED_CASES))
def test_fused_encoder_layer_is_the_layer_on_the_half_lookup(lgu, name, probe):
    """lgu_defcorr_pyramid_enc_fwd_f32: lookup + first corr_encoder layer in one launch == relu(W1 . x + b1) evaluated
    in fp32 on x = the half lookup (LGU_PYR_OUT_F16, itself bit-exact against the planar tensor) with the half weights,
    rounded to half: half products are exact in fp32 and accumulation is fp32 on both sides, so the difference is
    summation order + one half rounding (tolerance 2^-10 relative + 1e-4 absolute).  Channel counts 196 / 147 / 98
    exercise the K padding (224 / 160 / 128); W1 = 40 and 24 the partial 32-pixel tile.  Same offset side effects."""
    seed, E, H1, W1, L, sigma, osc, dense = TILED_CASES[name]
    case = inputs.pyramid_case(seed, E, H1, W1, L, 3, sigma, osc, dense)
    hw = [tuple(v.shape[3:]) for v in case["volumes"]]
    tv = [lgu.ops.volume_retile(dev(v)) for v in case["volumes"]]
    coords = dev(case["coords"])
    offs_a = [dev(o) if o is not None else None for o in case["offsets"]]
    offs_b = [dev(o) if o is not None else None for o in case["offsets"]]
    K = L * 49
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    conv_w = torch.randn(128, K, 1, 1, device="cuda", generator=g) * 0.5
    conv_b = torch.randn(128, device="cuda", generator=g) * 0.1
    w, b = lgu.ops.pack_encoder_layer(conv_w, conv_b)
    assert tuple(w.shape) == (128, -(-K // 32) * 32) and float(w[:, K:].abs().max() if w.shape[1] > K else 0.0) == 0.0
    x = lgu.ops.defcorr_pyramid_forward(tv, coords, offs_a, 3, probe=probe, tiled=True, level_hw=hw, out_format="nhwc_f16")
    plan = lgu.ops.DefcorrPyramidPlan(tv, offs_b, 3, probe=probe, tiled=True, level_hw=hw, encoder=(w, b))
    got = plan(coords)
    assert got.dtype == torch.float16 and tuple(got.shape) == (E, 128, H1, W1) and got.permute(0, 2, 3, 1).is_contiguous()
    want = torch.relu(x.permute(0, 2, 3, 1).float() @ w[:, :K].float().t() + b.float()).permute(0, 3, 1, 2)
    err = (got.float() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -10 + 1e-4).all()), float(err.max())
    assert float(want.max()) > 0.05   # the comparison is not vacuous
    for a_, b_ in zip(offs_a, offs_b):
        assert a_ is None or torch.equal(a_, b_)


def test_fused_encoder_argument_checks(lgu):
    v = [torch.randn(1, 8, 16, 8, 16, device="cuda"), torch.randn(1, 8, 16, 4, 8, device="cuda")]
    tv = [lgu.ops.volume_retile(x) for x in v]
    c = torch.zeros(1, 2, 8, 16, device="cuda")
    w, b = lgu.ops.pack_encoder_layer(torch.randn(128, 98, 1, 1, device="cuda"), torch.randn(128, device="cuda"))
    with pytest.raises(lgu._lib.UnsupportedShape):   # row-major pyramid
        lgu.ops.DefcorrPyramidPlan(v, [None, None], 3, encoder=(w, b))(c)
    with pytest.raises(RuntimeError):                # weight rows packed for another channel count
        lgu.ops.DefcorrPyramidPlan(tv, [None, None], 3, tiled=True, encoder=(w[:, :96].contiguous(), b))
    with pytest.raises(RuntimeError):
        lgu.ops.pack_encoder_layer(torch.randn(64, 98, 1, 1, device="cuda"), torch.randn(64, device="cuda"))
    got = lgu.ops.DefcorrPyramidPlan(tv, [None, None], 3, tiled=True, encoder=(w, b))(c)
    assert tuple(got.shape) == (1, 128, 8, 16)


def test_corrblock_out_format_and_corr_encoder(lgu, monkeypatch):
    """SURVEY f4 hand-over: CorrBlock.OUT_FORMAT = "nhwc_f16" returns the reference's logical (1,E,196,h,w) tensor
    as channel-last half = .half() of the planar result (to the 2e-5 two constructions differ by, see above), same
    persistent mask state; CorrEncoder over it == the reference's corr_encoder Sequential under autocast on the
    planar tensor (droid_net.py:76-80,116), to half rounding, and == the fp32 evaluation of the same op to 2e-3."""
    torch.manual_seed(9)
    E, h, w = 3, 24, 32
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    GA = lgu.GaussianMask(h, w).cuda()
    enc = torch.nn.Sequential(torch.nn.Conv2d(196, 128, 1), torch.nn.ReLU(inplace=True),
                              torch.nn.Conv2d(128, 128, 3, padding=1), torch.nn.ReLU(inplace=True)).cuda().eval()
    f1 = torch.randn(1, E, 128, h, w, device="cuda") * 0.5
    f2 = torch.randn(1, E, 128, h, w, device="cuda") * 0.5
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, h, w, 2, device="cuda")
    with torch.no_grad():
        outs = {}
        for fmt in ("planar", "nhwc", "nhwc_f16"):
            monkeypatch.setattr(lgu.CorrBlock, "OUT_FORMAT", fmt)
            blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
            outs[fmt] = [blk(coords)[0], blk(coords)[0]]
            for o in outs[fmt]:
                assert tuple(o.shape) == (1, E, 196, h, w)
        for k in range(2):
            assert outs["nhwc"][k].dtype == torch.float32
            assert float((outs["nhwc"][k] - outs["planar"][k]).abs().max()) <= 2e-5
            assert outs["nhwc_f16"][k].dtype == torch.float16
            tol = 2e-5 + outs["planar"][k].abs() * 2.0 ** -11
            assert bool(((outs["nhwc_f16"][k].float() - outs["planar"][k]).abs() <= tol + 6e-8).all())
        planar, cl16 = outs["planar"][0].view(E, 196, h, w), outs["nhwc_f16"][0].view(E, 196, h, w)
        fused = lgu.CorrEncoder(enc)
        assert fused.takes(cl16) and not fused.takes(planar)
        got = fused(cl16)
        assert got.dtype == torch.float16 and tuple(got.shape) == (E, 128, h, w)
        with torch.autocast("cuda", dtype=torch.float16):
            want = enc(planar)
        exact = enc(cl16.float())                      # the same op in fp32 on the half inputs
        scale = float(exact.abs().max())
        assert float((got.float() - want.float()).abs().max()) <= 2.0 ** -9 * scale
        assert float((got.float() - exact).abs().max()) <= 2e-3 * scale
        assert torch.equal(fused(planar), enc(planar))  # anything else goes to the wrapped module unchanged
        # first layer inside the lookup launch: CorrBlock returns the 128-channel half tensor, CorrEncoder finishes it
        monkeypatch.setattr(lgu.CorrBlock, "ENCODER", fused)
        blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        mid = blk(coords)[0]
        assert tuple(mid.shape) == (1, E, 128, h, w) and mid.dtype == torch.float16
        got2 = fused(mid.view(E, 128, h, w))
        assert tuple(got2.shape) == (E, 128, h, w)
        assert float((got2.float() - want.float()).abs().max()) <= 2.0 ** -9 * scale
        assert float((got2.float() - exact).abs().max()) <= 2e-3 * scale


@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("cfg", [(3, 24, 32, 128, 3, 4), (9, 12, 20, 64, 3, 3), (2, 10, 13, 128, 2, 2), (1, 24, 32, 32, 1, 2)])
def test_lowmem_pyramid_fused_launch_equals_per_level_operators(lgu, oracle, cfg, half):
    """lgu_lowmem_pyramid_fwd_h16 / _f32 (all levels of AltCorrBlock.corr_fn in one launch) == the per-level
    operator called L times with coords / 2^l, BIT FOR BIT, written at the right channels of the concatenated
    tensor; None offsets == zero offset tensors; same in-place centre zeroing; level 0 also against the oracle
    on the float copies."""
    B, H, W, C, radius, L = cfg
    cast = (lambda t: t.half()) if half else (lambda t: t)
    per_level = lgu.ops.lowMem_defSample_mixed if half else lgu.ops.lowMem_defSample
    rng = np.random.default_rng(500 + B + C)
    rd = 2 * radius + 1
    f1 = cast((torch.from_numpy(rng.standard_normal((B, H, W, C)).astype(np.float32)) * 0.125).cuda())
    f2s = [cast((torch.from_numpy(rng.standard_normal((B, max(H >> l, 1), max(W >> l, 1), C)).astype(np.float32)) * 0.125).cuda())
           for l in range(L)]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    c_np = (np.stack([xs, ys], -1)[None, None].repeat(B, 0) + rng.standard_normal((B, 1, H, W, 2)) * 3).astype(np.float32)
    coords = dev(c_np)
    off_np = [(4 * np.tanh(rng.standard_normal((B, H, W, rd, rd, 2)))).astype(np.float32) if l < 2 else None for l in range(L)]
    offs_a = [dev(o) if o is not None else None for o in off_np]
    offs_b = [dev(o) if o is not None else torch.zeros(B, H, W, rd, rd, 2, device="cuda") for o in off_np]
    got = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, offs_a, radius)
    assert got.shape == (B, 1, L * rd * rd, H, W)
    for l in range(L):
        want, = per_level(f1, f2s[l], (coords / 2 ** l).contiguous(), offs_b[l], radius)
        assert torch.equal(got[:, :, l * rd * rd:(l + 1) * rd * rd], want.view(B, 1, rd * rd, H, W)), l
        if offs_a[l] is not None:
            assert torch.equal(offs_a[l], offs_b[l])
    ref0, = oracle.lowMem_defSample(host(f1.float()), host(f2s[0].float()), c_np, off_np[0].copy(), radius)
    assert np.abs(host(got[:, :, :rd * rd]).reshape(ref0.shape) - ref0).max() <= 1e-5
    # a plan is reusable and an unsupported channel count is reported, not mis-served
    plan = lgu.ops.LowmemPyramidPlan(f1, f2s, offs_a, radius)
    assert torch.equal(plan(coords), got)
    with pytest.raises(lgu._lib.UnsupportedShape):   # 48 channels: not a matrix-core K multiple this kernel instantiates
        lgu.ops.lowmem_pyramid_forward_mixed(cast(torch.zeros(B, H, W, 48, device="cuda")),
                                             [cast(torch.zeros(B, f.shape[1], f.shape[2], 48, device="cuda")) for f in f2s],
                                             coords, offs_a, radius)


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("probe", [False, True])
def test_interleaved_coords_equal_planar_coords(lgu, tiled, probe):
    """LGU_PYR_COORDS_LAST: coords as (E,H1,W1,2) (how the factor graph holds them) == the operator's (E,2,H1,W1)
    planes, bit for bit; kernels that do not read the interleaved form say so."""
    case = inputs.pyramid_case(61, 2, 48, 64, 4, 3, 3.0, 4.0, False)
    vols = [dev(v) for v in case["volumes"]]
    hw = [tuple(v.shape[3:]) for v in vols]
    if tiled:
        vols = [lgu.ops.volume_retile(v) for v in vols]
    coords = dev(case["coords"])
    coords_xy = coords.permute(0, 2, 3, 1).contiguous()
    oa = [dev(o) if o is not None else None for o in case["offsets"]]
    ob = [dev(o) if o is not None else None for o in case["offsets"]]
    a = lgu.ops.defcorr_pyramid_forward(vols, coords, oa, 3, probe=probe, tiled=tiled, level_hw=hw if tiled else None)
    b = lgu.ops.defcorr_pyramid_forward(vols, coords_xy, ob, 3, probe=probe, tiled=tiled, level_hw=hw if tiled else None, coords_last=True)
    assert torch.equal(a, b) and all(x is None or torch.equal(x, y) for x, y in zip(oa, ob))
    plan = lgu.ops.DefcorrPyramidPlan(vols, ob, 3, probe=False, tiled=tiled, level_hw=hw if tiled else None, coords_last=True)
    with pytest.raises(RuntimeError, match="coords must be"):
        plan(coords)
    if not tiled:
        set_variant(2)   # the generic kernel reads planes only
        with pytest.raises(lgu._lib.UnsupportedShape):
            lgu.ops.defcorr_pyramid_forward(vols, coords_xy, ob, 3, coords_last=True)


@pytest.mark.parametrize("half", [True, False])
def test_lowmem_pyramid_reads_frame_buffers_in_place(lgu, half):
    """ii / jj form of the fused low-memory launch: the frame buffers are indexed inside the kernel (what
    `self.pyramid[i][:, jj]` gathers in reference corr.py:193-194) == the launch over gathered per-edge copies,
    BIT FOR BIT; lbase shifts the coordinate scale (the level-1 probe is L = 1, lbase = 1, radius 1, no offsets)."""
    torch.manual_seed(21)
    F_, H, W, C, L = 5, 24, 32, 128, 3
    cast = (lambda t: t.half()) if half else (lambda t: t)
    frames = [cast(torch.randn(F_, H >> l, W >> l, C, device="cuda") * 0.125) for l in range(L)]
    ii = torch.tensor([0, 0, 1, 4, 3, 2, 2, 4, 1], device="cuda")
    jj = torch.tensor([1, 2, 3, 0, 2, 4, 0, 4, 0], device="cuda")
    E = ii.numel()
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(E, 1, H, W, 2, device="cuda")).contiguous()
    o0 = (4 * torch.tanh(torch.randn(E, H, W, 7, 7, 2, device="cuda"))).contiguous()
    oa, ob = [o0.clone(), None, None], [o0.clone(), None, None]
    got = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], frames, coords, oa, 3, ii=ii, jj=jj)
    want = lgu.ops.lowmem_pyramid_forward_mixed(frames[0][ii].contiguous(), [f[jj].contiguous() for f in frames], coords, ob, 3)
    assert torch.equal(got, want) and torch.equal(oa[0], ob[0])
    probe = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], [frames[1]], coords, [None], 1, ii=ii, jj=jj, lbase=1)
    alt = (lgu.ops.altcorr_forward_mixed if half else lgu.ops.altcorr_forward)(frames[0][ii].contiguous(), frames[1][jj].contiguous(),
                                                                              (coords / 2).contiguous(), 1)[0]
    assert torch.equal(probe, alt)
    with pytest.raises(RuntimeError, match="both ii and jj"):
        lgu.ops.LowmemPyramidPlan(frames[0], frames, oa, 3, ii=ii)


@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("big_offsets", [False, True])
def test_lowmem_chunk_planar_feature_maps_are_bitwise_the_channel_last_ones(lgu, half, big_offsets):
    """LowmemPyramidPlan(chunked=True): fmap2 levels stored (F, C/k, H, W, k) (ops.lowmem_chunked) give the channel-last
    launch's output BIT FOR BIT — same products, same summation order, other addresses — with frame indices and with
    per-edge maps, ragged level sizes, and offsets beyond +-4 (boxes larger than a patch take the per-tap fallback,
    which reads the chunk-planar form too)."""
    torch.manual_seed(33)
    F_, H, W, C, L = 4, 22, 30, 128 if half else 64, 3
    cast = (lambda t: t.half()) if half else (lambda t: t)
    frames = [cast(torch.randn(F_, max(H >> l, 1), max(W >> l, 1), C, device="cuda") * 0.125) for l in range(L)]
    chunked = [lgu.ops.lowmem_chunked(f) for f in frames]
    k = 8 if half else 4
    assert tuple(chunked[1].shape) == (F_, C // k, H >> 1, W >> 1, k)
    assert torch.equal(chunked[1][2, 3, 5, 7], frames[1][2, 5, 7, 3 * k:4 * k])
    ii = torch.tensor([0, 1, 3, 2, 2, 0], device="cuda")
    jj = torch.tensor([1, 2, 0, 3, 0, 3], device="cuda")
    E = ii.numel()
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(E, 1, H, W, 2, device="cuda")).contiguous()
    o0 = ((9.0 if big_offsets else 4.0) * torch.tanh(torch.randn(E, H, W, 7, 7, 2, device="cuda"))).contiguous()
    oa, ob = [o0.clone(), None, None], [o0.clone(), None, None]
    want = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], frames, coords, oa, 3, ii=ii, jj=jj)
    got = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], chunked, coords, ob, 3, ii=ii, jj=jj, chunked=True)
    assert torch.equal(got, want) and torch.equal(oa[0], ob[0])
    # per-edge maps (no frame indices), radius 1 probe form
    pe = lgu.ops.lowmem_pyramid_forward_mixed(frames[0][ii].contiguous(), [lgu.ops.lowmem_chunked(frames[1][jj].contiguous())], coords,
                                              [None], 1, lbase=1, chunked=True)
    assert torch.equal(pe, lgu.ops.lowmem_pyramid_forward_mixed(frames[0], [frames[1]], coords, [None], 1, ii=ii, jj=jj, lbase=1))
    with pytest.raises(RuntimeError, match="chunked fmap2"):
        lgu.ops.LowmemPyramidPlan(frames[0], frames, [None] * L, 3, ii=ii, jj=jj, chunked=True)


def test_corrblock_slot_store_cat_and_getitem_move_no_volume(lgu):
    """The inference CorrBlock keeps its pyramid in slot-indirected buffers: `cat` copies only the new edges,
    `__getitem__` (bool mask, as factor_graph.rm_factors uses, or an index tensor) only edits the slot list, freed
    slots are reused; the edge-ordered view of the pyramid and the lookup equal those of a block built from exactly
    those edges.  (Lookups are compared on fresh blocks only: a lookup leaves persistent state in offset[1].)"""
    torch.manual_seed(9)
    E, h, w = 6, 24, 32
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    GA = lgu.GaussianMask(h, w).cuda()
    f1 = torch.randn(1, E, 128, h, w, device="cuda") * 0.5
    f2 = torch.randn(1, E, 128, h, w, device="cuda") * 0.5
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
    coords = torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, E, h, w, 2, device="cuda")

    def block(sel):
        sel = torch.as_tensor(sel, device="cuda")
        return lgu.CorrBlock(ofsMap, ofsRes, GA, f1[:, sel], f2[:, sel])

    def same_pyramid(blk, sel):   # two constructions differ by conv / matmul rounding only
        return all(float((a - b).abs().max()) <= 2e-5 for a, b in zip(blk.corr_pyramid, block(sel).corr_pyramid))

    with torch.no_grad():
        blk = block([0, 1, 2])
        assert blk._store is not None and blk._tiled
        blk.cat(block([3, 4]))
        assert len(blk._slot_list) == 5 and blk._store[0].shape[0] >= 5          # grown once (doubling)
        grown_ptr = blk._store[0].data_ptr()
        assert same_pyramid(blk, [0, 1, 2, 3, 4])
        blk[torch.tensor([False, True, True, False, True], device="cuda")]       # rm_factors-style boolean mask
        assert blk._slot_list == [1, 2, 4] and sorted(blk._free)[:2] == [0, 3] and blk._store[0].data_ptr() == grown_ptr
        assert same_pyramid(blk, [1, 2, 4]) and blk.offset[0].shape[0] == 3
        blk.cat(block([5]))                                                      # reuses a freed slot, no growth
        assert blk._store[0].data_ptr() == grown_ptr and len(blk._slot_list) == 4 and blk._slot_list[-1] in (0, 3, 5)
        assert same_pyramid(blk, [1, 2, 4, 5])
        blk[torch.tensor([3, 0], device="cuda")]                                  # reorder with an index tensor
        assert same_pyramid(blk, [5, 1])
        got = blk(coords[:, [5, 1]])[0]                                           # first lookup of this block
        assert float((got - block([5, 1])(coords[:, [5, 1]])[0]).abs().max()) <= 2e-5
        # corr_pyramid stays readable in edge order (gathered copy), and assigning it leaves the slot form
        lv = [v.clone() for v in blk.corr_pyramid]
        blk.corr_pyramid = lv
        assert blk._store is None and blk.corr_pyramid[0].shape[0] == 2


def test_lowmem_pyramid_backend_scale_indexed(lgu):
    """BASELINE config 5 per-GPU scale: 250 edges over 40 frames of 60x80x128 half feature maps, all levels in one
    launch reading the frame buffers in place.  The last edges (largest work-item ids / output addresses) equal a
    small launch over just those edges; everything is finite."""
    torch.manual_seed(17)
    F_, H, W, C, L, E = 40, 60, 80, 128, 4, 250
    frames = [(torch.randn(F_, H >> l, W >> l, C, device="cuda") * 0.125).half() for l in range(L)]
    ii = torch.randint(0, F_, (E,), device="cuda")
    jj = torch.randint(0, F_, (E,), device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(E, 1, H, W, 2, device="cuda")).contiguous()
    o0 = (4 * torch.tanh(torch.randn(1, H, W, 7, 7, 2, device="cuda"))).contiguous()   # offset[b*n] -> edge 0's offsets
    out = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], frames, coords, [o0, None, None, None], 3, ii=ii, jj=jj)
    assert out.shape == (E, 1, 196, H, W) and torch.isfinite(out).all()
    tail = slice(E - 3, E)
    small = lgu.ops.lowmem_pyramid_forward_mixed(frames[0], frames, coords[tail].contiguous(), [o0.clone(), None, None, None], 3,
                                                 ii=ii[tail].contiguous(), jj=jj[tail].contiguous())
    assert torch.equal(out[tail], small)


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomized_differential_pyramid(lgu, seed):
    """Seeded random shapes (ragged sizes, 1-4 levels, border-heavy coords, huge offsets): the production kernels
    over both pyramid layouts against the independent one-thread-per-output kernel (variant 2)."""
    rng = np.random.default_rng(1000 + seed)
    E = int(rng.integers(1, 4))
    H1 = int(rng.integers(5, 40))
    W1 = int(rng.integers(5, 70))
    L = int(rng.integers(1, 5))
    while (H1 >> (L - 1)) < 1 or (W1 >> (L - 1)) < 1:
        L -= 1
    sigma = float(rng.choice([1.0, 3.0, 12.0, 40.0]))
    osc = float(rng.choice([2.0, 4.0, 9.0]))
    dense = bool(rng.integers(0, 2))
    case = inputs.pyramid_case(2000 + seed, E, H1, W1, L, 3, sigma, osc, dense)
    vols = [dev(v) for v in case["volumes"]]
    hw = [tuple(v.shape[3:]) for v in vols]
    coords = dev(case["coords"])

    def fresh():
        return [dev(o) if o is not None else None for o in case["offsets"]]

    set_variant(2)
    o_ref = fresh()
    want = lgu.ops.defcorr_pyramid_forward(vols, coords, o_ref, 3)
    set_variant(0)
    o_t = fresh()
    got_t = lgu.ops.defcorr_pyramid_forward([lgu.ops.volume_retile(v) for v in vols], coords, o_t, 3, tiled=True, level_hw=hw)
    assert torch.equal(got_t, want), (E, H1, W1, L, sigma, osc, dense)
    assert all(a is None or torch.equal(a, b) for a, b in zip(o_t, o_ref))
    o_r = fresh()
    got_r = lgu.ops.defcorr_pyramid_forward(vols, coords, o_r, 3)
    assert torch.equal(got_r, want)
    # the other output forms of the tiled kernel: channel-last fp32 / half, and the fused first encoder layer
    tv = [lgu.ops.volume_retile(v) for v in vols]
    assert torch.equal(lgu.ops.defcorr_pyramid_forward(tv, coords, fresh(), 3, tiled=True, level_hw=hw, out_format="nhwc"), want)
    x16 = lgu.ops.defcorr_pyramid_forward(tv, coords, fresh(), 3, tiled=True, level_hw=hw, out_format="nhwc_f16")
    assert torch.equal(x16, want.half())
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    K = L * 49
    w, b = lgu.ops.pack_encoder_layer(torch.randn(128, K, 1, 1, device="cuda", generator=g) * 0.5,
                                      torch.randn(128, device="cuda", generator=g) * 0.1)
    mixed = any(o is None for o in case["offsets"]) and not all(o is None for o in case["offsets"][:L])
    try:
        enc = lgu.ops.DefcorrPyramidPlan(tv, fresh(), 3, tiled=True, level_hw=hw, encoder=(w, b))(coords)
    except lgu._lib.UnsupportedShape:
        assert mixed   # null-offset patterns the sampler splits into several launches are not fused
        return
    ref = torch.relu(x16.permute(0, 2, 3, 1).float() @ w[:, :K].float().t() + b.float()).permute(0, 3, 1, 2)
    assert bool(((enc.float() - ref).abs() <= ref.abs() * 2.0 ** -10 + 1e-4).all())


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomized_differential_lowmem(lgu, seed):
    """Seeded random shapes for the low-memory path: matrix-core kernels (float and half maps, per level and fused)
    against the independent wave-per-pixel kernel (variant 1)."""
    rng = np.random.default_rng(3000 + seed)
    B = int(rng.integers(1, 11))
    H1 = int(rng.integers(3, 30))
    W1 = int(rng.integers(3, 40))
    C = int(rng.choice([32, 64, 128]))
    radius = int(rng.choice([1, 2, 3]))
    L = int(rng.integers(1, 4))
    sigma = float(rng.choice([1.0, 3.0, 10.0, 30.0]))
    osc = float(rng.choice([2.0, 4.0, 8.0]))
    rd = 2 * radius + 1
    f1 = dev((rng.standard_normal((B, H1, W1, C)) * 0.125).astype(np.float32)).half().float()   # exactly representable in half
    f2s = [dev((rng.standard_normal((B, max(H1 >> l, 1), max(W1 >> l, 1), C)) * 0.125).astype(np.float32)).half().float() for l in range(L)]
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    coords = dev((np.stack([xs, ys], -1)[None, None].repeat(B, 0) + rng.standard_normal((B, 1, H1, W1, 2)) * sigma).astype(np.float32))
    off = [dev((osc * np.tanh(rng.standard_normal((B, H1, W1, rd, rd, 2)))).astype(np.float32)) for _ in range(L)]
    os.environ["LGU_LOWMEM_VARIANT"] = "1"
    try:
        want = [lgu.ops.lowMem_defSample(f1, f2s[l], (coords / 2 ** l).contiguous(), off[l].clone(), radius)[0].view(B, 1, rd * rd, H1, W1)
                for l in range(L)]
    finally:
        os.environ.pop("LGU_LOWMEM_VARIANT")
    want = torch.cat(want, 2)
    got_f = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o.clone() for o in off], radius)
    got_h = lgu.ops.lowmem_pyramid_forward_mixed(f1.half(), [f.half() for f in f2s], coords, [o.clone() for o in off], radius)
    tag = (B, H1, W1, C, radius, L, sigma, osc)
    assert float((got_f - want).abs().max()) <= 1e-5, tag
    assert float((got_h - want).abs().max()) <= 1e-5, tag


@pytest.mark.parametrize("shape", [(16, 60, 80, 32), (3, 21, 27, 64), (9, 10, 13, 128)])
def test_lowmem_coop_kernel_scheduling_modes_and_the_one_wave_kernel(lgu, oracle, shape, monkeypatch):
    """csrc/lowmem_coop.hip (production for half maps with C <= 128): the same call served (a) with whole rounds of
    workgroups fused over all levels and the remainder split by level group (the default rule: BASELINE config 4's 2 400
    tiles over 768 slots), (b) all fused, (c) all split gives BIT-IDENTICAL results (same products, same order, other
    work units), equals the one-wave-per-block kernel of lowmem_mfma.hip to summation order, and the C oracle on the
    first and the last edge.  Shapes: config-4 size at C = 32, ragged sizes (W1 not a multiple of the 8-pixel tile, odd
    level widths: pairs of positions straddling the right edge), 9 edges (XCD dealing with a tail).  sigma = 6: a third
    of the taps fall outside the map (zero-filled border patches)."""
    B, H, W, C = shape
    torch.manual_seed(B * 100 + W)
    L, radius = 4, 3
    f1 = (torch.randn(B, H, W, C, device="cuda") * 0.125).half()
    f2s = [(torch.randn(B, max(H >> l, 1), max(W >> l, 1), C, device="cuda") * 0.125).half() for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 6 * torch.randn(B, 1, H, W, 2, device="cuda")).contiguous()
    o0 = (4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device="cuda"))).contiguous()
    o1 = ((4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device="cuda")) + o0) / 2).contiguous()

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o0.clone(), o1.clone(), None, None], radius)
        for k in env:
            monkeypatch.delenv(k)
        return out

    default = run()
    assert torch.equal(default, run(LGU_LOWMEM_COOP_SPLIT="0"))
    assert torch.equal(default, run(LGU_LOWMEM_COOP_SPLIT="1"))
    old = run(LGU_LOWMEM_COOP="0")
    assert float((default - old).abs().max()) <= 1e-5
    for b in (0, B - 1):
        for l, off in enumerate([o0, o1, None, None]):
            # the reference reads offset[b * n] (lowMem_defSample.cu:80-83): with S = 1 every edge samples with edge 0's offsets
            o_np = host(off[0:1]).copy() if off is not None else np.zeros((1, H, W, 7, 7, 2), np.float32)
            want, = oracle.lowMem_defSample(host(f1[b:b + 1].float()), host(f2s[l][b:b + 1].float()),
                                            host(coords[b:b + 1] / 2 ** l), o_np, radius)
            got = host(default[b:b + 1, :, l * 49:(l + 1) * 49]).reshape(want.shape)
            assert np.abs(got - want).max() <= 1e-5, (b, l)


def test_offset_head_cache_equals_the_per_edge_convolution(lgu):
    """ops.OffsetHeadCache (per-frame partial convolutions P_A[ii] + P_B[jj], found and convolved on the device, kept across
    calls) against ops.offset_conv_frames (one convolution per edge and call) and an fp64 evaluation of the reference
    expression ofsMap(cat(frames[ii] * 4, frames[jj] * 4).float()) (corr.py:174-189): same products, fp32 sums in another
    order.  Three calls over overlapping frame sets: the second and third meet frames whose partials exist (their flags
    are set, nothing is convolved twice) and frames that are new; the worklist counter is back at zero after every call.
    Also the two-part input of the residual head and a frame set with duplicates inside one call."""
    torch.manual_seed(5)
    NF, H, W, C, Cout = 9, 12, 20, 128, 98
    frames = (torch.randn(NF, H, W, C, device="cuda") * 0.125).half().contiguous()
    conv = torch.nn.Conv2d(2 * C, Cout, 3, padding=1).cuda()
    packed = lgu.ops.pack_offset_conv(conv.weight, conv.bias)
    cache = lgu.ops.OffsetHeadCache(frames, lgu.ops.pack_offset_conv_parts(conv.weight, conv.bias))
    calls = [([0, 0, 1, 2, 2], [1, 2, 0, 3, 1]), ([2, 3, 3, 4], [4, 2, 5, 3]), ([8, 0, 7], [0, 8, 7])]
    seen_a, seen_b = set(), set()
    for ii_l, jj_l in calls:
        ii, jj = torch.tensor(ii_l, device="cuda"), torch.tensor(jj_l, device="cuda")
        got = cache(ii, jj)
        want = lgu.ops.offset_conv_frames(frames, ii, jj, packed)
        x = torch.cat((frames[ii].double() * 4, frames[jj].double() * 4), dim=-1).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x, conv.weight.double(), conv.bias.double(), padding=1)
        scale = float(ref.abs().max())
        assert float((got - want).abs().max()) <= 2e-6 * scale
        assert float((got.double() - ref).abs().max()) <= 5e-6 * scale
        seen_a |= set(ii_l); seen_b |= set(jj_l)
        done = cache.done.cpu().numpy()
        assert set(np.nonzero(done[0])[0]) == seen_a and set(np.nonzero(done[1])[0]) == seen_b
        assert int(cache.count.item()) == 0
    # the residual head's form: input in two half parts (2 x 2 averages need up to 24 bits)
    xf = torch.randn(NF, H, W, C, device="cuda") * 0.125
    hi = xf.half()
    lo = (xf - hi.float()).half()
    cache2 = lgu.ops.OffsetHeadCache(hi.contiguous(), lgu.ops.pack_offset_conv_parts(conv.weight, conv.bias), frames_lo=lo.contiguous())
    ii, jj = torch.tensor([1, 1, 6], device="cuda"), torch.tensor([6, 1, 1], device="cuda")
    got = cache2(ii, jj)
    want = lgu.ops.offset_conv_frames(hi.contiguous(), ii, jj, packed, frames_lo=lo.contiguous())
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_altcorrblock_head_cache_on_and_off(lgu, monkeypatch):
    """AltCorrBlock with the per-frame head cache (corr.HEAD_CACHE = True, opt-in) and without (default): two chunks of a graph
    whose frames overlap give the same lookup (offsets equal to fp32 summation order; the lookup is continuous in them
    away from integer sample positions, so it is compared loosely) and the same offsets closely."""
    torch.manual_seed(12)
    N, C, H, W = 6, 128, 16, 24
    fmaps = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
    ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    with torch.no_grad():
        for m in (ofsMap, ofsRes):
            m.weight.mul_(0.3)
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    chunks = [(torch.tensor([0, 0, 1, 1], device="cuda"), torch.tensor([1, 2, 0, 2], device="cuda")),
              (torch.tensor([2, 2, 3], device="cuda"), torch.tensor([1, 3, 2], device="cuda"))]
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(lgu.corr, "HEAD_CACHE", flag)
        blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        res = []
        with torch.no_grad():
            for ii, jj in chunks:
                coords = (torch.stack([xs, ys], -1)[None, None] + 1.5 * torch.randn(1, ii.numel(), H, W, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))).contiguous()
                c = blk(coords, ii, jj)
                res.append((c.clone(), [o.clone() for o in blk.offset[:2]]))
        assert hasattr(blk, "_head0") == flag
        outs[flag] = res
    # the cache is bounded: beyond HEAD_CACHE_MAX_BYTES for the whole frame buffer the heads are convolved per edge
    monkeypatch.setattr(lgu.corr, "HEAD_CACHE", True)
    monkeypatch.setattr(lgu.corr, "HEAD_CACHE_MAX_BYTES", 1 << 20)
    blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
    with torch.no_grad():
        blk(coords, *chunks[1])
        _ = blk.offset
    assert not hasattr(blk, "_head0")
    for (ca, oa), (cb, ob) in zip(outs[True], outs[False]):
        for x, y in zip(oa, ob):
            assert float((x - y).abs().max()) <= 2e-5
        assert float((ca - cb).abs().mean()) <= 1e-4


def test_offsets_finalize_with_the_probe_mask_folded_in(lgu):
    """ops.offsets_finalize(probe=...) (lgu_offsets_finalize_masked): level 1 leaves the post-processing already scaled by
    the uncertainty mask — BIT FOR BIT what offsets_finalize followed by the in-place probe_mask_scale_ leaves (level 0
    untouched), also on a pixel count that is not a multiple of the 32-pixel tile."""
    torch.manual_seed(21)
    for E, C, H, W in ((3, 98, 12, 20), (2, 98, 7, 9)):
        o0 = torch.randn(E, C, H, W, device="cuda")
        o1 = torch.randn(E, C, (H + 1) // 2, (W + 1) // 2, device="cuda")
        probe = torch.randn(E, 1, 9, H, W, device="cuda")
        a0, a1 = lgu.ops.offsets_finalize(o0, o1)
        lgu.ops.probe_mask_scale_(probe, a1)
        b0, b1 = lgu.ops.offsets_finalize(o0, o1, probe=probe)
        assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_altcorrblock_lazy_first_edge_offsets(lgu, monkeypatch):
    """AltCorrBlock.LAZY_OFFSETS: with one sample per pixel the reference's sampler reads offset[b * n] with n = 0
    (lowMem_defSample.cu:80-83) — every edge of a call samples with the FIRST edge's offsets — so the fast path computes
    probe, heads and post-processing for that edge only.  The lookup is BIT-IDENTICAL to the block that computes every
    edge's offsets, and the `offset` attribute, materialised on access, holds the same tensors bit for bit (the first
    edge's centre taps zeroed by the sampler in both)."""
    torch.manual_seed(17)
    N, C, H, W = 7, 128, 20, 28
    fmaps = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
    ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    ii = torch.tensor([2, 2, 3, 4, 4, 5], device="cuda")
    jj = torch.tensor([3, 4, 2, 5, 6, 4], device="cuda")
    coords = (torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, ii.numel(), H, W, 2, device="cuda")).contiguous()
    res = {}
    with torch.no_grad():
        for lazy in (True, False):
            monkeypatch.setattr(lgu.AltCorrBlock, "LAZY_OFFSETS", lazy)
            blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
            out = blk(coords, ii, jj)
            assert (getattr(blk, "_lazy", None) is not None) == lazy
            offs = [o.clone() for o in blk.offset]       # materialises in lazy mode
            assert getattr(blk, "_lazy", None) is None
            res[lazy] = (out.clone(), offs)
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert a.shape == b.shape and torch.equal(a, b)
    cen = 3 * 7 + 3
    assert not res[True][1][0][0].view(H, W, 49, 2)[:, :, cen].any()          # edge 0: centre taps zeroed by the sampler
    assert res[True][1][0][1].view(H, W, 49, 2)[:, :, cen].any()              # other edges: never touched


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(20, 28), (18, 28)])
def test_altcorrblock_call_many_equals_the_calls_one_by_one(lgu, oracle, shape):
    """AltCorrBlock.call_many / ShardedAltCorr.lookup_all: the chunk loop of update_lowmem (factor_graph.py:272-279, one
    corr_fn call per chunk of source frames) in ONE lookup launch, every call contributing the offset row of its first edge
    (lgu_lowmem_pyramid_calls_fwd_h16).  Bit-identical to the calls issued one by one — and the first call additionally
    checked against the oracle's sampler over the first edge's offsets (1e-5) — and `offset` afterwards is the last
    call's.  Chunks of uneven sizes including a single-edge one; odd level sizes (9 x 14, 4 x 7, 2 x 3).  (Shapes the
    head cache does not serve run the torch composition call by call, whose convolutions are not bit-reproducible.)"""
    torch.manual_seed(23)
    H, W = shape
    N, C = 9, 128
    fmaps = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
    ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    ii = torch.tensor([0, 0, 1, 1, 1, 2, 3, 3, 4, 5, 5, 6, 7, 8, 8], device="cuda")
    jj = torch.tensor([1, 2, 0, 2, 3, 3, 4, 2, 5, 4, 6, 7, 8, 7, 6], device="cuda")
    counts = [5, 1, 3, 4, 2]
    coords = (torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, ii.numel(), H, W, 2, device="cuda")).contiguous()
    with torch.no_grad():
        blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        many = blk.call_many(coords, ii, jj, counts)
        assert getattr(blk, "one_launch_calls", 0) == 1             # the one-launch path ran, not the fallback loop
        off_many = [o.clone() for o in blk.offset]
        ref_blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        parts, s = [], 0
        for c in counts:
            parts.append(ref_blk(coords[:, s:s + c], ii[s:s + c], jj[s:s + c]))
            s += c
        off_loop = [o.clone() for o in ref_blk.offset]
    assert many.shape == (1, ii.numel(), 4 * 49, H, W)
    assert torch.equal(many, torch.cat(parts, dim=1))
    for a, b in zip(off_many, off_loop):
        assert a.shape == b.shape and torch.equal(a, b)
    # a second pass over the same partition reuses the tables and the head cache
    with torch.no_grad():
        assert torch.equal(blk.call_many(coords, ii, jj, counts), many)
        # six-dimensional coords (explicit sample axis) and the fallback loop (counts of one call)
        assert torch.equal(blk.call_many(coords.unsqueeze(-2), ii, jj, counts).squeeze(-1), many)
        assert torch.equal(blk.call_many(coords[:, :5], ii[:5], jj[:5], [5]), parts[0])
    with pytest.raises(RuntimeError):
        blk.call_many(coords, ii, jj, [5, 1, 3])
    # the third call (edges 6..8) against the oracle: every edge of a call samples with the offsets of the call's first edge
    lo, hi = 6, 9
    with torch.no_grad():
        b2 = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        b2(coords[:, lo:hi], ii[lo:hi], jj[lo:hi])
        rows = [host(x[:1].float()).reshape(1, H, W, 7, 7, 2).copy() for x in b2.offset]
    for lvl in range(4):
        f1 = host(b2.pyramid[0][0][ii[lo:hi]].float())
        f2 = host(b2.pyramid[lvl][0][jj[lo:hi]].float())
        want, = oracle.lowMem_defSample(f1, f2, host(coords[0, lo:hi].unsqueeze(1) / 2 ** lvl), rows[lvl], 3)
        got = host(many[0, lo:hi, lvl * 49:(lvl + 1) * 49]).reshape(want.shape)
        assert np.abs(got - want).max() <= 1e-5, lvl


@pytest.mark.gpu
def test_sharded_altcorr_lookup_all_equals_lookup(lgu):
    """ShardedAltCorr.lookup_all (one launch for all of a rank's chunks) against .lookup (chunk by chunk), world 1 and the
    two halves of a world-2 partition; run_chunks(corr_all=...) hands every chunk its slice."""
    torch.manual_seed(29)
    N, C, H, W = 36, 64, 16, 24
    fmaps = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
    ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    pairs = [(i, j) for i in range(N) for j in range(N) if i != j and abs(i - j) <= 2]
    ii = torch.tensor([p[0] for p in pairs], device="cuda")
    jj = torch.tensor([p[1] for p in pairs], device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, ii.numel(), H, W, 2, device="cuda")).contiguous()
    with torch.no_grad():
        for rank, world in ((0, 1), (0, 2), (1, 2)):
            sac = lgu.sharded.ShardedAltCorr(ofsMap, ofsRes, None, fmaps, ii, jj, rank=rank, world=world)
            idx, corr, counts = sac.lookup_all(coords)
            assert sac.block.one_launch_calls == 1
            loop = list(sac.lookup(coords))
            assert counts == [int(i.numel()) for i, _ in loop] and len(counts) >= 2
            assert torch.equal(idx, torch.cat([i for i, _ in loop]))
            assert torch.equal(corr, torch.cat([c for _, c in loop], dim=1))
            seen = []
            lgu.sharded.run_chunks(sac.edges, ii, lambda i_, s_, c_: seen.append(c_) or (c_, c_, c_), corr_all=corr)
            assert all(torch.equal(a, b) for a, (_, b) in zip(seen, loop))


@pytest.mark.gpu
def test_config5_shard_in_one_launch_equals_the_chunk_loop(lgu, oracle):
    """BASELINE config 5's per-GPU shard at full size (rank 0 of 8 over a 200-keyframe graph of 60x80x128 half maps, edges
    at most 5 frames apart: ~250 edges in 4 source-frame chunks): ShardedAltCorr.lookup_all — ONE lookup launch with
    per-edge offset rows — against the reference's chunk loop, bit for bit, and the first edge of the LAST chunk against
    the C oracle's lowMem_defSample over that chunk's first-edge offsets."""
    torch.manual_seed(31)
    N, C, H, W, span = 200, 128, 60, 80, 5
    fmaps = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
    ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
    pairs = [(i, j) for i in range(N) for j in range(N) if i != j and abs(i - j) <= span]
    ii = torch.tensor([p[0] for p in pairs], device="cuda")
    jj = torch.tensor([p[1] for p in pairs], device="cuda")
    sac = lgu.sharded.ShardedAltCorr(ofsMap, ofsRes, None, fmaps, ii, jj, rank=0, world=8)
    own = sac.edges.my_edges
    assert 200 <= own.numel() <= 300 and len(sac.edges.my_chunks) >= 3
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = torch.zeros(1, ii.numel(), H, W, 2, device="cuda")
    coords[:, own] = torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, own.numel(), H, W, 2, device="cuda")
    with torch.no_grad():
        idx, corr, counts = sac.lookup_all(coords)
        assert getattr(sac.block, "one_launch_calls", 0) == 1
        pos = 0
        for cidx, c in sac.lookup(coords):
            assert torch.equal(corr[:, pos:pos + cidx.numel()], c)
            pos += cidx.numel()
        assert pos == own.numel() and torch.isfinite(corr).all()
        last = sac.edges.my_chunks[-1]
        rows = [host(x[:1].float()).reshape(1, H, W, 7, 7, 2).copy() for x in sac.block.offset]   # the last call's, materialised
    e = int(last[0])
    s = own.numel() - last.numel()
    for lvl in range(4):
        f1 = host(sac.block.pyramid[0][0][ii[e]].float())[None]
        f2 = host(sac.block.pyramid[lvl][0][jj[e]].float())[None]
        want, = oracle.lowMem_defSample(f1, f2, host(coords[0, e][None, None] / 2 ** lvl), rows[lvl], 3)
        got = host(corr[0, s, lvl * 49:(lvl + 1) * 49]).reshape(want.shape)
        assert np.abs(got - want).max() <= 1e-5, lvl


@pytest.mark.gpu
def test_lowmem_pyramid_offset_rows_per_edge(lgu, oracle):
    """lgu_lowmem_pyramid_calls_fwd_h16 at the operator level: every edge samples with the offset row `off_row` names — an
    arbitrary (non-monotonic) edge -> row table — and equals, bit for bit, the ordinary launch over that edge alone with
    that row as its offsets; one edge against the C oracle; rows past the tensors are clamped on the device (no wild
    read); argument checks."""
    torch.manual_seed(41)
    B, H, W, C, L, K = 11, 16, 24, 64, 4, 3
    f1 = (torch.randn(B, H, W, C, device="cuda") * 0.125).half()
    f2s = [(torch.randn(B, H >> l, W >> l, C, device="cuda") * 0.125).half() for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device="cuda")).contiguous()
    o0 = (4 * torch.tanh(torch.randn(K, H, W, 7, 7, 2, device="cuda"))).contiguous()
    o1 = ((4 * torch.tanh(torch.randn(K, H, W, 7, 7, 2, device="cuda")) + o0) / 2).contiguous()
    rows = torch.tensor([2, 0, 1, 1, 2, 0, 0, 2, 1, 0, 2], dtype=torch.int32, device="cuda")
    out = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o0.clone(), o1.clone(), None, None], 3, off_row=rows)
    assert out.shape == (B, 1, L * 49, H, W)
    for b in range(B):
        r = int(rows[b])
        one = lgu.ops.lowmem_pyramid_forward_mixed(f1[b:b + 1].contiguous(), [f[b:b + 1].contiguous() for f in f2s],
                                                   coords[b:b + 1].contiguous(),
                                                   [o0[r:r + 1].clone(), o1[r:r + 1].clone(), None, None], 3)
        assert torch.equal(out[b:b + 1], one), b
    b, r = 4, 2
    for l, off in enumerate((o0, o1, None, None)):
        o_np = host(off[r:r + 1]).copy() if off is not None else np.zeros((1, H, W, 7, 7, 2), np.float32)
        want, = oracle.lowMem_defSample(host(f1[b:b + 1].float()), host(f2s[l][b:b + 1].float()), host(coords[b:b + 1] / 2 ** l),
                                        o_np, 3)
        assert np.abs(host(out[b:b + 1, :, l * 49:(l + 1) * 49]).reshape(want.shape) - want).max() <= 1e-5, l
    # out-of-range rows: clamped to [0, K) on the device
    wild = rows.clone(); wild[3] = 1000; wild[5] = -7
    got = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o0.clone(), o1.clone(), None, None], 3, off_row=wild)
    ref = rows.clone(); ref[3] = K - 1; ref[5] = 0
    assert torch.equal(got, lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o0.clone(), o1.clone(), None, None], 3, off_row=ref))
    with pytest.raises(RuntimeError):   # one int32 entry per edge
        lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o0, o1, None, None], 3, off_row=rows[:5].contiguous())
    with pytest.raises(RuntimeError):
        lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o0, o1, None, None], 3, off_row=rows.long())
    with pytest.raises(RuntimeError):   # half feature maps only
        lgu.ops.lowmem_pyramid_forward_mixed(f1.float(), [f.float() for f in f2s], coords, [o0, o1, None, None], 3, off_row=rows)
    with pytest.raises(RuntimeError):   # one sample per pixel
        lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords.repeat(1, 2, 1, 1, 1).contiguous(), [o0, o1, None, None], 3, off_row=rows)


# ---- guard bands: no entry point writes outside the tensors it was given ------------------------------------------------
# An out-of-range device write that lands inside the caching allocator's pool raises nothing and corrupts a neighbour.
# Here every written tensor (outputs AND the in/out offsets) is a view into the middle of a larger buffer filled with a
# sentinel; after the call the bands in front of and behind it must still hold the sentinel bit for bit.  Shapes include
# the radii whose tap count is below the write-out's item width (9 / 25 taps: round 3 shipped, for an hour, a write-out
# that wrote 32 tap rows whatever the radius), ragged sizes and several samples per pixel.
_SENT = 1234.5


def _banded(shape, dtype=None, guard=4096):
    dtype = dtype or torch.float32
    n = int(np.prod(shape))
    big = torch.full((n + 2 * guard,), _SENT, dtype=dtype, device="cuda")
    return big, big[guard:guard + n].view(shape), guard


def _bands_intact(big, guard):
    return bool((big[:guard] == _SENT).all()) and bool((big[-guard:] == _SENT).all())


@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("cfg", [(2, 1, 12, 16, 64, 1, 2), (3, 1, 10, 13, 128, 2, 3), (2, 1, 24, 32, 128, 3, 4), (1, 2, 8, 16, 32, 1, 1),
                                 (9, 1, 9, 11, 64, 3, 2)])
def test_lowmem_entry_points_write_nothing_outside_their_tensors(lgu, cfg, half):
    B, S, H, W, C, radius, L = cfg
    rng = np.random.default_rng(77 + B + C + radius)
    rd = 2 * radius + 1
    cast = (lambda t: t.half()) if half else (lambda t: t)
    f1 = cast(dev((rng.standard_normal((B, H, W, C)) * 0.125).astype(np.float32)))
    f2s = [cast(dev((rng.standard_normal((B, max(H >> l, 1), max(W >> l, 1), C)) * 0.125).astype(np.float32))) for l in range(L)]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    coords = dev((np.stack([xs, ys], -1)[None, None].repeat(B, 0).repeat(S, 1) + rng.standard_normal((B, S, H, W, 2)) * 4).astype(np.float32))
    offs, bigs = [], []
    for l in range(L):
        if l < 2:
            big, view, g = _banded((B, H, W, rd, rd, 2))
            view.copy_(dev((4 * np.tanh(rng.standard_normal((B, H, W, rd, rd, 2)))).astype(np.float32)))
            offs.append(view); bigs.append((big, g))
        else:
            offs.append(None)
    obig, out, og = _banded((B, S, L * rd * rd, H, W))
    bigs.append((obig, og))
    if S == 1:
        got = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, offs, radius, out=out)
        assert got.data_ptr() == out.data_ptr()
        want = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [o.clone() if o is not None else None for o in offs], radius)
        assert torch.equal(got, want) and not bool((got == _SENT).any())        # fully written, same as into a fresh tensor
    else:   # several samples per pixel: the per-level operator (its output is allocated inside; the offsets are the banded tensor)
        per_level = lgu.ops.lowMem_defSample_mixed if half else lgu.ops.lowMem_defSample
        for l in range(L):
            c, = per_level(f1, f2s[l], (coords / 2 ** l).contiguous(), offs[l], radius)
            assert tuple(c.shape) == (B, S, rd, rd, H, W) and bool(torch.isfinite(c).all())
    torch.cuda.synchronize()
    for big, g in bigs:
        assert _bands_intact(big, g), "a kernel wrote outside the tensor it was given"


@pytest.mark.parametrize("shape", [(2, 32, 16, 32), (1, 32, 8, 16), (1, 128, 48, 64)], ids=["16x32", "8x16", "48x64"])
def test_volume_build_entry_points_write_nothing_outside_their_tensors(lgu, shape):
    """lgu_volume_build_pyramid_f32 / _h16 called through the C ABI with every written buffer (the four tiled levels, the
    half form's workspace) embedded in sentinel-filled memory: results equal the operator's into fresh tensors, every level
    is fully written (padding included) and the bands are untouched."""
    ops, lib = lgu.ops, lgu._lib.load()
    E, C, H, W = shape
    rng = np.random.default_rng(31 + C + W)
    f1 = dev((rng.standard_normal((E, C, H, W)) * 0.5).astype(np.float32))
    f2 = dev((rng.standard_normal((E, C, H, W)) * 0.5).astype(np.float32))
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    means = dev((np.stack([xs, ys], -1)[None].repeat(E, 0) + rng.standard_normal((E, H, W, 2))).astype(np.float32))
    covs = dev(rng.uniform(0.05, 5.05, (E, H, W, 2)).astype(np.float32))
    th = torch.cat((f1.half(), f2.half()), 1).permute(0, 2, 3, 1).contiguous()
    for half in (False, True):
        want = ops.volume_build_pyramid(th, None, means, covs) if half else ops.volume_build_pyramid(f1, f2, means, covs)
        bands = [_banded(ops.tiled_shape(E, H, W, H >> l, W >> l)) for l in range(4)]
        lp = (ops._vp * 4)(*[b[1].data_ptr() for b in bands])
        if half:
            wbig, work, wg = _banded(tuple(th.shape), torch.float16)
            rc = lib.lgu_volume_build_pyramid_h16(ops._ptr(th), ops._ptr(work), ops._ptr(means), ops._ptr(covs), None, 0, lp, 4,
                                                  E, C, H, W, 4, ops._stream(th))
        else:
            rc = lib.lgu_volume_build_pyramid_f32(ops._ptr(f1), ops._ptr(f2), ops._ptr(means), ops._ptr(covs), None, 0, lp, 4,
                                                  E, C, H, W, 4, ops._stream(f1))
        assert rc == 0
        torch.cuda.synchronize()
        for l, (big, view, g) in enumerate(bands):
            assert torch.equal(view, want[l]) and not bool((view == _SENT).any()), (half, l)
            assert _bands_intact(big, g), (half, l)
        if half:
            assert _bands_intact(wbig, wg) and not bool((work == _SENT).any())


@pytest.mark.parametrize("cfg", [(1, 2, 13, 21), (1, 2, 60, 80), (16, 8, 60, 80), (3, 3, 30, 40)])
def test_offset_convolution_writes_nothing_outside_its_tensor(lgu, cfg):
    """lgu_offset_conv_frames_h16 (small-grid and full-grid kernels, one- and two-part input) with the output embedded in
    sentinel-filled memory."""
    ops, lib = lgu.ops, lgu._lib.load()
    E, NF, H, W = cfg
    torch.manual_seed(3 * E + H)
    conv = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    frames = (torch.randn(NF, H, W, 128, device="cuda") * 0.125).half()
    lo = (torch.randn(NF, H, W, 128, device="cuda") * 1e-4).half()
    ii = torch.randint(0, NF, (E,), device="cuda")
    jj = torch.randint(0, NF, (E,), device="cuda")
    wpack, bias, Cout, C = ops.pack_offset_conv(conv.weight, conv.bias)
    for part in (None, lo):
        want = ops.offset_conv_frames(frames, ii, jj, (wpack, bias, Cout, C), frames_lo=part)
        big, out, g = _banded((E, Cout, H, W))
        rc = lib.lgu_offset_conv_frames_h16(ops._ptr(frames), ops._ptr(part) if part is not None else None, ops._ptr(ii), ops._ptr(jj),
                                            ops._ptr(wpack), ops._ptr(bias), ops._ptr(out), E, H, W, C, Cout, ops._stream(frames))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, want) and not bool((out == _SENT).any())
        assert _bands_intact(big, g), "the offset convolution wrote outside its output"


@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("probe", [True, False])
@pytest.mark.parametrize("shape", [(2, 16, 16), (1, 24, 40), (3, 12, 16)])
def test_volume_path_entry_points_write_nothing_outside_their_tensors(lgu, shape, probe, tiled):
    E, H, W = shape
    L, R = 4, 3
    rng = np.random.default_rng(5 + E + W)
    vols = [dev(rng.standard_normal((E, H, W, max(H >> l, 1), max(W >> l, 1))).astype(np.float32)) for l in range(L)]
    hw = [tuple(v.shape[3:]) for v in vols]
    use = [lgu.ops.volume_retile(v) for v in vols] if tiled else vols
    coords = dev(inputs.grid_coords(rng, E, H, W, 3.0))
    bigs, offs = [], []
    for l in range(L):
        if l < 2:
            big, view, g = _banded((E, H, W, 7, 7, 2))
            view.copy_(dev((4 * np.tanh(rng.standard_normal((E, H, W, 7, 7, 2)))).astype(np.float32)))
            offs.append(view); bigs.append((big, g))
        else:
            offs.append(None)
    obig, out, og = _banded((E, L * 49, H, W))
    bigs.append((obig, og))
    try:
        got = lgu.ops.defcorr_pyramid_forward(use, coords, offs, R, probe=probe, out=out, tiled=tiled, level_hw=hw if tiled else None)
    except lgu._lib.UnsupportedShape:
        pytest.skip("this shape / layout has no fused probe")
    assert got.data_ptr() == out.data_ptr() and not bool((got == _SENT).any())
    # the pyramid builder: every level it writes sits in a banded buffer of its own
    means = dev((np.stack(np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32)), -1)[None].repeat(E, 0)
                 + rng.standard_normal((E, H, W, 2))).astype(np.float32))
    covs = dev(rng.uniform(0.05, 5.05, (E, H, W, 2)).astype(np.float32))
    raw = dev(rng.standard_normal((E, H, W, H, W)).astype(np.float32))
    try:
        levels = lgu.ops.volume_pyramid(means, covs, raw, L, 4, inplace=False, tiled=tiled)
        assert all(bool(torch.isfinite(v).all()) for v in levels)
    except lgu._lib.UnsupportedShape:
        pass
    torch.cuda.synchronize()
    for big, g in bigs:
        assert _bands_intact(big, g), "a kernel wrote outside the tensor it was given"


@pytest.mark.parametrize("det_mode", ["none", "f32", "half"])
@pytest.mark.parametrize("shape", [(2, 128, 16, 16), (1, 128, 24, 32), (2, 64, 48, 64), (3, 16, 8, 16), (1, 128, 48, 64)])
def test_volume_built_on_the_matrix_cores_equals_matmul_plus_fused_builder(lgu, oracle, shape, det_mode):
    """lgu_volume_build_pyramid_f32 (csrc/volbuild.hip): CorrBlock.__init__'s volume formed on the fp32 matrix cores and written
    straight into the tiled 4-level pyramid == torch.matmul of the maps / 4 (reference corr.py:145-152) followed by the fused
    post-processing (lgu_volume_pyramid_det, itself held to the oracle and the reference build), level by level INCLUDING the
    zero padding of the tiled slices, to the GEMM's fp32 summation order (1e-5 of the level's scale); level 0 of the first
    edge also against the C oracle's composition on the host."""
    E, C, H, W = shape
    rng = np.random.default_rng(900 + C + W)
    f1 = dev((rng.standard_normal((E, C, H, W)) * 0.5).astype(np.float32))
    f2 = dev((rng.standard_normal((E, C, H, W)) * 0.5).astype(np.float32))
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    means = dev((np.stack([xs, ys], -1)[None].repeat(E, 0) + rng.standard_normal((E, H, W, 2)) * 1.5).astype(np.float32))
    covs = dev(rng.uniform(0.05, 5.05, (E, H, W, 2)).astype(np.float32))
    det = None
    if det_mode != "none":
        det = (covs[..., 0] * covs[..., 1]).reshape(E, H * W).contiguous()
        if det_mode == "half":
            det = det.half()
    got = lgu.ops.volume_build_pyramid(f1, f2, means, covs, det)
    raw = torch.matmul((f1.reshape(E, C, H * W) / 4.0).transpose(1, 2), f2.reshape(E, C, H * W) / 4.0).view(E, H, W, H, W).contiguous()
    want = lgu.ops.volume_pyramid(means, covs, raw.clone(), 4, 4, inplace=False, tiled=True, det=det)
    for l in range(4):
        assert got[l].shape == want[l].shape, l
        scale = float(want[l].abs().max())
        assert float((got[l] - want[l]).abs().max()) <= 1e-5 * scale, (l, float((got[l] - want[l]).abs().max()), scale)
    if det_mode == "none":
        lv = oracle.volume_pyramid(host(means[:1]), host(covs[:1]), host(raw[:1]), 4, 4)
        mine0 = lgu.ops.volume_retile(got[0][:1].contiguous(), to_tiled=False, hw=(H, W))
        assert np.abs(host(mine0) - lv[0]).max() <= 1e-5 * float(np.abs(lv[0]).max())


@pytest.mark.parametrize("shape", [(2, 128, 48, 64), (3, 64, 16, 32), (2, 32, 8, 16)], ids=["48x64x128", "16x32x64", "8x16x32"])
@pytest.mark.parametrize("det_mode", ["half", "fp32"])
def test_half_volume_build_equals_the_half_gemm_plus_fused_builder(lgu, shape, det_mode):
    """lgu_volume_build_pyramid_h16: half maps (the reference under autocast: corr.py:145-152 is a half GEMM), product summed in
    fp32 on the matrix cores and rounded to half in the kernel, then the same post-processing.
    (a) Inputs whose product sums are exactly representable (multiples of 1/64 below 8): BIT-identical to the library half
        GEMM + lgu_volume_pyramid_det at every level, padding included — the indexing, the channel assignment of the MFMA
        fragments and the rounding step carry no freedom there.
    (b) Random inputs: the two fp32 summation orders can put a sum on different sides of a half rounding boundary; every
        level-0 entry is within ONE half ulp of its raw product times the re-weighting's gain (1 + 3 / den) of the library
        path, and fewer than 2 % of the entries differ at all."""
    E, C, H, W = shape
    rng = np.random.default_rng(1200 + C + W)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    means = dev((np.stack([xs, ys], -1)[None].repeat(E, 0) + rng.standard_normal((E, H, W, 2)) * 1.5).astype(np.float32))
    covs = dev(rng.uniform(0.05, 5.05, (E, H, W, 2)).astype(np.float32))
    det = (covs[..., 0] * covs[..., 1]).reshape(E, H * W).contiguous()
    if det_mode == "half":
        det = det.half()

    def both(f1, f2):
        t = torch.cat((f1, f2), 1).permute(0, 2, 3, 1).contiguous()                     # CorrBlock's `t` (corr.py:57-62)
        got = lgu.ops.volume_build_pyramid(t, None, means, covs, det)
        raw = torch.matmul((f1.reshape(E, C, H * W) / 4.0).transpose(1, 2), f2.reshape(E, C, H * W) / 4.0)
        assert raw.dtype == torch.float16
        want = lgu.ops.volume_pyramid(means, covs, raw.view(E, H, W, H, W).contiguous(), 4, 4, inplace=False, tiled=True, det=det)
        return got, want, raw

    # (a) exact sums: entries in {-1, -.5, 0, .5, 1}
    f1 = dev((rng.integers(-2, 3, (E, C, H, W)) * 0.5).astype(np.float16))
    f2 = dev((rng.integers(-2, 3, (E, C, H, W)) * 0.5).astype(np.float16))
    got, want, _ = both(f1, f2)
    for l in range(4):
        assert got[l].shape == want[l].shape and torch.equal(got[l], want[l]), l
    # (b) random maps
    f1 = dev((rng.standard_normal((E, C, H, W)) * 0.5).astype(np.float16))
    f2 = dev((rng.standard_normal((E, C, H, W)) * 0.5).astype(np.float16))
    got, want, raw = both(f1, f2)
    g0 = lgu.ops.volume_retile(got[0], to_tiled=False, hw=(H, W)).view(E, H * W, H * W)
    w0 = lgu.ops.volume_retile(want[0], to_tiled=False, hw=(H, W)).view(E, H * W, H * W)
    rawf = raw.float().abs().clamp_min(2.0 ** -14)
    ulp = torch.exp2(torch.floor(torch.log2(rawf)) - 10.0)
    den = 6.28 * torch.sqrt(det.float())
    gain = (1.0 + 3.0 / den).view(E, H * W, 1)
    d = (g0 - w0).abs()
    assert bool((d <= ulp * gain * 1.01 + 1e-12).all()), float((d / (ulp * gain)).max())
    frac = float((d > 0).float().mean())
    assert frac < 0.02, frac
    for l in range(1, 4):
        sc = float(want[l].abs().max())
        assert float((got[l] - want[l]).abs().max()) <= 2e-3 * sc, l


def test_corrblock_half_build_is_opt_in_and_agrees_with_the_library_path_to_the_half_rounding(lgu, monkeypatch):
    """CorrBlock under autocast with half maps: the default keeps the library half GEMM; FUSED_BUILD_HALF (opt-in) builds on
    the matrix cores (no raw volume); lookups of the two blocks agree to the half rounding of the raw products."""
    torch.manual_seed(11)
    h, w = 16, 32
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    GA = lgu.GaussianMask(h, w).cuda()
    f1 = (torch.randn(1, 3, 128, h, w, device="cuda") * 0.5).half()
    f2 = (torch.randn(1, 3, 128, h, w, device="cuda") * 0.5).half()
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None].repeat(1, 3, 1, 1, 1) + torch.randn(1, 3, h, w, 2, device="cuda")).contiguous()
    calls = []
    real = lgu.ops.volume_build_pyramid
    monkeypatch.setattr(lgu.ops, "volume_build_pyramid", lambda *a, **k: (calls.append(a[1] is None), real(*a, **k))[1])
    assert lgu.CorrBlock.FUSED_BUILD_HALF is False or os.environ.get("LGU_FUSED_BUILD_HALF") == "1"   # opt-in
    outs, stores, offsets = [], [], None
    for flag in (False, True):
        monkeypatch.setattr(lgu.CorrBlock, "FUSED_BUILD_HALF", flag)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
            # the half convolutions of the offset heads are not run-to-run identical under autocast (the library picks its
            # algorithm per call): both blocks sample with the first one's offsets, so only the pyramids differ
            if offsets is None:
                offsets = [o.clone() for o in blk.offset]
            blk.offset = [o.clone() for o in offsets]
            outs.append(blk(coords)[0].float())
        stores.append([v.clone() for v in blk._store])
        assert calls == ([True] if flag else []), (flag, calls)
    for l in range(4):
        sc = float(stores[0][l].abs().max())
        assert float((stores[0][l] - stores[1][l]).abs().max()) <= 1e-3 * sc, l    # one half ulp of a raw product, re-weighted
    # the sampler zeroes a whole tap whose window leaves the map (defCorr_sampler_kernel.cu:76-79) and the level-1 offsets pass
    # through the probe's mask, so a last-bit change there can switch an isolated entry: all but a sliver agree to the raw
    # products' half rounding
    sc = float(outs[0].abs().max())
    d = (outs[0] - outs[1]).abs()
    frac_off = float((d > 2e-3 * sc).float().mean())
    assert frac_off < 1e-4, (frac_off, float(d.max()), sc)


def test_volume_build_refuses_what_it_does_not_serve_and_corrblock_falls_back(lgu, monkeypatch):
    # no edges: empty levels of the right shapes, nothing launched (fp32 and half forms)
    z = torch.zeros(0, 32, 8, 16, device="cuda")
    m0 = torch.zeros(0, 8, 16, 2, device="cuda")
    for lv in (lgu.ops.volume_build_pyramid(z, z, m0, m0), lgu.ops.volume_build_pyramid(torch.zeros(0, 8, 16, 64, device="cuda").half(), None, m0, m0)):
        assert [tuple(v.shape) for v in lv] == [lgu.ops.tiled_shape(0, 8, 16, 8 >> l, 16 >> l) for l in range(4)]
    f = torch.zeros(1, 128, 12, 48, device="cuda")
    m = torch.zeros(1, 12, 48, 2, device="cuda")
    with pytest.raises(lgu._lib.UnsupportedShape):
        lgu.ops.volume_build_pyramid(f, f, m, m + 1.0)        # H % 8 != 0, W = 48
    with pytest.raises(lgu._lib.UnsupportedShape):
        lgu.ops.volume_build_pyramid(torch.zeros(1, 12, 16, 16, device="cuda"), torch.zeros(1, 12, 16, 16, device="cuda"),
                                     torch.zeros(1, 16, 16, 2, device="cuda"), torch.ones(1, 16, 16, 2, device="cuda"))   # C % 16
    # CorrBlock: the matrix-core build and the library GEMM + fused post-processing give the same block (1e-5), and a shape the
    # build does not serve takes the latter silently
    torch.manual_seed(5)
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
    for h, w, served in ((16, 32, True), (12, 16, False)):
        GA = lgu.GaussianMask(h, w).cuda()
        torch.nn.init.normal_(GA.meanMap.weight, 0, 0.3)
        f1 = torch.randn(1, 2, 128, h, w, device="cuda") * 0.5
        f2 = torch.randn(1, 2, 128, h, w, device="cuda") * 0.5
        with torch.no_grad():
            monkeypatch.setattr(lgu.CorrBlock, "FUSED_BUILD", True)
            a = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
            monkeypatch.setattr(lgu.CorrBlock, "FUSED_BUILD", False)
            b = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        assert a._store is not None and b._store is not None and a._tiled and b._tiled
        for l in range(4):
            sc = float(b._store[l].abs().max())
            d = float((a._store[l] - b._store[l]).abs().max())
            assert d <= 1e-5 * sc and (served or d == 0.0), (h, w, l, d, sc)
