"""Direct parity against the reference's OWN kernels on the same GPU: oracle/_ref/*.so are the
reference extension sources compiled unmodified for gfx950 by oracle/build_ref.py (they ship to
the GPU box with the snapshot; nothing here reads /root/reference).  Every operator of the hot
path, forward and backward, on seeded inputs incl. BASELINE shapes.  Skipped when the reference
build is absent.

Tolerance 1e-5 (north_star).  The forward samplers are expected bit-identical: same fp32 order,
and FMA contraction in the reference build cannot change a 4-term blend by more than 1 ulp...
so they are held to 1e-6 and reported exactly by tools/compare_ref.py.
"""
import importlib.util
import os

import numpy as np
import pytest

from tests import inputs

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _load(name):
    path = os.path.join(REFDIR, name + ".so")
    if not os.path.exists(path):
        pytest.skip("reference build %s not present (run oracle/build_ref.py where /root/reference exists)" % name)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ref():
    assert torch.cuda.is_available()
    return _load("ref_defCorrSample")


@pytest.fixture(scope="module")
def refalt():
    assert torch.cuda.is_available()
    return _load("ref_altcorr")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol=1e-5):
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("shape", [(2, 12, 16, 12, 16, 3, 3.0, 1.0), (2, 48, 64, 24, 32, 3, 3.0, 0.5), (1, 30, 40, 30, 40, 3, 12.0, 1.0),
                                   (2, 24, 32, 12, 16, 1, 3.0, 0.5)])
def test_volume_samplers_forward_backward(lgu, ref, shape):
    E, H1, W1, H2, W2, r, sigma, cs = shape
    rng = np.random.default_rng(1000 + H2 + r)
    rd = 2 * r + 1
    v = dev(rng.standard_normal((E, H1, W1, H2, W2)).astype(np.float32))
    c = dev((inputs.grid_coords(rng, E, H1, W1, sigma) * cs).astype(np.float32))
    off = (4 * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2)))).astype(np.float32)
    g = dev(rng.standard_normal((E, rd, rd, H1, W1)).astype(np.float32))
    o_r, o_m = dev(off), dev(off)
    a, = ref.defCorr_index_forward(v, c, o_r, r)
    b, = lgu.ops.defCorr_index_forward(v, c, o_m, r)
    assert close(b, a, 1e-6) and torch.equal(o_r, o_m)
    vg_r, og_r = ref.defCorr_index_backward(v, c, dev(off), g, r)
    vg_m, og_m = lgu.ops.defCorr_index_backward(v, c, dev(off), g, r)
    assert close(vg_m, vg_r) and close(og_m, og_r)
    pa, = ref.corr_index_forward(v, c, r)
    pb, = lgu.ops.corr_index_forward(v, c, r)
    assert close(pb, pa, 1e-6)
    pg_r, = ref.corr_index_backward(v, c, g, r)
    pg_m, = lgu.ops.corr_index_backward(v, c, g, r)
    assert close(pg_m, pg_r)


def test_fused_pyramid_and_probe_vs_reference_call_sequence(lgu, ref):
    """CorrBlock.__call__ exactly as the reference drives its ops (corr.py:94-109), E = 4 at the
    BASELINE 48x64 shape, two consecutive calls (persistent offset[1] *= mask)."""
    case = inputs.pyramid_case(2024, 4, 48, 64, 4, 3, 3.0, 4.0, False)
    E, H1, W1 = 4, 48, 64
    vols = [dev(v) for v in case["volumes"]]
    c = dev(case["coords"])
    z = torch.zeros(E, H1, W1, 7, 7, 2, device="cuda")
    ro = [dev(case["offsets"][0]), dev(case["offsets"][1]), z.clone(), z.clone()]
    mo = [dev(case["offsets"][0]), dev(case["offsets"][1]), None, None]
    for call in range(2):
        probe, = ref.corr_index_forward(vols[1], c / 2, 1)
        mask = torch.sigmoid(torch.var(probe.permute(0, 3, 4, 1, 2), dim=[3, 4])).view(E, H1, W1, 1, 1, 1)
        ro[1] = (ro[1] * mask).contiguous()
        want = torch.cat([ref.defCorr_index_forward(vols[l], (c / 2 ** l).contiguous(), ro[l], 3)[0].view(E, 49, H1, W1)
                          for l in range(4)], 1)
        got = lgu.ops.defcorr_pyramid_forward(vols, c, mo, 3, probe=True)
        assert float((mo[1] - ro[1]).abs().max()) <= 2e-6, call
        assert close(got, want), call


def test_gaussian_mask_and_postprocessing(lgu, ref):
    rng = np.random.default_rng(5)
    E, H1, W1 = 2, 48, 64
    v = dev(rng.standard_normal((E, H1, W1, H1, W1)).astype(np.float32))
    ys, xs = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
    means = dev((np.stack([xs, ys], -1)[None].repeat(E, 0) + 2 * rng.standard_normal((E, H1, W1, 2))).astype(np.float32))
    covs = dev(rng.uniform(0.05, 5.05, (E, H1, W1, 2)).astype(np.float32))
    g = dev(rng.standard_normal((E, H1, W1, H1, W1)).astype(np.float32))
    a, = ref.gaussianMask(means, covs, v, 4)
    b, = lgu.ops.gaussianMask(means, covs, v, 4)
    assert close(b, a)
    mg_r, cg_r = ref.gaussianMask_backward(means, covs, v, g, 4)
    mg_m, cg_m = lgu.ops.gaussianMask_backward(means, covs, v, g, 4)
    assert close(mg_m, mg_r) and close(cg_m, cg_r)
    # CorrBlock.__init__ post-processing as the reference composes it
    lvl = a / (6.28 * torch.sqrt(covs[..., 0] * covs[..., 1]))[..., None, None] + v
    fused = lgu.ops.volume_pyramid(means, covs, v, 4, 4)
    for l in range(4):
        assert close(fused[l], lvl, 1e-6), l
        lvl = torch.nn.functional.avg_pool2d(lvl.view(-1, 1, H1 >> l, W1 >> l), 2, stride=2).view(E, H1, W1, H1 >> (l + 1), W1 >> (l + 1))


@pytest.mark.parametrize("cfg", [(3, 1, 60, 80, 60, 80, 128, 3, 3.0, 1.0), (2, 1, 60, 80, 30, 40, 128, 3, 3.0, 0.5),
                                 (2, 1, 60, 80, 7, 10, 128, 3, 3.0, 0.125), (2, 2, 8, 16, 8, 16, 64, 3, 4.0, 1.0)])
def test_lowmem_vs_reference(lgu, ref, cfg):
    B, S, H1, W1, H2, W2, C, r, sigma, cs = cfg
    case = inputs.fmap_case(3000 + H2, B, S, H1, W1, H2, W2, C, r, sigma, cs)
    o_r, o_m = dev(case["offset"]), dev(case["offset"])
    a, = ref.lowMem_defSample(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), o_r, r)
    b, = lgu.ops.lowMem_defSample(dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"]), o_m, r)
    assert close(b, a) and torch.equal(o_r, o_m)


@pytest.mark.parametrize("cfg", [(3, 1, 60, 80, 30, 40, 128, 1, 3.0, 0.5), (1, 2, 8, 16, 8, 16, 64, 3, 4.0, 1.0)])
def test_altcorr_vs_reference(lgu, refalt, cfg):
    B, S, H1, W1, H2, W2, C, r, sigma, cs = cfg
    case = inputs.fmap_case(4000 + H2, B, S, H1, W1, H2, W2, C, r, sigma, cs)
    f1, f2, c = dev(case["fmap1"]), dev(case["fmap2"]), dev(case["coords"])
    a, = refalt.altcorr_forward(f1, f2, c, r)
    b, = lgu.ops.altcorr_forward(f1, f2, c, r)
    assert close(b, a)
    g = torch.randn_like(a)
    r1, r2, r3 = refalt.altcorr_backward(f1, f2, c, g, r)
    m1, m2, m3 = lgu.ops.altcorr_backward(f1, f2, c, g, r)
    assert close(m1, r1) and close(m2, r2) and not bool(m3.any()) and not bool(r3.any())


def test_non_finite_and_wild_coordinates_match_reference(lgu, ref):
    """Reprojection can hand the lookup NaN / inf / huge coordinates (points behind the camera).
    The reference's behaviour there is whatever float->int conversion the hardware does; both
    builds run on the same device, so outputs must agree exactly (NaN positions included) and
    nothing may fault."""
    rng = np.random.default_rng(77)
    E, H1, W1 = 1, 16, 32
    case = inputs.pyramid_case(555, E, H1, W1, 2, 3, 3.0, 4.0, True)
    c = case["coords"].copy()
    c[0, 0, 0, :8] = np.nan
    c[0, 1, 1, :8] = np.inf
    c[0, 0, 2, :8] = -np.inf
    c[0, :, 3, :8] = 3.0e9
    c[0, :, 4, :8] = -3.0e9
    c[0, 0, 5, :8] = 1e-40  # subnormal
    vols = [dev(v) for v in case["volumes"]]
    cd = dev(c)
    for variant in ("0", "1", "2", "3"):
        os.environ["LGU_DEFCORR_VARIANT"] = variant
        try:
            for l in range(2):
                cl = (cd / 2 ** l).contiguous()
                a, = ref.defCorr_index_forward(vols[l], cl, dev(case["offsets"][l]), 3)
                b, = lgu.ops.defCorr_index_forward(vols[l], cl, dev(case["offsets"][l]), 3)
                assert torch.equal(torch.isnan(a), torch.isnan(b)), (variant, l)
                assert torch.allclose(a, b, rtol=0, atol=1e-6, equal_nan=True), (variant, l)
        finally:
            os.environ.pop("LGU_DEFCORR_VARIANT", None)
    # low-memory path
    fc = inputs.fmap_case(556, 2, 1, 8, 16, 8, 16, 64, 3, 3.0, 1.0)
    cc = fc["coords"].copy()
    cc[0, 0, 0, :4, 0] = np.nan
    cc[0, 0, 1, :4, 1] = np.inf
    cc[1, 0, 2, :4, :] = -2.5e9
    a, = ref.lowMem_defSample(dev(fc["fmap1"]), dev(fc["fmap2"]), dev(cc), dev(fc["offset"]), 3)
    b, = lgu.ops.lowMem_defSample(dev(fc["fmap1"]), dev(fc["fmap2"]), dev(cc), dev(fc["offset"]), 3)
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    assert torch.allclose(a, b, rtol=0, atol=1e-5, equal_nan=True)


def test_half_feature_maps_and_fused_levels_vs_reference_call_sequence(lgu, ref, refalt):
    """What AltCorrBlock issues for half feature buffers — the level-1 probe and all levels in one launch on the
    matrix cores — against the reference's own call sequence on the `.float()` copies (corr.py:192-213):
    altcorr_forward + 4 x lowMem_defSample + cat, BASELINE config 4 shapes."""
    torch.manual_seed(31)
    B, H, W, C, L = 3, 60, 80, 128, 4
    f1 = (torch.randn(B, H, W, C, device="cuda") * 0.125).half()
    f2s = [(torch.randn(B, H >> l, W >> l, C, device="cuda") * 0.125).half() for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3 * torch.randn(B, 1, H, W, 2, device="cuda")).contiguous()
    offs = [(4 * torch.tanh(torch.randn(B, H, W, 7, 7, 2, device="cuda"))).contiguous() for _ in range(2)]
    offs += [torch.zeros(B, H, W, 7, 7, 2, device="cuda") for _ in range(2)]
    want = torch.cat([ref.lowMem_defSample(f1.float(), f2s[l].float(), (coords / 2 ** l).contiguous(), offs[l].clone(), 3)[0]
                      .view(B, 1, 49, H, W) for l in range(L)], 2)
    got = lgu.ops.lowmem_pyramid_forward_mixed(f1, f2s, coords, [offs[0].clone(), offs[1].clone(), None, None], 3)
    assert close(got, want)
    pa, = refalt.altcorr_forward(f1.float(), f2s[1].float(), (coords / 2).contiguous(), 1)
    pb = lgu.ops.lowmem_pyramid_forward_mixed(f1, [f2s[1]], coords, [None], 1, lbase=1)
    assert close(pb, pa)
    # float maps through the fp32 matrix-core kernel
    gotf = lgu.ops.lowmem_pyramid_forward_mixed(f1.float(), [f.float() for f in f2s], coords, [offs[0].clone(), offs[1].clone(), None, None], 3)
    assert close(gotf, want)


def test_non_finite_coordinates_new_paths_match_reference(lgu, ref):
    """NaN / inf / huge coordinates through the tiled pyramid layout, the interleaved-coords form and the half
    matrix-core low-memory kernel: same NaN positions and values as the reference kernels, nothing faults."""
    case = inputs.pyramid_case(557, 1, 16, 32, 2, 3, 3.0, 4.0, True)
    c = case["coords"].copy()
    c[0, 0, 0, :8] = np.nan
    c[0, 1, 1, :8] = np.inf
    c[0, 0, 2, :8] = -np.inf
    c[0, :, 3, :8] = 3.0e9
    c[0, :, 4, :8] = -3.0e9
    vols = [dev(v) for v in case["volumes"]]
    hw = [tuple(v.shape[3:]) for v in vols]
    cd = dev(c)
    want = torch.cat([ref.defCorr_index_forward(vols[l], (cd / 2 ** l).contiguous(), dev(case["offsets"][l]), 3)[0].view(1, 49, 16, 32)
                      for l in range(2)], 1)
    got = lgu.ops.defcorr_pyramid_forward([lgu.ops.volume_retile(v) for v in vols], cd.permute(0, 2, 3, 1).contiguous(),
                                          [dev(o) for o in case["offsets"]], 3, tiled=True, level_hw=hw, coords_last=True)
    assert torch.equal(torch.isnan(want), torch.isnan(got))
    assert torch.allclose(want, got, rtol=0, atol=1e-6, equal_nan=True)
    fc = inputs.fmap_case(558, 2, 1, 8, 16, 8, 16, 64, 3, 3.0, 1.0)
    cc = fc["coords"].copy()
    cc[0, 0, 0, :4, 0] = np.nan
    cc[0, 0, 1, :4, 1] = np.inf
    cc[1, 0, 2, :4, :] = -2.5e9
    f1h, f2h = dev(fc["fmap1"]).half(), dev(fc["fmap2"]).half()
    a, = ref.lowMem_defSample(f1h.float(), f2h.float(), dev(cc), dev(fc["offset"]), 3)
    b, = lgu.ops.lowMem_defSample_mixed(f1h, f2h, dev(cc), dev(fc["offset"]), 3)
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    assert torch.allclose(a, b, rtol=0, atol=1e-5, equal_nan=True)
