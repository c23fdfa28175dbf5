"""CPU checks of the oracle itself (no GPU): independent formulations and the identities
the reference satisfies.  The pin against the REAL reference kernels is
tests/test_golden.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import inputs


def test_pyramid_oracle_vs_grid_sample(oracle):
    from oracle import grid_sample_baseline as G
    case = inputs.pyramid_case(1, 2, 12, 16, 3, 3, 3.0)
    want = G.defcorr_pyramid([torch.from_numpy(v) for v in case["volumes"]], torch.from_numpy(case["coords"]),
                             [torch.from_numpy(o) if o is not None else None for o in case["offsets"]], 3).numpy()
    got = oracle.defcorr_pyramid_forward(case["volumes"], case["coords"],
                                         [o.copy() if o is not None else None for o in case["offsets"]], 3)
    # grid_sample's coordinate normalisation costs ~5e-6 absolute
    assert np.abs(got - want).max() < 5e-5


def test_zero_offset_identity_and_centre_zeroing(oracle):
    case = inputs.pyramid_case(2, 2, 12, 16, 1, 3, 6.0, dense_offsets=True)
    v, c, off = case["volumes"][0], case["coords"], case["offsets"][0]
    a, = oracle.defCorr_index_forward(v, c, np.zeros_like(off), 3)
    b, = oracle.corr_index_forward(v, c, 3)
    assert np.array_equal(a, b)  # defCorr(offset=0) == corr_index, bit-exact
    o = off.copy()
    assert o[:, :, :, 3, 3].any()
    oracle.defCorr_index_forward(v, c, o, 3)
    assert not o[:, :, :, 3, 3].any()
    o[:, :, :, 3, 3] = off[:, :, :, 3, 3]
    assert np.array_equal(o, off)  # nothing else touched


def test_whole_tap_rule_differs_from_per_corner_padding(oracle):
    """The volume samplers zero the WHOLE tap when its top-left corner is out of bounds
    (defCorrSample_kernel.cu:67); plain zero padding would keep partial taps."""
    v = np.ones((1, 1, 1, 4, 4), np.float32)
    c = np.array([-0.5, 1.0], np.float32).reshape(1, 2, 1, 1)  # x0 = -0.5: floor = -1
    out, = oracle.corr_index_forward(v, c, 0)
    assert out.shape == (1, 1, 1, 1, 1) and out[0, 0, 0, 0, 0] == 0.0
    g = F.grid_sample(torch.ones(1, 1, 4, 4), torch.tensor([[[[2 * -0.5 / 3 - 1, 2 * 1.0 / 3 - 1]]]]),
                      padding_mode="zeros", align_corners=True)
    assert abs(float(g) - 0.5) < 1e-6


@pytest.mark.parametrize("radius", [1, 3])
def test_backward_is_the_adjoint_of_forward(oracle, radius):
    rng = np.random.default_rng(3)
    E, H1, W1 = 1, 6, 8
    rd = 2 * radius + 1
    v = rng.standard_normal((E, H1, W1, H1, W1)).astype(np.float32)
    c = inputs.grid_coords(rng, E, H1, W1, 2.0)
    off = (3 * np.tanh(rng.standard_normal((E, H1, W1, rd, rd, 2)))).astype(np.float32)
    g = rng.standard_normal((E, rd, rd, H1, W1)).astype(np.float32)
    dv = rng.standard_normal(v.shape).astype(np.float32)
    vg, og = oracle.defCorr_index_backward(v, c, off.copy(), g, radius)
    f1, = oracle.defCorr_index_forward(v + dv, c, off.copy(), radius)
    f0, = oracle.defCorr_index_forward(v, c, off.copy(), radius)
    assert abs(float(((f1 - f0).astype(np.float64) * g).sum()) - float((vg.astype(np.float64) * dv).sum())) < 1e-3
    pvg, = oracle.corr_index_backward(v, c, g, radius)
    p1, = oracle.corr_index_forward(v + dv, c, radius)
    p0, = oracle.corr_index_forward(v, c, radius)
    assert abs(float(((p1 - p0).astype(np.float64) * g).sum()) - float((pvg.astype(np.float64) * dv).sum())) < 1e-3
    assert og.shape == off.shape and np.isfinite(og).all()


def test_lowmem_equals_sampling_the_on_the_fly_volume_with_edge0_offsets(oracle):
    """lowMem_defSample == per-corner-zero-padded bilinear sample of fmap1.fmap2^T, and every
    edge uses offset[0] (the reference's offset[b*n] indexing, lowMem_defSample.cu:80-83)."""
    B, S, H1, W1, H2, W2, C, r = 2, 1, 6, 8, 6, 8, 32, 3
    case = inputs.fmap_case(5, B, S, H1, W1, H2, W2, C, r, sigma=1.5)
    off = case["offset"].copy()
    got, = oracle.lowMem_defSample(case["fmap1"], case["fmap2"], case["coords"], off, r)
    assert not off[0, :, :, 3, 3].any() and np.array_equal(off[1], case["offset"][1])
    vol = np.einsum("bijc,bklc->bijkl", case["fmap1"].astype(np.float64), case["fmap2"].astype(np.float64))
    o0 = case["offset"][0].copy()
    o0[:, :, 3, 3] = 0
    for b in range(B):
        for (h, w, ix, iy) in [(0, 0, 0, 0), (2, 3, 3, 3), (5, 7, 6, 1), (3, 1, 2, 5)]:
            x = case["coords"][b, 0, h, w, 0] + o0[h, w, ix, iy, 0]
            y = case["coords"][b, 0, h, w, 1] + o0[h, w, ix, iy, 1]
            fx, fy = int(np.floor(x)), int(np.floor(y))
            dx, dy = np.float32(x) - np.float32(fx), np.float32(y) - np.float32(fy)
            acc = 0.0
            for (yy, xx, wgt) in [(fy - r + iy, fx - r + ix, (1 - dy) * (1 - dx)), (fy - r + iy, fx - r + ix + 1, (1 - dy) * dx),
                                  (fy - r + iy + 1, fx - r + ix, dy * (1 - dx)), (fy - r + iy + 1, fx - r + ix + 1, dy * dx)]:
                if 0 <= yy < H2 and 0 <= xx < W2:
                    acc += vol[b, h, w, yy, xx] * wgt
            assert abs(got[b, 0, ix, iy, h, w] - acc) < 1e-5


def test_altcorr_equals_plain_sample_of_matmul_volume(oracle):
    B, S, H1, W1, H2, W2, C, r = 2, 1, 6, 8, 6, 8, 64, 1
    case = inputs.fmap_case(6, B, S, H1, W1, H2, W2, C, r, sigma=4.0)
    got, = oracle.altcorr_forward(case["fmap1"], case["fmap2"], case["coords"], r)
    vol = torch.einsum("bijc,bklc->bijkl", torch.from_numpy(case["fmap1"]), torch.from_numpy(case["fmap2"]))
    rd = 2 * r + 1
    xy = torch.from_numpy(case["coords"][:, 0])  # (B,H1,W1,2)
    d = torch.arange(-r, r + 1).float()
    gx = xy[..., 0].reshape(-1, 1, 1) + d.view(1, rd, 1)   # channel = ix*rd + iy: ix major
    gy = xy[..., 1].reshape(-1, 1, 1) + d.view(1, 1, rd)
    grid = torch.stack([2 * gx.expand(-1, rd, rd) / (W2 - 1) - 1, 2 * gy.expand(-1, rd, rd) / (H2 - 1) - 1], -1)
    s = F.grid_sample(vol.reshape(-1, 1, H2, W2), grid, padding_mode="zeros", align_corners=True)[:, 0]
    want = s.view(B, H1, W1, rd * rd).permute(0, 3, 1, 2).numpy()
    assert np.abs(got[:, 0] - want).max() < 5e-5


def test_gaussian_mask_support_and_peak(oracle):
    E, H1, W1 = 1, 4, 4
    v = np.ones((E, H1, W1, 12, 16), np.float32)
    means = np.full((E, H1, W1, 2), 6.0, np.float32)   # exactly on a grid point
    covs = np.ones((E, H1, W1, 2), np.float32)
    out, = oracle.gaussianMask(means, covs, v, 4)
    assert out[0, 0, 0, 6, 6] == pytest.approx(3.0)  # 3 * exp(0)
    assert out[0, 0, 0, 6, 7] == pytest.approx(3.0 * np.exp(-0.5), rel=1e-6)
    assert out[0, 0, 0, 1, 6] == 0 and out[0, 0, 0, 2, 6] > 0  # window = +-4 rows around floor(mean)
    assert out[0, 0, 0, 6, 11] == 0 and out[0, 0, 0, 6, 10] > 0


def test_volume_pyramid_oracle_vs_torch_composition(oracle):
    rng = np.random.default_rng(9)
    E, H1, W1, H2, W2, L = 1, 4, 6, 12, 16, 3
    v = rng.standard_normal((E, H1, W1, H2, W2)).astype(np.float32)
    means = rng.uniform(2, 10, (E, H1, W1, 2)).astype(np.float32)
    covs = rng.uniform(0.05, 5.05, (E, H1, W1, 2)).astype(np.float32)
    got = oracle.volume_pyramid(means, covs, v, L, 4)
    c1, = oracle.gaussianMask(means, covs, v, 4)
    lvl = torch.from_numpy(c1) / (6.28 * torch.sqrt(torch.from_numpy(covs[..., 0] * covs[..., 1])))[..., None, None] + torch.from_numpy(v)
    for l in range(L):
        assert np.abs(got[l] - lvl.numpy()).max() <= 1e-6
        lvl = F.avg_pool2d(lvl.view(-1, 1, H2 >> l, W2 >> l), 2, stride=2).view(E, H1, W1, H2 >> (l + 1), W2 >> (l + 1))
