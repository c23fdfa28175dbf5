#!/usr/bin/env python3
"""Generates tests/golden/glue_*.npz by running the REFERENCE's own Python glue, unchanged.

TEST INFRASTRUCTURE, build container only (needs /root/reference; the GPU box never has it).

What runs: `/root/reference/droid_slam/modules/corr.py` (CorrBlock, AltCorrBlock, the two
autograd Functions, per_Corr_Normalization) and `/root/reference/droid_slam/gaussianMask_cuda.py`
(GaussianMask, GaussianMaskCuda), imported as they lie there — nothing of their text is copied,
no bytecode is written (`sys.dont_write_bytecode`).  The three extension modules they import at
module level do not exist on a CPU-only machine, so stand-ins are injected into `sys.modules`
(SURVEY.md App. B.2):
    defCorrSample, droid_backends -> the operators of oracle/oracle.py, i.e. the C restatement
                                     of the reference kernels that tests/golden/*.npz (outputs of
                                     the reference's own kernels) pin — same signatures, list
                                     returns, in-place `offset` side effect, contiguity errors;
    cv2                            -> an empty module (imported by corr.py, never used).
Everything else — pyramid construction, offset heads and their post-processing, the Gaussian
head, the probe / uncertainty mask compounding into offset[1], cat / __getitem__ bookkeeping,
AltCorrBlock's per-call offsets and `/4`, `*4` scalings — is the reference's code executing.

What is stored (arrays only): seeded inputs (feature maps as float16 VALUES so a half and a float
run see the same numbers), the state dict of the three learned heads (non-zero: the reference
initialises ofsMap / ofs_residual / meanMap to zero, which would make every offset zero), the raw
convolution / linear outputs caught by forward hooks (so a test can separate "same post-processing
given the same head outputs" from end-to-end agreement), offsets before / after each call,
pyramid levels, mean_n, theta and every returned tensor.

    python oracle/gen_glue_golden.py [outdir=tests/golden]
"""
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
_here = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != _here]   # `oracle` must be the package, not oracle/oracle.py
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ----------------------------------------------------------------------------- stand-in extension modules
def _arr(name, t):
    """What the reference's launchers accept: a contiguous fp32 tensor (offersample_LGS/droid.cpp:48-49)."""
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float but found %s" % str(t.dtype).replace("torch.", "").capitalize())
    return t.detach().numpy()          # shares memory: the oracle's in-place centre zeroing lands in the tensor


def _out(arrs):
    return [torch.from_numpy(a) for a in arrs]


def make_stub_modules():
    from oracle import oracle as O
    O.build()
    d = types.ModuleType("defCorrSample")
    d.gaussianMask = lambda means, covs, volume, radius: _out(
        O.gaussianMask(_arr("means", means), _arr("covs", covs), _arr("volume", volume), int(radius)))
    d.gaussianMask_backward = lambda means, covs, volume, volume_grad, radius: _out(
        O.gaussianMask_backward(_arr("means", means), _arr("covs", covs), _arr("volume", volume),
                                _arr("volume_grad", volume_grad), int(radius)))
    d.lowMem_defSample = lambda fmap1, fmap2, coords, offset, radius: _out(
        O.lowMem_defSample(_arr("fmap1", fmap1), _arr("fmap2", fmap2), _arr("coords", coords), _arr("offset", offset), int(radius)))
    d.corr_index_forward = lambda volume, coords, radius: _out(
        O.corr_index_forward(_arr("volume", volume), _arr("coords", coords), int(radius)))
    d.corr_index_backward = lambda volume, coords, corr_grad, radius: _out(
        O.corr_index_backward(_arr("volume", volume), _arr("coords", coords), _arr("corr_grad", corr_grad), int(radius)))
    d.defCorr_index_forward = lambda volume, coords, offset, radius: _out(
        O.defCorr_index_forward(_arr("volume", volume), _arr("coords", coords), _arr("offset", offset), int(radius)))
    d.defCorr_index_backward = lambda volume, coords, offset, corr_grad, radius: _out(
        O.defCorr_index_backward(_arr("volume", volume), _arr("coords", coords), _arr("offset", offset),
                                 _arr("corr_grad", corr_grad), int(radius)))
    b = types.ModuleType("droid_backends")
    b.altcorr_forward = lambda fmap1, fmap2, coords, radius: _out(
        O.altcorr_forward(_arr("fmap1", fmap1), _arr("fmap2", fmap2), _arr("coords", coords), int(radius)))
    b.altcorr_backward = lambda fmap1, fmap2, coords, corr_grad, radius: _out(
        O.altcorr_backward(_arr("fmap1", fmap1), _arr("fmap2", fmap2), _arr("coords", coords), _arr("corr_grad", corr_grad),
                           int(radius)))
    return d, b, types.ModuleType("cv2")


_loaded = None


def load_reference_glue():
    """(corr module, gaussianMask_cuda module) of the reference, imported unchanged over the stand-ins.
    sys.modules / sys.path are restored afterwards; the two returned modules keep their bindings."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not os.path.isdir(REF):
        raise RuntimeError("%s is not present: the glue fixtures can only be generated in the build container" % REF)
    import importlib
    d, b, cv2 = make_stub_modules()
    names = ("defCorrSample", "droid_backends", "cv2", "modules", "modules.corr", "gaussianMask_cuda")
    saved = {k: sys.modules.get(k) for k in names}
    saved_path = list(sys.path)
    try:
        sys.modules["defCorrSample"], sys.modules["droid_backends"], sys.modules["cv2"] = d, b, cv2
        for k in ("modules", "modules.corr", "gaussianMask_cuda"):
            sys.modules.pop(k, None)
        sys.path.insert(0, os.path.join(REF, "droid_slam"))
        ref_corr = importlib.import_module("modules.corr")
        ref_ga = importlib.import_module("gaussianMask_cuda")
    finally:
        sys.path[:] = saved_path
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert ref_corr.__file__.startswith(REF) and ref_ga.__file__.startswith(REF)
    _loaded = (ref_corr, ref_ga)
    return _loaded


# ----------------------------------------------------------------------------- seeded inputs and heads
def make_heads(ref_ga, h, w, seed=20260):
    """ofsMap, ofs_residual (droid_net.py:145-146) and GaussianMask(h, w) (:144) with NON-ZERO seeded weights."""
    g = torch.Generator().manual_seed(seed)
    ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1)
    ofs_residual = torch.nn.Conv2d(256, 98, 3, padding=1)
    GA = ref_ga.GaussianMask(h, w)
    with torch.no_grad():
        for conv in (ofsMap, ofs_residual):
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.03)
            conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
        GA.map.weight.copy_(torch.randn(GA.map.weight.shape, generator=g) * 0.2)
        GA.map.bias.copy_(torch.randn(GA.map.bias.shape, generator=g) * 0.1)
        GA.meanMap.weight.copy_(torch.randn(GA.meanMap.weight.shape, generator=g) * 0.6)
        GA.meanMap.bias.copy_(torch.randn(GA.meanMap.bias.shape, generator=g) * 0.3)
        GA.covMap.weight.copy_(torch.randn(GA.covMap.weight.shape, generator=g) * 0.7)
        GA.covMap.bias.copy_(torch.randn(GA.covMap.bias.shape, generator=g) * 0.2)
    for m in (ofsMap, ofs_residual, GA):
        m.eval()
    return ofsMap, ofs_residual, GA


def heads_state(ofsMap, ofs_residual, GA):
    st = {}
    for prefix, m in (("ofsMap.", ofsMap), ("ofs_residual.", ofs_residual), ("GA.", GA)):
        for k, v in m.state_dict().items():
            st["w:" + prefix + k] = v.detach().numpy().copy()
    return st


def half_valued(g, shape, scale=0.5):
    """N(0, scale^2) rounded to half precision (returned as float16; .float() of it is exact)."""
    return (torch.randn(shape, generator=g) * scale).half()


def grid_coords(g, E, h, w, sigma):
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    grid = torch.stack([xs, ys], dim=-1)[None, None].expand(1, E, h, w, 2)
    return (grid + torch.randn((1, E, h, w, 2), generator=g) * sigma).contiguous()


class Tap:
    """Forward hooks that keep the raw outputs of the learned heads, in call order."""

    def __init__(self, **mods):
        self.got = {k: [] for k in mods}
        self.handles = [m.register_forward_hook(lambda _m, _i, o, k=k: self.got[k].append(o.detach().clone()))
                        for k, m in mods.items()]

    def pop(self, k):
        v = self.got[k]
        self.got[k] = []
        return v

    def close(self):
        for hd in self.handles:
            hd.remove()


def n_(t):
    return t.detach().contiguous().numpy().copy()


# ----------------------------------------------------------------------------- scenarios
def corrblock_scenario(h, w, E_a, E_b, seed, sub=None):
    """CorrBlock life as factor_graph.py drives it: build A (E_a edges), look up, build B (E_b), A.cat(B)
    (add_factors, :121-123), look up all, drop the middle edge (rm_factors, :139-140), look up again.
    `sub` = (ystep, xstep): store per-pixel tensors on that pixel sub-grid only (production shape)."""
    ref_corr, ref_ga = load_reference_glue()
    g = torch.Generator().manual_seed(seed)
    ofsMap, ofs_residual, GA = make_heads(ref_ga, h, w)
    E = E_a + E_b
    f1 = half_valued(g, (1, E, 128, h, w))
    f2 = half_valued(g, (1, E, 128, h, w))
    coords = [grid_coords(g, E, h, w, s) for s in (2.5, 4.0, 1.5)]
    tap = Tap(ofsMap=ofsMap, ofs_residual=ofs_residual, meanMap=GA.meanMap, covMap=GA.covMap)
    out = dict(fmap1=f1.numpy(), fmap2=f2.numpy(),
               coords1=n_(coords[0]), coords2=n_(coords[1]), coords3=n_(coords[2]), E_a=E_a, E_b=E_b, h=h, w=w)
    ys, xs = (sub if sub else (1, 1))

    def px(t):     # (E,h,w,...) -> pixel sub-grid
        return n_(t[:, ::ys, ::xs])

    with torch.no_grad():
        A = ref_corr.CorrBlock(ofsMap, ofs_residual, GA, f1[:, :E_a].float(), f2[:, :E_a].float(), num_levels=4, radius=3)
        out["A_raw_o0"], out["A_raw_o1_low"] = n_(tap.pop("ofsMap")[0]), n_(tap.pop("ofs_residual")[0])
        out["A_raw_mean_ofs"], out["A_raw_cov"] = n_(tap.pop("meanMap")[0]), n_(tap.pop("covMap")[0])
        out["A_mean_n"], out["A_theta"] = n_(A.mean_n), n_(A.theta)
        for l in range(4):
            if l < 2:
                out["A_offset%d_init" % l] = px(A.offset[l])
            else:   # zero by construction (corr.py:132-135): shape and the fact are enough
                assert float(A.offset[l].abs().max()) == 0.0
                out["A_offset%d_shape" % l] = np.array(A.offset[l].shape)
            lv = A.corr_pyramid[l]
            out["A_pyr%d" % l] = n_(lv) if not sub else n_(lv[:, ::ys * 4, ::xs * 4])   # source-pixel sub-grid of slices
        r1, mean_n, theta = A(coords[0][:, :E_a])
        assert mean_n is A.mean_n and theta is A.theta
        out["out1"] = n_(r1[:, :, :, ::ys, ::xs])
        out["A_offset0_after1"], out["A_offset1_after1"] = px(A.offset[0]), px(A.offset[1])

        B = ref_corr.CorrBlock(ofsMap, ofs_residual, GA, f1[:, E_a:].float(), f2[:, E_a:].float(), num_levels=4, radius=3)
        tap.pop("ofsMap"), tap.pop("ofs_residual")
        out["B_mean_n"], out["B_theta"] = n_(B.mean_n), n_(B.theta)
        for l in range(2):
            out["B_offset%d_init" % l] = px(B.offset[l])
        A = A.cat(B)
        r2, _, _ = A(coords[1])
        out["out2"] = n_(r2[:, :, :, ::ys, ::xs])
        out["AB_offset0_after2"], out["AB_offset1_after2"] = px(A.offset[0]), px(A.offset[1])
        out["AB_pyr1_shape"] = np.array(A.corr_pyramid[1].shape)

        keep = torch.ones(E, dtype=torch.bool)
        keep[E // 2] = False                       # rm_factors: self.corr = self.corr[~mask]
        A = A[keep]
        r3, _, _ = A(coords[2][:, keep])
        out["keep"] = keep.numpy()
        out["out3"] = n_(r3[:, :, :, ::ys, ::xs])
        out["AB_offset1_after3"] = px(A.offset[1])
        assert float(A.offset[2].abs().max()) == 0.0 and A.offset[2].shape[0] == E - 1
    tap.close()
    out["pixel_step"] = np.array([ys, xs])
    return out


def altcorr_scenario(h, w, N, edges_calls, seed, store_float_run=True, store_raw=True):
    """AltCorrBlock as update_lowmem drives it (factor_graph.py:262-279): one block over the frame buffer, one call per
    chunk of edges.  Run twice by the reference: on the half buffer (what depth_video holds) and on its float copy."""
    ref_corr, ref_ga = load_reference_glue()
    g = torch.Generator().manual_seed(seed)
    ofsMap, ofs_residual, GA = make_heads(ref_ga, h, w)
    fmaps = half_valued(g, (1, N, 128, h, w))
    out = dict(fmaps=fmaps.numpy(), h=h, w=w, n_calls=len(edges_calls))
    tap = Tap(ofsMap=ofsMap, ofs_residual=ofs_residual)
    runs = [("h", fmaps)] + ([("f", fmaps.float())] if store_float_run else [])
    calls = []
    for c, edges in enumerate(edges_calls):
        ii = torch.tensor([e[0] for e in edges], dtype=torch.long)
        jj = torch.tensor([e[1] for e in edges], dtype=torch.long)
        co = grid_coords(g, len(edges), h, w, 2.0 + c)
        calls.append((ii, jj, co))
        out["c%d_ii" % c], out["c%d_jj" % c], out["c%d_coords" % c] = ii.numpy(), jj.numpy(), n_(co)
    with torch.no_grad():
        for tag, fm in runs:
            blk = ref_corr.AltCorrBlock(ofsMap, ofs_residual, GA, fm, num_levels=4, radius=3)
            for l in range(4):
                lv = blk.pyramid[l]
                assert lv.dtype == fm.dtype and lv.shape == (1, N, h >> l, w >> l, 128)
                if l >= 1:      # level 0 is fmaps / 4, channel-last
                    out["%s_pyr%d" % (tag, l)] = n_(lv)
                else:
                    assert torch.equal(lv[0], (fm[0] / 4.0).permute(0, 2, 3, 1))
            for c, (ii, jj, co) in enumerate(calls):
                r = blk(co, ii, jj)
                out["%s_c%d_out" % (tag, c)] = n_(r)
                r0, r1 = tap.pop("ofsMap")[0], tap.pop("ofs_residual")[0]
                if store_raw:
                    out["%s_c%d_raw_o0" % (tag, c)], out["%s_c%d_raw_o1_low" % (tag, c)] = n_(r0), n_(r1)
                for l in range(2):
                    out["%s_c%d_offset%d_after" % (tag, c, l)] = n_(blk.offset[l])
                assert float(blk.offset[2].abs().max()) == 0.0 and float(blk.offset[3].abs().max()) == 0.0
    tap.close()
    return out


def head_functions_case(seed=5):
    """Inputs/outputs of the reference's three pure-torch pieces, at a shape no class hard-wires."""
    ref_corr, ref_ga = load_reference_glue()
    g = torch.Generator().manual_seed(seed)
    x4 = torch.randn((3, 98, 6, 8), generator=g) * 1.7 + 0.3
    x3 = torch.randn((3, 48, 2), generator=g) * 0.8 - 0.2
    return dict(pcn4_in=n_(x4), pcn4_out=n_(ref_corr.per_Corr_Normalization(x4, [1, 2, 3])),
                pcn3_in=n_(x3), pcn3_out=n_(ref_ga.per_Corr_Normalization(x3, [1, 2])))


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))

    def save(name, arrs):
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
        print("wrote %-28s %6.2f MB  (%d arrays)" % (name, os.path.getsize(path) / 1e6, len(arrs)))

    _, ref_ga = load_reference_glue()
    save("glue_heads", heads_state(*make_heads(ref_ga, 16, 16)))    # the same seeded heads in every scenario
    save("glue_functions", head_functions_case())
    save("glue_corrblock_16x16", corrblock_scenario(16, 16, 2, 1, seed=101))
    save("glue_corrblock_48x64", corrblock_scenario(48, 64, 1, 1, seed=102, sub=(3, 4)))
    save("glue_altcorr_16x16", altcorr_scenario(16, 16, 4, [[(0, 1), (0, 2), (1, 0), (1, 1)], [(2, 3), (3, 0)]], seed=103))
    save("glue_altcorr_24x32", altcorr_scenario(24, 32, 3, [[(1, 2), (0, 1), (2, 0)]], seed=104, store_float_run=False,
                                                   store_raw=False))
    print("GLUE_GOLDEN_DONE")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden"))
