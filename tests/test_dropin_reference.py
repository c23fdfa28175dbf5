"""Drop-in check against the reference's OWN glue (only where /root/reference exists, i.e. in
the build container; the GPU box never has it): with lgu_slam_amd's drop-in modules installed,
the reference's droid_slam/modules/corr.py and droid_slam/gaussianMask_cuda.py import unchanged
and bind to this library's operators.  No reference code is executed on data here (that needs
the GPU; the kernels themselves are pinned by tests/test_golden.py)."""
import os
import sys
import types

import pytest

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")
def test_reference_glue_imports_against_dropins(lgu):
    d, b = lgu.install_dropins()
    saved_path = list(sys.path)
    saved_mods = {k: sys.modules.get(k) for k in ("cv2", "droid_slam", "droid_slam.modules", "droid_slam.modules.corr",
                                                  "droid_slam.gaussianMask_cuda", "modules", "modules.corr", "gaussianMask_cuda")}
    dont_write = sys.dont_write_bytecode
    try:
        sys.dont_write_bytecode = True          # never write into the read-only reference tree
        sys.modules.setdefault("cv2", types.ModuleType("cv2"))   # the reference imports cv2 for unrelated code
        sys.path.insert(0, os.path.join(REF, "droid_slam"))
        import importlib
        ref_corr = importlib.import_module("modules.corr")
        ref_ga = importlib.import_module("gaussianMask_cuda")
        # the names the reference glue resolves at call time are this library's functions
        assert ref_corr.defCorrSample is d and ref_corr.droid_backends is b and ref_ga.defCorrSample is d
        assert ref_corr.defCorrSample.defCorr_index_forward is lgu.ops.defCorr_index_forward
        assert ref_corr.droid_backends.altcorr_forward is lgu.ops.altcorr_forward
        # same public surface as this build's counterparts
        for name in ("CorrSampler", "DefCorrSampler", "CorrBlock", "AltCorrBlock", "per_Corr_Normalization"):
            assert hasattr(ref_corr, name) and hasattr(lgu.corr, name)
        for name in ("GaussianMask", "GaussianMaskCuda"):
            assert hasattr(ref_ga, name) and hasattr(lgu.gaussian_mask, name)
        import inspect
        assert list(inspect.signature(ref_corr.CorrBlock.__init__).parameters) == list(inspect.signature(lgu.CorrBlock.__init__).parameters)
        assert list(inspect.signature(ref_corr.AltCorrBlock.__init__).parameters) == list(inspect.signature(lgu.AltCorrBlock.__init__).parameters)
        # identical parameter names -> a reference state dict loads into this build's GaussianMask
        assert set(dict(ref_ga.GaussianMask(4, 4).named_parameters())) == set(dict(lgu.GaussianMask(4, 4).named_parameters()))
    finally:
        sys.path[:] = saved_path
        sys.dont_write_bytecode = dont_write
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
