"""The C-ABI library builds, loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lgu_corr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lgu_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_nine_reference_ops_plus_fused():
    syms = declared_symbols()
    for s in ("lgu_defcorr_fwd_f32", "lgu_defcorr_bwd_f32", "lgu_corridx_fwd_f32", "lgu_corridx_bwd_f32",
              "lgu_gaussmask_fwd_f32", "lgu_gaussmask_bwd_f32", "lgu_lowmem_defsample_fwd_f32",
              "lgu_altcorr_fwd_f32", "lgu_altcorr_bwd_f32", "lgu_defcorr_pyramid_fwd_f32",
              "lgu_version", "lgu_error_string"):
        assert s in syms


def test_library_builds_loads_and_exports_every_declared_symbol(lgu):
    so = lgu.build()  # no-op when up to date; hipcc cross-compiles without a GPU
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    for s in declared_symbols():
        assert hasattr(lib, s), "missing export %s" % s
    assert set(lgu._lib.SIGNATURES) <= set(declared_symbols())
    assert "gfx950" in lgu._lib.version()
    assert lgu._lib.load().lgu_error_string(100002).decode().startswith("lgu:")


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "lgu_corr.h")).read()
    assert "torch" not in text.replace("torch::zeros", "").replace("PyTorch", "").lower().replace("pytorch", "")
    assert "#include" not in text  # plain C, self-contained
