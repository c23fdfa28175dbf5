"""ctypes binding of liblgu_corr.so (C ABI declared in include/lgu_corr.h).

There is NO fallback: if the shared library is missing or fails to load, every
operator raises.  Build it with `python __graft_entry__.py build` (or
`lgu_slam_amd._build.build()`), which needs only hipcc.
"""
import ctypes
import os

from . import _build

_c_float_p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_int = ctypes.c_int

# name -> argument types (return type is always int)
SIGNATURES = {
    "lgu_defcorr_fwd_f32": [_vp, _vp, _vp, _vp] + [_int] * 6 + [_vp],
    "lgu_defcorr_bwd_f32": [_vp] * 6 + [_int] * 6 + [_vp],
    "lgu_corridx_fwd_f32": [_vp, _vp, _vp] + [_int] * 6 + [_vp],
    "lgu_corridx_bwd_f32": [_vp] * 4 + [_int] * 6 + [_vp],
    "lgu_gaussmask_fwd_f32": [_vp] * 4 + [_int] * 6 + [_vp],
    "lgu_gaussmask_bwd_f32": [_vp] * 6 + [_int] * 6 + [_vp],
    "lgu_defcorr_pyramid_fwd_f32": [ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int,
                                    ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _vp],
    "lgu_defcorr_pyramid_slots_fwd_f32": [ctypes.POINTER(_vp), _vp, _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int,
                                          ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _vp],
    "lgu_volume_pyramid_f32": [_vp, _vp, _vp, ctypes.POINTER(_vp), _int, _int, _int, _int, _int, _int, _int, _vp],
    "lgu_volume_pyramid_tiled_f32": [_vp, _vp, _vp, ctypes.POINTER(_vp), _int, _int, _int, _int, _int, _int, _int, _vp],
    "lgu_volume_pyramid_h16": [_vp, _vp, _vp, ctypes.POINTER(_vp), _int, _int, _int, _int, _int, _int, _int, _int, _vp],
    "lgu_volume_pyramid_det": [_vp, _vp, _vp, _int, _vp, _int, ctypes.POINTER(_vp)] + [_int] * 8 + [_vp],
    "lgu_volume_build_pyramid_f32": [_vp] * 5 + [_int, ctypes.POINTER(_vp)] + [_int] * 6 + [_vp],
    "lgu_volume_build_pyramid_h16": [_vp] * 5 + [_int, ctypes.POINTER(_vp)] + [_int] * 6 + [_vp],
    "lgu_gaussian_params": [_vp] * 5 + [_int] * 4 + [ctypes.c_float, _vp],
    "lgu_probe_mask_scale_f32": [_vp, _vp, _int, _int, _int, _int, _vp],
    "lgu_offset_conv_frames_h16": [_vp] * 7 + [_int] * 5 + [_vp],
    "lgu_offset_heads_mark": [_vp, _vp, _int, _vp, _int, _vp, _vp, _vp],
    "lgu_offset_conv_worklist_h16": [_vp] * 4 + [_int] + [_vp] * 5 + [_int] * 4 + [_vp],
    "lgu_offset_heads_combine_f32": [_vp] * 5 + [_int] * 2 + [_vp, _vp],
    "lgu_offsets_finalize": [_vp] * 5 + [_int] * 7 + [ctypes.c_float, _vp],
    "lgu_offsets_finalize_masked": [_vp] * 3 + [_int] + [_vp] * 3 + [_int] * 7 + [ctypes.c_float, _vp],
    "lgu_volume_retile_f32": [_vp, _vp, ctypes.c_longlong, _int, _int, _int, _vp],
    "lgu_lowmem_defsample_fwd_f32": [_vp] * 5 + [_int] * 9 + [_vp],
    "lgu_altcorr_fwd_f32": [_vp] * 4 + [_int] * 8 + [_vp],
    "lgu_lowmem_defsample_fwd_h16": [_vp] * 5 + [_int] * 9 + [_vp],
    "lgu_altcorr_fwd_h16": [_vp] * 4 + [_int] * 8 + [_vp],
    "lgu_lowmem_pyramid_fwd_h16": [_vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int, _int, _int,
                                   ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _int, _vp, _vp, _vp],
    "lgu_lowmem_pyramid_chunked_fwd_h16": [_vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int, _int, _int,
                                   ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _int, _vp, _vp, _vp],
    "lgu_lowmem_pyramid_chunked_fwd_f32": [_vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int, _int, _int,
                                   ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _int, _vp, _vp, _vp],
    # fmap1, fmap2[], coords, offsets[], out, L, lbase, B, H1, W1, H2[], W2[], C, NO, off_row, radius, ii, jj, chunked, stream
    "lgu_lowmem_pyramid_calls_fwd_h16": [_vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int, _int,
                                         ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _vp, _int, _vp, _vp, _int, _vp],
    "lgu_lowmem_pyramid_fwd_f32": [_vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), _vp, _int, _int, _int, _int, _int, _int,
                                   ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _int, _vp, _vp, _vp],
    "lgu_ba_build_f32": [_vp] * 14 + [_int] * 3 + [_vp],
    "lgu_ba_build_slices": [_int],
    "lgu_ba_accum_f32": [_vp] * 4 + [_int] * 2 + [_vp],
    "lgu_ba_depth_system_f32": [_vp] * 8 + [_int] + [_vp] * 2 + [_int] * 2 + [_vp],
    "lgu_ba_depth_update_f32": [_vp] * 8 + [_int] * 2 + [_vp],
    "lgu_ba_scatter_sum_f64": [_vp] * 5 + [_int] * 2 + [ctypes.c_double, _vp],
    "lgu_ba_eet_f32": [_vp] * 4 + [_int] * 2 + [_vp],
    "lgu_ba_ev_f32": [_vp] * 5 + [_int] * 2 + [_vp],
    "lgu_ba_evt_f32": [_vp] * 4 + [_int] * 3 + [_vp],
    "lgu_ba_solve_f64": [_vp, _vp, _vp, _int, ctypes.c_double, ctypes.c_double, _vp],
    "lgu_ba_solve_blocked_f64": [_vp, _vp, _vp, _vp, _int, ctypes.c_double, ctypes.c_double, _vp],
    "lgu_ba_pose_retr_f32": [_vp] * 2 + [_int] * 2 + [_vp],
    "lgu_ba_assemble_f64": [_vp] * 14 + [_int, _vp],
    "lgu_ba_disp_retr_f32": [_vp] * 3 + [_int] * 2 + [_vp],
    "lgu_altcorr_bwd_f32": [_vp] * 6 + [_int] * 8 + [_vp],
    "lgu_defcorr_pyramid_enc_fwd_f32": [ctypes.POINTER(_vp), _vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp, _int, _int, _int,
                                        _int, ctypes.POINTER(_int), ctypes.POINTER(_int), _int, _int, _int, _vp],
}

_lib = None


class LguLibraryError(RuntimeError):
    pass


class UnsupportedShape(RuntimeError):
    """The kernels do not serve this shape / argument pattern (LGU_E_UNSUPPORTED)."""


LGU_E_BADARG, LGU_E_UNSUPPORTED = 100001, 100002


def so_path():
    """The library this process loads.  LGU_LIB_PATH (experiments only: tools/ab_lib_*.sh) points at another build —
    variants are loaded from where they were built, the in-tree default library is never overwritten."""
    return os.environ.get("LGU_LIB_PATH") or _build.SO_PATH


def load():
    """Load the HIP library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = so_path()
    if not os.path.exists(path):
        raise LguLibraryError(
            "lgu_slam_amd: %s is missing — the gfx950 HIP kernels are not built and there is no CPU "
            "fallback. Run `python __graft_entry__.py build`." % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:  # pragma: no cover - depends on the ROCm install
        raise LguLibraryError("lgu_slam_amd: cannot load %s: %s" % (path, exc)) from exc
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, let it surface
        fn.argtypes = argtypes
        fn.restype = _int
    lib.lgu_ba_solve_blocked_work_doubles.restype = ctypes.c_longlong
    lib.lgu_ba_solve_blocked_work_doubles.argtypes = [_int]
    lib.lgu_offsets_finalize_scratch_bytes.restype = ctypes.c_longlong
    lib.lgu_offsets_finalize_scratch_bytes.argtypes = [_int]
    lib.lgu_version.restype = ctypes.c_char_p
    lib.lgu_debug_knobs_enabled.restype = _int
    lib.lgu_debug_knobs_enabled.argtypes = []
    lib.lgu_error_string.restype = ctypes.c_char_p
    lib.lgu_error_string.argtypes = [_int]
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().lgu_error_string(code).decode()
        raise (UnsupportedShape if code == LGU_E_UNSUPPORTED else RuntimeError)("%s failed: %s (code %d)" % (what, msg, code))


def version():
    return load().lgu_version().decode()


def debug_knobs_enabled():
    """True if the LGU_* debug variables are live in this process (LGU_DEBUG_KNOBS=1 when the library was loaded)."""
    return bool(load().lgu_debug_knobs_enabled())
