"""ctypes front-end of the CPU oracle (oracle/lgu_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (lgu-slam_amd/).

All functions take/return C-contiguous float32 numpy arrays, mirror the reference's
operator signatures (offersample_LGS/droid.cpp:138-147, src/droid.cpp:246-247) and
return lists like the reference does.  In/out arguments (`offset`) are modified in
place exactly as the reference modifies them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblgu_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "lgu_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_get_threads.restype = _i
    return _lib


def set_threads(n):
    lib().oracle_set_threads(_i(int(n)))


def get_threads():
    return int(lib().oracle_get_threads())


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "oracle wants contiguous float32"
    return a.ctypes.data_as(_f)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def defCorr_index_forward(volume, coords, offset, radius):
    E, H1, W1, H2, W2 = volume.shape
    rd = 2 * radius + 1
    assert coords.shape == (E, 2, H1, W1) and offset.shape == (E, H1, W1, rd, rd, 2)
    corr = np.empty((E, rd, rd, H1, W1), np.float32)
    lib().oracle_defcorr_fwd(_p(volume), _p(coords), _p(offset), _p(corr), _i(E), _i(H1), _i(W1), _i(H2), _i(W2), _i(radius))
    return [corr]


def defCorr_index_backward(volume, coords, offset, corr_grad, radius):
    E, H1, W1, H2, W2 = volume.shape
    volume_grad = np.empty_like(volume)
    offset_grad = np.empty_like(offset)
    lib().oracle_defcorr_bwd(_p(volume), _p(coords), _p(offset), _p(corr_grad), _p(volume_grad), _p(offset_grad),
                             _i(E), _i(H1), _i(W1), _i(H2), _i(W2), _i(radius))
    return [volume_grad, offset_grad]


def corr_index_forward(volume, coords, radius):
    E, H1, W1, H2, W2 = volume.shape
    rd = 2 * radius + 1
    corr = np.empty((E, rd, rd, H1, W1), np.float32)
    lib().oracle_corridx_fwd(_p(volume), _p(coords), _p(corr), _i(E), _i(H1), _i(W1), _i(H2), _i(W2), _i(radius))
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    E, H1, W1, H2, W2 = volume.shape
    volume_grad = np.empty_like(volume)
    lib().oracle_corridx_bwd(_p(coords), _p(corr_grad), _p(volume_grad), _i(E), _i(H1), _i(W1), _i(H2), _i(W2), _i(radius))
    return [volume_grad]


def gaussianMask(means, covs, volume, radius):
    E, H1, W1, H2, W2 = volume.shape
    volume1 = np.empty_like(volume)
    lib().oracle_gaussmask_fwd(_p(means), _p(covs), _p(volume), _p(volume1), _i(E), _i(H1), _i(W1), _i(H2), _i(W2), _i(radius))
    return [volume1]


def gaussianMask_backward(means, covs, volume, volume_grad, radius):
    E, H1, W1, H2, W2 = volume.shape
    means_grad = np.empty_like(means)
    covs_grad = np.empty_like(covs)
    lib().oracle_gaussmask_bwd(_p(means), _p(covs), _p(volume), _p(volume_grad), _p(means_grad), _p(covs_grad),
                               _i(E), _i(H1), _i(W1), _i(H2), _i(W2), _i(radius))
    return [means_grad, covs_grad]


def lowMem_defSample(fmap1, fmap2, coords, offset, radius):
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    assert C % 32 == 0 and coords.shape == (B, S, H1, W1, 2)
    assert (B - 1) * (S - 1) < offset.shape[0]
    corr = np.empty((B, S, rd, rd, H1, W1), np.float32)
    lib().oracle_lowmem_defsample_fwd(_p(fmap1), _p(fmap2), _p(coords), _p(offset), _p(corr),
                                      _i(B), _i(S), _i(H1), _i(W1), _i(H2), _i(W2), _i(C), _i(radius))
    return [corr]


def altcorr_forward(fmap1, fmap2, coords, radius):
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    assert C % 32 == 0
    corr = np.empty((B, S, rd * rd, H1, W1), np.float32)
    lib().oracle_altcorr_fwd(_p(fmap1), _p(fmap2), _p(coords), _p(corr),
                             _i(B), _i(S), _i(H1), _i(W1), _i(H2), _i(W2), _i(C), _i(radius))
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    fmap1_grad = np.empty_like(fmap1)
    fmap2_grad = np.empty_like(fmap2)
    lib().oracle_altcorr_bwd(_p(fmap1), _p(fmap2), _p(coords), _p(corr_grad), _p(fmap1_grad), _p(fmap2_grad),
                             _i(B), _i(S), _i(H1), _i(W1), _i(H2), _i(W2), _i(C), _i(radius))
    coords_grad = np.zeros((B, S, H1, W1, 2), np.float32)  # never written by the reference (altcorr_kernel.cu:336)
    return [fmap1_grad, fmap2_grad, coords_grad]


def defcorr_pyramid_forward(volumes, coords, offsets, radius, probe=False):
    """CorrBlock.__call__ body (reference droid_slam/modules/corr.py:88-109).

    volumes: list of L arrays (E,H1,W1,H2l,W2l); offsets: list of L arrays or None
    (None = structurally zero); offsets are modified in place (centre zeroing and, with
    probe=True, the level-1 mask).  Returns out (E, L*rd*rd, H1, W1).
    """
    L = len(volumes)
    E, H1, W1 = volumes[0].shape[:3]
    rd = 2 * radius + 1
    out = np.empty((E, L * rd * rd, H1, W1), np.float32)
    vp = (_f * L)(*[_p(v) for v in volumes])
    op = (_f * L)(*[(_p(o) if o is not None else None) for o in offsets])
    h2 = (_i * L)(*[v.shape[3] for v in volumes])
    w2 = (_i * L)(*[v.shape[4] for v in volumes])
    sc = np.empty((E, 2, H1, W1), np.float32)
    so = np.empty((E, H1, W1, rd, rd, 2), np.float32)
    scr = np.empty((E, rd, rd, H1, W1), np.float32)
    sp = np.empty((E, 9, H1, W1), np.float32)
    lib().oracle_defcorr_pyramid_fwd(vp, _p(coords), op, _p(out), _i(L), _i(E), _i(H1), _i(W1), h2, w2,
                                     _i(radius), _i(1 if probe else 0), _p(sc), _p(so), _p(scr), _p(sp))
    return out


def volume_pyramid(means, covs, volume, num_levels, radius=4):
    """CorrBlock.__init__'s volume post-processing (gaussianMask_cuda.py:84-86 + corr.py:79-86).
    Returns the list of num_levels pyramid levels (E,H1,W1,H2>>l,W2>>l)."""
    E, H1, W1, H2, W2 = volume.shape
    levels = [np.empty((E, H1, W1, H2 >> l, W2 >> l), np.float32) for l in range(num_levels)]
    lp = (_f * num_levels)(*[_p(v) for v in levels])
    scratch = np.empty_like(volume)
    lib().oracle_volume_pyramid(_p(means), _p(covs), _p(volume), lp, _i(num_levels), _i(E), _i(H1), _i(W1), _i(H2),
                                _i(W2), _i(radius), _p(scratch))
    return levels
