"""lgu_slam_amd — MI355X (gfx950) implementation of LGU-SLAM's deformable
correlation-sampling hot path: hand-written HIP kernels behind a C ABI
(include/lgu_corr.h), the reference's operator signatures (`ops`), and host-side
counterparts of its correlation glue (`corr`, `gaussian_mask`).

The on-disk directory is `lgu-slam_amd/`; `import lgu_slam_amd` works through the alias
module `lgu_slam_amd.py` at the repository root.
"""
import os
import sys

from . import _build, _lib, ba, encoder, ops, sharded  # noqa: F401
from .corr import AltCorrBlock, CorrBlock, CorrSampler, DefCorrSampler, per_Corr_Normalization  # noqa: F401
from .encoder import CorrEncoder  # noqa: F401
from .gaussian_mask import GaussianMask, GaussianMaskCuda  # noqa: F401

__version__ = "0.7.0"

DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def build(force=False, verbose=False):
    """Compile the HIP kernels for gfx950 into lgu-slam_amd/liblgu_corr.so."""
    return _build.build(force=force, verbose=verbose)


def install_dropins(experimental_ba=False):
    """Make `import defCorrSample` / `import droid_backends` resolve to this library, so the
    reference's droid_slam package runs on it unmodified.

    experimental_ba=True additionally binds `droid_backends.ba` to this build's device-side bundle adjustment
    (lgu_slam_amd.ba.ba) — a first version whose parity with the reference is unpinned (the reference BA needs Eigen
    and cannot be built here); by default that name raises, like the other out-of-scope entries."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    import defCorrSample  # noqa: F401
    import droid_backends  # noqa: F401
    if experimental_ba:
        sys.modules["droid_backends"].ba = ba.ba
    return sys.modules["defCorrSample"], sys.modules["droid_backends"]
