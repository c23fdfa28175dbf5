"""Drop-in for the correlation part of the reference's `droid_backends` CUDA extension
(src/droid.cpp:246-247): `altcorr_forward` / `altcorr_backward`.

The rest of `droid_backends` (frame_distance, projmap, depth_filter, iproj and the original DROID
corr_index_*: src/droid.cpp:237-249) is outside this build's hot-path scope; those names raise with a
pointer to the reference extension instead of silently doing something else.  `ba` raises too unless
`lgu_slam_amd.install_dropins(experimental_ba=True)` bound it to this build's first device-side bundle
adjustment (lgu_slam_amd.ba.ba: parity with the reference unpinned).
"""
import lgu_slam_amd.ops as _ops

altcorr_forward = _ops.altcorr_forward
altcorr_backward = _ops.altcorr_backward


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            "droid_backends.%s is not part of the lgu_slam_amd hot-path library; "
            "use the reference's droid_backends build for it" % name)
    fn.__name__ = name
    return fn


for _n in ("ba", "frame_distance", "projmap", "depth_filter", "iproj", "corr_index_forward", "corr_index_backward"):
    globals()[_n] = _out_of_scope(_n)
