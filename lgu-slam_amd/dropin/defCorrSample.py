"""Drop-in for the reference's `defCorrSample` CUDA extension (offersample_LGS/droid.cpp:138-147).

Put this directory on sys.path (or call lgu_slam_amd.install_dropins()) and
`import defCorrSample` in droid_slam/modules/corr.py:8 and droid_slam/gaussianMask_cuda.py:5
resolves to the gfx950 kernels: same seven functions, same positional arguments, same
list-of-tensors returns, same in-place centre zeroing of `offset`.
"""
import lgu_slam_amd.ops as _ops

gaussianMask = _ops.gaussianMask
gaussianMask_backward = _ops.gaussianMask_backward
lowMem_defSample = _ops.lowMem_defSample
corr_index_forward = _ops.corr_index_forward
corr_index_backward = _ops.corr_index_backward
defCorr_index_forward = _ops.defCorr_index_forward
defCorr_index_backward = _ops.defCorr_index_backward
