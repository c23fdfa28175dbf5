"""`torch.ops.lgu.*`: the operator layer registered with torch.library, for callers that
dispatch through `torch.ops` (torch.compile graphs, serialized programs) instead of importing
the drop-in modules.  Same names, arguments and in-place side effects as `ops` /
the reference's pybind11 entries (offersample_LGS/droid.cpp:138-147, src/droid.cpp:246-247);
multi-tensor returns are Python lists exactly like the reference's std::vector<Tensor>.
Importing this module performs the registration once.
"""
from typing import List

import torch

from . import ops as _ops

_NS = "lgu"


def _define(name, mutates, fn):
    return torch.library.custom_op("%s::%s" % (_NS, name), mutates_args=mutates, device_types="cuda")(fn)


def _defCorr_index_forward(volume: torch.Tensor, coords: torch.Tensor, offset: torch.Tensor, radius: int) -> List[torch.Tensor]:
    return _ops.defCorr_index_forward(volume, coords, offset, radius)


def _defCorr_index_backward(volume: torch.Tensor, coords: torch.Tensor, offset: torch.Tensor, corr_grad: torch.Tensor,
                            radius: int) -> List[torch.Tensor]:
    return _ops.defCorr_index_backward(volume, coords, offset, corr_grad, radius)


def _corr_index_forward(volume: torch.Tensor, coords: torch.Tensor, radius: int) -> List[torch.Tensor]:
    return _ops.corr_index_forward(volume, coords, radius)


def _corr_index_backward(volume: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor, radius: int) -> List[torch.Tensor]:
    return _ops.corr_index_backward(volume, coords, corr_grad, radius)


def _gaussianMask(means: torch.Tensor, covs: torch.Tensor, volume: torch.Tensor, radius: int) -> List[torch.Tensor]:
    return _ops.gaussianMask(means, covs, volume, radius)


def _gaussianMask_backward(means: torch.Tensor, covs: torch.Tensor, volume: torch.Tensor, volume_grad: torch.Tensor,
                           radius: int) -> List[torch.Tensor]:
    return _ops.gaussianMask_backward(means, covs, volume, volume_grad, radius)


def _lowMem_defSample(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, offset: torch.Tensor,
                      radius: int) -> List[torch.Tensor]:
    return _ops.lowMem_defSample(fmap1, fmap2, coords, offset, radius)


def _altcorr_forward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, radius: int) -> List[torch.Tensor]:
    return _ops.altcorr_forward(fmap1, fmap2, coords, radius)


def _altcorr_backward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor,
                      radius: int) -> List[torch.Tensor]:
    return _ops.altcorr_backward(fmap1, fmap2, coords, corr_grad, radius)


REGISTERED = {}
if not hasattr(torch.ops, _NS) or not hasattr(getattr(torch.ops, _NS), "defCorr_index_forward"):
    for _name, _mut, _fn in (
        ("defCorr_index_forward", ("offset",), _defCorr_index_forward),
        ("defCorr_index_backward", ("offset",), _defCorr_index_backward),
        ("corr_index_forward", (), _corr_index_forward),
        ("corr_index_backward", (), _corr_index_backward),
        ("gaussianMask", (), _gaussianMask),
        ("gaussianMask_backward", (), _gaussianMask_backward),
        ("lowMem_defSample", ("offset",), _lowMem_defSample),
        ("altcorr_forward", (), _altcorr_forward),
        ("altcorr_backward", (), _altcorr_backward),
    ):
        REGISTERED[_name] = _define(_name, _mut, _fn)
