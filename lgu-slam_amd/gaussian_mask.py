"""Learnable per-pixel 2-D Gaussian mask on the correlation volume: this build's
counterpart of the reference's droid_slam/gaussianMask_cuda.py (same class names,
constructor and forward signatures, same parameter names so state dicts load)."""
import math

import torch
import torch.nn as nn

from . import ops


class GaussianMaskCuda(torch.autograd.Function):
    """reference gaussianMask_cuda.py:7-23: grads for mean and cov only."""

    @staticmethod
    def forward(ctx, mean, cov, corr, radius):
        mean = mean.float()
        cov = cov.float()
        ctx.save_for_backward(mean, cov, corr)
        ctx.radius = radius
        corr1, = ops.gaussianMask(mean, cov, corr, radius)
        return corr1

    @staticmethod
    def backward(ctx, grad_output):
        mean, cov, corr = ctx.saved_tensors
        means_grad, covs_grad = ops.gaussianMask_backward(mean, cov, corr, grad_output.contiguous(), ctx.radius)
        return means_grad, covs_grad, None, None


def per_Corr_Normalization(x, normalIndex, eps=1e-5):
    """(b, h*w, 2) standardised over `normalIndex` per sample, biased variance
    (reference gaussianMask_cuda.py:26-33)."""
    mean = x.mean(dim=normalIndex, keepdim=True)
    std = torch.sqrt(x.var(dim=normalIndex, unbiased=False, keepdim=True) + eps)
    return (x - mean) / std


FUSED_PARAMS = True   # False: the torch composition of gaussian_parameters also in inference (A/B and tests)


class GaussianMask(nn.Module):
    """Predicts a mean (grid + learned shift) and a diagonal covariance per source pixel
    from the concatenated feature pair and re-weights the volume with it
    (reference gaussianMask_cuda.py:35-88)."""

    RADIUS = 4  # window radius hard-wired at the reference call site (:84)

    def __init__(self, h, w):
        super().__init__()
        self.meanMap = nn.Linear(16, 2)
        self.covMap = nn.Linear(16, 2)
        self.map = nn.Linear(256, 16)
        self.cov = torch.eye(2)
        # reference initialisation (:43-56): zero mean head, He-style normal elsewhere
        nn.init.zeros_(self.meanMap.weight); nn.init.zeros_(self.meanMap.bias)
        nn.init.normal_(self.covMap.weight, 0, math.sqrt(2.0 / self.covMap.out_features)); nn.init.zeros_(self.covMap.bias)
        nn.init.normal_(self.map.weight, 0, math.sqrt(2.0 / self.map.out_features)); nn.init.zeros_(self.map.bias)
        self.mapA = nn.Sequential(self.map, nn.Tanh())
        ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        self.coord = torch.stack([xs, ys], dim=-1).view(h, w, 2)  # (x, y) of every source pixel

    def gaussian_parameters(self, x):
        """mean (b,h,w,2), cov (b,h,w,2) float32 contiguous, det (b,h*w) from the feature pair
        x (b,h,w,256) — everything in forward() up to the kernel call (:66-83)."""
        b, h, w, _ = x.shape
        tt = self.mapA(x)
        mean_ofs = self.meanMap(tt).view(b, h, w, 2)
        cov_raw = self.covMap(tt).view(b, h * w, 2)
        if (FUSED_PARAMS and cov_raw.is_cuda and cov_raw.dtype in (torch.float32, torch.float16)
                and mean_ofs.dtype == cov_raw.dtype and not (torch.is_grad_enabled() and cov_raw.requires_grad)
                and self.coord.shape[:2] == (h, w)):
            # inference: everything below in one launch, with torch's arithmetic (incl. the half roundings under autocast)
            return ops.gaussian_params(mean_ofs.contiguous(), cov_raw.contiguous(), h, w)
        cov = per_Corr_Normalization(cov_raw, [1, 2])
        cov = torch.sigmoid(cov) * 5 + 0.05
        det = cov[:, :, 0] * cov[:, :, 1]
        cov = cov.view(b, h, w, 2).float()
        mean = self.coord.to(device=cov.device, dtype=cov.dtype).expand(b, h, w, 2) + mean_ofs
        return mean.contiguous(), cov.contiguous(), det

    def forward(self, x, corr):
        b, h, w, _ = x.shape
        mean, cov, det = self.gaussian_parameters(x)
        corr1 = GaussianMaskCuda.apply(mean, cov, corr, self.RADIUS)
        corr1 = corr1 / (6.28 * torch.sqrt(det).view(b, h, w, 1, 1)) + corr
        return corr1, mean, det
