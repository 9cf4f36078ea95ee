"""Masked equivariance metrics — reference afldm/shift_utils/metrics.py:5-20.  A handful of
reductions over one image per call: plain tensor plumbing, not a hot-path kernel."""
import torch
import torch.nn.functional as F


def mask_mse(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor):
    per_sample = (a * mask - b * mask).square().sum((1, 2, 3)) / mask.sum((1, 2, 3))
    return per_sample.mean()


def mask_psnr(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor):
    am, bm = a * mask, b * mask
    rng = torch.max(am.max(), bm.max()) - torch.min(am.min(), bm.min())
    return 10 * torch.log10(rng * rng / mask_mse(a, b, mask))


def psnr(a: torch.Tensor, b: torch.Tensor, i_max=None):
    if i_max is None:
        i_max = torch.max(a.max(), b.max()) - torch.min(a.min(), b.min())
    return 10 * torch.log10(i_max * i_max / F.mse_loss(a, b))
