"""Masked equivariance metrics — the surface of reference afldm/shift_utils/metrics.py
(`mask_mse` :5-8, `mask_psnr` :11-15, `psnr` :17-20).

Tensors on the MI355X go through ONE reduction kernel (afldm_masked_metrics: per-sample squared error
under the mask, mask weight and the value range of both masked images in a single pass; the reference
launches ~10 elementwise / reduction kernels per call).  Host tensors (fixtures, file-based evaluation)
are reduced with torch in the reference's order.
"""
import torch

from .. import ops


def _stats(a, b, mask):
    """per-sample [B, 6] = (sum ((a-b) m)^2, sum m, max a m, min a m, max b m, min b m)"""
    if a.is_cuda:
        return ops.masked_metrics(a, b, mask)
    am, bm = a * mask, b * mask
    dims = tuple(range(1, a.ndim))
    # denominator: the mask summed over ITS OWN shape (metrics.py:6-7) - a [B,1,H,W] mask counts a pixel once
    msum = (mask.sum(tuple(range(1, mask.ndim))) if mask.ndim == a.ndim else mask.expand_as(a).sum(dims)).expand(a.shape[0])
    cols = [(am - bm).square().sum(dims), msum,
            am.amax(dims), am.amin(dims), bm.amax(dims), bm.amin(dims)]
    return torch.stack([c.to(torch.float32) for c in cols], 1)


def _mse_from(st):
    return (st[:, 0] / st[:, 1]).mean()


def _range_from(st):
    return torch.maximum(st[:, 2].max(), st[:, 4].max()) - torch.minimum(st[:, 3].min(), st[:, 5].min())


def mask_mse(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor):
    """mean over the batch of  sum ((a - b) mask)^2 / sum mask"""
    return _mse_from(_stats(a, b, mask))


def mask_psnr(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor):
    """PSNR of the masked pair, peak = value range of the two masked images"""
    st = _stats(a, b, mask)
    rng = _range_from(st)
    return 10 * torch.log10(rng * rng / _mse_from(st))


def psnr(a: torch.Tensor, b: torch.Tensor, i_max=None):
    """plain PSNR; i_max defaults to the joint value range"""
    ones = torch.ones((), dtype=torch.float32, device=a.device)
    st = _stats(a.reshape(1, -1), b.reshape(1, -1), ones)
    if i_max is None:
        i_max = _range_from(st)
    return 10 * torch.log10(i_max * i_max / (st[0, 0] / st[0, 1]))
