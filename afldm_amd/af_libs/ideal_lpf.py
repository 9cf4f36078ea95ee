"""Ideal (brick-wall) low-pass / reconstruction filters — the module surface of reference
afldm/af_libs/ideal_lpf.py (LPF_RFFT :52-93, LPF_RECON_RFFT :96-134, UpsampleRFFT :137-158,
subpixel_shift :161-172, mask builders :12-49) executed as dense separable circulant products
on MI355X (afldm_af_resample) instead of rfft2 -> mask -> irfft2.

Public tensors are NCHW like the reference; square planes only (the reference builds its mask
from the width alone, ideal_lpf.py:80).  cutoff is restricted to what the hot path uses
(LPF: 1/2; recon: 1/up) — other values raise instead of silently differing.
"""
import torch
import torch.nn as nn

from .. import _lib, ops


def _rect_1d(N, cutoff, nyquist):
    lo = int((N * cutoff) // 2)
    hi = int(N - lo)
    r = torch.ones(N)
    r[lo + 1:hi] = 0
    if N % 4 == 0:
        r[lo] = nyquist     # raises IndexError when lo == 0, exactly like the reference
        r[hi] = nyquist
    return r


def create_lpf_rect(N, cutoff=0.5):
    r = _rect_1d(N, cutoff, 0.0)
    return r[:, None] * r[None, :]


def create_recon_rect(N, cutoff=0.5):
    r = _rect_1d(N, cutoff, 0.5)
    return r[:, None] * r[None, :]


def create_fixed_lpf_rect(N, size):
    r = torch.ones(N)
    if size < N:
        lo = size // 2
        r[lo + 1:int(N - lo)] = 0
    return r[:, None] * r[None, :]


def _plane_op(x, M):
    """y = M x M^T on every (b, c) plane of an NCHW CUDA tensor."""
    if not x.is_cuda:
        raise RuntimeError("afldm_amd ideal filters run on MI355X only (no CPU path)")
    assert x.ndim == 4 and x.shape[-1] == x.shape[-2], "square planes only (reference ideal_lpf.py:80)"
    dtype = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
    xh = ops.to_nhwc(x.to(torch.float32).contiguous(), dtype)
    return ops.to_nchw(ops.af_resample(xh, M)).to(x.dtype)


class LPF_RFFT(nn.Module):
    def __init__(self, cutoff=0.5, transform_mode="rfft", fixed_size=None):
        super().__init__()
        assert transform_mode in ("fft", "rfft"), f"transform_mode={transform_mode} is not supported"
        if cutoff != 0.5 or fixed_size is not None:
            raise NotImplementedError("afldm_amd.LPF_RFFT implements cutoff=1/2 (the only value on the AF-LDM path)")
        self.cutoff = cutoff
        self.fixed_size = fixed_size
        self.transform_mode = transform_mode
        self.rect_dict = {}

    def forward(self, x):
        return _plane_op(x, ops.lpf_matrix(x.shape[-1], x.device))


class LPF_RECON_RFFT(nn.Module):
    """Recon filter on an ALREADY zero-stuffed tensor is not separately exposed on the HIP path:
    use UpsampleRFFT (zero-stuffing + recon + gain fused into one matrix)."""

    def __init__(self, cutoff=0.5, transform_mode="rfft"):
        super().__init__()
        self.cutoff = cutoff
        self.transform_mode = transform_mode
        self.rect_dict = {}

    def forward(self, x):
        raise NotImplementedError("use UpsampleRFFT: the zero-stuffed intermediate never exists on the HIP path")


class UpsampleRFFT(nn.Module):
    def __init__(self, up=2, transform_mode="rfft", factor=1):
        super().__init__()
        if factor != 1:
            raise NotImplementedError("factor != 1 is not used on the AF-LDM path")
        self.up = up
        self.recon_filter = LPF_RECON_RFFT(cutoff=1 / up * factor, transform_mode=transform_mode)

    def forward(self, x):
        return _plane_op(x, ops.up_matrix(x.shape[-1], self.up, x.device))


def subpixel_shift(images, up=2, shift_x=1, shift_y=1, up_method="ideal"):
    assert up_method == "ideal", 'Only "ideal" interpolation kenrel is supported'
    u = UpsampleRFFT(up=up)(images)
    return torch.roll(u, shifts=(-shift_x, -shift_y), dims=(2, 3))[:, :, ::up, ::up]
