"""Ideal (brick-wall) low-pass / reconstruction filters — the module surface of reference
afldm/af_libs/ideal_lpf.py (LPF_RFFT :52-93, LPF_RECON_RFFT :96-134, UpsampleRFFT :137-158,
subpixel_shift :161-172, mask builders :12-49) executed as dense separable circulant products
on MI355X (afldm_af_resample) instead of rfft2 -> mask -> irfft2.

Public tensors are NCHW like the reference; square planes only (the reference builds its mask
from the width alone, ideal_lpf.py:80).  The UNet's own filters (cutoff 1/2, up 2 / 8) take their
matrices from the C library (afldm_filter_matrix); any other cutoff / up / factor builds the
circulant of the reference's 1-D mask on the host in fp64 (`_circulant`).
"""
import numpy as np
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


_MATRIX_CACHE = {}


def _circulant(r):
    """Real circulant C = F^-1 diag(r) F of a 1-D frequency mask (fp64 host arithmetic).  The masks are
    even (r[k] == r[N - k]), so rfft2 -> mask -> irfft2 with the separable mask r r^T is exactly
    C x C^T on every plane."""
    r = np.asarray(r, dtype=np.float64)
    N = r.shape[0]
    k = np.arange(N)
    first_col = (r[None, :] * np.cos(2 * np.pi * np.outer(k, k) / N)).sum(1) / N      # Re(ifft(r))
    return first_col[(k[:, None] - k[None, :]) % N]


def _device_matrix(key, build, device):
    key = key + (str(device),)
    if key not in _MATRIX_CACHE:
        _MATRIX_CACHE[key] = torch.from_numpy(np.ascontiguousarray(build())).to(torch.float32).to(device)
    return _MATRIX_CACHE[key]


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
        self.cutoff = cutoff
        self.fixed_size = fixed_size          # stored and, like the reference's forward (:69-93), never used
        self.transform_mode = transform_mode
        self.rect_dict = {}

    def forward(self, x):
        if self.transform_mode == "fft":
            # the reference's 'fft' inverse is a one-argument lambda called with s= (ideal_lpf.py:67-68,91)
            raise TypeError("LPF_RFFT(transform_mode='fft'): itransform() got an unexpected keyword argument 's' "
                            "(same failure as the reference)")
        N = x.shape[-1]
        if self.cutoff == 0.5:
            return _plane_op(x, ops.lpf_matrix(N, x.device))
        M = _device_matrix(("lpf", N, float(self.cutoff)), lambda: _circulant(_rect_1d(N, self.cutoff, 0.0).numpy()),
                           x.device)
        return _plane_op(x, M)


class LPF_RECON_RFFT(nn.Module):
    """Reconstruction filter on an already zero-stuffed plane (ideal_lpf.py:96-134); the inverse
    transform is the same in both transform modes up to rounding.  Even planes only (the reference's
    irfft2 without `s` returns an even-width plane)."""

    def __init__(self, cutoff=0.5, transform_mode="rfft"):
        super().__init__()
        assert transform_mode in ("fft", "rfft"), f"mode={transform_mode} is not supported"
        self.cutoff = cutoff
        self.transform_mode = transform_mode
        self.rect_dict = {}

    def matrix(self, N, device):
        return _device_matrix(("recon", N, float(self.cutoff)),
                              lambda: _circulant(_rect_1d(N, self.cutoff, 0.5).numpy()), device)

    def forward(self, x):
        N = x.shape[-1]
        if N % 2:
            raise NotImplementedError("LPF_RECON_RFFT on an odd plane changes its width in the reference (irfft2)")
        return _plane_op(x, self.matrix(N, x.device))


class UpsampleRFFT(nn.Module):
    def __init__(self, up=2, transform_mode="rfft", factor=1):
        super().__init__()
        self.up = up
        self.factor = factor
        self.recon_filter = LPF_RECON_RFFT(cutoff=1 / up * factor, transform_mode=transform_mode)

    def forward(self, x):
        N, up = x.shape[-1], self.up
        if self.factor == 1:
            return _plane_op(x, ops.up_matrix(N, up, x.device))
        # zero-stuffing keeps every up-th column of the recon circulant; gain up per axis (ideal_lpf.py:148-158)
        cutoff = self.recon_filter.cutoff
        M = _device_matrix(("up", N, up, float(cutoff)),
                           lambda: up * _circulant(_rect_1d(N * up, cutoff, 0.5).numpy())[:, ::up], x.device)
        return _plane_op(x, M)


def subpixel_shift(images, up=2, shift_x=1, shift_y=1, up_method="ideal"):
    assert up_method == "ideal", 'Only "ideal" interpolation kenrel is supported'
    u = UpsampleRFFT(up=up)(images)
    return torch.roll(u, shifts=(-shift_x, -shift_y), dims=(2, 3))[:, :, ::up, ::up]
