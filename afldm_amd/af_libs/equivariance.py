"""Translation operators of reference afldm/af_libs/equivariance.py (StyleGAN3's EQ-T helpers):
`sinc` :24-27, `lanczos_window` :30-32, `apply_integer_translation` :49-63 and
`apply_fractional_translation` :70-103 — the ones ImageShifter('lanczos') and
flow_warp_with_occ_bg reach (shifters.py:158-161, flow_utils.py:103-108).

The reference filters a padded copy, crops it and pastes the crop into a zero image.  Pixels
outside the image are zero in upfirdn2d, so the same result is ONE pad/crop FIR pass per axis
straight into the final W x H frame: the 2a-tap Lanczos filter with padding [a + i, a - 1 - i]
(i = integer part of the shift; negative padding crops), and for the integer translation a 1-tap
filter with padding [i, -i].  Both run on the HIP kernel (csrc/fir.hip).  The tap weights are
host arithmetic, evaluated exactly like the reference's.  The generator-level metrics
(`compute_equivariance_metrics` :186-289) and the rotation operators need a StyleGAN3 generator
and are out of scope.
"""
import math

import torch

from .. import ops


def sinc(x):
    """sin(pi x) / (pi x) with the reference's 1e-30 guard."""
    y = (x * math.pi).abs()
    return torch.where(y < 1e-30, torch.ones_like(x), torch.sin(y) / y.clamp(1e-30, float("inf")))


def lanczos_window(x, a):
    r = x.abs() / a
    return torch.where(r < 1, sinc(r), torch.zeros_like(r))


def _pixels(t, extent):
    """shift in pixels as the reference computes it: fp32 product of the normalised shift and the extent"""
    return torch.as_tensor(t * extent).to(torch.float32)


def _box_mask(x, y0, y1, x0, x1):
    """1 inside rows [y0, y1) x columns [x0, x1), 0 elsewhere (empty when a range is empty)."""
    H, W = x.shape[-2:]
    rows = torch.arange(H, device=x.device)
    cols = torch.arange(W, device=x.device)
    box = ((rows >= y0) & (rows < y1))[:, None] & ((cols >= x0) & (cols < x1))[None, :]
    return box.to(x.dtype).expand_as(x).contiguous()


def _fir(x, fx, fy, padx, pady):
    """one FIR pass along W (fx, 1 x taps) and one along H (fy, taps x 1) with explicit [before, after] padding"""
    dev = x.device
    y = ops.upfirdn2d(x.contiguous(), fx.reshape(1, -1).to(dev).contiguous(), padx0=padx[0], padx1=padx[1])
    return ops.upfirdn2d(y, fy.reshape(-1, 1).to(dev).contiguous(), pady0=pady[0], pady1=pady[1])


def apply_integer_translation(x, tx, ty):
    """T_x of StyleGAN3 Appendix E.1 (tx, ty in units of the image extent) -> (image, mask)."""
    H, W = x.shape[-2:]
    ix, iy = int(_pixels(tx, W).round()), int(_pixels(ty, H).round())
    if abs(ix) >= W or abs(iy) >= H:
        return torch.zeros_like(x), torch.zeros_like(x)
    one = torch.ones(1, dtype=torch.float32)
    z = _fir(x, one, one, (ix, -ix), (iy, -iy))
    return z, _box_mask(x, max(iy, 0), H + min(iy, 0), max(ix, 0), W + min(ix, 0))


def apply_fractional_translation(x, tx, ty, a=3):
    """T_x of StyleGAN3 Appendix E.2: separable 2a-tap Lanczos shift -> (image, validity mask)."""
    H, W = x.shape[-2:]
    px, py = _pixels(tx, W), _pixels(ty, H)
    ix, iy = px.floor().to(torch.int64), py.floor().to(torch.int64)
    fx, fy = px - ix, py - iy
    ix, iy = int(ix), int(iy)
    b = a - 1
    taps = torch.arange(a * 2) - b
    wx = sinc(taps - fx) * sinc((taps - fx) / a)
    wy = sinc(taps - fy) * sinc((taps - fy) / a)
    # true convolution (the kernel flips the taps), output frame = input frame
    z = _fir(x, wx / wx.sum(), wy / wy.sum(), (a + ix, b - ix), (a + iy, b - iy))
    m = _box_mask(x, max(iy + a, 0), min(iy - b, 0) + H, max(ix + a, 0), min(ix - b, 0) + W)
    return z, m
