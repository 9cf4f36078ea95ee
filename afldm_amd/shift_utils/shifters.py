"""Image-space shifters and samplers — the surface of reference afldm/shift_utils/shifters.py
(`gen_valid_mask` :31-49, `gen_random_offset` :52-76, `fourier_shift_batch` :103-132,
`ImageShifter` :135-264, `get_blur_kernel` :266-276, `upsample_pad_zero` :279-289,
`ImageUpsampler` :292-338, `ImageDownsampler` :341-365) with every filter on MI355X kernels:

  ideal / ideal_crop   x-ratio periodic-sinc upsample as a dense circulant product (afldm_af_resample),
                       integer roll, crop mask, stride-ratio slice
  lanczos              one pad/crop 6-tap FIR pass per axis (afldm_upfirdn2d; af_libs/equivariance.py)
  fourier(_crop)       the phase ramp over fft2 is a complex circulant per axis; its real part is
                       Re(A_h) x Re(A_w)^T - Im(A_h) x Im(A_w)^T (afldm_af_resample_hw)
  bilinear (default)   a constant-offset bilinear warp with zero padding is a 2-tap FIR per axis
                       (afldm_upfirdn2d) - the reference goes through flow_warp / grid_sample
                       (flow_utils.py:34-86)
  blur                 upfirdn2d with the [1,3,3,1] binomial kernel (afldm_upfirdn2d)

`image_random_translate` / `image_latent_random_translate` (training-time augmentation on
flow_warp) are outside the inference path.
"""
from enum import Enum

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..af_libs.equivariance import apply_fractional_translation
from ..af_libs.ideal_lpf import LPF_RFFT, UpsampleRFFT
from ..af_libs.torch_utils.ops.upfirdn2d import upfirdn2d

FILTER_CHOICES = ["bilinear", "lanczos", "ideal", "ideal_crop", "fourier", "fourier_crop"]


def gen_valid_mask(shape, ti, tj):
    _, _, h, w = shape
    ti = int(np.ceil(ti)) if ti >= 0 else int(np.floor(ti))
    tj = int(np.ceil(tj)) if tj >= 0 else int(np.floor(tj))
    # (numpy, not torch: a torch CPU op on a 1 MB tensor fans out over every host core - milliseconds per op on the 256-thread
    #  hosts of the MI355X boxes, and the harness builds ~100 masks per run: 0.4 s of its 0.8 s, profiles/r06/harness_c1.txt)
    mask = np.ones(tuple(shape), dtype=np.float32)
    if ti >= 0:
        mask[:, :, 0:ti, :] = 0
    else:
        mask[:, :, ti:h, :] = 0
    if tj >= 0:
        mask[:, :, :, 0:tj] = 0
    else:
        mask[:, :, :, tj:w] = 0
    return torch.from_numpy(mask)


def gen_random_offset(max_offset_i, max_offset_j, int_offset, int_stride, bs=1, min_offset_i=0, min_offset_j=0):
    """Random (i, j) offsets, integer multiples of int_stride or uniform (shifters.py:52-76)."""
    span_i, span_j = max_offset_i - min_offset_i, max_offset_j - min_offset_j
    if int_offset:
        ri, rj = int(span_i // int_stride), int(span_j // int_stride)
        oi = torch.randint(-ri, ri + 1, (bs,)).to(torch.float32) * int_stride
        oj = torch.randint(-rj, rj + 1, (bs,)).to(torch.float32) * int_stride
    else:
        oi = (torch.rand((bs,)) * 2 - 1) * span_i
        oj = (torch.rand((bs,)) * 2 - 1) * span_j
    return oi + min_offset_i, oj + min_offset_j


def _phase_circulant(N, shift):
    """A = F^-1 diag(exp(-2 pi i shift fftfreq(N))) F as (Re A, Im A) in fp64 (host)."""
    k = np.arange(N)
    p = np.exp(-2j * np.pi * float(shift) * np.fft.fftfreq(N))
    first_col = np.fft.ifft(p)
    A = first_col[(k[:, None] - k[None, :]) % N]
    return A.real, A.imag


def fourier_shift_batch(image, shift_x, shift_y, device="cuda"):
    """Shift by (shift_x along dim 2, shift_y along dim 3) with a phase ramp over the 2-D DFT, real part
    (shifters.py:103-132; square planes: the reference broadcasts a (W, H) grid over [H, W])."""
    if not image.is_cuda:
        raise RuntimeError("afldm_amd fourier_shift_batch runs on MI355X only (no CPU path)")
    n, c, h, w = image.shape
    assert h == w, "square planes only (shifters.py:118 builds the phase grid transposed)"
    dev = image.device
    (hr, hi), (wr, wi) = _phase_circulant(h, shift_x), _phase_circulant(w, shift_y)
    m = [torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32).to(dev) for a in (hr, wr, hi, wi)]
    xh = ops.to_nhwc(image.to(torch.float32).contiguous(), torch.float32)
    y = ops.af_resample_hw(xh, m[0], m[1])
    if np.abs(hi).max() > 0 and np.abs(wi).max() > 0:     # only the Nyquist bin of an even plane is complex
        y = y - ops.af_resample_hw(xh, m[2], m[3])
    return ops.to_nchw(y).to(image.dtype)


def _shift_bilinear(img, ti, tj):
    """out[y, x] = bilinear sample of img at (y - ti, x - tj), zeros outside: per axis the two taps
    frac * in[p - floor(t) - 1] + (1 - frac) * in[p - floor(t)]."""
    def taps(t):
        it = int(np.floor(t))
        ft = float(np.float32(t) - np.float32(it))
        return it, torch.tensor([ft, 1.0 - ft], dtype=torch.float32)

    (ii, fi), (ij, fj) = taps(ti), taps(tj)
    dev = img.device
    y = ops.upfirdn2d(img.contiguous(), fj.reshape(1, 2).to(dev), padx0=ij + 1, padx1=-ij, flip_filter=True)
    return ops.upfirdn2d(y, fi.reshape(2, 1).to(dev), pady0=ii + 1, pady1=-ii, flip_filter=True)


class ImageShifter:
    class BgType(Enum):
        NO_BG = 0
        RANDN = 1
        FULL_COLOR = 2
        ORIGINAL_IMG = 3

    def __init__(self, filter=None, upsample_ratio=None):
        filter = "bilinear" if filter is None else filter
        assert filter in FILTER_CHOICES, f"Wrong filter type {filter}"
        self._filter = filter
        self._cache_key = None
        self._cache_up = None
        if filter in ("ideal", "ideal_crop"):
            assert upsample_ratio is not None
            self.upsample_ratio = upsample_ratio
            self.up_layer = UpsampleRFFT(upsample_ratio)

    def _upsampled(self, img):
        # the reference keys its cache on data_ptr() alone (shifters.py:167), which can return a
        # stale image after the allocator re-uses the block; keying on version + shape avoids that.
        key = (img.data_ptr(), img._version, tuple(img.shape))
        if self.upsample_ratio == 1:
            return img
        if key != self._cache_key:
            self._cache_key, self._cache_up = key, self.up_layer(img)
        return self._cache_up

    def shift(self, img, ti, tj):
        ti, tj = float(ti), float(tj)
        n, _, h, w = img.shape
        if self._filter in ("ideal", "ideal_crop"):
            r = self.upsample_ratio
            si, sj = int(np.round(ti * r)), int(np.round(tj * r))
            warped = torch.roll(self._upsampled(img), shifts=(si, sj), dims=(2, 3))
            if self._filter == "ideal":
                warped = warped[:, :, ::r, ::r]
                return warped, torch.ones_like(warped)
            warped = warped * gen_valid_mask(warped.shape, si, sj).to(warped.device)
            warped = warped[:, :, ::r, ::r]
            return warped, gen_valid_mask(warped.shape, ti, tj).to(warped.device)
        if self._filter == "lanczos":
            warped, mask = apply_fractional_translation(img, tj / w, ti / h)
            return warped, mask[:, 0:1, :, :]
        if self._filter in ("fourier", "fourier_crop"):
            warped = fourier_shift_batch(img, ti, tj, img.device)
            if self._filter == "fourier":
                return warped, torch.ones_like(warped)
            mask = gen_valid_mask(warped.shape, ti, tj).to(warped.device)
            return warped * mask, mask
        # bilinear backward warp by (-ti, -tj), zeros outside (flow_warp with mask, flow_utils.py:34-86):
        # the mask is the in-bounds test of the align_corners grid
        xg = 2 * (torch.arange(w, dtype=torch.float32, device=img.device) - tj) / (w - 1) - 1
        yg = 2 * (torch.arange(h, dtype=torch.float32, device=img.device) - ti) / (h - 1) - 1
        inside = ((yg >= -1) & (yg <= 1))[:, None] & ((xg >= -1) & (xg <= 1))[None, :]
        mask = inside.to(torch.float32)[None, None].repeat(n, 1, 1, 1)
        return _shift_bilinear(img, ti, tj), mask

    def translate_with_occ_bg(self, img, ti, tj, bg_type, mask=None, return_mask=False):
        """Shift and fill the disoccluded region with a background (shifters.py:208-236)."""
        if bg_type == ImageShifter.BgType.RANDN:
            background = torch.randn_like(img)
        elif bg_type == ImageShifter.BgType.FULL_COLOR:
            n, c = img.shape[0:2]
            background = (torch.rand((n, c, 1, 1)) * 2 - 1).to(device=img.device, dtype=img.dtype)
        elif bg_type == ImageShifter.BgType.ORIGINAL_IMG:
            background = img
        elif bg_type != ImageShifter.BgType.NO_BG:
            raise ValueError(f"No such background type {bg_type} in image shifter")
        warped, translate_mask = self.shift(img, ti, tj)
        if mask is None:
            mask = translate_mask
        if bg_type != ImageShifter.BgType.NO_BG:
            warped = warped * mask + background * (1 - mask)
        return (warped, mask) if return_mask else warped


def get_blur_kernel(channels, len=4):
    """Normalised binomial kernel, [channels, channels, len, len] (shifters.py:266-276)."""
    taps = torch.tensor((1, 3, 3, 1) if len == 4 else (1, 3, 6, 3, 1), dtype=torch.float32)
    k2 = torch.outer(taps, taps)
    return (k2 / k2.sum()).reshape(1, 1, len, len).repeat(channels, channels, 1, 1)


def upsample_pad_zero(x, scale):
    """Zero-stuffing up-sample (shifters.py:279-289) as a 1-tap upfirdn2d."""
    one = torch.ones(1, 1, dtype=torch.float32, device=x.device)
    return ops.upfirdn2d(x.contiguous(), one, upx=scale, upy=scale)


class ImageUpsampler:
    def __init__(self, scale=2, mode="nearest", device="cuda"):
        self.scale = scale
        self.mode = mode
        if mode == "ideal":
            self.up = UpsampleRFFT(scale).to(device)
        elif mode == "blur":
            self.blur_kernel = get_blur_kernel(1)[0, 0].to(device)
        elif mode == "learn":
            raise NotImplementedError("ImageUpsampler('learn') is a trainable ConvTranspose2d (training only)")

    def low_pass(self, x):
        if self.mode == "blur":
            return upfirdn2d(x, self.blur_kernel * 4, 2, padding=(2, 1, 2, 1))
        if self.mode == "ideal":
            return self.up.recon_filter(x)
        return F.interpolate(x, scale_factor=self.scale, mode=self.mode)

    def upsample(self, x):
        n, c, h, w = x.shape
        x = x.reshape(n * c, 1, h, w)
        if self.mode == "blur":
            x = upfirdn2d(x, self.blur_kernel * self.scale ** 2, self.scale, padding=(2, 1, 2, 1))
        elif self.mode == "ideal":
            x = self.up(x)
        else:
            x = F.interpolate(x, scale_factor=self.scale, mode=self.mode)
        return x.reshape(n, c, h * self.scale, w * self.scale)


class ImageDownsampler:
    def __init__(self, scale=2, mode="nearest", device="cuda"):
        self.scale = scale
        self.mode = mode
        if mode == "ideal":
            self.low_pass = LPF_RFFT(scale).to(device)     # cutoff = scale, as the reference passes it (:348)
        elif mode == "blur":
            self.blur_kernel = get_blur_kernel(1)[0, 0].to(device)

    def downsample(self, x):
        n, c, h, w = x.shape
        x = x.reshape(n * c, 1, h, w)
        if self.mode == "blur":
            # `scale` lands in upfirdn2d's `up` slot exactly as in the reference (:357), so the plane comes
            # back at the input size and the reshape below raises there too
            x = upfirdn2d(x, self.blur_kernel, self.scale, padding=(2, 1, 2, 1))[:, :, ::2, ::2]
        elif self.mode == "ideal":
            x = self.low_pass(x)[:, :, ::2, ::2]
        else:
            x = F.interpolate(x, scale_factor=1 / self.scale, mode=self.mode)
        return x.reshape(n, c, h // self.scale, w // self.scale)
