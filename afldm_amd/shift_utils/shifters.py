"""ImageShifter — surface of reference afldm/shift_utils/shifters.py:135-206 for the filters
the FFHQ harness uses: 'ideal' / 'ideal_crop' (x-ratio periodic-sinc upsample on the HIP
kernel, integer roll, crop mask, stride-ratio slice) and the default bilinear warp used only
for the ground-truth visualisation strip (torch grid_sample, plumbing).  'lanczos' / 'fourier'
need the vendored StyleGAN3 upfirdn2d path, which is out of scope."""
import numpy as np
import torch
import torch.nn.functional as F

from ..af_libs.ideal_lpf import UpsampleRFFT

FILTER_CHOICES = ["bilinear", "lanczos", "ideal", "ideal_crop", "fourier", "fourier_crop"]


def gen_valid_mask(shape, ti, tj):
    _, _, h, w = shape
    ti = int(np.ceil(ti)) if ti >= 0 else int(np.floor(ti))
    tj = int(np.ceil(tj)) if tj >= 0 else int(np.floor(tj))
    mask = torch.ones(shape, dtype=torch.float32)
    if ti >= 0:
        mask[:, :, 0:ti, :] = 0
    else:
        mask[:, :, ti:h, :] = 0
    if tj >= 0:
        mask[:, :, :, 0:tj] = 0
    else:
        mask[:, :, :, tj:w] = 0
    return mask


class ImageShifter:
    def __init__(self, filter=None, upsample_ratio=None):
        filter = "bilinear" if filter is None else filter
        assert filter in FILTER_CHOICES, f"Wrong filter type {filter}"
        if filter in ("lanczos", "fourier", "fourier_crop"):
            raise NotImplementedError(f"ImageShifter('{filter}') is outside the AF-LDM hot path")
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
        # bilinear backward warp by (-ti, -tj), zeros outside, align_corners=True
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                indexing="ij")
        xg = (2 * (xs - tj) / (w - 1) - 1).to(img.device)
        yg = (2 * (ys - ti) / (h - 1) - 1).to(img.device)
        grid = torch.stack([xg, yg], dim=-1)[None].repeat(n, 1, 1, 1).to(img.dtype)
        out = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        mask = ((xg >= -1) & (yg >= -1) & (xg <= 1) & (yg <= 1))[None].repeat(n, 1, 1)
        return out, mask.unsqueeze(1).to(torch.float32)
