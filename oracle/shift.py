"""Oracle: ImageShifter ('ideal', 'ideal_crop', default bilinear), valid masks and the
masked metrics.  Restated from reference afldm/shift_utils/shifters.py:31-49,142-206,
afldm/shift_utils/flow_utils.py:34-86 and afldm/shift_utils/metrics.py:5-20; pinned
against the imported reference by oracle/gen_golden.py.  Test infrastructure."""
import numpy as np
import torch
import torch.nn.functional as F

from .ideal_filters import upsample_rfft


def gen_valid_mask(shape, ti, tj):
    """shifters.py:31-49."""
    _, _, h, w = shape
    if ti >= 0:
        ti = int(np.ceil(ti)); i1, i2 = 0, ti
    else:
        ti = int(np.floor(ti)); i1, i2 = ti, h
    if tj >= 0:
        tj = int(np.ceil(tj)); j1, j2 = 0, tj
    else:
        tj = int(np.floor(tj)); j1, j2 = tj, w
    mask = torch.ones(shape, dtype=torch.float32)
    mask[:, :, i1:i2, :] = 0
    mask[:, :, :, j1:j2] = 0
    return mask


def shift_ideal(img, ti, tj, ratio, crop):
    """ImageShifter('ideal'|'ideal_crop', ratio).shift (shifters.py:163-191)."""
    ti, tj = float(ti), float(tj)
    up = img if ratio == 1 else upsample_rfft(img, ratio)
    si = int(np.round(ti * ratio))
    sj = int(np.round(tj * ratio))
    warped = torch.roll(up, shifts=(si, sj), dims=(2, 3))
    if not crop:
        warped = warped[:, :, ::ratio, ::ratio]
        return warped, torch.ones_like(warped)
    warped = warped * gen_valid_mask(warped.shape, si, sj)
    warped = warped[:, :, ::ratio, ::ratio]
    return warped, gen_valid_mask(warped.shape, ti, tj)


def shift_bilinear(img, ti, tj):
    """ImageShifter().shift default branch (shifters.py:200-205) via flow_warp
    (flow_utils.py:79-86) -> grid_sample(align_corners=True, zeros)."""
    ti, tj = float(ti), float(tj)
    n, _, h, w = img.shape
    flow = torch.tensor([-ti, -tj]).reshape(1, 2, 1, 1).repeat(n, 1, h, w)
    flow = torch.flip(flow, (1,))
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack([x, y], dim=0).float()[None].repeat(n, 1, 1, 1) + flow
    grid = grid.to(img.dtype)
    xg = 2 * grid[:, 0] / (w - 1) - 1
    yg = 2 * grid[:, 1] / (h - 1) - 1
    out = F.grid_sample(img, torch.stack([xg, yg], dim=-1), mode="bilinear",
                        padding_mode="zeros", align_corners=True)
    mask = (xg >= -1) & (yg >= -1) & (xg <= 1) & (yg <= 1)
    return out, mask.unsqueeze(1).to(torch.float32)


def mask_mse(a, b, mask):
    """metrics.py:5-8."""
    loss = (a * mask - b * mask).square().sum((1, 2, 3)) / mask.sum((1, 2, 3))
    return loss.mean()


def mask_psnr(a, b, mask):
    """metrics.py:11-15."""
    a_, b_ = a * mask, b * mask
    i_max = torch.max(a_.max(), b_.max()) - torch.min(a_.min(), b_.min())
    return 10 * torch.log10(i_max * i_max / mask_mse(a, b, mask))


def psnr(a, b, i_max=None):
    """metrics.py:17-20."""
    if i_max is None:
        i_max = torch.max(a.max(), b.max()) - torch.min(a.min(), b.min())
    return 10 * torch.log10(i_max * i_max / F.mse_loss(a, b))
