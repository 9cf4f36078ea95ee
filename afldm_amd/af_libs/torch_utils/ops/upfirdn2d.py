"""upfirdn2d — the module surface of reference afldm/af_libs/torch_utils/ops/upfirdn2d.py
(`setup_filter` :72-118, `upfirdn2d` :122-140, `filter2d` :273-309, `upsample2d` :313-347,
`downsample2d` :351-387) on the hand-written HIP kernel (csrc/fir.hip, afldm_upfirdn2d) instead of
the vendored CUDA plugin / its conv2d fallback.

Same names, argument meaning and assertions as the reference.  `impl` is accepted for source
compatibility: both 'ref' and 'cuda' run the HIP kernel (there is no CPU path — CPU tensors raise).
Forward only: the reference's custom autograd (another upfirdn2d with the flipped filter,
upfirdn2d.py:216-236) belongs to training, which is out of scope.
"""
import numpy as np
import torch

from .... import ops

_DEVICE_FILTERS = {}


def _ints(v, n, what):
    if isinstance(v, int):
        v = [v] * min(n, 2)
    assert isinstance(v, (list, tuple)) and all(isinstance(e, int) for e in v), f"{what} must be int(s)"
    return list(v)


def _parse_scaling(scaling):
    """int | [x, y] -> (sx, sy), both >= 1"""
    sx, sy = _ints(scaling, 2, "scaling")
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    """int | [x, y] | [x_before, x_after, y_before, y_after] -> the 4-tuple (negative = crop)"""
    p = _ints(padding, 4, "padding")
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    padx0, padx1, pady0, pady1 = p
    return padx0, padx1, pady0, pady1


def _get_filter_size(f):
    """(fw, fh) of a [fh, fw] / [taps] filter, (1, 1) for None"""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device("cpu"), normalize=True, flip_filter=False, gain=1, separable=None):
    """Convenience function to set up a FIR filter for upfirdn2d() (reference :72-118)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2]
    assert f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Pad, upsample, filter, and downsample a batch of 2D images (reference :122-194).
    x: [batch, channels, H, W] CUDA tensor, fp32 or bf16; f: float32 [fh, fw], [taps] (separable) or None."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert impl in ["ref", "cuda"]
    if not x.is_cuda:
        raise RuntimeError("afldm_amd upfirdn2d runs on MI355X only (no CPU path)")
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    assert f.dtype == torch.float32 and not f.requires_grad
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    _, _, in_height, in_width = x.shape
    # reference :159-162: the up-sampled buffer must not be smaller than the filter (either axis, either rank)
    upW = in_width * upx + padx0 + padx1
    upH = in_height * upy + pady0 + pady1
    assert upW >= f.shape[-1] and upH >= f.shape[0]
    f = f.to(x.device)
    xc = x if x.dtype in (torch.float32, torch.bfloat16) else x.to(torch.float32)
    xc = xc.contiguous()
    if f.ndim == 2:
        y = ops.upfirdn2d(xc, f.contiguous(), upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
    else:
        # separable: each pass carries sqrt(gain) (`gain ** (f.ndim / 2)`, reference :173); W first, then H
        g = gain ** 0.5
        y = ops.upfirdn2d(xc, f.unsqueeze(0).contiguous(), upx, 1, downx, 1, padx0, padx1, 0, 0, flip_filter, g)
        y = ops.upfirdn2d(y, f.unsqueeze(1).contiguous(), 1, upy, 1, downy, 0, 0, pady0, pady1, flip_filter, g)
    return y if y.dtype == x.dtype else y.to(x.dtype)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Filter a batch of 2D images; by default the result keeps the input shape (reference :273-309)."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Upsample a batch of 2D images with the given filter (reference :313-347)."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Downsample a batch of 2D images with the given filter (reference :351-387)."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2,
         pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
