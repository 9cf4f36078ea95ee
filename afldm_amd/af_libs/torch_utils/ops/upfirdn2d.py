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
    """FIR taps for upfirdn2d() with the reference's conventions (upfirdn2d.py:72-118): a scalar or None is the 1-tap
    identity; 1-D taps stay separable from 8 taps on (or when asked), otherwise they become the outer-product 2-D
    filter; optional unit DC gain, optional flip, and `gain` spread evenly over the filter's axes."""
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert 0 <= taps.ndim <= 2 and taps.numel() > 0
    taps = taps.reshape(1) if taps.ndim == 0 else taps
    keep_1d = taps.ndim == 1 and (taps.numel() >= 8 if separable is None else bool(separable))
    if taps.ndim == 1 and not keep_1d:
        taps = torch.outer(taps, taps)
    assert taps.ndim == (1 if (separable if separable is not None else keep_1d) else 2)
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = taps.flip(tuple(range(taps.ndim)))
    return (taps * gain ** (taps.ndim / 2)).to(device=device)


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


def _centred(x, f, up, down, padding, flip_filter, gain, impl, lead, trail):
    """upfirdn2d with the filter's footprint absorbed into the padding: `lead(taps, u, d)` / `trail(taps, u, d)` give
    the extra samples in front of / behind each axis - the three conventions of the reference's wrappers
    (upfirdn2d.py:273-387) differ only there."""
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pad = [px0 + lead(fw, upx, downx), px1 + trail(fw, upx, downx), py0 + lead(fh, upy, downy), py1 + trail(fh, upy, downy)]
    return upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip_filter, gain=gain, impl=impl)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Filter a batch of 2D images; by default the result keeps the input shape (reference :273-309)."""
    return _centred(x, f, 1, 1, padding, flip_filter, gain, impl,
                    lambda n, u, d: n // 2, lambda n, u, d: (n - 1) // 2)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Upsample a batch of 2D images with the given filter: output = input x up, gain compensated for the
    zero-stuffing (reference :313-347)."""
    upx, upy = _parse_scaling(up)
    return _centred(x, f, up, 1, padding, flip_filter, gain * upx * upy, impl,
                    lambda n, u, d: (n + u - 1) // 2, lambda n, u, d: (n - u) // 2)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Downsample a batch of 2D images with the given filter: output = input / down (reference :351-387)."""
    return _centred(x, f, 1, down, padding, flip_filter, gain, impl,
                    lambda n, u, d: (n - d + 1) // 2, lambda n, u, d: (n - d) // 2)
