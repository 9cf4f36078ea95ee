"""Oracle: upfirdn2d and what the reference builds on it — test infrastructure (CPU, torch fp32).

Restated from the reference's PyTorch fallback of its vendored StyleGAN3 operator
(afldm/af_libs/torch_utils/ops/upfirdn2d.py: `_parse_*` :35-68, `setup_filter` :72-118,
`_upfirdn2d_ref` :144-194, `filter2d` :273-309, `upsample2d` :313-347, `downsample2d` :351-387),
the translation operators of afldm/af_libs/equivariance.py (`sinc` :24-27,
`apply_integer_translation` :49-63, `apply_fractional_translation` :70-103) and the image-space
helpers of afldm/shift_utils/shifters.py (`fourier_shift_batch` :103-132, `get_blur_kernel` :266-276,
`ImageUpsampler` :292-338, `ImageDownsampler` :341-365).  Pinned against outputs of the imported
reference by oracle/gen_golden.py part e (tests/golden/g10_upfirdn.npz).  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this module; the product never does.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .ideal_filters import lpf_recon_rfft, lpf_rfft, upsample_rfft


# ----------------------------------------------------------------------------- argument parsing
def parse_scaling(scaling):
    """upfirdn2d.py:35-42."""
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and all(isinstance(v, int) for v in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def parse_padding(padding):
    """upfirdn2d.py:44-53: int | [x, y] | [x0, x1, y0, y1]."""
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    return tuple(padding)


def filter_size(f):
    """upfirdn2d.py:55-66 -> (fw, fh)."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, normalize=True, flip_filter=False, gain=1, separable=None):
    """upfirdn2d.py:72-118."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in (0, 1, 2) and f.numel() > 0
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


# ----------------------------------------------------------------------------- the operator
def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """`_upfirdn2d_ref` (upfirdn2d.py:144-194): zero-stuff, pad / crop, grouped conv, decimate."""
    assert x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    assert f.ndim in (1, 2) and f.dtype == torch.float32
    B, C, H, W = x.shape
    upx, upy = parse_scaling(up)
    downx, downy = parse_scaling(down)
    padx0, padx1, pady0, pady1 = parse_padding(padding)
    assert W * upx + padx0 + padx1 >= f.shape[-1] and H * upy + pady0 + pady1 >= f.shape[0]
    x = x.reshape(B, C, H, 1, W, 1)
    x = F.pad(x, [0, upx - 1, 0, 0, 0, upy - 1])
    x = x.reshape(B, C, H * upy, W * upx)
    x = F.pad(x, [max(padx0, 0), max(padx1, 0), max(pady0, 0), max(pady1, 0)])
    x = x[:, :, max(-pady0, 0):x.shape[2] - max(-pady1, 0), max(-padx0, 0):x.shape[3] - max(-padx1, 0)]
    f = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f[None, None].repeat([C, 1] + [1] * f.ndim)
    if f.ndim == 4:
        x = F.conv2d(x, f, groups=C)
    else:
        x = F.conv2d(x, f.unsqueeze(2), groups=C)
        x = F.conv2d(x, f.unsqueeze(3), groups=C)
    return x[:, :, ::downy, ::downx]


def filter2d(x, f, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:273-309: same-size FIR."""
    padx0, padx1, pady0, pady1 = parse_padding(padding)
    fw, fh = filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:313-347."""
    upx, upy = parse_scaling(up)
    padx0, padx1, pady0, pady1 = parse_padding(padding)
    fw, fh = filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:351-387."""
    downx, downy = parse_scaling(down)
    padx0, padx1, pady0, pady1 = parse_padding(padding)
    fw, fh = filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2,
         pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


# ----------------------------------------------------------------------------- translations (equivariance.py)
def sinc(x):
    """equivariance.py:24-27."""
    y = (x * np.pi).abs()
    z = torch.sin(y) / y.clamp(1e-30, float("inf"))
    return torch.where(y < 1e-30, torch.ones_like(x), z)


def lanczos_window(x, a):
    """equivariance.py:30-32."""
    x = x.abs() / a
    return torch.where(x < 1, sinc(x), torch.zeros_like(x))


def apply_integer_translation(x, tx, ty):
    """equivariance.py:49-63 (tx, ty in units of the image extent)."""
    _N, _C, H, W = x.shape
    ix = int(torch.as_tensor(tx * W).to(torch.float32).round())
    iy = int(torch.as_tensor(ty * H).to(torch.float32).round())
    z, m = torch.zeros_like(x), torch.zeros_like(x)
    if abs(ix) < W and abs(iy) < H:
        y = x[:, :, max(-iy, 0):H + min(-iy, 0), max(-ix, 0):W + min(-ix, 0)]
        z[:, :, max(iy, 0):H + min(iy, 0), max(ix, 0):W + min(ix, 0)] = y
        m[:, :, max(iy, 0):H + min(iy, 0), max(ix, 0):W + min(ix, 0)] = 1
    return z, m


def apply_fractional_translation(x, tx, ty, a=3):
    """equivariance.py:70-103: separable 2a-tap Lanczos shift + validity mask."""
    _N, _C, H, W = x.shape
    tx = torch.as_tensor(tx * W).to(torch.float32)
    ty = torch.as_tensor(ty * H).to(torch.float32)
    ix, iy = tx.floor().to(torch.int64), ty.floor().to(torch.int64)
    fx, fy = tx - ix, ty - iy
    ix, iy = int(ix), int(iy)
    b = a - 1
    z = torch.zeros_like(x)
    zx0, zy0 = max(ix - b, 0), max(iy - b, 0)
    zx1, zy1 = min(ix + a, 0) + W, min(iy + a, 0) + H
    if zx0 < zx1 and zy0 < zy1:
        taps = torch.arange(a * 2) - b
        filter_x = (sinc(taps - fx) * sinc((taps - fx) / a)).unsqueeze(0)
        filter_y = (sinc(taps - fy) * sinc((taps - fy) / a)).unsqueeze(1)
        y = filter2d(x, filter_x / filter_x.sum(), padding=[b, a, 0, 0])
        y = filter2d(y, filter_y / filter_y.sum(), padding=[0, 0, b, a])
        y = y[:, :, max(b - iy, 0):H + b + a + min(-iy - a, 0), max(b - ix, 0):W + b + a + min(-ix - a, 0)]
        z[:, :, zy0:zy1, zx0:zx1] = y
    m = torch.zeros_like(x)
    mx0, my0 = max(ix + a, 0), max(iy + a, 0)
    mx1, my1 = min(ix - b, 0) + W, min(iy - b, 0) + H
    if mx0 < mx1 and my0 < my1:
        m[:, :, my0:my1, mx0:mx1] = 1
    return z, m


# ----------------------------------------------------------------------------- shifters.py helpers
def fourier_shift_batch(image, shift_x, shift_y):
    """shifters.py:103-132: phase ramp over fft2; shift_x runs along dim 2 (the 'ij' meshgrid of the
    (W, H) frequency vectors is broadcast as [H, W], so H == W is implied)."""
    N, C, H, W = image.shape
    fft_image = torch.fft.fft2(image)
    u, v = torch.fft.fftfreq(W), torch.fft.fftfreq(H)
    U, V = torch.meshgrid(u, v, indexing="ij")
    phase = torch.exp(-2j * np.pi * (shift_x * U + shift_y * V))[None, None]
    return torch.real(torch.fft.ifft2(fft_image * phase))


def get_blur_kernel(channels, len=4):
    """shifters.py:266-276."""
    k = torch.tensor((1, 3, 3, 1) if len == 4 else (1, 3, 6, 3, 1), dtype=torch.float32)
    k = torch.outer(k, k)
    k = (k / k.sum()).reshape(1, 1, len, len)
    return k.repeat(channels, channels, 1, 1)


def image_upsample(x, scale=2, mode="nearest"):
    """ImageUpsampler(scale, mode).upsample (shifters.py:315-338)."""
    n, c, h, w = x.shape
    x = x.reshape(n * c, 1, h, w)
    if mode == "blur":
        x = upfirdn2d(x, get_blur_kernel(1)[0, 0] * scale ** 2, scale, padding=(2, 1, 2, 1))
    elif mode == "ideal":
        x = upsample_rfft(x, scale)
    else:
        x = F.interpolate(x, scale_factor=scale, mode=mode)
    return x.reshape(n, c, h * scale, w * scale)


def image_low_pass(x, scale=2, mode="nearest"):
    """ImageUpsampler(scale, mode).low_pass (shifters.py:303-313)."""
    if mode == "blur":
        return upfirdn2d(x, get_blur_kernel(1)[0, 0] * 4, 2, padding=(2, 1, 2, 1))
    if mode == "ideal":
        return lpf_recon_rfft(x, 1 / scale)
    return F.interpolate(x, scale_factor=scale, mode=mode)


def image_downsample(x, scale=2, mode="nearest"):
    """ImageDownsampler(scale, mode).downsample (shifters.py:352-365)."""
    n, c, h, w = x.shape
    x = x.reshape(n * c, 1, h, w)
    if mode == "blur":
        x = upfirdn2d(x, get_blur_kernel(1)[0, 0], scale, padding=(2, 1, 2, 1))[:, :, ::2, ::2]
    elif mode == "ideal":
        x = lpf_rfft(x, scale)[:, :, ::2, ::2]
    else:
        x = F.interpolate(x, scale_factor=1 / scale, mode=mode)
    return x.reshape(n, c, h // scale, w // scale)
