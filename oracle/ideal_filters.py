"""Oracle: ideal (brick-wall) FFT-domain filters, restated from
reference afldm/af_libs/ideal_lpf.py.  CPU / torch.fft / fp32.  Test infrastructure.

Every function cites the reference lines it follows.  The op ORDER is kept
identical to the reference so that results are bit-identical on the same
PyTorch CPU build (checked by tests/test_oracle_golden.py against fixtures
produced from the imported reference).
"""
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- masks
def lpf_rect_1d(N: int, cutoff: float = 0.5) -> torch.Tensor:
    """1-D low-pass mask in FFT bin order (ideal_lpf.py:12-21)."""
    lo = int((N * cutoff) // 2)
    hi = int(N - lo)
    r = torch.ones(N)
    r[lo + 1:hi] = 0
    if N % 4 == 0:          # Nyquist of the decimated signal -> 0 (ideal_lpf.py:17-21)
        r[lo] = 0
        r[hi] = 0
    return r


def recon_rect_1d(N: int, cutoff: float = 0.5) -> torch.Tensor:
    """1-D reconstruction mask (ideal_lpf.py:38-47): Nyquist bins get 0.5."""
    lo = int((N * cutoff) // 2)
    hi = int(N - lo)
    r = torch.ones(N)
    r[lo + 1:hi] = 0
    if N % 4 == 0:
        r[lo] = 0.5
        r[hi] = 0.5
    return r


def lpf_rect_2d(N, cutoff=0.5):
    r = lpf_rect_1d(N, cutoff)
    return r[:, None] * r[None, :]                       # ideal_lpf.py:23


def recon_rect_2d(N, cutoff=0.5):
    r = recon_rect_1d(N, cutoff)
    return r[:, None] * r[None, :]                       # ideal_lpf.py:48


# --------------------------------------------------------------------------- FFT operators
def lpf_rfft(x: torch.Tensor, cutoff: float = 0.5) -> torch.Tensor:
    """LPF_RFFT.forward (ideal_lpf.py:69-93).  Mask is built from the WIDTH only
    (square planes assumed, ideal_lpf.py:80)."""
    x_fft = torch.fft.rfft2(x)
    N = x.shape[-1]
    rect = lpf_rect_2d(N, cutoff)[:, :int(N / 2 + 1)]
    x_fft *= rect
    return torch.fft.irfft2(x_fft, s=(x.shape[-2], x.shape[-1]))


def lpf_recon_rfft(x: torch.Tensor, cutoff: float = 0.5) -> torch.Tensor:
    """LPF_RECON_RFFT.forward (ideal_lpf.py:112-134); irfft2 default (even) size."""
    x_fft = torch.fft.rfft2(x)
    N = x.shape[-1]
    rect = recon_rect_2d(N, cutoff)[:, :int(N / 2 + 1)]
    x_fft *= rect
    return torch.fft.irfft2(x_fft)


def upsample_rfft(x: torch.Tensor, up: int = 2, factor: float = 1) -> torch.Tensor:
    """UpsampleRFFT.forward (ideal_lpf.py:148-158): zero-stuff, recon filter, gain up^2."""
    b, c, h, w = x.shape
    x = x.reshape([b, c, h, 1, w, 1])
    x = F.pad(x, [0, up - 1, 0, 0, 0, up - 1])
    x = x.reshape([b, c, h * up, w * up])
    return lpf_recon_rfft(x, cutoff=1 / up * factor) * (up ** 2)


def subpixel_shift(images, up=2, shift_x=1, shift_y=1):
    """subpixel_shift (ideal_lpf.py:161-172)."""
    u = upsample_rfft(images, up)
    return torch.roll(u, shifts=(-shift_x, -shift_y), dims=(2, 3))[:, :, ::up, ::up]


# --------------------------------------------------------------------------- AF block bodies
def warped_nonlinearity(x: torch.Tensor, act=F.silu) -> torch.Tensor:
    """WarpedNonlinearity.forward (af_blocks.py:19-28)."""
    if x.ndim < 4:
        return act(x)
    x = upsample_rfft(x, 2)
    x = act(x)
    x = lpf_rfft(x, 0.5)
    return x[:, :, ::2, ::2]


def af_downsample(x, weight, bias, padding=1):
    """AliasFreeDownsample2D.forward (af_blocks.py:135-152): conv stride FORCED to 1
    (af_blocks.py:129) -> LPF -> [::2, ::2].  padding==0 => explicit (1,1,1,1) zero pad."""
    if padding == 0:
        x = F.pad(x, (1, 1, 1, 1), mode="constant", value=0)
    x = F.conv2d(x, weight, bias, stride=1, padding=padding)
    x = lpf_rfft(x, 0.5)
    return x[:, :, ::2, ::2]


def af_upsample(x, weight, bias):
    """AliasFreeUpsample2D.forward (af_blocks.py:64-106): UpsampleRFFT(2) -> conv3x3 pad 1."""
    x = upsample_rfft(x, 2)
    return F.conv2d(x, weight, bias, stride=1, padding=1)


# --------------------------------------------------------------------------- dense-matrix form
def circulant_from_mask(mask_1d: np.ndarray) -> np.ndarray:
    """Real MxM circulant C with C[i,j] = h[(i-j) mod M], h = ifft(mask).real
    (irfft2 drops the imaginary part of the DC/Nyquist bins,
    which is what taking .real of a symmetric-mask ifft reproduces)."""
    M = mask_1d.shape[0]
    h = np.fft.ifft(mask_1d.astype(np.float64)).real
    idx = (np.arange(M)[:, None] - np.arange(M)[None, :]) % M
    return h[idx]


def up_matrix(N: int, up: int = 2) -> np.ndarray:
    """U in R^{upN x N}: UpsampleRFFT(up)(X) == U X U^T (fp64)."""
    M = N * up
    C = circulant_from_mask(recon_rect_1d(M, 1.0 / up).numpy())
    return up * C[:, ::up]


def down_matrix(M: int) -> np.ndarray:
    """D in R^{M/2 x M}: LPF_RFFT(0.5)(Z)[::2, ::2] == D Z D^T (fp64)."""
    C = circulant_from_mask(lpf_rect_1d(M, 0.5).numpy())
    return C[::2, :]
