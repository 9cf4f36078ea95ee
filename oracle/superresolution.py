"""CPU restatement (test infrastructure only) of the reference's x4 super-resolution degrade
operator `build_sr4x` (afldm/af_libs/superresolution.py:288-320).

The reference applies H = U diag(s) V^T of a separable down-sampling matrix through SVD factors
(`SRConv`, :160-260, singular values below 3e-2 zeroed :196-197; `SuperResolution`, :89-150 for
the 4x4 mean) and then nearest-upsamples by 4 (:296, :312).  Written out, the whole operator is
    y = (R Ht) x (R Ht)^T   per channel,
with Ht the truncated [n/4 x n] matrix and R the [n x n/4] row-replication matrix.
Pinned by tests/golden/g9_sr4x.npz, recorded from the imported reference (oracle/gen_golden.py part d).
"""
import numpy as np
import torch


def bicubic_kernel_taps(factor=4, a=-0.5):
    """superresolution.py:263-277 (kernel of 4*factor taps, normalised twice)."""
    def k1(x):
        ax = abs(x)
        if ax <= 1:
            return (a + 2) * ax ** 3 - (a + 3) * ax ** 2 + 1
        if 1 < ax < 2:
            return a * ax ** 3 - 5 * a * ax ** 2 + 8 * a * ax - 4 * a
        return 0.0
    k = np.zeros(factor * 4)
    for i in range(factor * 4):
        k[i] = k1((1 / factor) * (i - np.floor(factor * 4 / 2) + 0.5))
    k = k / np.sum(k)
    kernel = torch.from_numpy(k).float()
    return kernel / kernel.sum()


def conv_matrix_1d(kernel, img_dim, stride):
    """The 1-D strided correlation matrix with reflective padding (SRConv.__init__, :177-189)."""
    small = img_dim // stride
    H = torch.zeros(small, img_dim)
    K = kernel.shape[0]
    for i in range(stride // 2, img_dim + stride // 2, stride):
        for j in range(i - K // 2, i + K // 2):
            je = j
            if je < 0:
                je = -je - 1
            if je >= img_dim:
                je = (img_dim - 1) - (je - img_dim)
            H[i // stride, je] += kernel[j - i + K // 2]
    return H


def truncated(H, zero=3e-2):
    """U diag(s) V^T with singular values below `zero` dropped (:191-197)."""
    U, s, V = torch.svd(H, some=False)
    s = s.clone()
    s[s < zero] = 0
    return (U[:, : s.shape[0]] * s) @ V[:, : s.shape[0]].T


def degrade_matrix(sr_filter, image_size, factor=4):
    """M [n x n]: nearest x4 of the (truncated) x1/4 operator, as one matrix."""
    assert sr_filter in ("pool", "bicubic")
    small = image_size // factor
    if sr_filter == "bicubic":
        Ht = truncated(conv_matrix_1d(bicubic_kernel_taps(factor), image_size, factor))
    else:   # SuperResolution: each factor x factor patch -> its mean (H = ones / factor^2 per patch)
        Ht = torch.zeros(small, image_size)
        for i in range(small):
            Ht[i, factor * i: factor * (i + 1)] = 1.0 / factor
    R = torch.zeros(image_size, small)
    R[torch.arange(image_size), torch.arange(image_size) // factor] = 1.0
    return R @ Ht


def build_sr4x(sr_filter, image_size):
    M = degrade_matrix(sr_filter, image_size).double()

    def sr4x(img):
        squeeze = img.dim() == 3
        x = img.unsqueeze(0) if squeeze else img
        y = torch.einsum("rh,bchw,sw->bcrs", M, x.double(), M).float()
        return y[0] if squeeze else y
    return sr4x
