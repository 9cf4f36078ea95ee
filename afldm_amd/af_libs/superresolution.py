"""x4 super-resolution degrade operator - the `build_sr4x` surface of reference
afldm/af_libs/superresolution.py:288-320 (the only part of that file the SR harness calls:
scripts/shift_ldm_sr.py:47,109).

The reference routes the image through the SVD factors of a separable down-sampling matrix
(`SRConv`, :160-260, singular values below 3e-2 zeroed; `SuperResolution`, :89-150, for the
4x4 mean) and then nearest-upsamples by 4.  Written out, the operator is one separable product
    y = M x M^T,   M = R Ht   [n x n]
with Ht the truncated [n/4 x n] matrix and R the x4 row-replication matrix.  M is built once on the
host (setup-time linear algebra, like the filter matrices) and applied on MI355X by
afldm_af_resample; there is no CPU path for the product itself.
"""
import numpy as np
import torch

from .. import ops


def _bicubic_taps(factor, a=-0.5):
    def k1(x):
        ax = abs(x)
        if ax <= 1:
            return (a + 2) * ax ** 3 - (a + 3) * ax ** 2 + 1
        if 1 < ax < 2:
            return a * ax ** 3 - 5 * a * ax ** 2 + 8 * a * ax - 4 * a
        return 0.0
    k = np.array([k1((1 / factor) * (i - np.floor(factor * 4 / 2) + 0.5)) for i in range(factor * 4)])
    k = torch.from_numpy(k / k.sum()).float()
    return k / k.sum()


def _strided_reflect_matrix(kernel, img_dim, stride):
    """1-D strided correlation with reflective padding as a [img_dim/stride, img_dim] matrix."""
    K = kernel.shape[0]
    H = torch.zeros(img_dim // stride, img_dim)
    for i in range(stride // 2, img_dim + stride // 2, stride):
        for j in range(i - K // 2, i + K // 2):
            je = -j - 1 if j < 0 else ((img_dim - 1) - (j - img_dim) if j >= img_dim else j)
            H[i // stride, je] += kernel[j - i + K // 2]
    return H


def degrade_matrix(sr_filter, image_size, factor=4):
    assert sr_filter in ["pool", "bicubic"]
    small = image_size // factor
    if sr_filter == "bicubic":
        H = _strided_reflect_matrix(_bicubic_taps(factor), image_size, factor)
        U, s, V = torch.svd(H, some=False)
        s = torch.where(s < 3e-2, torch.zeros_like(s), s)      # the reference's ZERO threshold
        Ht = (U[:, : s.shape[0]] * s) @ V[:, : s.shape[0]].T
    else:
        Ht = torch.zeros(small, image_size)
        for i in range(small):
            Ht[i, factor * i: factor * (i + 1)] = 1.0 / factor
    R = torch.zeros(image_size, small)
    R[torch.arange(image_size), torch.arange(image_size) // factor] = 1.0
    return (R @ Ht).contiguous()


def build_sr4x(device, sr_filter, image_size):
    """-> sr4x(img): [B,3,n,n] or [3,n,n] in, same shape out (x1/4 `sr_filter` then nearest x4)."""
    assert sr_filter in ["pool", "bicubic"]
    M = degrade_matrix(sr_filter, image_size).to(device)

    def sr4x(img):
        is3d = img.dim() == 3
        x = (img.unsqueeze(0) if is3d else img).to(device=device, dtype=torch.float32).contiguous()
        assert x.shape[-1] == image_size and x.shape[-2] == image_size
        y = ops.to_nchw(ops.af_resample(ops.to_nhwc(x), M))
        return y[0] if is3d else y

    return sr4x
