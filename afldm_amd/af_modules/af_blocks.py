"""Alias-free blocks — module surface of reference afldm/af_modules/af_blocks.py
(WarpedNonlinearity :12-28, AliasFreeUpsample2D :45-106, AliasFreeDownsample2D :109-152),
executing on the HIP kernels of libafldm_hip.so.

Inside a UNet forward these modules see NHWC tensors (see models/blocks.py).  Note that
ResnetBlock2D fuses GroupNorm + WarpedNonlinearity into one kernel when its `nonlinearity` is a
WarpedNonlinearity, so `WarpedNonlinearity.forward` itself only runs for stand-alone use.
"""
import torch
import torch.nn as nn

from .. import ops
from ..af_libs.ideal_lpf import LPF_RFFT, UpsampleRFFT  # noqa: F401  (re-exported like the reference)
from ..models.blocks import Downsample2D, Upsample2D, conv_forward


class WarpedNonlinearity(nn.Module):
    """Any module can be wrapped, as in the reference (af_blocks.py:12-28).  nn.SiLU - the activation of every AF-LDM
    UNet / VAE config - runs as ONE fused kernel (afldm_af_act; ResnetBlock2D additionally fuses its GroupNorm into it);
    any other module runs as HIP x2 upsample -> the wrapped module itself (the caller's code, elementwise on the 2N x 2N
    plane) -> HIP low-pass + decimate."""

    def __init__(self, nonlinearity):
        super().__init__()
        self.up_layer = UpsampleRFFT()
        self.lpf = LPF_RFFT(1 / 2)
        self.nonlinearity = nonlinearity

    @property
    def fused_silu(self):
        return type(self.nonlinearity) is nn.SiLU

    def forward(self, x):
        """x: NHWC [B, N, N, C] (internal layout) or a <4-D tensor (the plain nonlinearity, af_blocks.py:20-21)."""
        if self.fused_silu:
            return ops.silu(x) if x.ndim < 4 else ops.af_act(x)
        if x.ndim < 4:
            return self.nonlinearity(x)
        # The wrapped module is the caller's code and expects the reference's NCHW tensor (af_blocks.py:24-27): modules with
        # channel semantics - PReLU(num_parameters=C), Softmax / GLU over dim 1, channel-wise norms - would silently
        # compute something else on the internal NHWC layout.  A permuted VIEW carries the NCHW shape and
        # strides over the NHWC memory; the result is brought back to contiguous NHWC for the HIP low-pass.
        up = ops.af_up2(x).permute(0, 3, 1, 2)
        return ops.af_lpf_down2(self.nonlinearity(up).permute(0, 2, 3, 1).contiguous())


class AliasFreeUpsample2D(Upsample2D):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 kernel_size=None, padding=1, norm_type=None, eps=None, elementwise_affine=None, bias=True,
                 interpolate=True, ori_conv=None):
        super().__init__(channels, use_conv, use_conv_transpose, out_channels, name, kernel_size, padding, norm_type,
                         eps, elementwise_affine, bias, interpolate)
        self.up_layer = UpsampleRFFT()
        self.conv = ori_conv

    def forward(self, hidden_states, output_size=None, *args, **kwargs):
        assert hidden_states.shape[-1] == self.channels          # NHWC
        if self.interpolate:
            hidden_states = ops.af_up2(hidden_states)            # filters stay fp32-accumulated in any dtype
        if self.use_conv:
            hidden_states = conv_forward(self.conv if self.name == "conv" else self.Conv2d_0, hidden_states,
                                         want_stats=True)
        return hidden_states


class AliasFreeDownsample2D(Downsample2D):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", kernel_size=3,
                 norm_type=None, eps=None, elementwise_affine=None, bias=True, ori_conv=None):
        super().__init__(channels, use_conv, out_channels, padding, name, kernel_size, norm_type, eps,
                         elementwise_affine, bias)
        self.conv = ori_conv
        self.conv.stride = 1           # reference af_blocks.py:129: the stride-2 conv runs at stride 1
        self.lpf = LPF_RFFT()
        if getattr(self, "Conv2d_0", None) is not None:
            self.Conv2d_0 = None

    def forward(self, hidden_states, *args, **kwargs):
        assert hidden_states.shape[-1] == self.channels          # NHWC
        # padding == 0 (VAE encoder): the reference zero-pads (1,1,1,1) and runs the conv unpadded
        # (af_blocks.py:142-144) == a 'same' 3x3 conv, which is what the implicit GEMM computes;
        # padding == 1 (UNet): the conv's own padding.  Both are the stride-1 'same' convolution.
        assert self.padding in (0, 1) and tuple(self.conv.kernel_size) == (3, 3)
        hidden_states = conv_forward(self.conv, hidden_states)
        return ops.af_lpf_down2(hidden_states, want_stats=True)
