"""diffusers-compatible building blocks of UNet2DModel, executing on libafldm_hip.so.

Constructor arguments, attribute names and state-dict keys follow diffusers 0.32.1 (the
package the reference builds on: reference afldm/af_modules/af_blocks.py:6-7,
afldm/pipelines/cross_frame_attn.py:3), so unmodified diffusers checkpoints load and the
reference's module surgery (af_api.py) applies as-is.

INTERNAL TENSOR CONVENTION: inside a UNet forward every activation is an NHWC ("channels
last") contiguous CUDA tensor [B, H, W, C] in the model dtype — the layout the MFMA implicit
GEMM, the attention kernel ([B, HW, C] is the same memory) and the alias-free kernels want.
`UNet2DModel.forward` converts from/to the public NCHW layout at its boundary.  A "virtual
concat" (the up-block skip torch.cat) is passed around as a tuple (x1, x2) and resolved inside
the kernels (two base pointers), never materialised.
"""
import math

import os

import torch
import torch.nn as nn

from .. import aql, ops

# AFLDM_SHORTCUT_ORDER: where a ResnetBlock2D issues its 1x1 conv_shortcut (independent of norm1 -> conv1 -> norm2):
# 0 (default) between the second activation and conv2, 1 behind the first activation, 2 in front of it.  Same results;
# with an AQL policy armed (afldm_amd/aql.py) the launch marked `independent` runs beside its neighbour.
_SC_ORDER = int(os.environ.get("AFLDM_SHORTCUT_ORDER", "0"))
# Edges of a ResnetBlock2D that travel in 8-channel blocks at the 32^2 / 16^2 levels: 1 act1 -> conv1, 2 conv1 -> act2, 4 act2 -> conv2.
# Default 5: the activations WRITE blocks (an item's output is one contiguous run: -9 % per launch) and the convolutions read
# them; a convolution writing blocks loses more in its epilogue (16-byte pieces to 24 planes per row) than the activation
# behind it gains: 4.804 (0) / 4.785 (7) / 4.827 (2) / 4.754 (5) ms/step, same box (profiles/r05/c8_layout_ab.txt).
_C8_EDGES = int(os.environ.get("AFLDM_C8_EDGES", "5"))


def _pair(x):
    return x if isinstance(x, tuple) else (x, None)


class PackedMixin:
    """Caches kernel-ready copies of parameters (OHWI weights in the activation dtype, fp32
    biases / affine params).  Invalidated by _apply (.to / .cuda / .half ...) and load_state_dict."""

    def _packed(self, key, build):
        cache = self.__dict__.setdefault("_afldm_cache", {})
        if key not in cache:
            cache[key] = build()
        return cache[key]

    def _invalidate(self):
        self.__dict__.pop("_afldm_cache", None)


def invalidate_packed(module: nn.Module):
    for m in module.modules():
        m.__dict__.pop("_afldm_cache", None)


def packed_conv(mod, dtype):
    """(OHWI weight in `dtype`, fp32 bias) for an nn.Conv2d / nn.Linear, cached on the module."""
    cache = mod.__dict__.setdefault("_afldm_cache", {})
    key = ("w", dtype)
    if key not in cache:
        w = ops.pack_weight(mod.weight, dtype)
        b = None if mod.bias is None else mod.bias.detach().to(torch.float32).contiguous()
        cache[key] = (w, b)
    return cache[key]


def packed_qkv(attn, dtype, which):
    """Row-concatenated (to_q | to_k | to_v subset) OHWI weight + fp32 bias, cached on the module."""
    cache = attn.__dict__.setdefault("_afldm_cache", {})
    key = ("qkv", dtype, which)
    if key not in cache:
        mods = [getattr(attn, "to_" + n) for n in which]
        w = torch.cat([m.weight.detach().float() for m in mods], 0)
        b = torch.cat([m.bias.detach().float() for m in mods], 0).contiguous()
        cache[key] = (ops.pack_weight(w, dtype), b)
    return cache[key]


def packed_norm(mod):
    cache = mod.__dict__.setdefault("_afldm_cache", {})
    if "gn" not in cache:
        cache["gn"] = (mod.weight.detach().to(torch.float32).contiguous(),
                       mod.bias.detach().to(torch.float32).contiguous())
    return cache["gn"]


def packed_conv_dense2x2(conv, dtype, C1, C2):
    """A 3x3 'same' convolution on a 2x2 plane is ONE dense layer over the flattened plane: output
    pixel p = (oh, ow) sees input pixel q = (ih, iw) through tap (ih - oh + 1, iw - ow + 1), which
    always lies inside the kernel.  W2[(p, n), (q, c)] = W[n, c, ih-oh+1, iw-ow+1], columns ordered
    like the NHWC flattening of the (virtually concatenated) input: [x1: q*C1 + c | x2: q*C2 + c].
    The implicit GEMM would spend 5 of its 9 taps on zero padding here (2.25x the flops)."""
    cache = conv.__dict__.setdefault("_afldm_cache", {})
    key = ("dense2x2", dtype, C1, C2)
    if key not in cache:
        W = conv.weight.detach().float()                       # [Cout, Cin, 3, 3]
        Cout = W.shape[0]
        assert W.shape[1] == C1 + C2 and tuple(W.shape[2:]) == (3, 3)
        parts = []
        for lo, hi in ((0, C1), (C1, C1 + C2)):
            if hi == lo:
                continue
            blk = torch.zeros(4, Cout, 4, hi - lo, dtype=torch.float32, device=W.device)
            for pidx in range(4):
                oh, ow = divmod(pidx, 2)
                for qidx in range(4):
                    ih, iw = divmod(qidx, 2)
                    blk[pidx, :, qidx, :] = W[:, lo:hi, ih - oh + 1, iw - ow + 1]
            parts.append(blk.reshape(4 * Cout, 4 * (hi - lo)))
        w2 = torch.cat(parts, 1).contiguous()
        b2 = None if conv.bias is None else conv.bias.detach().float().repeat(4).contiguous()
        cache[key] = (ops.pack_weight(w2, dtype), b2)
    return cache[key]


# The alias-free activation of a 2x2 plane is plane-constant (lpf(4) = [1,0,0,0]: reference ideal_lpf.py:17-21), so the 3x3 convolutions
# behind it are dense layers over Cin columns with tap-summed weights (packed_conv_dense2x2_const): a quarter of the 2x2 level's weight
# bytes (330 -> 83 MB per step).  AFLDM_NO_CONST2=1: the full flattened-plane form (A/B).
_CONST2 = os.environ.get("AFLDM_NO_CONST2", "0") != "1"


def packed_conv_dense2x2_const(conv, dtype):
    """A 3x3 'same' convolution of a PLANE-CONSTANT 2x2 input a[b, c] (the output of WarpedNonlinearity at N = 2): output pixel
    p = (oh, ow) sees every input pixel q through tap (ih - oh + 1, iw - ow + 1), and all four carry the same value, so
    y[b, (p, n)] = sum_c (sum_q W[n, c, tap(p, q)]) a[b, c]: the dense layer of packed_conv_dense2x2 with its four column blocks
    added up - Ws[(p, n), c] = W[n, c, 1-oh : 3-oh, 1-ow : 3-ow].sum().  Summed in fp64, rounded once to `dtype`."""
    cache = conv.__dict__.setdefault("_afldm_cache", {})
    key = ("dense2x2_const", dtype)
    if key not in cache:
        W = conv.weight.detach().double()                      # [Cout, Cin, 3, 3]
        assert tuple(W.shape[2:]) == (3, 3)
        blocks = [W[:, :, 1 - oh:3 - oh, 1 - ow:3 - ow].sum((2, 3)) for oh in (0, 1) for ow in (0, 1)]
        ws = torch.cat(blocks, 0).float().contiguous()         # [4 * Cout, Cin], row = pixel * Cout + n
        b2 = None if conv.bias is None else conv.bias.detach().float().repeat(4).contiguous()
        cache[key] = (ops.pack_weight(ws, dtype), b2)
    return cache[key]


def packed_conv_dense2x2_const_cm(conv, dtype):
    """packed_conv_dense2x2_const with its rows ordered channel-major (row = 4 n + pixel instead of pixel * Cout + n): a GroupNorm
    group of the output - cpg channels x 4 pixels - is then 4 cpg ADJACENT rows, what afldm_conv2x2_const_norm_act tiles over."""
    cache = conv.__dict__.setdefault("_afldm_cache", {})
    key = ("dense2x2_const_cm", dtype)
    if key not in cache:
        W = conv.weight.detach().double()
        Cout = W.shape[0]
        blocks = [W[:, :, 1 - oh:3 - oh, 1 - ow:3 - ow].sum((2, 3)) for oh in (0, 1) for ow in (0, 1)]      # [pixel][Cout, Cin]
        ws = torch.stack(blocks, 1).reshape(4 * Cout, -1).float().contiguous()                             # row = 4 n + pixel
        cache[key] = ops.pack_weight(ws, dtype)
    return cache[key]


def conv_forward(conv: nn.Conv2d, x, **kw):
    """F.conv2d(x, conv.weight, conv.bias, stride 1, 'same') on NHWC (or a virtual concat)."""
    x1, x2 = _pair(x)
    if (x1.ndim == 4 and x1.shape[1] == 2 and x1.shape[2] == 2 and tuple(conv.kernel_size) == (3, 3)
            and kw.get("out_mode", 0) == 0 and "out" not in kw and x1.shape[-1] % 8 == 0
            and (x2 is None or x2.shape[-1] % 8 == 0) and conv.out_channels % 8 == 0
            and not os.environ.get("AFLDM_NO_DENSE2X2")):
        B, C1, C2, Cout = x1.shape[0], x1.shape[-1], 0 if x2 is None else x2.shape[-1], conv.out_channels
        w2, b2 = packed_conv_dense2x2(conv, x1.dtype, C1, C2)
        kw = dict(kw)
        res = kw.pop("residual", None)
        if kw.get("temb") is not None:
            kw["temb_mod"] = Cout
        y = ops.conv2d(x1.reshape(B, 4 * C1), w2, b2, x2=None if x2 is None else x2.reshape(B, 4 * C2),
                       residual=None if res is None else res.reshape(B, 4 * Cout), **kw)
        out = y.view(B, 2, 2, Cout)
        st = getattr(y, "gn_partial", None)
        if st is not None:                                      # [B, S, 4*Cout, 2] -> per-pixel splits of Cout channels
            out.gn_partial = st.reshape(B, st.shape[1] * 4, Cout, 2)
        return out
    if (x2 is None and x1.ndim == 4 and x1.shape[-1] == 3 and tuple(conv.kernel_size) == (3, 3)
            and x1.dtype == torch.bfloat16 and conv.out_channels % 16 == 0 and (x1.shape[1] * x1.shape[2]) % 128 == 0):
        # RGB conv_in of the VAE encoder: a zero 4th channel puts it on the MFMA conv_in kernel
        # (k_conv_cin4_mfma) instead of the scalar small-Cin kernel (580 -> ~90 us at 8 x 256^2)
        cache = conv.__dict__.setdefault("_afldm_cache", {})
        key = ("w_cin4", x1.dtype)
        if key not in cache:
            w4 = torch.nn.functional.pad(conv.weight.detach().float(), (0, 0, 0, 0, 0, 1))      # [Cout, 4, 3, 3]
            cache[key] = (ops.pack_weight(w4, x1.dtype),
                          None if conv.bias is None else conv.bias.detach().to(torch.float32).contiguous())
        w, b = cache[key]
        return ops.conv2d(torch.nn.functional.pad(x1, (0, 1)), w, b, **kw)
    w, b = packed_conv(conv, x1.dtype)
    return ops.conv2d(x1, w, b, x2=x2, **kw)


def linear_forward(lin: nn.Linear, x, **kw):
    w, b = packed_conv(lin, x.dtype)
    return ops.conv2d(x, w, b, **kw)


# ----------------------------------------------------------------------------- embeddings
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps, dtype=torch.float32):
        """timesteps: device float32 [rows] -> [rows, num_channels] in `dtype`."""
        return ops.timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      float(self.downscale_freq_shift), dtype)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        h = linear_forward(self.linear_1, sample)
        h = ops.silu(h)
        return linear_forward(self.linear_2, h)


# ----------------------------------------------------------------------------- resampling
class Downsample2D(nn.Module):
    """diffusers.models.downsampling.Downsample2D (conv variant, stride 2).  The vanilla
    stride-2 path is not on the alias-free hot path: only AliasFreeDownsample2D executes."""

    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", kernel_size=3,
                 norm_type=None, eps=None, elementwise_affine=None, bias=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.padding = padding
        self.name = name
        self.norm = None
        assert norm_type is None, "norm_type is not used by the FFHQ / AF-VAE configs"
        if use_conv:
            conv = nn.Conv2d(self.channels, self.out_channels, kernel_size=kernel_size, stride=2, padding=padding,
                             bias=bias)
        else:
            assert self.channels == self.out_channels
            conv = nn.AvgPool2d(kernel_size=2, stride=2)
        if name == "conv":
            self.Conv2d_0 = conv
            self.conv = conv
        elif name == "Conv2d_0":
            self.conv = conv
        else:
            self.conv = conv

    def forward(self, hidden_states, *args, **kwargs):
        raise NotImplementedError(
            "afldm_amd executes the alias-free model only: call make_af_unet(unet) "
            "(afldm.af_modules.af_api) before running; the strided Downsample2D has no HIP path")


class Upsample2D(nn.Module):
    """diffusers.models.upsampling.Upsample2D (nearest x2 + conv).  See Downsample2D."""

    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 kernel_size=None, padding=1, norm_type=None, eps=None, elementwise_affine=None, bias=True,
                 interpolate=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_conv_transpose = use_conv_transpose
        self.name = name
        self.interpolate = interpolate
        self.norm = None
        assert norm_type is None and not use_conv_transpose
        conv = None
        if use_conv:
            if kernel_size is None:
                kernel_size = 3
            conv = nn.Conv2d(self.channels, self.out_channels, kernel_size=kernel_size, padding=padding, bias=bias)
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv

    def forward(self, hidden_states, output_size=None, *args, **kwargs):
        raise NotImplementedError(
            "afldm_amd executes the alias-free model only: call make_af_unet(unet) first; the "
            "nearest-neighbour Upsample2D has no HIP path")


def _bias_f32(conv):
    cache = conv.__dict__.setdefault("_afldm_cache", {})
    if "bias_f32" not in cache:
        cache["bias_f32"] = None if conv.bias is None else conv.bias.detach().to(torch.float32).contiguous()
    return cache["bias_f32"]


def _plane2_view(y, B, Cout):
    """[B, 4 * Cout] output of a flattened-plane dense layer as the NHWC tensor [B, 2, 2, Cout], statistics carried along."""
    out = y.view(B, 2, 2, Cout)
    st = getattr(y, "gn_partial", None)
    if st is not None:                                      # [B, S, 4*Cout, 2] -> per-pixel splits of Cout channels
        out.gn_partial = st.reshape(B, st.shape[1] * 4, Cout, 2)
    return out


# ----------------------------------------------------------------------------- resnet
class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D (time_embedding_norm='default', no up/down, output_scale 1).

    forward(x, temb_proj): `temb_proj` is this block's slice [rows, out_channels] of the
    batched time_emb_proj GEMM computed once per step by UNet2DModel (rows = 1 broadcast or B),
    i.e. time_emb_proj(nonlinearity(emb)) of the reference forward."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish",
                 time_embedding_norm="default", output_scale_factor=1.0, use_in_shortcut=None):
        super().__init__()
        assert time_embedding_norm == "default" and output_scale_factor == 1.0
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.output_scale_factor = output_scale_factor
        self.time_embedding_norm = time_embedding_norm
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.upsample = self.downsample = None
        self.use_in_shortcut = self.in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=True)

    def _c8_plan(self, input_tensor, temb_proj, temb_stride=0):
        """(conv1 takes / writes 8-channel blocks, conv2 takes them): the layout of the tensors between this block's
        alias-free activations and its 3x3 convolutions at the 32^2 / 16^2 levels (ops._C8; afldm_conv2d_c8_ok)."""
        from ..af_modules.af_blocks import WarpedNonlinearity
        x1, x2 = _pair(input_tensor)
        if (not ops._C8 or x1.ndim != 4 or x1.dtype != torch.bfloat16 or x1.shape[1] != x1.shape[2] or x1.shape[1] not in (16, 32)
                or not isinstance(self.nonlinearity, WarpedNonlinearity) or not self.nonlinearity.fused_silu
                or tuple(self.conv1.kernel_size) != (3, 3) or tuple(self.conv2.kernel_size) != (3, 3)):
            return False, False
        B, N = x1.shape[0], x1.shape[1]
        cin = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
        key = (B, N, cin, x1.device, int(temb_stride))
        cache = self.__dict__.setdefault("_afldm_c8", {})
        if key not in cache:
            w1, b1 = packed_conv(self.conv1, x1.dtype)
            w2, b2 = packed_conv(self.conv2, x1.dtype)
            a_in = torch.empty((B, N, N, cin), dtype=x1.dtype, device=x1.device)
            h_in = torch.empty((B, N, N, self.out_channels), dtype=x1.dtype, device=x1.device)
            cache[key] = (ops.conv2d_c8_ok(a_in, w1, b1, temb=temb_proj, temb_stride=temb_stride),
                          ops.conv2d_c8_ok(h_in, w2, b2, residual=h_in))
        return cache[key]

    def _norm_act(self, norm, x, out_c8=False, out_const=False):
        """norm -> self.nonlinearity fused: GroupNorm statistics, then either the fused
        GN + WarpedNonlinearity kernel (alias-free model) or GN + SiLU."""
        from ..af_modules.af_blocks import WarpedNonlinearity
        x1, x2 = _pair(x)
        gamma, beta = packed_norm(norm)
        stats = ops.gn_stats(x1, norm.num_groups, x2=x2)
        if isinstance(self.nonlinearity, WarpedNonlinearity):
            if self.nonlinearity.fused_silu:
                return ops.af_act(x1, x2, stats, gamma, beta, norm.num_groups, norm.eps, out_c8=out_c8, out_const=out_const)
            # a wrapped module other than SiLU: GroupNorm pass, then the module's own (unfused) alias-free form
            return self.nonlinearity(ops.gn_apply(x1, stats, gamma, beta, norm.num_groups, norm.eps, act=0, x2=x2))
        return ops.gn_apply(x1, stats, gamma, beta, norm.num_groups, norm.eps, act=1, x2=x2)

    def _norm_act_conv(self, norm, x, conv, force=False, **kw):
        """conv(nonlinearity(norm(x))): ONE merged launch at the 32^2 / 16^2 levels of the alias-free bf16 model
        (ops.af_act_conv2d, csrc/actconv.hip), else the activation kernel followed by the convolution."""
        from ..af_modules.af_blocks import WarpedNonlinearity
        x1, x2 = _pair(x)
        if (isinstance(self.nonlinearity, WarpedNonlinearity) and self.nonlinearity.fused_silu and x1.ndim == 4
                and x1.dtype == torch.bfloat16 and (force or x1.shape[1] in ops._ACTCONV_N) and tuple(conv.kernel_size) == (3, 3)):
            gamma, beta = packed_norm(norm)
            stats = ops.gn_stats(x1, norm.num_groups, x2=x2)
            w, b = packed_conv(conv, x1.dtype)
            out = ops.af_act_conv2d(x1, x2, stats, gamma, beta, norm.num_groups, norm.eps, w, b, **kw)
            if out is not None:
                return out
        return conv_forward(conv, self._norm_act(norm, x), **kw)

    def _conv1_norm2_act_fused(self, h, temb_proj, temb_stride):
        """conv1 -> norm2 -> WarpedNonlinearity on the 2x2 / 4x4 planes when conv1 splits K: the activation kernel takes
        the convolution's fp32 slabs and finishes them itself (afldm_af_act_slabs) - no reduction launch, no stored
        intermediate.  Returns None when this shape / plan does not qualify (the caller runs the ordinary sequence)."""
        from ..af_modules.af_blocks import WarpedNonlinearity
        if (os.environ.get("AFLDM_NO_FUSED_ACT") or not isinstance(self.nonlinearity, WarpedNonlinearity)
                or not self.nonlinearity.fused_silu or isinstance(h, tuple) or h.ndim != 4 or h.shape[1] != h.shape[2] or h.shape[1] not in (2, 4)):
            return None
        B, N, _, Cin = h.shape
        conv, norm = self.conv1, self.norm2
        Cout = conv.out_channels
        if tuple(conv.kernel_size) != (3, 3) or Cout % norm.num_groups or Cin % 8 or Cout % 8:
            return None
        if N == 2 and not os.environ.get("AFLDM_NO_DENSE2X2"):
            w2, _ = packed_conv_dense2x2(conv, h.dtype, Cin, 0)          # one dense layer over the flattened plane
            got = ops.conv2d_slabs(h.reshape(B, 4 * Cin), w2)
        else:
            w, _ = packed_conv(conv, h.dtype)
            got = ops.conv2d_slabs(h, w)
        if got is None:
            return None
        slabs, nslab = got
        gamma, beta = packed_norm(norm)
        cache = conv.__dict__.setdefault("_afldm_cache", {})
        if "bias_f32" not in cache:
            cache["bias_f32"] = None if conv.bias is None else conv.bias.detach().to(torch.float32).contiguous()
        bias = cache["bias_f32"]
        return ops.af_act_slabs(slabs, nslab, bias, temb_proj, temb_stride, gamma, beta, norm.num_groups, norm.eps,
                                B, N, Cout, h.dtype)

    def _conv2_to_next_norm_fused(self, h, res, next_gn):
        """conv2 (+ shortcut) of this block straight into the GroupNorm of the attention block that follows, on the
        2x2 / 4x4 planes when conv2 splits K (afldm_af_act_slabs, act = 0): the reduction launch and the GroupNorm
        pass become one.  Returns this block's output with the normalised tensor attached as `.gn_applied =
        (tensor, norm module)` for AttnProcessor2_0, or None when the shape / plan does not qualify."""
        if (os.environ.get("AFLDM_NO_FUSED_ACT") or isinstance(h, tuple) or h.ndim != 4 or h.shape[1] != h.shape[2]
                or h.shape[1] not in (2, 4)):
            return None
        B, N, _, Cin = h.shape
        conv = self.conv2
        Cout = conv.out_channels
        if (tuple(conv.kernel_size) != (3, 3) or Cout % next_gn.num_groups or Cin % 8 or Cout % 8
                or next_gn.num_channels != Cout or res.shape[-1] != Cout or not res.is_contiguous()):
            return None          # (afldm_af_act_slabs reads the residual as dense [B, N, N, Cout]: no res_ld)
        if N == 2 and not os.environ.get("AFLDM_NO_DENSE2X2"):
            w2, _ = packed_conv_dense2x2(conv, h.dtype, Cin, 0)
            got = ops.conv2d_slabs(h.reshape(B, 4 * Cin), w2)
        else:
            w, _ = packed_conv(conv, h.dtype)
            got = ops.conv2d_slabs(h, w)
        if got is None:
            return None
        slabs, nslab = got
        gamma, beta = packed_norm(next_gn)
        cache = conv.__dict__.setdefault("_afldm_cache", {})
        if "bias_f32" not in cache:
            cache["bias_f32"] = None if conv.bias is None else conv.bias.detach().to(torch.float32).contiguous()
        hn, y = ops.af_act_slabs(slabs, nslab, cache["bias_f32"], None, 0, gamma, beta, next_gn.num_groups, next_gn.eps,
                                 B, N, Cout, h.dtype, residual=res, want_raw=True, act=False)
        y.gn_applied = (hn, next_gn)
        return y

    def _const2_ok(self, input_tensor):
        from ..af_modules.af_blocks import WarpedNonlinearity
        x1, x2 = _pair(input_tensor)
        return (_CONST2 and not os.environ.get("AFLDM_NO_DENSE2X2") and x1.ndim == 4 and x1.shape[1] == 2 and x1.shape[2] == 2
                and isinstance(self.nonlinearity, WarpedNonlinearity) and self.nonlinearity.fused_silu
                and tuple(self.conv1.kernel_size) == (3, 3) and tuple(self.conv2.kernel_size) == (3, 3)
                and x1.shape[-1] % 8 == 0 and (x2 is None or x2.shape[-1] % 8 == 0) and self.out_channels % 8 == 0
                and self.out_channels % self.norm2.num_groups == 0
                # K of the two dense layers = Cin / Cout: whole K steps of the GEMM kernels (128 bytes of elements)
                and (x1.shape[-1] + (0 if x2 is None else x2.shape[-1])) % (128 // x1.element_size()) == 0
                and self.out_channels % (128 // x1.element_size()) == 0)

    def _forward_const2(self, input_tensor, temb_proj, temb_stride, next_gn):
        """The block on 2x2 planes: both alias-free activations are plane-constant there, so they are stored once per plane
        ([B, C]) and conv1 / conv2 run as dense layers over Cin / Cout columns with tap-summed weights
        (packed_conv_dense2x2_const) - same function, a quarter of the weight bytes and of the GEMM's K."""
        x1, x2 = _pair(input_tensor)
        B, dt, Cout = x1.shape[0], x1.dtype, self.out_channels
        a = self._norm_act(self.norm1, input_tensor, out_const=True)                 # [B, Cin]
        g2, be2 = packed_norm(self.norm2)
        h = None
        got = None
        if ops.conv2x2_const_norm_act_ok(a.shape[-1], Cout, self.norm2.num_groups, dt, batch=B) and not os.environ.get("AFLDM_NO_FUSED_ACT"):
            # conv1 + temb -> norm2 -> activation in ONE launch: a workgroup owns a whole GroupNorm group of the dense layer's columns
            h = ops.conv2x2_const_norm_act(a, packed_conv_dense2x2_const_cm(self.conv1, dt), _bias_f32(self.conv1), temb_proj,
                                           temb_stride, g2, be2, self.norm2.num_groups, self.norm2.eps)
        else:
            w1, b1 = packed_conv_dense2x2_const(self.conv1, dt)
            got = None if os.environ.get("AFLDM_NO_FUSED_ACT") else ops.conv2d_slabs(a, w1)
        if h is not None:
            pass
        elif got is not None:                                                          # the plan splits K: the slabs' consumer finishes them
            slabs, nslab = got
            h = ops.af_act_slabs(slabs, nslab, _bias_f32(self.conv1), temb_proj, temb_stride, g2, be2, self.norm2.num_groups,
                                 self.norm2.eps, B, 2, Cout, dt, act=2)
        else:
            kw = dict(temb=temb_proj, temb_stride=temb_stride, temb_mod=Cout) if temb_proj is not None else {}
            y = ops.conv2d(a, w1, b1, want_stats=True, **kw)                         # [B, 4 * Cout]
            h = self._norm_act(self.norm2, _plane2_view(y, B, Cout), out_const=True)
        res = conv_forward(self.conv_shortcut, input_tensor) if self.conv_shortcut is not None else x1
        assert self.conv_shortcut is not None or x2 is None
        w2, b2 = packed_conv_dense2x2_const(self.conv2, dt)
        if (next_gn is not None and not os.environ.get("AFLDM_NO_FUSED_ACT") and Cout % next_gn.num_groups == 0
                and next_gn.num_channels == Cout and res.shape[-1] == Cout and res.is_contiguous()):
            got = ops.conv2d_slabs(h, w2)
            if got is not None:                # conv2 (+ shortcut) straight into the attention block's GroupNorm (see _conv2_to_next_norm_fused)
                slabs, nslab = got
                gamma, beta = packed_norm(next_gn)
                hn, y = ops.af_act_slabs(slabs, nslab, _bias_f32(self.conv2), None, 0, gamma, beta, next_gn.num_groups, next_gn.eps,
                                         B, 2, Cout, dt, residual=res, want_raw=True, act=False)
                y.gn_applied = (hn, next_gn)
                return y
        y = ops.conv2d(h, w2, b2, residual=res.reshape(B, 4 * Cout), want_stats=True)
        return _plane2_view(y, B, Cout)

    def forward(self, input_tensor, temb_proj=None, temb_stride=0, next_gn=None):
        """next_gn: the GroupNorm module of an attention block that consumes this block's output next (the block loops
        pass it): lets conv2 hand its result over already normalised where that saves launches."""
        x1, x2 = _pair(input_tensor)
        if self._const2_ok(input_tensor):
            return self._forward_const2(input_tensor, temb_proj, temb_stride, next_gn)
        if x1.ndim == 4 and x1.shape[1] in ops._ACTCONV_N:
            # 32^2 / 16^2 levels, opt-in (AFLDM_ACTCONV_N): norm -> activation -> conv pairs as merged launches where there is a kernel for them
            h = self._norm_act_conv(self.norm1, input_tensor, self.conv1, temb=temb_proj, temb_stride=temb_stride, want_stats=True)
            res = conv_forward(self.conv_shortcut, input_tensor) if self.conv_shortcut is not None else x1
            assert self.conv_shortcut is not None or x2 is None
            return self._norm_act_conv(self.norm2, h, self.conv2, residual=res, want_stats=True)
        c8_1, c8_2 = self._c8_plan(input_tensor, temb_proj, temb_stride)
        # which of the three edges travel in blocks (AFLDM_C8_EDGES, A/B): 1 act1 -> conv1, 2 conv1 -> act2, 4 act2 -> conv2
        e1, e2, e3 = c8_1 and bool(_C8_EDGES & 1), c8_1 and bool(_C8_EDGES & 2), c8_2 and bool(_C8_EDGES & 4)
        res = None
        merged1 = bool(ops._ACTCONV_SITES) and x1.ndim == 4 and (x1.shape[1], x1.shape[-1] + (0 if x2 is None else x2.shape[-1])) in ops._ACTCONV_SITES
        if merged1:
            h = None
        elif self.conv_shortcut is not None and _SC_ORDER == 2:
            # shortcut first, the activation beside it (AQL policy: afldm_amd/aql.py; tools/aql_shortcut_ab.py)
            res = conv_forward(self.conv_shortcut, input_tensor)
            with aql.independent("act1"):
                h = self._norm_act(self.norm1, input_tensor, out_c8=e1)
        else:
            h = self._norm_act(self.norm1, input_tensor, out_c8=e1)
        if self.conv_shortcut is not None and _SC_ORDER == 1:
            with aql.independent("shortcut"):
                res = conv_forward(self.conv_shortcut, input_tensor)
        fused = self._conv1_norm2_act_fused(h, temb_proj, temb_stride) if h is not None else None
        if fused is not None:
            h = fused
        elif h is None:
            # per-site policy (ops._ACTCONV_SITES): norm1 -> activation -> conv1 as the merged launch (experimental library)
            h = self._norm_act_conv(self.norm1, input_tensor, self.conv1, force=True, temb=temb_proj, temb_stride=temb_stride, want_stats=True)
            h = self._norm_act(self.norm2, h, out_c8=e3)
        else:
            # (the convs whose outputs feed a GroupNorm emit its statistics from their epilogue)
            h = conv_forward(self.conv1, h, temb=temb_proj, temb_stride=temb_stride, want_stats=True, out_c8=e2)
            h = self._norm_act(self.norm2, h, out_c8=e3)
        if res is not None:
            pass
        elif self.conv_shortcut is not None:
            with aql.independent("shortcut"):
                res = conv_forward(self.conv_shortcut, input_tensor)
        else:
            assert x2 is None
            res = x1
        if next_gn is not None:
            out = self._conv2_to_next_norm_fused(h, res, next_gn)
            if out is not None:
                return out
            # 8x8 planes: the attention block's GroupNorm from the epilogue of conv2 itself (one tile = one whole sample)
            gamma, beta = packed_norm(next_gn)
            out = conv_forward(self.conv2, h, residual=res, want_stats=True,
                               norm_out=(gamma, beta, next_gn.num_groups, next_gn.eps))
            hn = getattr(out, "norm_applied", None)
            if hn is not None:
                out.gn_applied = (hn, next_gn)
            return out
        return conv_forward(self.conv2, h, residual=res, want_stats=True)


def _next_gn(attn):
    """The GroupNorm an attention block will apply to its input, when its processor is the plain self-attention one
    (a cross-frame processor normalises / stores on its own terms: no hand-over)."""
    if attn is None or attn.group_norm is None or type(attn.processor) is not AttnProcessor2_0:
        return None
    return attn.group_norm


# ----------------------------------------------------------------------------- attention
class AttnProcessor2_0:
    """diffusers AttnProcessor2_0 for the self-attention 'attn block' configuration
    (residual_connection=True, rescale_output_factor=1), on NHWC tensors.

    encoder_hidden_states, when given, is the already group-normed K/V source [Bk, HW, C]
    (the protocol CrossFrameAttnProcessor uses, reference cross_frame_attn.py:125)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, kv=None, kv_sink=None):
        """kv / kv_sink (CrossFrameAttnProcessor with cache_kv): `kv_sink(k, vt)` receives this call's projected keys
        [B, T, C] (a view) and channel-major values [B, C, T] - the self-attention then runs on the three-launch path, where
        they exist in memory; `kv = (k, vt)` runs the attention against such a stored pair (Bk divides B) instead of projecting
        an encoder_hidden_states map: K / V of the stored pass are the same numbers whichever pass projects them."""
        assert attention_mask is None
        B, H, W, C = hidden_states.shape
        gamma, beta = packed_norm(attn.group_norm)
        gn = attn.group_norm
        pre = getattr(hidden_states, "gn_applied", None)
        plain = kv is None and kv_sink is None
        assert kv is None or encoder_hidden_states is None
        if (plain and pre is None and encoder_hidden_states is None and C // attn.heads <= 32
                and ops.attn_block_fused_ok(hidden_states.view(B, H * W, C), attn.heads, gn.num_groups)):
            # 32^2 / 16^2 levels, bf16: GroupNorm-apply + q | k | v projection + attention in ONE launch (csrc/attnf.hip)
            stats = ops.gn_stats(hidden_states, gn.num_groups)
            w, b = packed_qkv(attn, hidden_states.dtype, ("q", "k", "v"))
            if ops.attn_block_fused_out_ok(hidden_states.view(B, H * W, C), attn.heads, gn.num_groups):
                # ... and to_out + the residual connection in the same launch (32^2 level)
                wo, bo = packed_conv(attn.to_out[0], hidden_states.dtype)
                y = ops.attn_block_fused_out(hidden_states.view(B, H * W, C), stats, gamma, beta, gn.num_groups, gn.eps, w, b,
                                             attn.heads, attn.scale, wo, bo)
                return ops.carry_stats(y.view(B, H, W, C), y)
            o = ops.attn_block_fused(hidden_states.view(B, H * W, C), stats, gamma, beta, gn.num_groups, gn.eps, w, b,
                                     attn.heads, attn.scale)
            return linear_forward(attn.to_out[0], o.view(B, H, W, C), residual=hidden_states, want_stats=True)
        if pre is not None and pre[1] is gn:
            hn = pre[0]                                   # the producing resnet block already applied this GroupNorm
        else:
            stats = ops.gn_stats(hidden_states, gn.num_groups)
            hn = ops.gn_apply(hidden_states, stats, gamma, beta, gn.num_groups, gn.eps, act=0)
        tokens = hn.view(B, H * W, C)
        if C // attn.heads > 32:
            # large head_dim (the VAE mid block: one head of 512): GEMM - row softmax - GEMM per sample
            assert plain, "kv / kv_sink cover head_dim <= 32 (the UNet's attention blocks)"
            src = tokens if encoder_hidden_states is None else encoder_hidden_states
            q = linear_forward(attn.to_q, tokens)
            k = linear_forward(attn.to_k, src)
            vt = linear_forward(attn.to_v, src, out_mode=1)
            o = ops.attention_dense(q, k, vt, attn.scale)
            return linear_forward(attn.to_out[0], o.view(B, H, W, C), residual=hidden_states, want_stats=True)
        if plain and encoder_hidden_states is None and ops.attn_small_fused_ok(tokens, attn.heads):
            # 8^2 / 4^2 levels, bf16: projection + attention in ONE launch on the normalised tokens (csrc/attns.hip)
            w, b = packed_qkv(attn, tokens.dtype, ("q", "k", "v"))
            o = ops.attn_small_fused(tokens, w, b, attn.heads, attn.scale)
            return linear_forward(attn.to_out[0], o.view(B, H, W, C), residual=hidden_states, want_stats=True)
        if kv is not None:
            q = linear_forward(attn.to_q, tokens)
            k, vt = kv
        elif encoder_hidden_states is None:
            # fused Q|K|V projection: one GEMM reads the normed tokens once; Q and K land token-major
            # side by side, V channel-major (the attention kernel's V^T operand)
            w, b = packed_qkv(attn, tokens.dtype, ("q", "k", "v"))
            qk, vt = ops.linear_split(tokens, w, b, 2 * C)
            q, k = qk[:, :, :C], qk[:, :, C:]
            if kv_sink is not None:
                kv_sink(k, vt)
        else:
            q = linear_forward(attn.to_q, tokens)
            w, b = packed_qkv(attn, tokens.dtype, ("k", "v"))
            k, vt = ops.linear_split(encoder_hidden_states, w, b, C)
        o = ops.attention(q, k, vt, attn.heads, scale=attn.scale)
        return linear_forward(attn.to_out[0], o.view(B, H, W, C), residual=hidden_states, want_stats=True)


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention restricted to what UNet2DModel's attention
    blocks use (self-attention, group_norm, bias, residual connection)."""

    def __init__(self, query_dim, heads=8, dim_head=64, rescale_output_factor=1.0, eps=1e-5, norm_num_groups=None,
                 residual_connection=False, bias=False, upcast_softmax=False, _from_deprecated_attn_block=False,
                 processor=None, **unused):
        super().__init__()
        assert rescale_output_factor == 1.0
        self.inner_dim = dim_head * heads
        self.query_dim = query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.norm_cross = None
        self.spatial_norm = None
        self.group_norm = (nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=True), nn.Dropout(0.0)])
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kwargs)


# ----------------------------------------------------------------------------- UNet blocks
class _BlockBase(nn.Module):
    def _resnets(self, in_cs, out_c, temb_channels, eps, groups, dropout):
        return nn.ModuleList([
            ResnetBlock2D(in_channels=c, out_channels=out_c, temb_channels=temb_channels, eps=eps, groups=groups,
                          dropout=dropout) for c in in_cs])

    def _attns(self, n, c, head_dim, eps, groups):
        return nn.ModuleList([
            Attention(c, heads=c // head_dim, dim_head=head_dim, eps=eps, norm_num_groups=groups,
                      residual_connection=True, bias=True, upcast_softmax=True, _from_deprecated_attn_block=True)
            for _ in range(n)])


class DownBlock2D(_BlockBase):
    has_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 dropout=0.0, add_downsample=True, downsample_padding=1, attention_head_dim=1):
        super().__init__()
        self.resnets = self._resnets([in_channels if i == 0 else out_channels for i in range(num_layers)],
                                     out_channels, temb_channels, resnet_eps, resnet_groups, dropout)
        if self.has_attention:
            self.attentions = self._attns(num_layers, out_channels, attention_head_dim, resnet_eps, resnet_groups)
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=downsample_padding, name="op")])
                             if add_downsample else None)

    def forward(self, hidden_states, temb_slices):
        outs = ()
        for i, resnet in enumerate(self.resnets):
            hidden_states = resnet(hidden_states, *temb_slices[i], next_gn=_next_gn(self.attentions[i]) if self.has_attention else None)
            if self.has_attention:
                hidden_states = self.attentions[i](hidden_states)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class AttnDownBlock2D(DownBlock2D):
    has_attention = True


class UNetMidBlock2D(_BlockBase):
    def __init__(self, in_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_groups=None, add_attention=True, attention_head_dim=1):
        super().__init__()
        self.add_attention = add_attention
        if attn_groups is None:
            attn_groups = resnet_groups
        self.resnets = self._resnets([in_channels] * (num_layers + 1), in_channels, temb_channels, resnet_eps,
                                     resnet_groups, dropout)
        self.attentions = (self._attns(num_layers, in_channels, attention_head_dim, resnet_eps, attn_groups)
                           if add_attention else nn.ModuleList([None] * num_layers))

    def forward(self, hidden_states, temb_slices=None):
        if temb_slices is None:                       # VAE: no time embedding
            temb_slices = [(None, 0)] * len(self.resnets)
        hidden_states = self.resnets[0](hidden_states, *temb_slices[0], next_gn=_next_gn(self.attentions[0]) if len(self.attentions) else None)
        for i, (attn, resnet) in enumerate(zip(self.attentions, self.resnets[1:])):
            if attn is not None:
                hidden_states = attn(hidden_states)
            hidden_states = resnet(hidden_states, *temb_slices[i + 1])
        return hidden_states


class UpBlock2D(_BlockBase):
    has_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, dropout=0.0, add_upsample=True, attention_head_dim=1):
        super().__init__()
        in_cs = []
        for i in range(num_layers):
            res_skip = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            in_cs.append(rin + res_skip)
        self.resnets = self._resnets(in_cs, out_channels, temb_channels, resnet_eps, resnet_groups, dropout)
        if self.has_attention:
            self.attentions = self._attns(num_layers, out_channels, attention_head_dim, resnet_eps, resnet_groups)
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def forward(self, hidden_states, res_hidden_states_tuple, temb_slices):
        for i, resnet in enumerate(self.resnets):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = resnet((hidden_states, res), *temb_slices[i],      # virtual torch.cat([h, res], 1)
                                   next_gn=_next_gn(self.attentions[i]) if self.has_attention else None)
            if self.has_attention:
                hidden_states = self.attentions[i](hidden_states)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class AttnUpBlock2D(UpBlock2D):
    has_attention = True


DOWN_BLOCKS = {"DownBlock2D": DownBlock2D, "AttnDownBlock2D": AttnDownBlock2D}
UP_BLOCKS = {"UpBlock2D": UpBlock2D, "AttnUpBlock2D": AttnUpBlock2D}
