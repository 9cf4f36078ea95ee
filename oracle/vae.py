"""Oracle: functional CPU fp32 restatement of diffusers' AutoencoderKL encode / decode after the
reference's alias-free surgery (make_af_vae, reference af_api.py:34-67; switches from
configs/vae/model_afvae.json:17-22,32,46-56).  diffusers semantics restated (parity unpinned, see
oracle/__init__.py); the alias-free pieces call oracle.ideal_filters (pinned).  Test infrastructure."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ideal_filters as idf

AF_VAE = dict(   # values restated from reference configs/vae/model_afvae.json
    in_channels=3, out_channels=3, latent_channels=4, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
    norm_num_groups=32, scaling_factor=0.6, mid_act=True, down_filtered_act=[False, True, True, True],
    up_filtered_act=[True, True, True, False], up_rescale=[True, True, True],
)


def tiny_vae(**over):
    cfg = dict(AF_VAE)
    cfg.update(block_out_channels=[64, 128, 128, 128], layers_per_block=1)
    cfg.update(over)
    return cfg


def init_vae_params(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def _u(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def conv(name, cin, cout, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        sd[name + ".weight"] = _u((cout, cin, k, k), bound)
        sd[name + ".bias"] = _u((cout,), bound)

    def linear(name, cin, cout):
        bound = 1.0 / math.sqrt(cin)
        sd[name + ".weight"] = _u((cout, cin), bound)
        sd[name + ".bias"] = _u((cout,), bound)

    def gn(name, c):
        sd[name + ".weight"] = 1.0 + 0.2 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    def resnet(p, cin, cout):
        gn(p + ".norm1", cin); conv(p + ".conv1", cin, cout, 3)
        gn(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    def attn(p, c):
        gn(p + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            linear(p + "." + n, c, c)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c); attn(p + ".attentions.0", c); resnet(p + ".resnets.1", c, c)

    boc, L, lc = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    conv("encoder.conv_in", cfg["in_channels"], boc[0], 3)
    out_c = boc[0]
    for i in range(len(boc)):
        in_c, out_c = out_c, boc[i]
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    mid("encoder.mid_block", boc[-1])
    gn("encoder.conv_norm_out", boc[-1]); conv("encoder.conv_out", boc[-1], 2 * lc, 3)
    conv("decoder.conv_in", lc, boc[-1], 3)
    mid("decoder.mid_block", boc[-1])
    rev = list(reversed(boc))
    out_c = rev[0]
    for i in range(len(rev)):
        prev, out_c = out_c, rev[i]
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    gn("decoder.conv_norm_out", boc[0]); conv("decoder.conv_out", boc[0], cfg["out_channels"], 3)
    conv("quant_conv", 2 * lc, 2 * lc, 1)
    conv("post_quant_conv", lc, lc, 1)
    return sd


def _gn(sd, name, x, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def _resnet(sd, p, x, groups, af):
    act = idf.warped_nonlinearity if af else F.silu
    h = act(_gn(sd, p + ".norm1", x, groups))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = act(_gn(sd, p + ".norm2", h, groups))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attn(sd, p, x, groups):
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".group_norm", x.view(b, c, hh * ww), groups).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]      # one head of dim c
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


def _mid(sd, p, x, groups, af):
    x = _resnet(sd, p + ".resnets.0", x, groups, af)
    x = _attn(sd, p + ".attentions.0", x, groups)
    return _resnet(sd, p + ".resnets.1", x, groups, af)


@torch.no_grad()
def encode_moments(sd, cfg, x):
    """AutoencoderKL.encode(x) -> moments (mean | logvar) before DiagonalGaussianDistribution."""
    boc, L, G = cfg["block_out_channels"], cfg["layers_per_block"], cfg["norm_num_groups"]
    down_resample = list(reversed(cfg["up_rescale"]))      # af_api.py:42
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(boc)):
        for j in range(L):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, G, cfg["down_filtered_act"][i])
        if i != len(boc) - 1:
            w, b = sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"]
            if down_resample[i]:
                h = idf.af_downsample(h, w, b, padding=0)
            else:      # vanilla diffusers Downsample2D(padding=0): asymmetric (0,1,0,1) pad, stride 2
                h = F.conv2d(F.pad(h, (0, 1, 0, 1)), w, b, stride=2)
    h = _mid(sd, "encoder.mid_block", h, G, cfg["mid_act"])
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, G))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


@torch.no_grad()
def decode(sd, cfg, z):
    boc, L, G = cfg["block_out_channels"], cfg["layers_per_block"], cfg["norm_num_groups"]
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _mid(sd, "decoder.mid_block", h, G, cfg["mid_act"])
    for i in range(len(boc)):
        for j in range(L + 1):
            h = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, G, cfg["up_filtered_act"][i])
        if i != len(boc) - 1:
            w, b = sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"]
            if cfg["up_rescale"][i]:
                h = idf.af_upsample(h, w, b)
            else:
                h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, G))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
