"""Oracle: functional CPU fp32 restatement of diffusers' UNet2DModel.forward after
the reference's alias-free surgery (make_af_unet, reference af_api.py:70-83).

diffusers is a third-party dependency of the reference that is neither vendored nor
installed here; the block semantics below are restated from diffusers 0.32.1
 -> "parity unpinned" for these parts (see oracle/__init__.py).
The alias-free pieces call oracle.ideal_filters, which IS pinned to the reference.

Parameters live in a flat dict with diffusers' state-dict key names, so the same
dict loads into the product's afldm_amd UNet2DModel.  Test infrastructure.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ideal_filters as idf


# --------------------------------------------------------------------------- topology
def unet_topology(cfg):
    """Enumerate blocks exactly as diffusers UNet2DModel.__init__ does."""
    boc = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    downs = []
    out_c = boc[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        final = i == len(boc) - 1
        resnets = [(in_c if j == 0 else out_c, out_c) for j in range(L)]
        downs.append(dict(type=typ, resnets=resnets, attn=typ.startswith("Attn"),
                          channels=out_c, downsample=not final))
    ups = []
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, typ in enumerate(cfg["up_block_types"]):
        prev_out, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        final = i == len(boc) - 1
        resnets = []
        for j in range(L + 1):
            skip_c = in_c if j == L else out_c
            rin = prev_out if j == 0 else out_c
            resnets.append((rin + skip_c, out_c))
        ups.append(dict(type=typ, resnets=resnets, attn=typ.startswith("Attn"),
                        channels=out_c, upsample=not final))
    return downs, ups


# --------------------------------------------------------------------------- parameter init
def init_unet_params(cfg, seed=0, conv_out_scale=1.0):
    """Seeded PyTorch-default init for every conv/linear; GN affine=(1,0)
    (SURVEY.md 8(d) 'Synthetic inputs').  Keys = diffusers state-dict names."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def _uniform(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def conv(name, cin, cout, k):
        fan_in = cin * k * k
        bound = 1.0 / math.sqrt(fan_in)         # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), ..)
        sd[name + ".weight"] = _uniform((cout, cin, k, k), bound)
        sd[name + ".bias"] = _uniform((cout,), bound)

    def linear(name, cin, cout):
        bound = 1.0 / math.sqrt(cin)
        sd[name + ".weight"] = _uniform((cout, cin), bound)
        sd[name + ".bias"] = _uniform((cout,), bound)

    def gn(name, c):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)

    boc = cfg["block_out_channels"]
    temb = boc[0] * 4

    def resnet(prefix, cin, cout):
        gn(prefix + ".norm1", cin)
        conv(prefix + ".conv1", cin, cout, 3)
        linear(prefix + ".time_emb_proj", temb, cout)
        gn(prefix + ".norm2", cout)
        conv(prefix + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(prefix + ".conv_shortcut", cin, cout, 1)

    def attn(prefix, c):
        gn(prefix + ".group_norm", c)
        for p in ("to_q", "to_k", "to_v", "to_out.0"):
            linear(prefix + "." + p, c, c)

    conv("conv_in", cfg["in_channels"], boc[0], 3)
    linear("time_embedding.linear_1", boc[0], temb)
    linear("time_embedding.linear_2", temb, temb)
    downs, ups = unet_topology(cfg)
    for i, b in enumerate(downs):
        for j, (cin, cout) in enumerate(b["resnets"]):
            resnet(f"down_blocks.{i}.resnets.{j}", cin, cout)
            if b["attn"]:
                attn(f"down_blocks.{i}.attentions.{j}", cout)
        if b["downsample"]:
            conv(f"down_blocks.{i}.downsamplers.0.conv", b["channels"], b["channels"], 3)
    cm = boc[-1]
    resnet("mid_block.resnets.0", cm, cm)
    if cfg.get("add_attention", True):
        attn("mid_block.attentions.0", cm)
    resnet("mid_block.resnets.1", cm, cm)
    for i, b in enumerate(ups):
        for j, (cin, cout) in enumerate(b["resnets"]):
            resnet(f"up_blocks.{i}.resnets.{j}", cin, cout)
            if b["attn"]:
                attn(f"up_blocks.{i}.attentions.{j}", cout)
        if b["upsample"]:
            conv(f"up_blocks.{i}.upsamplers.0.conv", b["channels"], b["channels"], 3)
    gn("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg["out_channels"], 3)
    if conv_out_scale != 1.0:       # keeps a random-weight DDIM trajectory bounded
        sd["conv_out.weight"] *= conv_out_scale
        sd["conv_out.bias"] *= conv_out_scale
    return sd


def randomize_norm_affine(sd, seed=1):
    """Perturb GN affine params so tests exercise gamma/beta (default init is (1,0))."""
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if ".norm" in k or "group_norm" in k or k.startswith("conv_norm_out"):
            if k.endswith(".weight"):
                sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
            elif k.endswith(".bias"):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return sd


# --------------------------------------------------------------------------- pieces
def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    """diffusers get_timestep_embedding."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class AttnCache:
    """State of the reference's cross-frame attention hook
    (AttnState + CrossFrameAttnProcessor.maps, cross_frame_attn.py:6-64)."""
    STORE, LOAD, IDLE = 0, 1, 2

    def __init__(self, enable_interp=False):
        self.state = AttnCache.IDLE
        self.timestep = 0
        self.maps = {}          # site -> {t: pre-norm NCHW map}            (store_id 0)
        self.maps1 = {}         # site -> {t: pre-norm NCHW map}            (store_id 1, cross_frame_attn.py:64)
        self.store_id = 0       # AttnState.store_id (:29-31): which of the two map sets a STORE pass fills
        self.alpha = 0.0        # AttnState.alpha (:33-35)
        self.enable_interp = enable_interp   # CrossFrameAttnProcessor(enable_interp=...) (:61-64)


class _Ctx:
    def __init__(self, sd, cfg, af, cache, taps):
        self.sd, self.cfg, self.af, self.cache, self.taps = sd, cfg, af, cache, taps
        self.groups = cfg["norm_num_groups"]
        self.eps = cfg["norm_eps"]

    def tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x.detach().clone()


def _gn(c, name, x):
    return F.group_norm(x, c.groups, c.sd[name + ".weight"], c.sd[name + ".bias"], c.eps)


def _act(c, x):
    # ResnetBlock2D.nonlinearity after wrap_resblock_nonlinearity (af_api.py:28-31):
    # WarpedNonlinearity for 4-D tensors, plain SiLU for the 2-D temb (af_blocks.py:20-21)
    if c.af:
        return idf.warped_nonlinearity(x, F.silu)
    return F.silu(x)


def resnet_block(c, prefix, x, emb):
    sd = c.sd
    h = _gn(c, prefix + ".norm1", x)
    h = _act(c, h)
    h = F.conv2d(h, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"], padding=1)
    t = F.silu(emb)
    t = F.linear(t, sd[prefix + ".time_emb_proj.weight"], sd[prefix + ".time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = _gn(c, prefix + ".norm2", h)
    h = _act(c, h)
    h = F.conv2d(h, sd[prefix + ".conv2.weight"], sd[prefix + ".conv2.bias"], padding=1)
    if prefix + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[prefix + ".conv_shortcut.weight"], sd[prefix + ".conv_shortcut.bias"])
    out = x + h
    c.tap(prefix, out)
    return out


def _cached_kv_tokens(c, prefix, m, b, ch, hw):
    """cross_frame_attn.py:82-98 (and :103-117 for the second map): stored pre-norm NCHW map ->
    [n, hw, c] tokens -> attn.group_norm -> batch repeat up to the current batch."""
    n0 = m.shape[0]
    m = m.view(n0, ch, hw)
    m = F.group_norm(m, c.groups, c.sd[prefix + ".group_norm.weight"], c.sd[prefix + ".group_norm.bias"],
                     c.eps).transpose(1, 2)
    if n0 < b:
        m = m.unsqueeze(1).repeat(1, b // n0, 1, 1).reshape(b, hw, ch)
    return m


def _attention_core(c, prefix, x, kv_src):
    """AttnProcessor2_0.__call__ for the deprecated attention-block configuration (group norm on the
    flattened input, q from the input, k/v from `kv_src` tokens or the normed input, SDPA, to_out,
    residual connection, rescale_output_factor 1)."""
    sd, cfg = c.sd, c.cfg
    b, ch, hh, ww = x.shape
    heads = ch // cfg["attention_head_dim"]
    h = x.view(b, ch, hh * ww)
    h = F.group_norm(h, c.groups, sd[prefix + ".group_norm.weight"],
                     sd[prefix + ".group_norm.bias"], c.eps).transpose(1, 2)
    q = F.linear(h, sd[prefix + ".to_q.weight"], sd[prefix + ".to_q.bias"])
    src = h if kv_src is None else kv_src
    k = F.linear(src, sd[prefix + ".to_k.weight"], sd[prefix + ".to_k.bias"])
    v = F.linear(src, sd[prefix + ".to_v.weight"], sd[prefix + ".to_v.bias"])
    d = ch // heads
    q = q.view(b, -1, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(b, -1, ch)
    o = F.linear(o, sd[prefix + ".to_out.0.weight"], sd[prefix + ".to_out.0.bias"])
    o = o.transpose(-1, -2).reshape(b, ch, hh, ww)
    return o + x


def attention_block(c, prefix, x):
    """diffusers Attention (deprecated attn-block config) via AttnProcessor2_0, with the
    reference's CrossFrameAttnProcessor semantics (cross_frame_attn.py:66-130), including the
    `enable_interp` blend of two stored passes (:100-122)."""
    b, ch, hh, ww = x.shape
    kv_src = None
    if c.cache is not None and c.cache.state != AttnCache.IDLE:
        t = c.cache.timestep
        if c.cache.state == AttnCache.STORE:
            maps = c.cache.maps if c.cache.store_id == 0 else c.cache.maps1
            maps.setdefault(prefix, {})[t] = x.detach().clone()
        else:   # LOAD: K/V from the cached pre-norm map of the unshifted pass, group-normed
            kv_src = _cached_kv_tokens(c, prefix, c.cache.maps[prefix][t], b, ch, hh * ww)
            if c.cache.enable_interp:
                kv1 = _cached_kv_tokens(c, prefix, c.cache.maps1[prefix][t], b, ch, hh * ww)
                alpha = c.cache.alpha
                out = (1 - alpha) * _attention_core(c, prefix, x, kv_src) + alpha * _attention_core(c, prefix, x, kv1)
                c.tap(prefix, out)
                return out
    out = _attention_core(c, prefix, x, kv_src)
    c.tap(prefix, out)
    return out


def downsample(c, prefix, x):
    w, b = c.sd[prefix + ".conv.weight"], c.sd[prefix + ".conv.bias"]
    if c.af:
        return idf.af_downsample(x, w, b, padding=c.cfg["downsample_padding"])
    return F.conv2d(x, w, b, stride=2, padding=c.cfg["downsample_padding"])


def upsample(c, prefix, x):
    w, b = c.sd[prefix + ".conv.weight"], c.sd[prefix + ".conv.bias"]
    if c.af:
        return idf.af_upsample(x, w, b)
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return F.conv2d(x, w, b, padding=1)


# --------------------------------------------------------------------------- forward
@torch.no_grad()
def unet_forward(sd, cfg, sample, timestep, af=True, cache=None, taps=None):
    """UNet2DModel.forward(sample, timestep)[0] after make_af_unet (af=True)
    or of the vanilla model (af=False)."""
    c = _Ctx(sd, cfg, af, cache, taps)
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.long)
    elif timestep.ndim == 0:
        timestep = timestep[None]
    timesteps = timestep * torch.ones(sample.shape[0], dtype=timestep.dtype)
    boc = cfg["block_out_channels"]
    t_emb = timestep_embedding(timesteps, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    emb = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.silu(emb)
    emb = F.linear(emb, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    c.tap("emb", emb)

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    c.tap("conv_in", x)
    skips = [x]
    downs, ups = unet_topology(cfg)
    for i, b in enumerate(downs):
        for j in range(len(b["resnets"])):
            x = resnet_block(c, f"down_blocks.{i}.resnets.{j}", x, emb)
            if b["attn"]:
                x = attention_block(c, f"down_blocks.{i}.attentions.{j}", x)
            skips.append(x)
        if b["downsample"]:
            x = downsample(c, f"down_blocks.{i}.downsamplers.0", x)
            c.tap(f"down_blocks.{i}.downsamplers.0", x)
            skips.append(x)
    x = resnet_block(c, "mid_block.resnets.0", x, emb)
    if cfg.get("add_attention", True):
        x = attention_block(c, "mid_block.attentions.0", x)
    x = resnet_block(c, "mid_block.resnets.1", x, emb)
    for i, b in enumerate(ups):
        for j in range(len(b["resnets"])):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(c, f"up_blocks.{i}.resnets.{j}", x, emb)
            if b["attn"]:
                x = attention_block(c, f"up_blocks.{i}.attentions.{j}", x)
        if b["upsample"]:
            x = upsample(c, f"up_blocks.{i}.upsamplers.0", x)
            c.tap(f"up_blocks.{i}.upsamplers.0", x)
    x = F.group_norm(x, c.groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], c.eps)
    x = F.silu(x)          # unet.conv_act is NOT wrapped by make_af_unet (af_api.py:70-83)
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return x


def attention_sites(cfg):
    """Attention module paths in forward order (= get_unet_attn_processors keys minus '.processor')."""
    downs, ups = unet_topology(cfg)
    sites = []
    for i, b in enumerate(downs):
        if b["attn"]:
            sites += [f"down_blocks.{i}.attentions.{j}" for j in range(len(b["resnets"]))]
    if cfg.get("add_attention", True):
        sites.append("mid_block.attentions.0")
    for i, b in enumerate(ups):
        if b["attn"]:
            sites += [f"up_blocks.{i}.attentions.{j}" for j in range(len(b["resnets"]))]
    return sites
