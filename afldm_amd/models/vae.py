"""AutoencoderKL with the diffusers module surface (config, attribute names, state-dict keys,
encode / decode signatures) executing on libafldm_hip.so, plus the reference's
AliasFreeAutoencoderKL (afldm/models/af_vae.py:8-55).

Reference call sites: scripts/shift_ldm_ffhq.py:38-46 (`vae.encode(x).latent_dist.sample()`,
`vae.decode(z / scaling_factor, return_dict=False)[0]`), ldm_pipeline.py:117-119, and the surgery of
af_api.make_af_vae(_from_config) (af_api.py:34-67).  Activations are NHWC inside (see
models/blocks.py); the alias-free activations of the 64^2 / 128^2 levels run as three separable
MFMA passes (csrc/sep.hip) because a 2N x 2N plane no longer fits in LDS."""
import json
import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import ops
from ..configs import FrozenConfig
from ..utils import randn_tensor
from . import blocks as B


class DiagonalGaussianDistribution:
    """diffusers DiagonalGaussianDistribution (public NCHW tensors; a few elementwise ops)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator=None):
        noise = randn_tensor(self.mean.shape, generator=generator, device=self.parameters.device,
                             dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DownEncoderBlock2D(B._BlockBase):
    def __init__(self, in_channels, out_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, dropout=0.0,
                 add_downsample=True, downsample_padding=1):
        super().__init__()
        self.resnets = self._resnets([in_channels if i == 0 else out_channels for i in range(num_layers)], out_channels,
                                     None, resnet_eps, resnet_groups, dropout)
        self.downsamplers = (nn.ModuleList([B.Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                           padding=downsample_padding, name="op")])
                             if add_downsample else None)

    def forward(self, hidden_states):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
        return hidden_states


class UpDecoderBlock2D(B._BlockBase):
    def __init__(self, in_channels, out_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, dropout=0.0,
                 add_upsample=True):
        super().__init__()
        self.resnets = self._resnets([in_channels if i == 0 else out_channels for i in range(num_layers)], out_channels,
                                     None, resnet_eps, resnet_groups, dropout)
        self.upsamplers = (nn.ModuleList([B.Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def forward(self, hidden_states):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


def _norm_act_out(norm, h):
    gamma, beta = B.packed_norm(norm)
    stats = ops.gn_stats(h, norm.num_groups)
    return ops.gn_apply(h, stats, gamma, beta, norm.num_groups, norm.eps, act=1)      # conv_act: plain SiLU


class Encoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 block_out_channels=(64,), layers_per_block=2, norm_num_groups=32, act_fn="silu", double_z=True,
                 mid_block_add_attention=True):
        super().__init__()
        assert act_fn == "silu" and all(t == "DownEncoderBlock2D" for t in down_block_types)
        boc = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[0], kernel_size=3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(DownEncoderBlock2D(in_c, out_c, num_layers=layers_per_block, resnet_eps=1e-6,
                                                       resnet_groups=norm_num_groups,
                                                       add_downsample=i != len(boc) - 1, downsample_padding=0))
        self.mid_block = B.UNetMidBlock2D(in_channels=boc[-1], temb_channels=None, resnet_eps=1e-6,
                                          resnet_groups=norm_num_groups, add_attention=mid_block_add_attention,
                                          attention_head_dim=boc[-1])
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[-1], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)

    def forward(self, sample):
        h = B.conv_forward(self.conv_in, sample)
        for blk in self.down_blocks:
            h = blk(h)
        h = self.mid_block(h)
        return B.conv_forward(self.conv_out, _norm_act_out(self.conv_norm_out, h))


class Decoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,),
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", mid_block_add_attention=True):
        super().__init__()
        assert act_fn == "silu" and all(t == "UpDecoderBlock2D" for t in up_block_types)
        boc = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[-1], kernel_size=3, stride=1, padding=1)
        self.mid_block = B.UNetMidBlock2D(in_channels=boc[-1], temb_channels=None, resnet_eps=1e-6,
                                          resnet_groups=norm_num_groups, add_attention=mid_block_add_attention,
                                          attention_head_dim=boc[-1])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i in range(len(rev)):
            prev, out_c = out_c, rev[i]
            self.up_blocks.append(UpDecoderBlock2D(prev, out_c, num_layers=layers_per_block + 1, resnet_eps=1e-6,
                                                   resnet_groups=norm_num_groups, add_upsample=i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, sample, latent_embeds=None):
        h = B.conv_forward(self.conv_in, sample)
        h = self.mid_block(h)
        for blk in self.up_blocks:
            h = blk(h)
        return B.conv_forward(self.conv_out, _norm_act_out(self.conv_norm_out, h))


class AutoencoderKL(nn.Module):
    config_name = "config.json"

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215, shift_factor=None,
                 latents_mean=None, latents_std=None, force_upcast=True, use_quant_conv=True,
                 use_post_quant_conv=True, mid_block_add_attention=True, **extra):
        super().__init__()
        cfg = dict(in_channels=in_channels, out_channels=out_channels, down_block_types=list(down_block_types),
                   up_block_types=list(up_block_types), block_out_channels=list(block_out_channels),
                   layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                   norm_num_groups=norm_num_groups, sample_size=sample_size, scaling_factor=scaling_factor,
                   shift_factor=shift_factor, latents_mean=latents_mean, latents_std=latents_std,
                   force_upcast=force_upcast, use_quant_conv=use_quant_conv, use_post_quant_conv=use_post_quant_conv,
                   mid_block_add_attention=mid_block_add_attention)
        # unknown JSON keys (mid_act, down_filtered_act, up_filtered_act, up_rescale, ...) stay readable on
        # .config: make_af_vae_from_config depends on it (reference af_api.py:63-67)
        cfg.update(extra)
        object.__setattr__(self, "config", FrozenConfig(cfg))
        self.encoder = Encoder(in_channels, latent_channels, down_block_types, block_out_channels, layers_per_block,
                               norm_num_groups, act_fn, True, mid_block_add_attention)
        self.decoder = Decoder(latent_channels, out_channels, up_block_types, block_out_channels, layers_per_block,
                               norm_num_groups, act_fn, mid_block_add_attention)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1) if use_quant_conv else None
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1) if use_post_quant_conv else None
        self.up_block_types = list(up_block_types)        # read by scripts/shift_ldm_ffhq.py:60
        self.requires_grad_(False)

    # ---- diffusers-style plumbing
    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            model = cls.from_config(json.load(f))
        from safetensors.torch import load_file
        model.load_state_dict(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")))
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    @property
    def dtype(self):
        return self.encoder.conv_in.weight.dtype

    @property
    def device(self):
        return self.encoder.conv_in.weight.device

    def _apply(self, fn, *a, **k):
        B.invalidate_packed(self)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        B.invalidate_packed(self)
        return super().load_state_dict(*a, **k)

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("afldm_amd.AutoencoderKL runs on MI355X only (hand-written HIP kernels); there is "
                               "no CPU path (the CPU restatement in oracle/ is test infrastructure)")

    # ---- public API (NCHW in / out)
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        self._check(x)
        h = self.encoder(ops.to_nhwc(x.to(torch.float32).contiguous(), self.dtype))
        if self.quant_conv is not None:
            h = B.conv_forward(self.quant_conv, h)
        moments = ops.to_nchw(h).to(x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        posterior = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(latent_dist=posterior) if return_dict else (posterior,)

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        self._check(z)
        h = ops.to_nhwc(z.to(torch.float32).contiguous(), self.dtype)
        if self.post_quant_conv is not None:
            h = B.conv_forward(self.post_quant_conv, h)
        dec = ops.to_nchw(self.decoder(h)).to(z.dtype if z.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        return DecoderOutput(sample=dec) if return_dict else (dec,)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        return self.decode(z, return_dict=return_dict)
