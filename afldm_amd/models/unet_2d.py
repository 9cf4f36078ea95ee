"""UNet2DModel with the diffusers module surface (config, attribute names, state-dict keys,
forward signature) whose forward runs on hand-written HIP kernels (libafldm_hip.so).

Reference call sites this serves: afldm/pipelines/ldm_pipeline.py:106 (`self.unet(x, t).sample`),
scripts/shift_ldm_ffhq.py:98-102 (`unet(x, t, return_dict=False)[0]`) and the in-place
surgery of afldm/af_modules/af_api.py:70-83.
"""
import json
import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import ops, trunk
from ..configs import FrozenConfig
from . import blocks as B


@dataclass
class UNet2DOutput:
    sample: torch.Tensor


class UNet2DModel(nn.Module):
    config_name = "config.json"

    def __init__(self, sample_size=None, in_channels=3, out_channels=3, center_input_sample=False,
                 time_embedding_type="positional", time_embedding_dim=None, freq_shift=0, flip_sin_to_cos=True,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1,
                 downsample_padding=1, downsample_type="conv", upsample_type="conv", dropout=0.0, act_fn="silu",
                 attention_head_dim=8, norm_num_groups=32, attn_norm_num_groups=None, norm_eps=1e-5,
                 resnet_time_scale_shift="default", add_attention=True, class_embed_type=None, num_class_embeds=None,
                 num_train_timesteps=None, **extra):
        super().__init__()
        cfg = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                   center_input_sample=center_input_sample, time_embedding_type=time_embedding_type,
                   time_embedding_dim=time_embedding_dim, freq_shift=freq_shift, flip_sin_to_cos=flip_sin_to_cos,
                   down_block_types=list(down_block_types), up_block_types=list(up_block_types),
                   block_out_channels=list(block_out_channels), layers_per_block=layers_per_block,
                   mid_block_scale_factor=mid_block_scale_factor, downsample_padding=downsample_padding,
                   downsample_type=downsample_type, upsample_type=upsample_type, dropout=dropout, act_fn=act_fn,
                   attention_head_dim=attention_head_dim, norm_num_groups=norm_num_groups,
                   attn_norm_num_groups=attn_norm_num_groups, norm_eps=norm_eps,
                   resnet_time_scale_shift=resnet_time_scale_shift, add_attention=add_attention,
                   class_embed_type=class_embed_type, num_class_embeds=num_class_embeds,
                   num_train_timesteps=num_train_timesteps)
        cfg.update({k: v for k, v in extra.items()})       # unknown keys are kept on .config, like diffusers
        object.__setattr__(self, "config", FrozenConfig(cfg))
        if (time_embedding_type != "positional" or act_fn != "silu" or downsample_type != "conv"
                or upsample_type != "conv" or resnet_time_scale_shift != "default" or class_embed_type is not None
                or mid_block_scale_factor != 1 or center_input_sample):
            raise NotImplementedError("afldm_amd.UNet2DModel covers the configuration family of the AF-LDM "
                                      "checkpoints (positional time embedding, conv resampling, SiLU)")
        assert len(down_block_types) == len(up_block_types) == len(block_out_channels)
        boc = list(block_out_channels)
        temb_dim = time_embedding_dim or boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_proj = B.Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = B.TimestepEmbedding(boc[0], temb_dim)

        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, typ in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            final = i == len(boc) - 1
            self.down_blocks.append(B.DOWN_BLOCKS[typ](
                in_channels=in_c, out_channels=out_c, temb_channels=temb_dim, num_layers=layers_per_block,
                resnet_eps=norm_eps, resnet_groups=norm_num_groups, dropout=dropout, add_downsample=not final,
                downsample_padding=downsample_padding,
                attention_head_dim=attention_head_dim if attention_head_dim is not None else out_c))
        self.mid_block = B.UNetMidBlock2D(
            in_channels=boc[-1], temb_channels=temb_dim, dropout=dropout, resnet_eps=norm_eps,
            resnet_groups=norm_num_groups, attn_groups=attn_norm_num_groups, add_attention=add_attention,
            attention_head_dim=attention_head_dim if attention_head_dim is not None else boc[-1])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, typ in enumerate(up_block_types):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            self.up_blocks.append(B.UP_BLOCKS[typ](
                in_channels=in_c, prev_output_channel=prev, out_channels=out_c, temb_channels=temb_dim,
                num_layers=layers_per_block + 1, resnet_eps=norm_eps, resnet_groups=norm_num_groups, dropout=dropout,
                add_upsample=not final,
                attention_head_dim=attention_head_dim if attention_head_dim is not None else out_c))
        groups_out = norm_num_groups if norm_num_groups is not None else min(boc[0] // 4, 32)
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=groups_out, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, kernel_size=3, padding=1)
        self.requires_grad_(False)          # inference engine: no autograd anywhere on this path

    # ------------------------------------------------------------------ diffusers-style plumbing
    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        """Load a diffusers-format directory (config.json + diffusion_pytorch_model.safetensors)."""
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            model = cls.from_config(json.load(f))
        from safetensors.torch import load_file
        fn = os.path.join(d, "diffusion_pytorch_model.safetensors")
        model.load_state_dict(load_file(fn))
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "UNet2DModel"
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _apply(self, fn, *a, **k):
        B.invalidate_packed(self)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        B.invalidate_packed(self)
        return super().load_state_dict(*a, **k)

    # ------------------------------------------------------------------ time embedding (once per step)
    def _resnets_in_order(self):
        for blk in self.down_blocks:
            yield from blk.resnets
        yield from self.mid_block.resnets
        for blk in self.up_blocks:
            yield from blk.resnets

    def _temb_weights(self, dtype):
        """All 27 time_emb_proj layers as ONE [sum(Cout), temb_dim] GEMM (packed once)."""
        cache = self.__dict__.setdefault("_afldm_cache", {})
        key = ("temb", dtype)
        if key not in cache:
            rs = list(self._resnets_in_order())
            w = torch.cat([r.time_emb_proj.weight.detach().float() for r in rs], 0)
            b = torch.cat([r.time_emb_proj.bias.detach().float() for r in rs], 0).contiguous()
            offs, o = [], 0
            for r in rs:
                offs.append(o)
                o += r.out_channels
            cache[key] = (ops.pack_weight(w, dtype), b, offs, o)
        return cache[key]

    def time_embed(self, timestep, batch):
        """-> (per-resnet list of (temb slice view, stride)), emb.  `timestep`: python number,
        0-dim / 1-elem tensor (shared by the batch -> 1 row, broadcast) or a [B] tensor."""
        dev, dtype = self.device, self.dtype
        if torch.is_tensor(timestep):
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        else:
            t = torch.full((1,), float(timestep), dtype=torch.float32, device=dev)
        assert t.numel() in (1, batch), "timestep must be a scalar or have one value per sample"
        t_emb = self.time_proj(t, dtype)
        emb = self.time_embedding(t_emb)                                   # [rows, temb_dim]
        w, b, offs, total = self._temb_weights(dtype)
        proj = ops.conv2d(ops.silu(emb), w, b)                            # [rows, sum Cout]
        return self.temb_slices(proj, t.numel() > 1), emb

    def temb_slices(self, proj, per_sample=False):
        """per-resnet (view into `proj` [rows, sum Cout], row stride) pairs."""
        _, _, offs, total = self._temb_weights(self.dtype)
        stride = total if per_sample else 0
        return [(proj.view(-1)[o:] if stride == 0 else proj[:, o:], stride) for o in offs]

    def temb_projection(self, timestep):
        """[1, sum Cout] time_emb_proj outputs of one shared timestep (what DenoiseEngine tabulates per step)."""
        dev, dtype = self.device, self.dtype
        t = torch.full((1,), float(timestep), dtype=torch.float32, device=dev)
        emb = self.time_embedding(self.time_proj(t, dtype))
        w, b, _, _ = self._temb_weights(dtype)
        return ops.conv2d(ops.silu(emb), w, b)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward_nhwc(self, x, timestep, temb_slices=None):
        """x: NHWC [B, H, W, in_channels] in the model dtype -> NHWC [B, H, W, out_channels].
        temb_slices: the per-resnet time_emb_proj slices when the caller already has them (DenoiseEngine's table)."""
        slices = temb_slices if temb_slices is not None else self.time_embed(timestep, x.shape[0])[0]
        it = iter(slices)

        def take(n):
            return [next(it) for _ in range(n)]

        h = B.conv_forward(self.conv_in, x, want_stats=True)
        skips = (h,)
        up_blocks = list(self.up_blocks)
        for bi, blk in enumerate(self.down_blocks):
            if bi == len(self.down_blocks) - 1 and trunk.eligible(self, h):
                # the 2x2 level (this block, the mid block, the resnets of up_blocks[0]) as ONE cooperative launch
                first, ups = self.down_blocks[-1], up_blocks[0]
                n_lvl = len(first.resnets) + len(self.mid_block.resnets) + len(ups.resnets)
                sl = take(n_lvl)
                tr = trunk.get(self, h, self._temb_weights(self.dtype)[2])
                skips = skips[:-1]                                # the level's input is the skip its last resnet consumes
                h = tr.run(h, sl[0][0].data_ptr(), sl[0][1])
                for u in (ups.upsamplers or ()):
                    h = u(h)
                up_blocks = up_blocks[1:]
                break
            h, outs = blk(h, take(len(blk.resnets)))
            skips += outs
        else:
            h = self.mid_block(h, take(len(self.mid_block.resnets)))
        for blk in up_blocks:
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            h = blk(h, res, take(n))
        gamma, beta = B.packed_norm(self.conv_norm_out)
        gno = self.conv_norm_out
        wo, bo = B.packed_conv(self.conv_out, h.dtype)
        fused = ops.conv_out_fused(h, wo, bo, gamma, beta, gno.num_groups, gno.eps)      # norm -> SiLU -> conv in one launch
        if fused is not None:
            return fused
        stats = ops.gn_stats(h, gno.num_groups)
        h = ops.gn_apply(h, stats, gamma, beta, gno.num_groups, gno.eps, act=1)          # conv_act is plain SiLU
        return B.conv_forward(self.conv_out, h)

    @torch.no_grad()
    def forward(self, sample, timestep, class_labels=None, return_dict=True):
        if not sample.is_cuda:
            raise RuntimeError("afldm_amd.UNet2DModel runs on MI355X only (hand-written HIP kernels); "
                               "move the model and inputs to 'cuda'.  The CPU restatement lives in oracle/ "
                               "and is test infrastructure, not a fallback.")
        assert class_labels is None
        x = ops.to_nhwc(sample.to(torch.float32).contiguous(), self.dtype)
        y = self.forward_nhwc(x, timestep)
        out = ops.to_nchw(y).to(sample.dtype if sample.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        if not return_dict:
            return (out,)
        return UNet2DOutput(sample=out)
