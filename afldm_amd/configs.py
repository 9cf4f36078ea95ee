"""Model / scheduler configurations of the reference's FFHQ-256 alias-free LDM
(values restated from reference configs/ldm/model_unet.json:1-49 and
configs/ldm/noise_scheduler.json:1-14), plus a tiny same-topology config for tests."""

FFHQ_UNET_CONFIG = {
    "_class_name": "UNet2DModel",
    "act_fn": "silu",
    "add_attention": True,
    "attention_head_dim": 24,
    "attn_norm_num_groups": None,
    "block_out_channels": [192, 384, 384, 768, 768],
    "center_input_sample": False,
    "class_embed_type": None,
    "down_block_types": ["AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D",
                         "DownBlock2D"],
    "downsample_padding": 1,
    "downsample_type": "conv",
    "dropout": 0.0,
    "flip_sin_to_cos": True,
    "freq_shift": 0,
    "in_channels": 4,
    "layers_per_block": 2,
    "mid_block_scale_factor": 1,
    "norm_eps": 1e-05,
    "norm_num_groups": 32,
    "num_class_embeds": None,
    "num_train_timesteps": None,
    "out_channels": 4,
    "resnet_time_scale_shift": "default",
    "sample_size": 32,
    "time_embedding_dim": None,
    "time_embedding_type": "positional",
    "up_block_types": ["UpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D"],
    "upsample_type": "conv",
}

FFHQ_DDIM_CONFIG = {
    "_class_name": "DDIMScheduler",
    "beta_end": 0.0195,
    "beta_schedule": "scaled_linear",
    "beta_start": 0.0015,
    "clip_sample": False,
    "num_train_timesteps": 1000,
    "prediction_type": "epsilon",
    "set_alpha_to_one": False,
    "steps_offset": 1,
    "timestep_spacing": "leading",
    "trained_betas": None,
}


def tiny_unet_config(**over):
    cfg = dict(FFHQ_UNET_CONFIG)
    cfg.update(block_out_channels=[64, 128, 128],
               down_block_types=["AttnDownBlock2D", "AttnDownBlock2D", "DownBlock2D"],
               up_block_types=["UpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D"],
               attention_head_dim=16, sample_size=16, layers_per_block=1)
    cfg.update(over)
    return cfg


class FrozenConfig(dict):
    """dict with attribute access, like diffusers' FrozenDict: unknown JSON keys are kept (the
    reference's make_af_vae_from_config relies on that, af_api.py:63-67)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        raise AttributeError("config is frozen")
