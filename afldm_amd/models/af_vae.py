"""AliasFreeAutoencoderKL — reference afldm/models/af_vae.py:8-55: an AutoencoderKL whose
constructor applies make_af_vae with the four alias-free switches of its config."""
from ..af_modules.af_api import make_af_vae
from .vae import AutoencoderKL


class AliasFreeAutoencoderKL(AutoencoderKL):
    def __init__(self, *args, mod_mid_act=True, down_filtered_act=(True, True, True, True),
                 up_filtered_act=(True, True, True, True), up_rescale=(True, True, True), **kwargs):
        super().__init__(*args, mod_mid_act=mod_mid_act, down_filtered_act=list(down_filtered_act),
                         up_filtered_act=list(up_filtered_act), up_rescale=list(up_rescale), **kwargs)
        make_af_vae(self, mod_mid_act, down_filtered_act, up_filtered_act, up_rescale)

    @property
    def downsample_ratio(self):
        return 2 ** (len(self.config.block_out_channels) - 1)

    def encode_scale(self, x):
        return self.encode(x).latent_dist.sample() * self.config.scaling_factor

    def decode_scale(self, x):
        return self.decode(x / self.config.scaling_factor).sample
