"""In-place alias-free surgery on afldm_amd's diffusers-compatible models.

Public functions, argument meaning and the set of modules touched are those of the reference's
afldm/af_modules/af_api.py (make_af_unet :70-83, make_af_vae :34-60, make_af_vae_from_config
:63-67, make_af_controlnet :86-93, replace_* :13-25, wrap_* :9-31): every down/up sampler is
swapped for its alias-free subclass (re-using the original conv's parameters) and every
ResnetBlock2D.nonlinearity is wrapped in WarpedNonlinearity.  NOT touched, as in the
reference: unet.conv_act, time_embedding.act, attention.  No parameter is added, so
state-dict keys stay vanilla diffusers.
"""
from .af_blocks import AliasFreeDownsample2D, AliasFreeUpsample2D, WarpedNonlinearity

__all__ = ["wrap_nonlinearity", "replace_upsampler", "replace_downsampler", "wrap_resblock_nonlinearity",
           "make_af_unet", "make_af_vae", "make_af_vae_from_config", "make_af_controlnet"]


def wrap_nonlinearity(nonlinearity):
    return WarpedNonlinearity(nonlinearity)


def replace_upsampler(ori_upsampler):
    u = ori_upsampler
    return AliasFreeUpsample2D(u.channels, u.use_conv, out_channels=u.out_channels, ori_conv=u.conv)


def replace_downsampler(ori_downsampler):
    d = ori_downsampler
    return AliasFreeDownsample2D(d.channels, d.use_conv, out_channels=d.out_channels, padding=d.padding,
                                 ori_conv=d.conv)


def wrap_resblock_nonlinearity(block):
    for resnet in block.resnets:
        resnet.nonlinearity = wrap_nonlinearity(resnet.nonlinearity)


def _convert(blocks, sampler_attr, replace, resample_flags=None, act_flags=None):
    """Shared walker: blocks[i].<sampler_attr>[0] -> alias-free version when resample_flags[i]
    (default: always), and its resnets' activations wrapped when act_flags[i] (default: always)."""
    for i, block in enumerate(blocks):
        samplers = getattr(block, sampler_attr, None)
        if samplers is not None and (resample_flags is None or resample_flags[i]):
            samplers[0] = replace(samplers[0])
        if act_flags is None or act_flags[i]:
            wrap_resblock_nonlinearity(block)


def make_af_unet(unet):
    _convert(unet.down_blocks, "downsamplers", replace_downsampler)
    wrap_resblock_nonlinearity(unet.mid_block)
    _convert(unet.up_blocks, "upsamplers", replace_upsampler)


def make_af_controlnet(model):
    _convert(model.down_blocks, "downsamplers", replace_downsampler)
    wrap_resblock_nonlinearity(model.mid_block)


def make_af_vae(vae, mod_mid_act=True, mod_down_filtered_act=(True, True, True, True),
                mod_up_filtered_act=(True, True, True, True), mod_resampling_layer=(True, True, True)):
    # encoder level i pairs with decoder level (last - i): the resampling flags are given in
    # decoder order, hence reversed for the encoder (reference af_api.py:42)
    enc_resample = list(reversed(list(mod_resampling_layer)))
    _convert(vae.encoder.down_blocks, "downsamplers", replace_downsampler, enc_resample + [False],
             mod_down_filtered_act)
    if mod_mid_act:
        wrap_resblock_nonlinearity(vae.encoder.mid_block)
        wrap_resblock_nonlinearity(vae.decoder.mid_block)
    _convert(vae.decoder.up_blocks, "upsamplers", replace_upsampler, list(mod_resampling_layer) + [False],
             mod_up_filtered_act)


def make_af_vae_from_config(vae):
    cfg = vae.config
    make_af_vae(vae, mod_mid_act=cfg.mid_act, mod_down_filtered_act=cfg.down_filtered_act,
                mod_up_filtered_act=cfg.up_filtered_act, mod_resampling_layer=cfg.up_rescale)
