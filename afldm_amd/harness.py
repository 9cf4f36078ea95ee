"""Fractional-shift equivariance harness — the procedure of reference
scripts/shift_ldm_ffhq.py::shift_ldm (:49-159) on the MI355X implementation:

  1. install a CrossFrameAttnProcessor on every attention of the UNet;
  2. STORE pass: denoise the initial latent, remembering every attention's input per timestep;
  3. decode the result (the un-shifted image);
  4. for each offset tj = 1/r .. n/r latent pixels: ideal-crop shift the INITIAL latent, denoise
     it in LOAD mode (K/V from the stored maps), decode (masked), and stack
     [shifted output | bilinear-shifted un-shifted image | abs difference] vertically;
  5. restore the processors, write the frames as a GIF.

Multi-GPU: the shift offsets are independent given the STORE pass, so ranks take interleaved
offsets (each recomputes the cheap STORE pass locally) and the frames are gathered once.
Returns the frames and, per offset, the masked latent-space equivariance MSE."""
import torch

from . import parallel
from .io_utils import image_to_tensor, save_gif_from_tensors
from .pipelines.cross_frame_attn import (AttnState, CrossFrameAttnProcessor, get_unet_attn_processors,
                                         set_unet_attn_processor)
from .shift_utils.metrics import mask_mse
from .shift_utils.shifters import ImageShifter
from .utils import randn_tensor


def vae_encode(vae, x):
    return vae.encode(x).latent_dist.sample() * vae.config.scaling_factor


def vae_decode(vae, x):
    return vae.decode(x / vae.config.scaling_factor, return_dict=False)[0]


@torch.no_grad()
def shift_ldm(pipeline, num_inference_steps=50, num_shift_steps=16, output_path="results/shift_ldm.gif",
              input_path=None, generator=None, rank=0, world=1, batch_offsets=True, reference_exact=True):
    """batch_offsets: the LOAD passes of this rank's offsets run as ONE batch (samples are independent; the
    cross-frame K/V of the stored pass is shared by the whole batch) instead of the reference's one B = 1
    sampler run per offset (shift_ldm_ffhq.py:124-151) - 50 UNet evaluations instead of 50 per offset.

    reference_exact=True (the default) follows the reference flow to the letter; reference_exact=False
    (scripts: --fixed_resize) opts into two deliberate deviations:
    * `input_path`: the image is resized to sample_size * VAE ratio (256) so that its latent has the UNet's
      sample_size; the reference resizes to (sample_size, sample_size) = 32 x 32 BEFORE the VAE
      (shift_ldm_ffhq.py:110-113), i.e. inverts a 4 x 4 latent.
    * the initial noise is drawn on the CPU (device independent) instead of on the GPU (:118-122)."""
    device = pipeline.device
    vae, unet, scheduler = pipeline.vae, pipeline.unet, pipeline.scheduler
    pipeline.set_progress_bar_config(disable=True)
    ratio = 2 ** (len(vae.up_block_types) - 1) if vae is not None else 8
    latent_shifter = ImageShifter("ideal_crop", ratio)
    image_shifter = ImageShifter()

    attn_state = AttnState()
    previous = get_unet_attn_processors(unet)
    set_unet_attn_processor(unet, {k: CrossFrameAttnProcessor(attn_state) for k in previous})

    def denoise(latents):
        latents = latents.to(device)
        scheduler.set_timesteps(num_inference_steps, device=device)
        for t in scheduler.timesteps:
            attn_state.set_timestep(t)
            eps = unet(scheduler.scale_model_input(latents, t), t, return_dict=False)[0]
            latents = scheduler.step(eps, t, latents, eta=0, return_dict=False)[0]
        return latents

    try:
        if input_path is not None:
            size = unet.config.sample_size * (1 if reference_exact else ratio)
            tensor = vae_encode(vae, image_to_tensor(input_path, (size, size)).to(device))
            scheduler.set_timesteps(num_inference_steps, device=device)
            init_latent = pipeline.ddim_inversion(tensor, bar=False)
        else:
            # CPU-side draw (seedable, device independent) — the reference draws on the GPU
            # (shift_ldm_ffhq.py:118-122), which is not reproducible across devices
            shape = (1, unet.config.in_channels, unet.config.sample_size, unet.config.sample_size)
            if reference_exact:
                init_latent = randn_tensor(shape, device=device, generator=generator)
            else:
                init_latent = randn_tensor(shape, generator=generator).to(device)
        attn_state.reset()
        denoised = denoise(init_latent)
        attn_state.to_load()
        rec_img = vae_decode(vae, denoised) if vae is not None else None

        offsets = torch.linspace(1 / ratio, num_shift_steps / ratio, num_shift_steps)
        mine = parallel.interleaved(num_shift_steps, rank, world)
        frames, errors = {}, {}
        shifted = {i: latent_shifter.shift(init_latent, 0, float(offsets[i])) for i in mine}
        if batch_offsets and len(mine) > 1:
            den_all = denoise(torch.cat([shifted[i][0] for i in mine], 0))
            dens = {i: den_all[k:k + 1] for k, i in enumerate(mine)}
        else:
            dens = {i: denoise(shifted[i][0]) for i in mine}
        for i in mine:
            tj = float(offsets[i])
            den, mask = dens[i], shifted[i][1]
            ref_lat, _ = latent_shifter.shift(denoised, 0, tj)
            errors[i] = float(mask_mse(den, ref_lat, mask))
            if vae is not None:
                gt, _ = image_shifter.shift(rec_img, 0, tj * ratio)
                img = vae_decode(vae, den * mask)
                frames[i] = torch.cat((img, gt, torch.abs(img - gt)), -2).float().cpu()
    finally:
        set_unet_attn_processor(unet, dict(previous))

    frames, errors = parallel.gather_indexed(world, frames, errors)
    ordered = [frames[i] for i in sorted(frames)]
    if ordered and rank == 0 and output_path:
        save_gif_from_tensors(ordered, output_path, denorm=True)
    return ordered, [errors[i] for i in sorted(errors)]


def vae_encode_mode(vae, x):
    """reference scripts/shift_ldm_sr.py:31-34 (the SR harness encodes with the posterior mode)."""
    return vae.encode(x).latent_dist.mode() * vae.config.scaling_factor


@torch.no_grad()
def shift_ldm_sr(pipeline, num_inference_steps=50, num_shift_steps=16, output_path="results/shift_ldm_sr.gif",
                 input_path=None, image=None, rank=0, world=1, batch_offsets=True):
    """Fractional-shift equivariance of x4 super-resolution with I2SB - the flow of reference
    scripts/shift_ldm_sr.py:43-150: degrade (bicubic x1/4, nearest x4), VAE-encode, denoise with
    cross-frame attention STORE, then for every offset shift the initial latent, denoise in LOAD mode
    and compare with the shifted reconstruction.  `image` ([1,3,H,W] in [-1,1]) replaces
    `input_path` when given; the offsets are sharded across ranks."""
    from .af_libs.superresolution import build_sr4x
    device = pipeline.device
    vae, unet, scheduler = pipeline.vae, pipeline.unet, pipeline.scheduler
    pipeline.set_progress_bar_config(disable=True)
    ratio = 2 ** (len(vae.up_block_types) - 1)
    size = unet.config.sample_size * ratio
    sr_func = build_sr4x(device, "bicubic", size)
    latent_shifter = ImageShifter("ideal_crop", ratio)
    image_shifter = ImageShifter()

    attn_state = AttnState()
    previous = get_unet_attn_processors(unet)
    set_unet_attn_processor(unet, {k: CrossFrameAttnProcessor(attn_state) for k in previous})

    def denoise(latents):
        latents = latents.to(device)
        scheduler.set_timesteps(num_inference_steps, device=device)
        ts = scheduler.timesteps
        for i, t in enumerate(ts):
            if i == num_inference_steps - 1:
                break
            attn_state.set_timestep(t)
            eps = unet(scheduler.scale_model_input(latents, t), t, return_dict=False)[0]
            latents = scheduler.step(eps, t, latents, is_ode=True, generator=None).prev_sample
        return latents

    try:
        if image is None:
            image = image_to_tensor(input_path, (size, size))
        tensor = sr_func(image.to(device).float()).clip(-1, 1)
        init_latent = vae_encode_mode(vae, tensor.to(vae.dtype)).to(unet.dtype)
        attn_state.reset()
        denoised = denoise(init_latent)
        attn_state.to_load()
        rec_img = vae_decode(vae, denoised)
        offsets = torch.linspace(1 / ratio, num_shift_steps / ratio, num_shift_steps)
        frames, errors = {}, {}
        mine = parallel.interleaved(num_shift_steps, rank, world)
        shifts = {i: latent_shifter.shift(init_latent, 0, float(offsets[i])) for i in mine}
        if batch_offsets and len(mine) > 1:      # one batched LOAD pass for this rank's offsets (see shift_ldm)
            den_all = denoise(torch.cat([shifts[i][0] for i in mine], 0))
            dens = {i: den_all[k:k + 1] for k, i in enumerate(mine)}
        else:
            dens = {i: denoise(shifts[i][0]) for i in mine}
        for i in mine:
            tj = float(offsets[i])
            (shifted, mask), den = shifts[i], dens[i]
            ref_lat, _ = latent_shifter.shift(denoised, 0, tj)
            errors[i] = float(mask_mse(den, ref_lat, mask))
            gt, _ = image_shifter.shift(rec_img, 0, tj * ratio)
            img_in = vae_decode(vae, shifted * mask)
            img = vae_decode(vae, den * mask)
            frames[i] = torch.cat((img_in, img, gt, torch.abs(img - gt)), -2).float().cpu()
    finally:
        set_unet_attn_processor(unet, dict(previous))

    frames, errors = parallel.gather_indexed(world, frames, errors)
    ordered = [frames[i] for i in sorted(frames)]
    if ordered and rank == 0 and output_path:
        save_gif_from_tensors(ordered, output_path, denorm=True)
    return ordered, [errors[i] for i in sorted(errors)]

