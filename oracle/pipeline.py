"""Oracle: DDIM sampling loop and the fractional-shift equivariance harness.
Restated from reference afldm/pipelines/ldm_pipeline.py:80-112 (sampling) and
scripts/shift_ldm_ffhq.py:85-151 (STORE pass, shifted LOAD passes), latent-space only
(the VAE is an §8(f) 'next' row).  Test infrastructure."""
import time

import torch

from .ddim import DDIM
from .shift import mask_mse, shift_ideal
from .unet import AttnCache, unet_forward


@torch.no_grad()
def ddim_sample(sd, cfg, latents, num_inference_steps=50, af=True, cache=None, ddim_cfg=None,
                return_traj=False):
    """MyLDMPipeline.__call__(latents=..., output_type='latent') (ldm_pipeline.py:80-112)."""
    sched = DDIM(ddim_cfg)
    latents = latents * sched.init_noise_sigma
    sched.set_timesteps(num_inference_steps)
    traj = []
    for t in sched.timesteps:
        if cache is not None:
            cache.timestep = int(t)
        eps = unet_forward(sd, cfg, sched.scale_model_input(latents, t), t, af=af, cache=cache)
        latents = sched.step(eps, t, latents, eta=0.0)
        if return_traj:
            traj.append(latents.clone())
    return (latents, traj) if return_traj else latents


@torch.no_grad()
def ddim_inversion(sd, cfg, latent, num_inference_steps=50, af=True, ddim_cfg=None):
    """MyLDMPipeline.ddim_inversion (ldm_pipeline.py:133-160): deterministic DDIM inversion over the
    REVERSED timesteps of the scheduler's current schedule.  `reversed(tensor)` is `tensor.flip(0)`, so
    `timesteps[i - 1]` (:144-147) is the next-SMALLER timestep; step 0 uses final_alpha_cumprod.
        pred_x0 = (latent - sigma_prev eps) / mu_prev ;  latent = mu pred_x0 + sigma eps     (:156-158)
    The caller must have called scheduler.set_timesteps (shift_ldm_ffhq.py:114-116 does)."""
    sched = DDIM(ddim_cfg)
    sched.set_timesteps(num_inference_steps)
    timesteps = sched.timesteps.flip(0)
    for i, t in enumerate(timesteps):
        alpha_prod_t = sched.alphas_cumprod[int(t)]
        alpha_prod_t_prev = sched.alphas_cumprod[int(timesteps[i - 1])] if i > 0 else sched.final_alpha_cumprod
        mu, mu_prev = alpha_prod_t ** 0.5, alpha_prod_t_prev ** 0.5
        sigma, sigma_prev = (1 - alpha_prod_t) ** 0.5, (1 - alpha_prod_t_prev) ** 0.5
        eps = unet_forward(sd, cfg, latent, t, af=af)
        pred_x0 = (latent - sigma_prev * eps) / mu_prev
        latent = mu * pred_x0 + sigma * eps
    return latent


@torch.no_grad()
def shift_equivariance(sd, cfg, init_latent, offsets, num_inference_steps=50, ratio=8, af=True,
                       cross_frame=True):
    """Latent-space core of shift_ldm (shift_ldm_ffhq.py:124-151): one STORE pass on the
    unshifted latent, then for each offset tj a LOAD pass on the ideal_crop-shifted latent.
    Returns the unshifted result, and per offset (denoised_shifted, mask, mask_mse vs the
    ideal-shifted unshifted result)."""
    cache = AttnCache() if cross_frame else None
    if cache is not None:
        cache.state = AttnCache.STORE
    base = ddim_sample(sd, cfg, init_latent, num_inference_steps, af=af, cache=cache)
    if cache is not None:
        cache.state = AttnCache.LOAD
    out = []
    for tj in offsets:
        shifted, mask = shift_ideal(init_latent, 0.0, tj, ratio, crop=True)
        den = ddim_sample(sd, cfg, shifted, num_inference_steps, af=af, cache=cache)
        ref, _ = shift_ideal(base, 0.0, tj, ratio, crop=True)
        out.append(dict(tj=float(tj), latent=den, mask=mask,
                        mse=float(mask_mse(den, ref, mask))))
    return base, out


def time_denoise_steps(sd, cfg, batch=1, steps=2, threads=None, seed=1234, budget_s=None):
    """cpu_baseline leg of bench.py: (seconds per UNet+scheduler step, steps timed) on the host cores.
    With budget_s the loop stops once that much wall time has been spent (at least 2, at most `steps`)."""
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(batch, cfg["in_channels"], cfg["sample_size"], cfg["sample_size"], generator=g)
    sched = DDIM()
    sched.set_timesteps(50)
    steps = min(steps, 49)
    ts = sched.timesteps[:steps + 1]
    eps = unet_forward(sd, cfg, lat, ts[0])        # warm-up (FFT plans, oneDNN primitives)
    t0 = time.perf_counter()
    done = 0
    for t in ts[1:]:
        eps = unet_forward(sd, cfg, lat, t)
        lat = sched.step(eps, t, lat)
        done += 1
        if budget_s is not None and done >= 2 and time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    return dt / done, done
