"""I2SBLDMPipeline — surface of reference afldm/pipelines/i2sb_pipeline.py:17-78: the degraded
image is VAE-encoded and used as the INITIAL latent of the (unconditional) alias-free UNet; the
loop breaks before the last timestep (num_inference_steps - 1 UNet evaluations)."""
import torch

from ..schedulers.i2sb import I2SBScheduler
from .ldm_pipeline import MyLDMPipeline


class I2SBLDMPipeline(MyLDMPipeline):
    def __init__(self, vae, unet, scheduler: I2SBScheduler):
        super().__init__(vae, unet, scheduler)

    @torch.no_grad()
    def __call__(self, images, generator=None, is_ode=False, num_inference_steps=50, output_type="pil",
                 return_dict=True, reference_exact=True, latent_dtype=torch.float32, use_graph=True, **kwargs):
        """images: [B, 3, H, W] tensor in [-1, 1] (the reference's VaeImageProcessor.preprocess leaves
        such tensors unchanged).  Like the reference (i2sb_pipeline.py:43) the posterior sample of the start latent is
        drawn by latent_dist.sample() WITHOUT the caller's generator; reference_exact=False opts into drawing it from
        `generator` (seeded runs become reproducible end to end).

        latent_dtype: the dtype the latent is CARRIED in between UNet evaluations.  Default fp32 - a deliberate deviation
        from the reference for bf16 / fp16 UNets, where the reference stores the latent in the UNet's dtype
        (i2sb_pipeline.py:41) and loses 0.11 rel-RMS over the 99 evaluations to storage rounding alone
        (tests/golden/g16_r04_floor.npz); with an fp32 UNet both are the same thing.  `latent_dtype=None` reproduces the
        reference's storage (the latent stays in the UNet's dtype)."""
        if self.vae is None:
            raise NotImplementedError("I2SBLDMPipeline needs a VAE to encode the degraded image")
        start = self.vae.encode(images.to(device=self.device, dtype=self.unet.dtype)).latent_dist.sample(
            None if reference_exact else generator)
        latents = self._bridge(start * self.vae.config.scaling_factor, num_inference_steps, is_ode, generator,
                               latent_dtype=latent_dtype, use_graph=use_graph)
        return self._deliver(latents, output_type, return_dict)

    def _ode_engine(self, batch, steps):
        """DenoiseEngine over the deterministic bridge (scheduler.ode_schedule): the same captured-graph loop the DDIM sampler
        uses - the unclipped ODE update is the DDIM kernel's linear form with another coefficient row per evaluation."""
        from ..engine import DenoiseEngine
        ode = self.scheduler.ode_schedule(steps)
        cfg_key = tuple(sorted((k, repr(v)) for k, v in dict(ode.config).items()))
        key = (batch, steps, self.unet.dtype, str(self.unet.device), cfg_key)
        cache = self.__dict__.setdefault("_ode_engines", {})
        if key not in cache:
            cache.clear()
            cache[key] = DenoiseEngine(self.unet, ode, batch, ode.evaluations, use_graph=True)
        return cache[key]

    def _bridge(self, latents, steps, is_ode, generator, latent_dtype=torch.float32, use_graph=True):
        """steps - 1 UNet evaluations from the encoded degraded image towards the clean latent: the reference loop
        leaves before its last timestep (i2sb_pipeline.py:48-50).  The deterministic, unclipped bridge with the latent carried
        in fp32 (is_ode, clip_sample off - what scripts/shift_ldm_sr.py runs) replays captured HIP graphs (use_graph); the
        stochastic / clipped forms and latent_dtype=None run the eager loop below."""
        sched, unet = self.scheduler, self.unet
        if (use_graph and is_ode and not sched.config.clip_sample and latent_dtype == torch.float32 and latents.is_cuda
                and steps >= 2):
            return self._ode_engine(latents.shape[0], steps).run(latents).to(latents.dtype)
        sched.set_timesteps(steps)
        # The latent is carried in fp32 BETWEEN evaluations whatever the UNet's dtype (as DenoiseEngine does for DDIM): a
        # step of the 100-step bridge moves the latent by about one bf16 ulp, and a latent stored in bf16 - what the
        # reference does with a bf16 UNet (i2sb_pipeline.py:41; scheduler.step returns the sample's dtype) - loses 0.11
        # rel-RMS over the 99 evaluations to storage rounding alone (oracle measurement: tests/golden/g16_r04_floor.npz).
        # The UNet still sees its own dtype; the result is returned in the caller's dtype.  latent_dtype=None: the
        # reference's storage (the latent keeps the dtype it came in with).
        dtype = latents.dtype
        if latent_dtype is not None:
            latents = latents.to(latent_dtype)
        for t in self.progress_bar(sched._timesteps_host[:steps - 1]):
            prediction = unet(sched.scale_model_input(latents, t).to(unet.dtype), t).sample
            latents = sched.step(prediction, t, latents, is_ode=is_ode, generator=generator).prev_sample
        return latents.to(dtype)
