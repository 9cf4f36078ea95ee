"""MyLDMPipeline — surface of reference afldm/pipelines/ldm_pipeline.py:17-160 on MI355X.

`__call__(output_type='latent')` is the throughput entry point: the DDIM loop runs as a
replayed HIP graph (afldm_amd.engine.DenoiseEngine); other output types decode through the
(alias-free) AutoencoderKL of afldm_amd.models.vae."""
import inspect
import json
import os

import torch

from ..engine import DenoiseEngine
from ..models.unet_2d import UNet2DModel
from ..schedulers.ddim import DDIMScheduler
from ..utils import randn_tensor
from .pipeline_utils import DiffusionPipeline, ImagePipelineOutput


class MyLDMPipeline(DiffusionPipeline):
    def __init__(self, vae, unet: UNet2DModel, scheduler: DDIMScheduler):
        super().__init__()
        self.register_modules(vae=vae, unet=unet, scheduler=scheduler)
        self._engines = {}

    @classmethod
    def from_pretrained(cls, path, **kw):
        """Local diffusers-format directory: unet/ (config.json + safetensors), scheduler/
        (scheduler_config.json), optional vae/.  There is no network access on the target boxes."""
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (afldm_amd cannot download checkpoints)")
        unet = UNet2DModel.from_pretrained(path, subfolder="unet")
        with open(os.path.join(path, "scheduler", "scheduler_config.json")) as f:
            scheduler = DDIMScheduler.from_config(json.load(f))
        vae = None
        if os.path.isdir(os.path.join(path, "vae")):
            from ..models.vae import AutoencoderKL
            vae = AutoencoderKL.from_pretrained(path, subfolder="vae")
        return cls(vae, unet, scheduler)

    def _engine(self, batch, steps, use_graph):
        # keyed on the scheduler's CONFIG (the coefficient / timestep tables are a function of it), not on the
        # scheduler object: __call__ re-creates the scheduler every time (like the reference, ldm_pipeline.py:80),
        # and an identity key made every call rebuild the engine and re-capture its HIP graphs
        cfg_key = tuple(sorted((k, repr(v)) for k, v in dict(self.scheduler.config).items()))
        key = (batch, steps, use_graph, self.unet.dtype, str(self.unet.device), cfg_key)
        if key not in self._engines:
            self._engines = {key: DenoiseEngine(self.unet, self.scheduler, batch, steps, use_graph)}
        return self._engines[key]

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, eta=0.0, num_inference_steps=50, latents=None,
                 output_type="pil", return_dict=True, use_graph=True, **kwargs):
        self.scheduler = DDIMScheduler.from_config(self.scheduler.config)
        if latents is None:
            latents = randn_tensor((batch_size, self.unet.config.in_channels, self.unet.config.sample_size,
                                    self.unet.config.sample_size), generator=generator)
        if eta != 0.0:
            # stochastic DDIM (reference ldm_pipeline.py:96-109 forwards eta to scheduler.step): the per-step noise comes
            # from the caller's generator on the host, so this is the eager loop, not the captured graph
            self.scheduler.set_timesteps(num_inference_steps)
            latents = latents.to(device=self.unet.device, dtype=self.unet.dtype)
            for t in self.progress_bar(self.scheduler._timesteps_host):
                eps = self.unet(latents, t).sample
                latents = self.scheduler.step(eps, t, latents, eta=eta, generator=generator).prev_sample
            return self._deliver(latents, output_type, return_dict)
        eng = self._engine(latents.shape[0], num_inference_steps, use_graph)
        eng.scheduler = self.scheduler
        latents = eng.run(latents).to(self.unet.dtype)
        return self._deliver(latents, output_type, return_dict)

    def _deliver(self, latents, output_type, return_dict):
        """'latent' -> the latents; 'pt' -> decoded tensor in [-1, 1]; 'np' / 'pil' -> [0, 1] NHWC arrays / PIL images
        (wrapped in ImagePipelineOutput unless return_dict is False)."""
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise NotImplementedError("this pipeline was built without a VAE: use output_type='latent'")
        decoded = self.vae.decode(latents.to(self.vae.dtype) / self.vae.config.scaling_factor).sample
        if output_type == "pt":
            return decoded
        arrays = (decoded / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).float().numpy()
        images = self.numpy_to_pil(arrays) if output_type == "pil" else arrays
        return ImagePipelineOutput(images=images) if return_dict else (images,)

    @torch.no_grad()
    def ddim_inversion(self, latent, bar=True):
        """Deterministic DDIM inversion over reversed timesteps (reference ldm_pipeline.py:133-160)."""
        from .. import ops
        ts = list(reversed(self.scheduler._timesteps_host))
        it = self.progress_bar(ts) if bar else ts
        ac = self.scheduler.alphas_cumprod
        for i, t in enumerate(it):
            a_t = ac[t]
            a_prev = ac[ts[i - 1]] if i > 0 else self.scheduler.final_alpha_cumprod
            mu, mu_prev = a_t ** 0.5, a_prev ** 0.5
            sigma, sigma_prev = (1 - a_t) ** 0.5, (1 - a_prev) ** 0.5
            eps = self.unet(latent, t).sample
            # latent <- mu * (latent - sigma_prev eps) / mu_prev + sigma eps : the DDIM update kernel
            # with (sqrt a_t, sqrt(1-a_t)) := (mu_prev, sigma_prev) and (sqrt a_prev, ..) := (mu, sigma)
            latent = ops.ddim_step_flat(latent.float().contiguous(), eps.float().contiguous(),
                                        (float(mu_prev), float(sigma_prev), float(mu), float(sigma))).to(latent.dtype)
        return latent
