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

    def _inversion_rows(self):
        """(timestep, (mu_prev, sigma_prev, mu, sigma)) per step of the inversion over the scheduler's current timesteps, reversed:
        latent <- mu (latent - sigma_prev eps) / mu_prev + sigma eps (reference ldm_pipeline.py:133-160) - the DDIM update kernel's
        linear form with (sqrt a_t, sqrt(1 - a_t)) := (mu_prev, sigma_prev) and (sqrt a_prev, sqrt(1 - a_prev)) := (mu, sigma)."""
        ts = list(reversed(self.scheduler._timesteps_host))
        ac = self.scheduler.alphas_cumprod
        rows = []
        for i, t in enumerate(ts):
            a_t = ac[t]
            a_prev = ac[ts[i - 1]] if i > 0 else self.scheduler.final_alpha_cumprod
            rows.append((t, (float(a_prev ** 0.5), float((1 - a_prev) ** 0.5), float(a_t ** 0.5), float((1 - a_t) ** 0.5))))
        return rows

    @torch.no_grad()
    def ddim_inversion(self, latent, bar=True, use_graph=True):
        """Deterministic DDIM inversion over reversed timesteps (reference ldm_pipeline.py:133-160).  use_graph (fp32 latents on the
        GPU, plain attention processors): the loop replays DenoiseEngine's captured graphs over the inversion's coefficient rows;
        otherwise - bf16 latents, which the reference carries in their own dtype between steps, or a UNet with cross-frame
        processors installed - the eager loop below."""
        from .. import ops
        rows = self._inversion_rows()
        if (use_graph and latent.is_cuda and latent.dtype == torch.float32 and len(rows) >= 1
                and tuple(latent.shape[1:]) == (self.unet.config.in_channels, self.unet.config.sample_size, self.unet.config.sample_size)
                and all(type(m.processor).__name__ == "AttnProcessor2_0" for m in self.unet.modules() if hasattr(m, "processor"))):
            sched = _InversionSchedule(self.scheduler, rows)
            cfg_key = tuple(sorted((k, repr(v)) for k, v in dict(sched.config).items()))
            key = (latent.shape[0], len(rows), self.unet.dtype, str(self.unet.device), cfg_key)
            cache = self.__dict__.setdefault("_inv_engines", {})
            if key not in cache:
                cache.clear()
                cache[key] = DenoiseEngine(self.unet, sched, latent.shape[0], len(rows), use_graph=True)
            return cache[key].run(latent).to(latent.dtype)
        it = self.progress_bar(rows) if bar else rows
        for t, coef in it:
            eps = self.unet(latent, t).sample
            latent = ops.ddim_step_flat(latent.float().contiguous(), eps.float().contiguous(), coef).to(latent.dtype)
        return latent


class _InversionSchedule:
    """Scheduler-shaped view of a DDIM inversion for DenoiseEngine: its timesteps (ascending) and one coefficient row per step."""

    def __init__(self, scheduler, rows):
        self.rows = rows
        self.init_noise_sigma = 1.0
        self.config = dict(scheduler.config, _inversion_timesteps=tuple(t for t, _ in rows))
        self._timesteps_host = [t for t, _ in rows]

    def set_timesteps(self, n=None, device=None):
        assert n in (None, len(self.rows))

    def coefficient_table(self, device):
        return torch.tensor([c for _, c in self.rows], dtype=torch.float32).to(device)
