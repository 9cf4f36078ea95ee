"""I2SBLDMPipeline — surface of reference afldm/pipelines/i2sb_pipeline.py:17-78: the degraded
image is VAE-encoded and used as the INITIAL latent of the (unconditional) alias-free UNet; the
loop breaks before the last timestep (num_inference_steps - 1 UNet evaluations)."""
import torch

from ..schedulers.i2sb import I2SBScheduler
from .ldm_pipeline import MyLDMPipeline
from .pipeline_utils import ImagePipelineOutput


class I2SBLDMPipeline(MyLDMPipeline):
    def __init__(self, vae, unet, scheduler: I2SBScheduler):
        super().__init__(vae, unet, scheduler)

    @torch.no_grad()
    def __call__(self, images, generator=None, is_ode=False, num_inference_steps=50, output_type="pil",
                 return_dict=True, **kwargs):
        """images: [B, 3, H, W] tensor in [-1, 1] (the reference's VaeImageProcessor.preprocess leaves
        such tensors unchanged)."""
        if self.vae is None:
            raise NotImplementedError("I2SBLDMPipeline needs a VAE to encode the degraded image")
        x = images.to(device=self.device, dtype=self.unet.dtype)
        latents = self.vae.encode(x).latent_dist.sample(generator) * self.vae.config.scaling_factor
        self.scheduler.set_timesteps(num_inference_steps)
        ts = self.scheduler._timesteps_host
        for i, t in enumerate(self.progress_bar(ts)):
            if i == num_inference_steps - 1:
                break
            eps = self.unet(self.scheduler.scale_model_input(latents, t), t).sample
            latents = self.scheduler.step(eps, t, latents, is_ode=is_ode, generator=generator).prev_sample
        if output_type == "latent":
            return latents
        image = self.vae.decode(latents.to(self.vae.dtype) / self.vae.config.scaling_factor).sample
        if output_type != "pt":
            image = (image / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).float().numpy()
            if output_type == "pil":
                image = self.numpy_to_pil(image)
            return ImagePipelineOutput(images=image) if return_dict else (image,)
        return image
