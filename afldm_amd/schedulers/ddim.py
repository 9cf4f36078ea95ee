"""DDIMScheduler with the diffusers surface the reference uses (ldm_pipeline.py:80-109,
scripts/shift_ldm_ffhq.py:85-106): from_config, set_timesteps, timesteps, scale_model_input,
step, alphas_cumprod, final_alpha_cumprod, init_noise_sigma, config.

The schedule tables are a few hundred host floats; `step` on CUDA tensors launches the HIP
update kernel (afldm_ddim_step_flat).  eta = 0 / epsilon prediction / no clipping — the
reference's configuration (configs/ldm/noise_scheduler.json:1-14); other settings raise."""
from dataclasses import dataclass

import numpy as np
import torch

from .. import ops
from ..utils import randn_tensor
from ..configs import FFHQ_DDIM_CONFIG, FrozenConfig


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", timestep_spacing="leading", **extra):
        cfg = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                   beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
                   set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
                   timestep_spacing=timestep_spacing)
        cfg.update(extra)
        self.config = FrozenConfig(cfg)
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError(f"num_inference_steps {num_inference_steps} > num_train_timesteps {T}")
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            ratio = T / num_inference_steps
            ts = np.round(np.arange(T, 0, -ratio)).astype(np.int64) - 1
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(sp)
        self._timesteps_host = [int(t) for t in ts]
        self.timesteps = torch.from_numpy(ts).to(device)

    def coefficients(self, timestep):
        """(sqrt a_t, sqrt(1-a_t), sqrt a_prev, sqrt(1-a_prev)) in fp32 arithmetic, as floats."""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))

    def _variance(self, timestep):
        """diffusers DDIMScheduler._get_variance: (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev), as a float."""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return float((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))

    def coefficient_table(self, device):
        """float32 [nsteps, 4] device table for the graph-replayed loop (afldm_ddim_step)."""
        rows = [self.coefficients(t) for t in self._timesteps_host]
        return torch.tensor(rows, dtype=torch.float32).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        if self.config.clip_sample or self.config.prediction_type != "epsilon":
            raise NotImplementedError("afldm_amd.DDIMScheduler implements the reference's setting: "
                                      "epsilon prediction, clip_sample=False")
        if not sample.is_cuda:
            raise RuntimeError("afldm_amd.DDIMScheduler.step runs on MI355X tensors only (no CPU path)")
        c = self.coefficients(timestep)
        x = sample.to(torch.float32).contiguous()
        e = model_output.to(torch.float32).contiguous()
        if eta == 0.0:
            prev = ops.ddim_step_flat(x, e, c).to(sample.dtype)
        else:
            # diffusers DDIMScheduler.step with eta > 0 (the reference forwards `eta`, ldm_pipeline.py:38,96-101):
            # sigma_t = eta sqrt((1 - a_prev) / (1 - a_t) (1 - a_t / a_prev)); the direction coefficient becomes
            # sqrt(1 - a_prev - sigma_t^2) and sigma_t * noise is added (noise drawn like diffusers: randn_tensor with the
            # caller's generator, on the CPU for a CPU generator)
            sigma = float(eta) * self._variance(timestep) ** 0.5
            a_prev = c[2] * c[2]
            prev = ops.ddim_step_flat(x, e, (c[0], c[1], c[2], float(max(1.0 - a_prev - sigma * sigma, 0.0) ** 0.5)))
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` "
                                 "or `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
            prev = (prev + sigma * variance_noise.to(torch.float32)).to(sample.dtype)
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev)

    def __len__(self):
        return self.config.num_train_timesteps


def ffhq_ddim_scheduler():
    return DDIMScheduler.from_config(FFHQ_DDIM_CONFIG)
