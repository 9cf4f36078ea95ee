"""I2SBScheduler — surface of reference afldm/schedulers/i2sb_scheduler.py used by the SR
pipeline (tables :188-197, set_timesteps :224-300, step :382-459, add_noise :461-485,
compute_label :507-513, previous_timestep :518-531).  The schedule tables are host floats; `step`
on CUDA tensors runs the same flat update kernel as DDIM (afldm_ddim_step_flat):
  x0 = x - s_t eps;  x_prev = mu_x0 x0 + mu_xt x = (mu_x0 + mu_xt) x0 + mu_xt s_t eps."""
from dataclasses import dataclass

import numpy as np
import torch

from .. import ops
from ..configs import FrozenConfig
from ..utils import randn_tensor


def compute_gaussian_product_coef(sigma1, sigma2):
    denom = sigma1 ** 2 + sigma2 ** 2
    return sigma2 ** 2 / denom, sigma1 ** 2 / denom, (sigma1 ** 2 * sigma2 ** 2) / denom


@dataclass
class I2SBSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class _OdeSchedule:
    """Scheduler-shaped view of an I2SBScheduler's deterministic bridge for the graph-replayed engines (afldm_amd/engine.py): the
    timesteps of its evaluations and one (c0, c1, c2, c3) row per evaluation for afldm_ddim_step."""

    def __init__(self, sched, steps):
        self.sched, self.steps = sched, steps
        self.evaluations = steps - 1
        self.init_noise_sigma = 1.0
        self.config = FrozenConfig(dict(sched.config, _i2sb_ode_steps=steps))
        self._timesteps_host = []

    def set_timesteps(self, num_evaluations=None, device=None):
        assert num_evaluations in (None, self.evaluations), "the schedule is fixed by ode_schedule(num_inference_steps)"
        self.sched.set_timesteps(self.steps)
        self._timesteps_host = list(self.sched._timesteps_host[:self.evaluations])

    def coefficient_table(self, device):
        return torch.tensor([self.sched.ode_coefficients(t) for t in self._timesteps_host], dtype=torch.float32).to(device)


class I2SBScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False, **extra):
        cfg = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                   beta_schedule=beta_schedule, trained_betas=trained_betas, variance_type=variance_type,
                   clip_sample=clip_sample, prediction_type=prediction_type, thresholding=thresholding,
                   dynamic_thresholding_ratio=dynamic_thresholding_ratio, clip_sample_range=clip_sample_range,
                   sample_max_value=sample_max_value, timestep_spacing=timestep_spacing, steps_offset=steps_offset,
                   rescale_betas_zero_snr=rescale_betas_zero_snr)
        cfg.update(extra)
        self.config = FrozenConfig(cfg)
        if thresholding or rescale_betas_zero_snr or prediction_type != "epsilon":
            raise NotImplementedError("afldm_amd.I2SBScheduler covers the reference's configuration "
                                      "(epsilon prediction, no thresholding / zero-SNR rescale)")
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        self.std_fwd = torch.sqrt(torch.cumsum(self.betas, 0))
        self.std_bwd = torch.sqrt(torch.flip(torch.cumsum(torch.flip(self.betas, dims=[0]), 0), dims=[0]))
        self.mu_x0, self.mu_x1, var = compute_gaussian_product_coef(self.std_fwd, self.std_bwd)
        self.std_sb = torch.sqrt(var)
        self.init_noise_sigma = 1.0
        self.custom_timesteps = False
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        T = self.config.num_train_timesteps
        if timesteps is not None:
            if any(timesteps[i] >= timesteps[i - 1] for i in range(1, len(timesteps))):
                raise ValueError("`custom_timesteps` must be in descending order.")
            if timesteps[0] >= T:
                raise ValueError(f"`timesteps` must start before `self.config.train_timesteps`: {T}.")
            ts = np.array(timesteps, dtype=np.int64)
            self.custom_timesteps = True
        else:
            if num_inference_steps > T:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {T}")
            self.num_inference_steps = num_inference_steps
            self.custom_timesteps = False
            sp = self.config.timestep_spacing
            if sp == "linspace":
                ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
            elif sp == "leading":
                ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64)
                ts += self.config.steps_offset
            elif sp == "trailing":
                ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
            else:
                raise ValueError(f"{sp} is not supported.")
        self._timesteps_host = [int(t) for t in ts]
        self.timesteps = torch.from_numpy(ts).to(device)

    def previous_timestep(self, timestep):
        t = int(timestep)
        if self.custom_timesteps:
            idx = self._timesteps_host.index(t)
            return -1 if idx == len(self._timesteps_host) - 1 else self._timesteps_host[idx + 1]
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def ode_coefficients(self, timestep):
        """(c0, c1, c2, c3) of the deterministic, unclipped update in the linear form the DDIM kernels take -
        x0 = (x - c1 eps) / c0, x_prev = c2 x0 + c3 eps - i.e. x0 = x - s_t eps, x_prev = (mu_x0 + mu_xt) x0 + mu_xt s_t eps
        (reference i2sb_scheduler.py:382-459 with is_ode and without clip_sample)."""
        t, prev_t = int(timestep), self.previous_timestep(timestep)
        std_fwd, std_prev = self.std_fwd[t], self.std_fwd[prev_t]
        std_delta = (std_fwd ** 2 - std_prev ** 2).sqrt()
        mu_x0, mu_xt, _ = compute_gaussian_product_coef(std_prev, std_delta)
        return (1.0, float(std_fwd), float(mu_x0 + mu_xt), float(mu_xt * std_fwd))

    def ode_schedule(self, num_inference_steps):
        """The ODE bridge of `num_inference_steps` (num_inference_steps - 1 UNet evaluations: the reference loop leaves before
        its last timestep, i2sb_pipeline.py:48-50) as the (timesteps, coefficient table) object DenoiseEngine / the harness's
        CrossFrameSampler replay as HIP graphs; None when the configuration clips x0 (the clamp breaks the linear form)."""
        return None if self.config.clip_sample else _OdeSchedule(self, int(num_inference_steps))

    def step(self, model_output, timestep, sample, is_ode=False, generator=None, return_dict=True):
        if not sample.is_cuda:
            raise RuntimeError("afldm_amd.I2SBScheduler.step runs on MI355X tensors only (no CPU path)")
        t, prev_t = int(timestep), self.previous_timestep(timestep)
        std_fwd, std_prev = self.std_fwd[t], self.std_fwd[prev_t]
        std_delta = (std_fwd ** 2 - std_prev ** 2).sqrt()
        mu_x0, mu_xt, var = compute_gaussian_product_coef(std_prev, std_delta)
        x = sample.to(torch.float32).contiguous()
        e = model_output.to(torch.float32).contiguous()
        if self.config.clip_sample:
            # the clamp breaks the linear form: one extra elementwise pass (tensor plumbing)
            x0 = ops.ddim_step_flat(x, e, (1.0, float(std_fwd), 1.0, 0.0)).clamp_(-self.config.clip_sample_range,
                                                                              self.config.clip_sample_range)
            prev = float(mu_x0) * x0 + float(mu_xt) * x
        else:
            x0 = None
            prev = ops.ddim_step_flat(x, e, self.ode_coefficients(timestep))
        if t > 0 and not is_ode:
            prev = prev + randn_tensor(e.shape, generator=generator, device=e.device, dtype=e.dtype) * float(var.sqrt())
        prev = prev.to(sample.dtype)
        if not return_dict:
            return (prev,)
        return I2SBSchedulerOutput(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, x0, x1, timesteps, is_ode=False, noise=None):
        shape = (-1,) + (1,) * (x0.ndim - 1)
        ts = timesteps.to("cpu")
        xt = self.mu_x0[ts].view(shape).to(x0.device) * x0 + self.mu_x1[ts].view(shape).to(x0.device) * x1
        if not is_ode:
            noise = torch.randn_like(xt) if noise is None else noise
            xt = xt + self.std_sb[ts].view(shape).to(x0.device) * noise
        return xt

    def compute_label(self, timesteps, x0, xt):
        shape = (-1,) + (1,) * (x0.ndim - 1)
        return (xt - x0) / self.std_fwd[timesteps.to("cpu")].view(shape).to(x0.device)

    def __len__(self):
        return self.config.num_train_timesteps
