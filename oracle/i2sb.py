"""Oracle: I2SBScheduler tables / step / add_noise / compute_label, restated from reference
afldm/schedulers/i2sb_scheduler.py:131-139 (Gaussian product), :188-197 (tables), :382-459 (step),
:461-485 (add_noise), :507-513 (compute_label), :518-531 (previous_timestep).  The reference file
imports diffusers for its config / output plumbing only; oracle/gen_golden.py part i imports it under a
plumbing-only stub of those names and records tables, timesteps, step (ODE + seeded stochastic, clip_sample on / off),
add_noise and compute_label: tests/test_oracle_golden.py::test_i2sb_oracle_vs_reference_file asserts this restatement
bit-equal to those fixtures (tests/golden/g15_r04_refpins.npz) - PINNED to the reference file.  Test infrastructure."""
import numpy as np
import torch

from .configs import FFHQ_DDIM


def gaussian_product_coef(sigma1, sigma2):
    denom = sigma1 ** 2 + sigma2 ** 2
    return sigma2 ** 2 / denom, sigma1 ** 2 / denom, (sigma1 ** 2 * sigma2 ** 2) / denom


class I2SB:
    def __init__(self, cfg=None, clip_sample=False):
        cfg = dict(FFHQ_DDIM if cfg is None else cfg)
        self.cfg = cfg
        self.clip_sample = clip_sample
        T = cfg["num_train_timesteps"]
        assert cfg["beta_schedule"] == "scaled_linear"
        self.betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, T, dtype=torch.float32) ** 2
        self.std_fwd = torch.sqrt(torch.cumsum(self.betas, 0))
        self.std_bwd = torch.sqrt(torch.flip(torch.cumsum(torch.flip(self.betas, dims=[0]), 0), dims=[0]))
        self.mu_x0, self.mu_x1, var = gaussian_product_coef(self.std_fwd, self.std_bwd)
        self.std_sb = torch.sqrt(var)
        self.num_inference_steps = None

    def set_timesteps(self, n):
        T = self.cfg["num_train_timesteps"]
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64) + self.cfg["steps_offset"]
        self.timesteps = torch.from_numpy(ts)

    def previous_timestep(self, t):
        """i2sb_scheduler.py:518-531, default (non-custom) timesteps."""
        n = self.num_inference_steps if self.num_inference_steps else self.cfg["num_train_timesteps"]
        return int(t) - self.cfg["num_train_timesteps"] // n

    def step(self, eps, t, sample, is_ode=True, generator=None, return_x0=False):
        """i2sb_scheduler.py:382-459.  is_ode=True is what the reference scripts pass (shift_ldm_sr.py); the
        stochastic branch (:444-450) draws randn(shape, generator) * sqrt(var) for t > 0."""
        t = int(t)
        prev_t = self.previous_timestep(t)
        std_fwd, std_prev = self.std_fwd[t], self.std_fwd[prev_t]
        std_delta = (std_fwd ** 2 - std_prev ** 2).sqrt()
        x0 = sample - std_fwd * eps
        if self.clip_sample:
            x0 = x0.clamp(-1.0, 1.0)
        mu_x0, mu_xt, var = gaussian_product_coef(std_prev, std_delta)
        prev = mu_x0 * x0 + mu_xt * sample
        if t > 0 and not is_ode:
            prev = prev + torch.randn(eps.shape, generator=generator, dtype=eps.dtype) * var.sqrt()
        return (prev, x0) if return_x0 else prev

    def add_noise(self, x0, x1, timesteps, noise=None, is_ode=False):
        shape = (-1,) + (1,) * (x0.ndim - 1)
        xt = self.mu_x0[timesteps].view(shape) * x0 + self.mu_x1[timesteps].view(shape) * x1
        if not is_ode:
            xt = xt + self.std_sb[timesteps].view(shape) * noise
        return xt

    def compute_label(self, timesteps, x0, xt):
        return (xt - x0) / self.std_fwd[timesteps].view((-1,) + (1,) * (x0.ndim - 1))
