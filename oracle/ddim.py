"""Oracle: DDIMScheduler restatement (diffusers; config restated from reference
configs/ldm/noise_scheduler.json:1-14; math per SURVEY.md Appendix C).  Parity
unpinned w.r.t. diffusers itself; anchored on the Appendix-C known-answer values
(tests/test_oracle_golden.py).  Test infrastructure."""
import numpy as np
import torch

from .configs import FFHQ_DDIM


class DDIM:
    def __init__(self, cfg=None):
        cfg = dict(FFHQ_DDIM if cfg is None else cfg)
        self.cfg = cfg
        T = cfg["num_train_timesteps"]
        if cfg["beta_schedule"] == "scaled_linear":
            self.betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, T,
                                        dtype=torch.float32) ** 2
        elif cfg["beta_schedule"] == "linear":
            self.betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], T, dtype=torch.float32)
        else:
            raise NotImplementedError(cfg["beta_schedule"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = (torch.tensor(1.0) if cfg["set_alpha_to_one"]
                                    else self.alphas_cumprod[0])
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, T)[::-1].copy().astype(np.int64))

    def set_timesteps(self, n):
        T = self.cfg["num_train_timesteps"]
        self.num_inference_steps = n
        assert self.cfg["timestep_spacing"] == "leading"
        step_ratio = T // n
        ts = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.cfg["steps_offset"]
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, eps, t, sample, eta=0.0, generator=None):
        t = int(t)
        prev_t = t - self.cfg["num_train_timesteps"] // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        x0 = (sample - beta_t ** 0.5 * eps) / a_t ** 0.5
        assert not self.cfg["clip_sample"]
        # diffusers 0.32.1 scheduling_ddim.py step(): sigma_t = eta sqrt(variance), direction sqrt(1 - a_prev - sigma_t^2) eps,
        # + sigma_t * randn(generator) for eta > 0 (parity unpinned like the rest of the diffusers restatement)
        variance = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std_dev_t = eta * variance ** 0.5
        direction = (1 - a_prev - std_dev_t ** 2) ** 0.5 * eps
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            prev = prev + std_dev_t * torch.randn(eps.shape, generator=generator, dtype=eps.dtype)
        return prev

    def coefficients(self, t):
        """(sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)) as python floats."""
        t = int(t)
        prev_t = t - self.cfg["num_train_timesteps"] // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float(a_t ** 0.5), float((1 - a_t) ** 0.5),
                float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))
