"""DenoiseEngine — the 50-step DDIM loop of the reference (afldm/pipelines/ldm_pipeline.py:103-109)
as ONE captured HIP graph replayed per step.

The reference loop costs, per step, ~1100 dependent kernel launches from Python plus a
`t.item()` device sync.  Here everything the step needs lives in device
memory — the timestep table, the DDIM coefficient table and a step counter the update kernel
increments itself — so a replay needs no host value and no synchronisation:

    step++; t <- t_table[step]                    (afldm_select_timestep, counter starts at -1)
    x_nhwc <- NCHW fp32 latents                   (afldm_nchw_to_nhwc)
    eps    <- UNet(x_nhwc, t)                     (~450 HIP kernels, all from libafldm_hip.so)
    lat    <- DDIM(lat, eps, coef[step])          (afldm_ddim_step, in place)
"""
import os

import torch

from . import ops


class DenoiseEngine:
    def __init__(self, unet, scheduler, batch_size, num_inference_steps=50, use_graph=True, steps_per_graph=5):
        if unet.device.type != "cuda":
            raise RuntimeError("DenoiseEngine needs the UNet on an MI355X ('cuda') device; there is no CPU path")
        self.unet, self.scheduler = unet, scheduler
        self.B, self.n = batch_size, num_inference_steps
        dev = unet.device
        c, s = unet.config.in_channels, unet.config.sample_size
        scheduler.set_timesteps(num_inference_steps)
        self.timesteps = list(scheduler._timesteps_host)
        self.t_table = torch.tensor(self.timesteps, dtype=torch.float32).to(dev)
        self.coef = scheduler.coefficient_table(dev).reshape(-1).contiguous()
        self.step_idx = torch.full((1,), -1, dtype=torch.int32, device=dev)
        self.t_cur = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lat = torch.zeros(batch_size, c, s, s, dtype=torch.float32, device=dev)
        self.x_nhwc = torch.empty(batch_size, s, s, c, dtype=unet.dtype, device=dev)
        self.graph = None
        self.graph_multi = None
        # a replay boundary costs ~9 us of idle GPU (host launch + graph start): besides the one-step graph a
        # `steps_per_graph`-step graph is captured and used for every full group of steps
        self.steps_per_graph = max(1, int(os.environ.get("AFLDM_STEPS_PER_GRAPH", steps_per_graph)))
        self.use_graph = use_graph
        self.kernels_per_step = None

    # one denoise step, entirely stream-ordered
    def _step(self):
        # the step counter starts at -1 and is advanced by the first kernel of the step
        ops.select_timestep(self.t_table, self.step_idx, self.t_cur, pre_advance=True)
        ops.to_nhwc(self.lat, self.unet.dtype, out=self.x_nhwc)
        eps = self.unet.forward_nhwc(self.x_nhwc, self.t_cur)
        ops.ddim_step(self.lat, eps, self.coef, self.step_idx, advance=False, out=self.lat)

    def _capture(self):
        keep = self.lat.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up: packs weights, fills caches, sizes workspaces
            self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.step_idx.fill_(-1)
        self.lat.copy_(keep)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        self.graph = g
        if self.steps_per_graph > 1:
            self.step_idx.fill_(-1)
            self.lat.copy_(keep)
            gm = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gm):
                for _ in range(self.steps_per_graph):
                    self._step()
            self.graph_multi = gm
        self.step_idx.fill_(-1)
        self.lat.copy_(keep)

    def reset(self, latents):
        """latents: [B, C, H, W] (any device / float dtype); scaled by init_noise_sigma like the reference."""
        self.lat.copy_(latents.to(device=self.lat.device, dtype=torch.float32) * self.scheduler.init_noise_sigma)
        self.step_idx.fill_(-1)

    def step(self, k=1):
        if self.use_graph and self.graph is None:
            self._capture()
        if not self.use_graph:
            for _ in range(k):
                self._step()
            return
        while self.graph_multi is not None and k >= self.steps_per_graph:
            self.graph_multi.replay()
            k -= self.steps_per_graph
        for _ in range(k):
            self.graph.replay()

    @torch.no_grad()
    def run(self, latents):
        self.reset(latents)
        self.step(self.n)
        return self.lat.clone()
