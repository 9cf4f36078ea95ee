"""Fractional-shift equivariance harness — the procedure of reference
scripts/shift_ldm_ffhq.py::shift_ldm (:49-159) on the MI355X implementation:

  1. install a CrossFrameAttnProcessor on every attention of the UNet;
  2. STORE pass: denoise the initial latent, remembering every attention's input per timestep;
  3. decode the result (the un-shifted image);
  4. for each offset tj = 1/r .. n/r latent pixels: ideal-crop shift the INITIAL latent, denoise
     it in LOAD mode (K/V from the stored maps), decode (masked), and stack
     [shifted output | bilinear-shifted un-shifted image | abs difference] vertically;
  5. restore the processors, write the frames as a GIF.

Multi-GPU: the shift offsets are independent given the STORE pass, so ranks take interleaved
offsets (each recomputes the cheap STORE pass locally) and the frames are gathered once.
Returns the frames and, per offset, the masked latent-space equivariance MSE."""
import time

import torch

from . import parallel
from .engine import DenoiseEngine, model_state_key
from .io_utils import image_to_tensor, save_gif_from_tensors
from .pipelines.cross_frame_attn import (AttnState, CrossFrameAttnProcessor, get_unet_attn_processors,
                                         set_unet_attn_processor)
from .shift_utils.metrics import mask_mse
from .shift_utils.shifters import ImageShifter
from .utils import randn_tensor


class _UnrolledEngine(DenoiseEngine):
    """DenoiseEngine whose ONE graph holds all n steps: the cross-frame processors key their stored tensors by timestep, and in
    an unrolled capture every step's slot is a fixed address - no device-side slot index, no `.item()`, no copy launches.
    `attn_state.set_timestep` gets the step's timestep as a host integer while the launches are recorded."""

    def __init__(self, unet, scheduler, batch_size, num_inference_steps, attn_state):
        super().__init__(unet, scheduler, batch_size, num_inference_steps, use_graph=True, steps_per_graph=1, branches=1)
        self.attn_state = attn_state
        self.graph_full = None
        self._host_step = 0

    def _step(self):
        self.attn_state.set_timestep(self.timesteps[self._host_step])
        super()._step()

    def _capture(self):
        keep = self.lat.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up: packs weights, sizes workspaces (what it stores is overwritten below)
            self._host_step = 0
            self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.check_errors()
        self.step_idx.fill_(-1)
        self.lat.copy_(keep)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(self.n):
                self._host_step = i
                self._step()
        self.graph_full = g
        self.step_idx.fill_(-1)
        self.lat.copy_(keep)

    def refresh_if_stale(self):
        return False            # CrossFrameSampler.run compares the model's fingerprint itself and drops every engine together

    def step(self, k=None):
        assert k in (None, self.n), "the unrolled graph runs the whole schedule"
        if self.graph_full is None:
            self._capture()
        self.graph_full.replay()


class CrossFrameSampler:
    """The denoising passes of the shift harness (reference scripts/shift_ldm_ffhq.py:85-151: one STORE pass, then LOAD
    passes of the shifted latents) as replayed HIP graphs.  Owns the AttnState, one CrossFrameAttnProcessor (cache_kv) per
    attention and one unrolled engine per (pass kind, batch): the STORE graph writes every attention's K / V^T of every step
    into tensors of its own pool, the LOAD graphs read them in place - a new STORE replay (next seed / image) refreshes what
    the LOAD graphs see.  Cached on the pipeline: the first call captures (about the cost of two eager passes), later calls
    replay."""

    def __init__(self, unet, scheduler, steps):
        self.unet, self.scheduler, self.steps = unet, scheduler, steps
        self.attn_state = AttnState()
        self.names = list(get_unet_attn_processors(unet))
        self.procs = {k: CrossFrameAttnProcessor(self.attn_state, cache_kv=True) for k in self.names}
        self.engines = {}
        self.model_key, self.stored = None, False
        self.key = (unet.dtype, str(unet.device), steps, tuple(sorted((k, repr(v)) for k, v in dict(scheduler.config).items())))

    def install(self):
        previous = get_unet_attn_processors(self.unet)
        set_unet_attn_processor(self.unet, dict(self.procs))
        return previous

    def run(self, latents, load):
        """One pass over the schedule (processors installed by the caller): STORE (load=False; attn_state is reset) or LOAD."""
        from .models.blocks import invalidate_packed
        fp = model_state_key(self.unet)
        if fp != self.model_key:                  # load_state_dict / .to() / surgery since the capture: every graph is stale
            torch.cuda.synchronize()
            self.engines.clear()
            for p in self.procs.values():
                p.maps, p.kv = [dict(), dict()], [dict(), dict()]
            invalidate_packed(self.unet)
            self.model_key, self.stored = fp, False
        if load:
            if not self.stored:
                raise RuntimeError("CrossFrameSampler: a LOAD pass needs a STORE pass first")
            self.attn_state.to_load()
        else:
            self.attn_state.reset()
        key = (bool(load), latents.shape[0])
        if key not in self.engines:
            if not load:                          # a new STORE capture moves the stored tensors: the LOAD graphs read the old ones
                self.engines.clear()
            self.engines[key] = _UnrolledEngine(self.unet, self.scheduler, latents.shape[0], self.steps, self.attn_state)
        out = self.engines[key].run(latents)
        self.stored = self.stored or not load
        return out


def _sampler(pipeline, steps, sched=None):
    """The pipeline's cached CrossFrameSampler for `steps` evaluations of `sched` (default: a fresh DDIMScheduler of the pipeline's
    configuration; the SR harness passes its I2SB ODE schedule)."""
    from .schedulers.ddim import DDIMScheduler
    unet = pipeline.unet
    if sched is None:
        sched = DDIMScheduler.from_config(pipeline.scheduler.config)
    smp = getattr(pipeline, "_xframe_sampler", None)
    probe = (unet.dtype, str(unet.device), steps, tuple(sorted((k, repr(v)) for k, v in dict(sched.config).items())))
    if smp is None or smp.unet is not unet or smp.key != probe or smp.names != list(get_unet_attn_processors(unet)):
        smp = CrossFrameSampler(unet, sched, steps)
        pipeline._xframe_sampler = smp
    return smp


def vae_encode(vae, x):
    return vae.encode(x).latent_dist.sample() * vae.config.scaling_factor


def vae_decode(vae, x):
    return vae.decode(x / vae.config.scaling_factor, return_dict=False)[0]


class _Clock:
    """Wall time per phase of a harness run ('unet_s', 'vae_s', the rest under 'other_s'), each bracketed by a device
    synchronisation; inactive (no synchronisation at all) unless the caller passes a dict to fill."""

    def __init__(self, sink):
        self.sink = sink
        if sink is not None:
            torch.cuda.synchronize()
            self.t0 = self.last = time.perf_counter()

    def lap(self, name):
        if self.sink is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.sink[name] = self.sink.get(name, 0.0) + now - self.last
            self.last = now

    def done(self):
        if self.sink is not None:
            torch.cuda.synchronize()
            self.sink["total_s"] = time.perf_counter() - self.t0


@torch.no_grad()
def shift_ldm(pipeline, num_inference_steps=50, num_shift_steps=16, output_path="results/shift_ldm.gif",
              input_path=None, generator=None, rank=0, world=1, batch_offsets=True, reference_exact=True, use_graph=True,
              timings=None):
    """batch_offsets: the LOAD passes of this rank's offsets run as ONE batch (samples are independent; the
    cross-frame K/V of the stored pass is shared by the whole batch) instead of the reference's one B = 1
    sampler run per offset (shift_ldm_ffhq.py:124-151) - 50 UNet evaluations instead of 50 per offset.

    use_graph (default): the STORE and LOAD passes run as replayed HIP graphs (CrossFrameSampler: no per-step `t.item()`
    synchronisation, no per-site clone launches, the LOAD passes attend to the stored pass's projected K / V); the graphs
    are cached on the pipeline, so the first call pays their capture and later calls (more seeds / images) replay.
    use_graph=False is the eager loop that follows reference shift_ldm_ffhq.py:85-108 statement by statement.
    timings: a dict that receives the wall time of the UNet passes ('unet_s'), the VAE calls ('vae_s'), the latent / image
    shifters ('shift_s'), the masked metrics ('metric_s'), frame assembly + device-to-host copies ('frames_s'), everything else
    ('other_s') and 'total_s' (synchronises at the phase boundaries; off when None).

    reference_exact=True (the default) follows the reference flow to the letter; reference_exact=False
    (scripts: --fixed_resize) opts into two deliberate deviations:
    * `input_path`: the image is resized to sample_size * VAE ratio (256) so that its latent has the UNet's
      sample_size; the reference resizes to (sample_size, sample_size) = 32 x 32 BEFORE the VAE
      (shift_ldm_ffhq.py:110-113), i.e. inverts a 4 x 4 latent.
    * the initial noise is drawn on the CPU (device independent) instead of on the GPU (:118-122)."""
    device = pipeline.device
    vae, unet, scheduler = pipeline.vae, pipeline.unet, pipeline.scheduler
    pipeline.set_progress_bar_config(disable=True)
    ratio = 2 ** (len(vae.up_block_types) - 1) if vae is not None else 8
    latent_shifter = ImageShifter("ideal_crop", ratio)
    image_shifter = ImageShifter()
    clock = _Clock(timings)

    init_latent = None
    if use_graph:
        if input_path is not None:
            # The reference inverts with its processors installed in their initial STORE state (shift_ldm_ffhq.py:110-116); what
            # that stores is overwritten by the STORE pass, so the inversion runs here, on the plain processors and the captured-graph
            # loop (MyLDMPipeline.ddim_inversion, latent carried in fp32): same procedure, no side effects to undo
            size = unet.config.sample_size * (1 if reference_exact else ratio)
            tensor = vae_encode(vae, image_to_tensor(input_path, (size, size)).to(device))
            clock.lap("vae_s")
            scheduler.set_timesteps(num_inference_steps, device=device)
            init_latent = pipeline.ddim_inversion(tensor.float(), bar=False)
            clock.lap("unet_s")
        sampler = _sampler(pipeline, num_inference_steps)
        attn_state = sampler.attn_state
        previous = sampler.install()

        def denoise(latents, load):
            return sampler.run(latents, load).to(latents.dtype)
    else:
        attn_state = AttnState()
        previous = get_unet_attn_processors(unet)
        set_unet_attn_processor(unet, {k: CrossFrameAttnProcessor(attn_state) for k in previous})

        def denoise(latents, load):
            if load:
                attn_state.to_load()
            else:
                attn_state.reset()
            latents = latents.to(device)
            scheduler.set_timesteps(num_inference_steps, device=device)
            for t in scheduler.timesteps:
                attn_state.set_timestep(t)
                eps = unet(scheduler.scale_model_input(latents, t), t, return_dict=False)[0]
                latents = scheduler.step(eps, t, latents, eta=0, return_dict=False)[0]
            return latents

    try:
        if init_latent is not None:
            pass
        elif input_path is not None:
            size = unet.config.sample_size * (1 if reference_exact else ratio)
            tensor = vae_encode(vae, image_to_tensor(input_path, (size, size)).to(device))
            clock.lap("vae_s")
            scheduler.set_timesteps(num_inference_steps, device=device)
            init_latent = pipeline.ddim_inversion(tensor, bar=False)
            clock.lap("unet_s")
        else:
            # CPU-side draw (seedable, device independent) — the reference draws on the GPU
            # (shift_ldm_ffhq.py:118-122), which is not reproducible across devices
            shape = (1, unet.config.in_channels, unet.config.sample_size, unet.config.sample_size)
            if reference_exact:
                init_latent = randn_tensor(shape, device=device, generator=generator)
            else:
                init_latent = randn_tensor(shape, generator=generator).to(device)
        clock.lap("other_s")
        denoised = denoise(init_latent, load=False)
        clock.lap("unet_s")
        rec_img = vae_decode(vae, denoised) if vae is not None else None
        clock.lap("vae_s")

        offsets = torch.linspace(1 / ratio, num_shift_steps / ratio, num_shift_steps)
        mine = parallel.interleaved(num_shift_steps, rank, world)
        frames, errors = {}, {}
        shifted = {i: latent_shifter.shift(init_latent, 0, float(offsets[i])) for i in mine}
        clock.lap("shift_s")
        if batch_offsets and len(mine) > 1:
            den_all = denoise(torch.cat([shifted[i][0] for i in mine], 0), load=True)
            dens = {i: den_all[k:k + 1] for k, i in enumerate(mine)}
        else:
            dens = {i: denoise(shifted[i][0], load=True) for i in mine}
        clock.lap("unet_s")
        for i in mine:
            tj = float(offsets[i])
            den, mask = dens[i], shifted[i][1]
            ref_lat, _ = latent_shifter.shift(denoised, 0, tj)
            clock.lap("shift_s")
            errors[i] = float(mask_mse(den, ref_lat, mask))
            clock.lap("metric_s")
            if vae is not None:
                gt, _ = image_shifter.shift(rec_img, 0, tj * ratio)
                clock.lap("shift_s")
                img = vae_decode(vae, den * mask)
                clock.lap("vae_s")
                frames[i] = torch.cat((img, gt, torch.abs(img - gt)), -2).float().cpu()
                clock.lap("frames_s")
    finally:
        set_unet_attn_processor(unet, dict(previous))

    frames, errors = parallel.gather_indexed(world, frames, errors)
    ordered = [frames[i] for i in sorted(frames)]
    if ordered and rank == 0 and output_path:
        save_gif_from_tensors(ordered, output_path, denorm=True)
    clock.lap("other_s")
    clock.done()
    return ordered, [errors[i] for i in sorted(errors)]


def vae_encode_mode(vae, x):
    """reference scripts/shift_ldm_sr.py:31-34 (the SR harness encodes with the posterior mode)."""
    return vae.encode(x).latent_dist.mode() * vae.config.scaling_factor


@torch.no_grad()
def shift_ldm_sr(pipeline, num_inference_steps=50, num_shift_steps=16, output_path="results/shift_ldm_sr.gif",
                 input_path=None, image=None, rank=0, world=1, batch_offsets=True, use_graph=True):
    """Fractional-shift equivariance of x4 super-resolution with I2SB - the flow of reference
    scripts/shift_ldm_sr.py:43-150: degrade (bicubic x1/4, nearest x4), VAE-encode, denoise with
    cross-frame attention STORE, then for every offset shift the initial latent, denoise in LOAD mode
    and compare with the shifted reconstruction.  `image` ([1,3,H,W] in [-1,1]) replaces
    `input_path` when given; the offsets are sharded across ranks.  use_graph (default): the STORE / LOAD passes replay HIP
    graphs (CrossFrameSampler over the scheduler's ODE schedule) when the scheduler does not clip x0; otherwise - and with
    use_graph=False - the eager loop below, which follows the reference statement by statement."""
    from .af_libs.superresolution import build_sr4x
    device = pipeline.device
    vae, unet, scheduler = pipeline.vae, pipeline.unet, pipeline.scheduler
    pipeline.set_progress_bar_config(disable=True)
    ratio = 2 ** (len(vae.up_block_types) - 1)
    size = unet.config.sample_size * ratio
    sr_func = build_sr4x(device, "bicubic", size)
    latent_shifter = ImageShifter("ideal_crop", ratio)
    image_shifter = ImageShifter()

    ode = scheduler.ode_schedule(num_inference_steps) if (use_graph and num_inference_steps >= 2) else None
    if ode is not None:
        sampler = _sampler(pipeline, ode.evaluations, ode)
        attn_state = sampler.attn_state
        previous = sampler.install()

        def denoise(latents, load):
            return sampler.run(latents, load).to(latents.dtype)
    else:
        attn_state = AttnState()
        previous = get_unet_attn_processors(unet)
        set_unet_attn_processor(unet, {k: CrossFrameAttnProcessor(attn_state) for k in previous})

        def denoise(latents, load):
            if load:
                attn_state.to_load()
            else:
                attn_state.reset()
            latents = latents.to(device)
            scheduler.set_timesteps(num_inference_steps, device=device)
            ts = scheduler.timesteps
            for i, t in enumerate(ts):
                if i == num_inference_steps - 1:
                    break
                attn_state.set_timestep(t)
                eps = unet(scheduler.scale_model_input(latents, t), t, return_dict=False)[0]
                latents = scheduler.step(eps, t, latents, is_ode=True, generator=None).prev_sample
            return latents

    try:
        if image is None:
            image = image_to_tensor(input_path, (size, size))
        tensor = sr_func(image.to(device).float()).clip(-1, 1)
        init_latent = vae_encode_mode(vae, tensor.to(vae.dtype)).to(unet.dtype)
        denoised = denoise(init_latent, load=False)
        rec_img = vae_decode(vae, denoised)
        offsets = torch.linspace(1 / ratio, num_shift_steps / ratio, num_shift_steps)
        frames, errors = {}, {}
        mine = parallel.interleaved(num_shift_steps, rank, world)
        shifts = {i: latent_shifter.shift(init_latent, 0, float(offsets[i])) for i in mine}
        if batch_offsets and len(mine) > 1:      # one batched LOAD pass for this rank's offsets (see shift_ldm)
            den_all = denoise(torch.cat([shifts[i][0] for i in mine], 0), load=True)
            dens = {i: den_all[k:k + 1] for k, i in enumerate(mine)}
        else:
            dens = {i: denoise(shifts[i][0], load=True) for i in mine}
        for i in mine:
            tj = float(offsets[i])
            (shifted, mask), den = shifts[i], dens[i]
            ref_lat, _ = latent_shifter.shift(denoised, 0, tj)
            errors[i] = float(mask_mse(den, ref_lat, mask))
            gt, _ = image_shifter.shift(rec_img, 0, tj * ratio)
            img_in = vae_decode(vae, shifted * mask)
            img = vae_decode(vae, den * mask)
            frames[i] = torch.cat((img_in, img, gt, torch.abs(img - gt)), -2).float().cpu()
    finally:
        set_unet_attn_processor(unet, dict(previous))

    frames, errors = parallel.gather_indexed(world, frames, errors)
    ordered = [frames[i] for i in sorted(frames)]
    if ordered and rank == 0 and output_path:
        save_gif_from_tensors(ordered, output_path, denorm=True)
    return ordered, [errors[i] for i in sorted(errors)]

