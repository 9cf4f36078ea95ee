"""DenoiseEngine — the 50-step DDIM loop of the reference (afldm/pipelines/ldm_pipeline.py:103-109)
as ONE captured HIP graph replayed per step.

The reference loop costs, per step, ~1100 dependent kernel launches from Python plus a
`t.item()` device sync.  Here everything the step needs lives in device
memory — the timestep table, the DDIM coefficient table and a step counter the update kernel
increments itself — so a replay needs no host value and no synchronisation:

    step++; t <- t_table[step]; temb <- table[step] (afldm_select_step_row, counter starts at -1: the time-embedding
                                                   MLP and the 27 time_emb_proj layers depend on the timestep only and
                                                   are tabulated once per schedule)
    x_nhwc <- NCHW fp32 latents                   (afldm_nchw_to_nhwc)
    eps    <- UNet(x_nhwc, t)                     (~250 HIP kernels at batch 64, all from libafldm_hip.so)
    lat    <- DDIM(lat, eps, coef[step])          (afldm_ddim_step, in place)
"""
import os

import torch

from . import ops


def model_state_key(model):
    """Fingerprint of everything a captured step bakes in: module identities (af_api surgery swaps modules), the
    attention processors (set_unet_attn_processor), and every parameter's storage + in-place version counter
    (load_state_dict copies in place -> _version moves; .to() / .half() re-allocate -> data_ptr moves).  ~1 ms for the
    FFHQ UNet; compared once per DenoiseEngine.reset()."""
    def ver(p):
        try:
            return p._version
        except RuntimeError:          # tensors created under torch.inference_mode() have no version counter
            return -1
    mods = tuple((id(m), id(getattr(m, "processor", None))) for m in model.modules())
    params = tuple((p.data_ptr(), ver(p), p.dtype, str(p.device)) for p in model.parameters())
    bufs = tuple((b.data_ptr(), ver(b), b.dtype, str(b.device)) for b in model.buffers())
    return hash((mods, params, bufs))


class DenoiseEngine:
    def __init__(self, unet, scheduler, batch_size, num_inference_steps=50, use_graph=True, steps_per_graph=5, branches=1):
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
        # everything of the UNet that depends on the timestep only (time_proj -> time_embedding -> SiLU -> the 27
        # time_emb_proj layers) is tabulated once for the schedule: [steps, sum Cout]; a step copies its row
        self.temb_table = torch.cat([unet.temb_projection(t) for t in self.timesteps], 0).contiguous()
        self.temb_row = torch.empty_like(self.temb_table[:1])
        self.temb_slices = unet.temb_slices(self.temb_row)
        self._model_key = model_state_key(unet)
        nb = int(os.environ.get("AFLDM_BRANCHES", branches))
        self.branches = nb if nb > 1 and batch_size % nb == 0 else 1
        self._side = [torch.cuda.Stream() for _ in range(self.branches - 1)]
        if self.branches > 1:
            from . import trunk
            trunk.block("DenoiseEngine with parallel branches")      # (the experimental cooperative trunk needs the GPU to itself)
        # counters of in-kernel reductions / cluster hand-overs: private to this engine (one per branch), so that its
        # captured graphs can replay next to another engine's on a different stream
        self._sync = [ops.new_sync_buffer(dev) for _ in range(self.branches)]

    # one denoise step, entirely stream-ordered
    def _step(self):
        # the step counter starts at -1 and is advanced by the first kernel of the step
        ops.select_step_row(self.t_table, self.step_idx, self.t_cur, self.temb_table, self.temb_row, pre_advance=True)
        nb = self.branches
        if nb <= 1:
            self._substep(self.lat, self.x_nhwc)
            return
        # The batch as `nb` independent sub-batches on parallel streams (parallel branches of the captured graph):
        # samples are independent, so the kernels of one branch fill the gaps the other leaves - prologues / epilogues
        # of the large convolutions (one tile per CU: nothing inside a launch overlaps them) and the latency-bound
        # launches of the 8x8 .. 2x2 levels.
        main = torch.cuda.current_stream()
        per = self.B // nb
        for i, s in enumerate(self._side):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                self._substep(self.lat[(i + 1) * per:(i + 2) * per], self.x_nhwc[(i + 1) * per:(i + 2) * per], branch=i + 1)
        self._substep(self.lat[:per], self.x_nhwc[:per])
        for s in self._side:
            main.wait_stream(s)

    def _substep(self, lat, x_nhwc, branch=0):
        with ops.sync_scope(self._sync[branch]):
            ops.to_nhwc(lat, self.unet.dtype, out=x_nhwc)
            eps = self.unet.forward_nhwc(x_nhwc, self.t_cur, temb_slices=self.temb_slices)
            ops.ddim_step(lat, eps, self.coef, self.step_idx, advance=False, out=lat)

    def _capture(self):
        keep = self.lat.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up: packs weights, fills caches, sizes workspaces
            self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.check_errors()                           # (the warm-up step ran every hand-over once: free to look, we just synchronised)
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

    def check_errors(self):
        """Read (and clear) the error words of this engine's sync buffers.  The attention block's in-launch hand-over
        (afldm_attn_block_fused_out, csrc/attnf.hip phase C) gives up after a bounded spin instead of hanging the device and
        then finishes on incomplete data: that must never pass silently.  On a hit the fused tail is switched off
        for the process, this engine's graphs are dropped (the next step re-captures on the two-launch path) and a
        RuntimeError says that the latents of this run are invalid.  Synchronises."""
        sk, ho = ops.sync_errors(self._sync)
        if not (sk or ho):
            return
        self.graph = self.graph_multi = None
        if ho:
            ops._FUSED_ATTN_OUT = False
        raise RuntimeError(
            f"afldm_amd: an in-launch hand-over gave up (split-K word {sk}, hand-over word {ho}: 1 = a workgroup waited out its "
            "cluster, 2 = a cluster straddled XCDs): the latents of this run are INVALID.  The fused attention tail is now off "
            "(two launches per block); run again.  Causes: a CU mask / partitioned device, or many concurrent heavy streams.")

    def refresh_if_stale(self):
        """The captured graphs hold the packed-weight pointers and the time-embedding table of the model AS IT WAS at
        capture: after load_state_dict / .to() / af_api surgery / set_unet_attn_processor they would replay the old
        model (or freed packed tensors).  Re-tabulate and drop the graphs when the model's fingerprint moved
        (ADVICE r02)."""
        key = model_state_key(self.unet)
        if key == self._model_key:
            return False
        from .models.blocks import invalidate_packed
        torch.cuda.synchronize()
        if self.unet.dtype != self.x_nhwc.dtype or self.unet.device != self.x_nhwc.device:
            # .to(dtype / device): every dtype-typed static buffer of the engine is stale, not just the table
            invalidate_packed(self.unet)
            self.__init__(self.unet, self.scheduler, self.B, self.n, self.use_graph, self.steps_per_graph, self.branches)
            return True
        self.graph = self.graph_multi = None
        invalidate_packed(self.unet)       # in-place parameter edits do not pass through the modules' own hooks
        self.temb_table.copy_(torch.cat([self.unet.temb_projection(t) for t in self.timesteps], 0))
        self._model_key = key
        return True

    def reset(self, latents):
        """latents: [B, C, H, W] (any device / float dtype); scaled by init_noise_sigma like the reference."""
        self.refresh_if_stale()
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
        out = self.lat.clone()
        self.check_errors()                           # one 8-byte read per run: a run whose hand-over failed must not return latents
        return out
