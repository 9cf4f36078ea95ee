#!/usr/bin/env python
"""bench.py — denoise-steps/sec of the FFHQ-256 alias-free UNet + DDIM update on MI355X.

A "step" is one denoising step of the whole local batch: timestep select -> NCHW->NHWC ->
AF-UNet forward (a few hundred HIP kernels from libafldm_hip.so) -> DDIM update, replayed as one
captured HIP graph.  Reported: the MEDIAN of --regions (5) timed regions of --steps steps each.  Workload = BASELINE.json configs[1] (batch 64 per GPU, bf16 storage /
MFMA with fp32 accumulation), synthetic: seeded random weights of the FFHQ architecture
(conv_out x0.1), CPU-seeded noise, inputs resident in HBM before the timed region.
N > 1: launched by torch.distributed.run, one rank per GPU, batch-sharded (weak scaling: 64 per
GPU), no per-step communication, ONE RCCL all-gather of the final latents inside the timed
region.  Rank 0 prints one JSON line.  At N = 1 the same line also carries, timed inside this run: the other
single-GPU configurations of north_star (batch 1 / 8 bf16, batch 64 / 1 fp32: `other_configs`), BASELINE configs[3]
(the AF-VAE at 256^2 x 128 with its shift-equivariance check: `vae_c4`; also `--workload vae` on its own), the north-star
harness procedure of BASELINE configs[0] on the graph-replayed path (`harness_c1`; `--workload harness`), the
roofline of the dominant kernel family and the CPU baseline on BASELINE configs[0].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def build_unet(dtype, device):
    from afldm_amd.af_modules.af_api import make_af_unet
    from afldm_amd.configs import FFHQ_UNET_CONFIG
    from afldm_amd.models.unet_2d import UNet2DModel
    torch.manual_seed(0)
    unet = UNet2DModel.from_config(FFHQ_UNET_CONFIG)      # PyTorch default init, seed 0
    with torch.no_grad():
        unet.conv_out.weight.mul_(0.1)
        unet.conv_out.bias.mul_(0.1)
    make_af_unet(unet)
    return unet.to(device).to(dtype)


def lib_sha256():
    import hashlib
    from afldm_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()


def roofline_pass(unet, batch, dtype):
    """One eager step with HIP events around every launch (ops.Profiler): per-kernel-family
    time and algorithmic work, measured live on the launch stream."""
    from afldm_amd import ops
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), batch, 50, use_graph=False)
    eng.reset(torch.randn(batch, 4, 32, 32, generator=torch.Generator().manual_seed(1)))
    eng.step(2)                                            # warm
    torch.cuda.synchronize()
    with ops.Profiler() as prof:
        eng.step(3)
    agg = prof.summary()
    total_ms = sum(d["ms"] for d in agg.values())
    fam = {}
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        fam[k] = dict(launches_per_step=d["launches"] // 3, ms_per_step=round(d["ms"] / 3, 4),
                      share=round(d["ms"] / total_ms, 4),
                      tflops=round(d["flops"] / d["ms"] / 1e9, 2) if d["ms"] > 0 else 0.0,
                      gbs=round(d["bytes"] / d["ms"] / 1e6, 1) if d["ms"] > 0 else 0.0)
    # Every family again WITHOUT host gaps: the launches of ONE step of the family - same argument blocks, same buffers
    # still holding the step's real activations - captured back to back into one HIP graph and timed with events around
    # its replay.  `ms_per_step` / `tflops` / `gbs` of a family are from this replay (kernel time incl. the boundaries
    # between the family's own launches); the eager figures (each launch bracketed by events, 10-20 us of Python in
    # front of it) are kept as eager_ms_per_step.
    replays = {}
    for k, d in agg.items():
        calls = [r for r in prof.records if r[0] == k]
        if not calls or any(r[5] is None for r in calls) or len(calls) % 3:
            continue
        calls = calls[:len(calls) // 3]                      # the launches of ONE step
        rp = replay_family(calls)
        replays[k] = rp
        f = fam[k]
        f["eager_ms_per_step"] = f["ms_per_step"]
        f["ms_per_step"] = round(rp["ms"], 4)
        f["tflops"] = round(rp["flops"] / rp["ms"] / 1e9, 2) if rp["ms"] > 0 else 0.0
        f["gbs"] = round(rp["bytes"] / rp["ms"] / 1e6, 1) if rp["ms"] > 0 else 0.0
        f["timing"] = "graph replay"
    tot = sum(f["ms_per_step"] for f in fam.values())
    for f in fam.values():
        f["share"] = round(f["ms_per_step"] / tot, 4) if tot > 0 else 0.0
    fam = dict(sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"]))
    name = next(iter(fam))
    d = agg[name]
    replay = replays.get(name)
    # HBM bytes per launch of the dominant family from the PMC counters (FETCH_SIZE / WRITE_SIZE in separate
    # rocprofv3 passes, gfx950 x2 read correction): a PMC pass cannot run inside the timed process, so it is
    # collected by profiles/run_pmc_conv3x3.sh over a replay of exactly this family's launches and committed
    # STAMPED WITH THE SHA-256 OF THE LIBRARY it was measured on; a stamp that does not match the library loaded
    # here is refused (traffic = null) instead of quoting a stale build.
    traffic, traffic_note = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "conv3x3_traffic.json")))
        if tj.get("family") != name or batch != 64 or dtype != torch.bfloat16:
            traffic_note = "no PMC record for this family / batch / dtype"
        elif tj.get("lib_sha256") != lib_sha256():
            traffic_note = "profiles/conv3x3_traffic.json was measured on another build of libafldm_hip.so (stale): refused"
        else:
            traffic = round(tj["hbm_bytes_per_launch"])
            traffic_note = (f"PMC (2*FETCH_SIZE+WRITE_SIZE)*1024 over {tj['launches_per_step']} launches, tag {tj['tag']}; "
                            f"algorithmic bytes per launch there: {round(tj['algorithmic_bytes_per_launch'])}")
    except (OSError, ValueError, KeyError) as e:
        traffic_note = f"no usable PMC record ({type(e).__name__})"
    mfma_bound = name.startswith("conv") or name in ("linear", "attention", "attn_fused") or name.startswith("af_act_N3") \
        or name.startswith("af_act_N16")
    if mfma_bound:
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        eager = d["flops"] / d["ms"] / 1e9
        achieved = replay["flops"] / replay["ms"] / 1e9 if replay else eager
        fam_ms = replay["ms"] if replay else d["ms"] / 3
        n = replay["launches"] if replay else d["launches"] // 3
        roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=traffic, traffic_note=traffic_note,
                    launches_per_step=n, avg_launch_us=round(1e3 * fam_ms / n, 2), family_ms_per_step=round(fam_ms, 4),
                    timing="HIP events around a graph replay of the family's launches of one step (no host gaps)" if replay
                    else "HIP events around each eager launch",
                    eager_achieved=round(eager, 2), eager_family_ms_per_step=round(d["ms"] / 3, 4),
                    flops_per_launch=d["flops"] / d["launches"], algorithmic_bytes_per_launch=round(d["bytes"] / d["launches"]))
    else:
        achieved = d["bytes"] / d["ms"] / 1e6
        roof = dict(kernel=name, bound="hbm", achieved=round(achieved, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(achieved / PEAK_HBM_GBS, 4), traffic=None,
                    launches_per_step=d["launches"] // 3, avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2),
                    bytes_per_launch=d["bytes"] / d["launches"])
    # whole-step dense rate: the algorithmic MFMA-shaped work of one step (3x3 / 1x1 convolutions, linear layers, attention; the
    # alias-free filters' dense separable form is kept apart) over the TIMED step - filled in by main()
    dense = sum(agg[k]["flops"] for k in agg if k.startswith("conv") or k in ("linear", "attention", "attn_fused")) / 3
    af = sum(agg[k]["flops"] for k in agg if k.startswith("af_")) / 3
    roof["step_dense_tflop"] = round(dense / 1e12, 4)
    roof["step_af_filter_tflop"] = round(af / 1e12, 4)
    return roof, fam


def replay_family(calls):
    """HIP-graph replay of recorded launches (ops.Profiler records): median of 5 replays after 2 warm ones."""
    for r in calls:
        r[5]()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for r in calls:
            r[5]()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    del g
    return dict(ms=median(ts[2:]), launches=len(calls), flops=sum(r[1] for r in calls), bytes=sum(r[2] for r in calls))


def _sysfs_first(paths):
    import glob
    for pat in paths:
        for fn in sorted(glob.glob(pat)):
            try:
                return open(fn).read().strip()
            except OSError:
                pass
    return None


def box_record(dev):
    """Fingerprint of THIS box, measured in this run: a fixed MFMA loop (~50 ms of back-to-back
    v_mfma_f32_32x32x16_bf16 on every SIMD: the sustained MFMA clock under load), a 1 GiB device-to-device stream copy,
    and the clock / power-cap state the driver exposes.  Box-to-box spread of one build was 3-12 % in rounds 1-2; dividing a
    kernel figure by `mfma_tflops` (MFMA-bound) or `copy_gbs` (HBM-bound) of its own run makes lines from different
    boxes comparable."""
    from afldm_amd import _lib
    lib = _lib.lib
    st = torch.cuda.current_stream().cuda_stream
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    wgs = cus * 4                                              # 4 waves per SIMD
    out = torch.zeros(wgs * 256, dtype=torch.float32, device=dev)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return median(ts)

    iters = 240000
    ms = timed(lambda: _lib.check(lib.afldm_probe_mfma(out.data_ptr(), wgs, iters, st), "probe_mfma"), 3)
    flops = wgs * 4.0 * iters * 4 * 32 * 32 * 16 * 2
    # the same loop on random bf16 operands: the chip's sustained MFMA rate depends on the data (power), this is the
    # ceiling a convolution on real activations sees
    ms_rnd = timed(lambda: _lib.check(lib.afldm_probe_mfma_random(out.data_ptr(), wgs, iters, st), "probe_mfma_random"), 3)
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    cms = timed(lambda: _lib.check(lib.afldm_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, st), "probe_copy"), 5)
    # latency side (the throughput probes did not explain a 4 % spread between two boxes): (a) the boundary
    # between two dependent EMPTY kernels inside a captured graph (200 launches of 256 workgroups, replayed), (b) the
    # same for a real streaming kernel (the copy probe on 64 MiB, captured 20 times: kernel + boundary), (c) a dependent-load
    # chain of 4096 steps spread over 1 GiB (256 KiB + 128 B apart), cold (one pass right after a 1 GiB copy has swept the
    # L2s and the 256 MiB Infinity Cache) and cache-resident (repeated passes: its 4096 lines are 512 KiB): ns per load
    def graph_time(fn, n, reps=5):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        t = timed(g.replay, reps)
        del g
        return t
    st2 = lambda: torch.cuda.current_stream().cuda_stream
    t_empty = graph_time(lambda: _lib.check(lib.afldm_probe_empty(cus, st2()), "probe_empty"), 200)
    small = 64 << 20
    t_copy20 = graph_time(lambda: _lib.check(lib.afldm_probe_copy(src.data_ptr(), dst.data_ptr(), small, st2()), "probe_copy"), 20)
    nent = (1 << 30) // 4
    stride = ((256 << 10) + 128) // 4
    steps = 4096
    idx = (torch.arange(steps + 1, dtype=torch.int64) * stride) % nent
    chain = torch.zeros(nent, dtype=torch.int32)
    chain[idx[:-1]] = idx[1:].to(torch.int32)
    chain = chain.to(dev)
    cout = torch.zeros(4, dtype=torch.int32, device=dev)
    chase = lambda: _lib.check(lib.afldm_probe_chase(chain.data_ptr(), cout.data_ptr(), steps, st), "probe_chase")
    t_chase_hot = timed(chase, 3)                  # the chain's 4 096 lines (512 KiB) are cache-resident after the first pass
    # cold: ONE launch right after a 1 GiB stream copy has swept the L2s and the 256 MiB Infinity Cache (no warm-up pass)
    t_chase = 1e30
    for _ in range(2):
        _lib.check(lib.afldm_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, st), "probe_copy")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        chase()
        e1.record()
        torch.cuda.synchronize()
        t_chase = min(t_chase, e0.elapsed_time(e1))
    ticks = int(cout.cpu()[1].item()) & 0xFFFFFFFF       # s_memtime ticks of the last (cold) chase loop
    rec = dict(mfma_tflops=round(flops / ms / 1e9, 1), mfma_probe_ms=round(ms, 2),
               mfma_tflops_random_operands=round(flops / ms_rnd / 1e9, 1),
               graph_empty_kernel_us=round(1e3 * t_empty / 200, 3),
               graph_copy_64mib_us=round(1e3 * t_copy20 / 20, 3),
               dependent_load_ns=round(1e6 * t_chase / steps, 1),                  # beyond every cache (HBM)
               dependent_load_cached_ns=round(1e6 * t_chase_hot / steps, 1),       # the same chain, cache-resident
               memtime_ticks_per_us=round(ticks / (1e3 * t_chase), 1),             # s_memtime rate while one wave chases
               # one 32x32x16 bf16 MFMA (32768 flop) occupies its SIMD's matrix pipe for 32 cycles
               mfma_clock_ghz=round(flops / (ms * 1e-3) / 32768.0 / (cus * 4) * 32 / 1e9, 3),
               copy_gbs=round(2.0 * nbytes / cms / 1e6, 1), copy_probe_ms=round(cms, 3), cus=cus,
               device=torch.cuda.get_device_properties(dev).name)
    # clocks / power cap as exposed by the amdgpu driver (absent inside some containers: recorded as null)
    base = "/sys/class/drm/card*/device/"
    def cur(txt):
        if not txt:
            return None
        for line in txt.splitlines():
            if line.rstrip().endswith("*"):
                return line.split(":")[1].strip(" *")
        return txt.splitlines()[-1].split(":")[-1].strip()
    rec["sclk"] = cur(_sysfs_first([base + "pp_dpm_sclk"]))
    rec["mclk"] = cur(_sysfs_first([base + "pp_dpm_mclk"]))
    cap = _sysfs_first([base + "hwmon/hwmon*/power1_cap"])
    rec["power_cap_w"] = round(int(cap) / 1e6, 1) if cap and cap.isdigit() else None
    pw = _sysfs_first([base + "hwmon/hwmon*/power1_average", base + "hwmon/hwmon*/power1_input"])
    rec["power_now_w"] = round(int(pw) / 1e6, 1) if pw and pw.isdigit() else None
    del src, dst, out, chain, cout
    torch.cuda.empty_cache()
    return rec


def cpu_baseline(budget_s=14.0):
    """BASELINE configs[0] (C1) on the host: the oracle (CPU restatement: torch.fft filters, F.conv2d, SDPA, eager
    PyTorch fp32) at batch 1 over the 50-step DDIM schedule, on the host cores - a bounded sample: as many of the
    50 steps as fit `budget_s` - plus a short single-thread sample.  A reported baseline, never the thing measured."""
    from oracle import configs as oc, pipeline as op, unet as ou
    # eager ops at this size do not scale past a few tens of threads (256 threads measured 100x SLOWER than 8 on the
    # MI355X host: oversubscribed OpenMP barriers), so "all cores" is capped at 32.
    cores = min(os.cpu_count() or 1, 32)
    sd = ou.init_unet_params(oc.FFHQ_UNET, seed=0, conv_out_scale=0.1)
    sec, steps = op.time_denoise_steps(sd, oc.FFHQ_UNET, batch=1, steps=49, threads=cores, budget_s=budget_s)
    sec1, steps1 = op.time_denoise_steps(sd, oc.FFHQ_UNET, batch=1, steps=6, threads=1, budget_s=6.0)
    torch.set_num_threads(cores)
    return dict(value=round(1.0 / sec, 4), unit="denoise-steps/s", cores=cores, kind="port",
                ms_per_step=round(sec * 1e3, 2),
                single_thread=dict(value=round(1.0 / sec1, 4), ms_per_step=round(sec1 * 1e3, 2), steps=steps1),
                sample=f"BASELINE configs[0]: oracle CPU restatement, FFHQ AF-UNet + DDIM update, batch 1 fp32, {steps} of the "
                       f"50 DDIM steps ({sec * steps:.1f} s) on {cores} threads after 1 warm-up step; single-thread: {steps1} steps")


def timed_regions(run_steps, k, regions, world, dev):
    """`regions` timed regions of exactly k steps each, every one bracketed by barrier + synchronize on both sides,
    MAX over ranks per region; returns the per-region seconds."""
    from afldm_amd import parallel
    out = []
    for _ in range(regions):
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(k)
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        out.append(dt)
    return out


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def side_config(batch, dtype, steps=20, regions=3):
    """north_star's other single-GPU configurations (batch 1 / 8 bf16, fp32 = the reference's precision), timed the
    same way inside this run (graph replay, median of `regions` regions of `steps` steps)."""
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    unet = build_unet(dtype, torch.device("cuda", torch.cuda.current_device()))
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), batch, 50, use_graph=True)
    noise = torch.randn(batch, 4, 32, 32, generator=torch.Generator().manual_seed(1234))
    eng.reset(noise)
    eng.step(6)
    ts = []
    for _ in range(regions):
        eng.reset(noise)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(steps)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = median(ts)
    finite = bool(torch.isfinite(eng.lat).all().item())
    del eng, unet
    torch.cuda.empty_cache()
    return dict(batch=batch, dtype="bf16" if dtype == torch.bfloat16 else "fp32", steps=steps, regions=regions,
                ms_per_step=round(1e3 * dt / steps, 4), value=round(batch * steps / dt, 2), unit="denoise-steps/s",
                latents_finite=finite)


def concurrent_engines(n=2, batch=64, dtype=torch.bfloat16, steps=20, regions=3):
    """NOT the bench configuration (one batch of 64 per GPU) - a throughput-serving data point: `n` independent batch-64
    jobs in flight on one GPU, each a DenoiseEngine replaying its own captured graph on its own stream.  The kernels of
    one job run under the launch ramps / write tails / cold first tiles of the other's (profiles/r04/two_streams_ab.txt)."""
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    unet = build_unet(dtype, torch.device("cuda", torch.cuda.current_device()))
    gen = torch.Generator().manual_seed(1234)
    noise = [torch.randn(batch, 4, 32, 32, generator=gen) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    engs = []
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            e = DenoiseEngine(unet, ffhq_ddim_scheduler(), batch, 50, use_graph=True)
            e.reset(noise[i])
            e.step(6)
            engs.append(e)
    torch.cuda.synchronize()
    spg = engs[0].steps_per_graph
    steps -= steps % spg
    ts = []
    for _ in range(regions):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                engs[i].reset(noise[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // spg):                       # the host alternates between the jobs' queues
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    engs[i].step(spg)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = median(ts)
    finite = all(bool(torch.isfinite(e.lat).all().item()) for e in engs)
    del engs, unet
    torch.cuda.empty_cache()
    return dict(config=f"{n} independent batch-{batch} jobs in flight on one GPU ({n} streams) - not the bench configuration",
                jobs=n, batch_per_job=batch, dtype="bf16" if dtype == torch.bfloat16 else "fp32", steps=steps, regions=regions,
                ms_per_step_all_jobs=round(1e3 * dt / steps, 4), ms_per_64_samples=round(1e3 * dt / steps * 64 / (n * batch), 4),
                value=round(n * batch * steps / dt, 2), unit="denoise-steps/s", latents_finite=finite)


def build_vae(dtype, device):
    """The reference's AF-VAE (configs/vae/model_afvae.json: [128, 256, 512, 512], 2 layers per block, 83.65 M
    parameters) with seeded PyTorch default-init weights."""
    from afldm_amd.af_modules.af_api import make_af_vae_from_config
    from afldm_amd.models.vae import AutoencoderKL
    torch.manual_seed(3)
    vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                        up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512],
                        layers_per_block=2, latent_channels=4, norm_num_groups=32, scaling_factor=0.6, mid_act=True,
                        down_filtered_act=[False, True, True, True], up_filtered_act=[True, True, True, False],
                        up_rescale=[True, True, True])
    make_af_vae_from_config(vae)
    return vae.to(device).to(dtype)


def vae_workload(batch=128, dtype=torch.bfloat16, passes=3):
    """BASELINE configs[3] (C4): alias-free AutoencoderKL encode + decode of `batch` 256x256 images on one GPU, with
    the fractional-shift equivariance check of SURVEY.md 8d (mask_psnr of decode(T z) against T(decode z), tj = 1/8
    and 1/2 latent pixels).  Images resident in HBM; median of `passes` encode+decode passes."""
    from afldm_amd.shift_utils.metrics import mask_psnr
    from afldm_amd.shift_utils.shifters import ImageShifter
    dev = torch.device("cuda", torch.cuda.current_device())
    vae = build_vae(dtype, dev)
    imgs = (torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)
    z = vae.encode(imgs).latent_dist.mode()
    vae.decode(z, return_dict=False)
    enc, dec = [], []
    for _ in range(passes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = vae.encode(imgs).latent_dist.mode()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        img = vae.decode(z, return_dict=False)[0]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enc.append(t1 - t0)
        dec.append(t2 - t1)
    te, td = median(enc), median(dec)
    eq = {}
    zr = torch.randn(8, 4, 32, 32, generator=torch.Generator().manual_seed(8)).to(dev).to(dtype)
    base = vae.decode(zr, return_dict=False)[0].float()
    for tj in (0.125, 0.5):
        zs, _ = ImageShifter("ideal", 8).shift(zr.float(), 0, tj)
        got = vae.decode(zs.to(dtype), return_dict=False)[0].float()
        gt, m = ImageShifter().shift(base, 0, tj * 8)
        mask = m.clone().expand_as(base).contiguous()
        mask[..., :32] = 0
        mask[..., -32:] = 0
        eq[f"mask_psnr_db_tj_{tj}"] = round(float(mask_psnr(got, gt, mask)), 2)
    finite = bool(torch.isfinite(img.float()).all().item())
    del vae
    torch.cuda.empty_cache()
    return dict(workload="AF-VAE encode + decode 256x256 (BASELINE configs[3])", batch=batch,
                dtype="bf16" if dtype == torch.bfloat16 else "fp32",
                encode_ms=round(te * 1e3, 2), decode_ms=round(td * 1e3, 2),
                value=round(batch / (te + td), 2), unit="images/s (encode+decode)",
                encode_images_per_s=round(batch / te, 1), decode_images_per_s=round(batch / td, 1),
                equivariance=eq, outputs_finite=finite)


def harness_c1(dtype=torch.bfloat16, steps=50, offsets=16):
    """The north-star harness itself (reference scripts/shift_ldm_ffhq.py:49-159 = BASELINE configs[0]'s procedure): one
    STORE pass at batch 1, `offsets` fractional shifts denoised in LOAD mode, every result decoded by the AF-VAE and compared
    with the shifted un-shifted image - FFHQ AF-UNet + the full AF-VAE, seeded random weights, no GIF.  Timed on the
    graph-replayed path (afldm_amd.harness.CrossFrameSampler; the first call captures and is reported as `first_call_s`):
    wall time per call with its UNet and VAE parts, for the batched LOAD pass (one batch-16 run) and for the reference's
    one-run-per-offset loop, next to 50 x the plain batch-1 / batch-16 step of the same process."""
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.harness import shift_ldm
    from afldm_amd.pipelines.ldm_pipeline import MyLDMPipeline
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    dev = torch.device("cuda", torch.cuda.current_device())
    unet, vae = build_unet(dtype, dev), build_vae(dtype, dev)
    pipe = MyLDMPipeline(vae, unet, ffhq_ddim_scheduler())
    pipe.set_progress_bar_config(disable=True)
    out = {"procedure": f"shift_ldm: STORE pass + {offsets} shifted LOAD passes, {steps} DDIM steps, batch 1, AF-VAE decode of "
                        f"{1 + offsets} latents, masked metrics (reference scripts/shift_ldm_ffhq.py:85-151)",
           "dtype": "bf16" if dtype == torch.bfloat16 else "fp32"}
    for name, batched in (("batched_load", True), ("sequential_load", False)):
        recs = []
        for rep in range(3):
            tm = {}
            t0 = time.perf_counter()
            _, errs = shift_ldm(pipe, steps, offsets, output_path=None, generator=torch.Generator().manual_seed(5 + rep),
                                batch_offsets=batched, reference_exact=False, timings=tm)
            tm["wall_s"] = time.perf_counter() - t0
            recs.append(tm)
        best = min(recs[1:], key=lambda r: r["total_s"])
        out[name] = dict(total_s=round(best["total_s"], 4), unet_s=round(best["unet_s"], 4), vae_s=round(best["vae_s"], 4),
                         shift_s=round(best.get("shift_s", 0.0), 4), metric_s=round(best.get("metric_s", 0.0), 4),
                         frames_s=round(best.get("frames_s", 0.0), 4), other_s=round(best.get("other_s", 0.0), 4),
                         first_call_s=round(recs[0]["total_s"], 3),
                         mask_mse_first_last=[float(f"{errs[0]:.4e}"), float(f"{errs[-1]:.4e}")])
    # the plain sampler at the two batch sizes of the procedure (no cross-frame processors, graph replay): the yardstick
    plain = {}
    for b in (1, offsets):
        eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), b, steps, use_graph=True)
        z = torch.randn(b, 4, 32, 32, generator=torch.Generator().manual_seed(1))
        eng.run(z)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(z)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        plain[b] = median(ts)
        del eng
    out["plain_sampler_s"] = {f"batch_{b}": round(v, 4) for b, v in plain.items()}
    out["unet_part_over_plain"] = round(out["batched_load"]["unet_s"] / (plain[1] + plain[offsets]), 3)
    # the eager loop that follows the reference statement by statement (use_graph=False), once warm
    tm = {}
    shift_ldm(pipe, steps, offsets, output_path=None, generator=torch.Generator().manual_seed(5), reference_exact=False,
              use_graph=False)
    shift_ldm(pipe, steps, offsets, output_path=None, generator=torch.Generator().manual_seed(5), reference_exact=False,
              use_graph=False, timings=tm)
    out["eager_batched_load"] = dict(total_s=round(tm["total_s"], 4), unet_s=round(tm["unet_s"], 4), vae_s=round(tm["vae_s"], 4))
    del pipe, unet, vae
    torch.cuda.empty_cache()
    return out


def i2sb_c5(batch=32, dtype=torch.bfloat16, steps=100):
    """BASELINE configs[4]'s sampler, per-GPU share (batch 256 over 8 GPUs = 32 per GPU): the 99 UNet evaluations of the 100-step
    I2SB bridge (deterministic, unclipped - what scripts/shift_ldm_sr.py runs; reference i2sb_pipeline.py:48-56) on the FFHQ-size AF-UNet,
    replayed as HIP graphs (I2SBLDMPipeline._bridge -> DenoiseEngine over scheduler.ode_schedule) and as the eager loop."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    dev = torch.device("cuda", torch.cuda.current_device())
    unet = build_unet(dtype, dev)
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    pipe = I2SBLDMPipeline(None, unet, I2SBScheduler.from_config(cfg))
    pipe.set_progress_bar_config(disable=True)
    start = (torch.randn(batch, 4, 32, 32, generator=torch.Generator().manual_seed(3)) * 0.5).to(dev).to(dtype)
    out = {}
    for name, graph, reps in (("graph", True, 3), ("eager", False, 1)):
        pipe._bridge(start, steps, True, None, use_graph=graph)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = pipe._bridge(start, steps, True, None, use_graph=graph)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name] = median(ts)
        finite = bool(torch.isfinite(y.float()).all().item())
    del pipe, unet
    torch.cuda.empty_cache()
    ev = steps - 1
    return dict(workload=f"I2SB ODE bridge, {ev} UNet evaluations, batch {batch} (BASELINE configs[4], per-GPU share), FFHQ-size AF-UNet",
                dtype="bf16" if dtype == torch.bfloat16 else "fp32", seconds=round(out["graph"], 4),
                ms_per_evaluation=round(1e3 * out["graph"] / ev, 4), value=round(batch * ev / out["graph"], 1), unit="evaluations/s",
                eager_seconds=round(out["eager"], 4), outputs_finite=finite)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--workload", default="unet", choices=["unet", "vae", "harness"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch 1 / 8 / fp32 / AF-VAE side measurements")
    args = ap.parse_args()

    from afldm_amd import parallel
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    rank, world, local = parallel.init_distributed()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    if args.workload == "vae":
        r = vae_workload(batch=args.batch if args.batch != 64 else 128, dtype=dtype)
        if rank == 0:
            print(json.dumps({"metric": "AF-VAE encode+decode images/sec, 256x256, batch 128 (BASELINE configs[3])",
                              "value": r["value"], "unit": "images/s", "n_gpus": 1, "higher_is_better": True,
                              "dtype": args.dtype, "data": "synthetic (seeded random AF-VAE weights, uniform images)",
                              "config": r}), flush=True)
        return

    if args.workload == "harness":
        r = harness_c1(dtype=dtype)
        if rank == 0:
            print(json.dumps({"metric": "shift_ldm wall seconds per call (STORE + 16 LOAD passes, 50 steps, FFHQ AF-UNet + AF-VAE)",
                              "value": r["batched_load"]["total_s"], "unit": "s", "n_gpus": 1, "higher_is_better": False,
                              "dtype": args.dtype, "data": "synthetic (seeded random weights)", "config": r}), flush=True)
        return

    B = args.batch
    total = B * world
    unet = build_unet(dtype, dev)
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), B, 50, use_graph=not args.no_graph)
    s, e = parallel.shard_range(total, rank, world)
    noise = parallel.global_noise(total, (4, 32, 32), 1234)[s:e].to(dev)     # resident before timing
    final = torch.empty((total, 4, 32, 32), dtype=torch.float32, device=dev) if world > 1 else None
    pos = [0]

    def run_steps(k, gather=True):
        done = 0
        while done < k:
            if pos[0] == 0:
                eng.reset(noise)                  # device->device copy of 1 MiB + counter reset
            n = min(50 - pos[0], k - done)
            eng.step(n)
            done += n
            pos[0] = (pos[0] + n) % 50
        if world > 1 and gather:
            parallel.all_gather_into(final, eng.lat)        # the ONE collective of the sampler (RCCL all_gather_into_tensor)

    run_steps(args.warmup if args.warmup > 0 else 1, gather=False)       # includes graph capture
    if world > 1:
        # RCCL builds its communicator rings / channels lazily on the first collective of a kind:
        # that one-off set-up belongs to the warm-up, not to the timed steps
        parallel.all_gather_into(final, eng.lat)
    pos[0] = 0
    regions = timed_regions(run_steps, args.steps, max(1, args.regions), world, dev)
    dt = median(regions)
    finite = bool(torch.isfinite(eng.lat).all().item())
    # the process group as it actually ran (every rank takes part): world size, backend, each rank's device, and the
    # sampler's one collective timed on its own + verified block by block
    rccl = parallel.rccl_record(dev, local, payload=eng.lat) if world > 1 else None

    if rank == 0:
        out = {
            "metric": "denoise-steps/sec + ms/step, FFHQ-256 AF-UNet, batch 64 @1/2/4/8 GPU",     # BASELINE.json "metric", verbatim
            "value": round(total * args.steps / dt, 2),
            "unit": "denoise-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (seeded random FFHQ AF-UNet weights, CPU-seeded noise)",
            "config": {"workload": "FFHQ-256 AF-UNet single denoise step, batch 64 per GPU (BASELINE configs[1])",
                       "batch_per_gpu": B, "global_batch": total, "sharding": "batch, one all-gather of final latents per region",
                       "hip_graph": not args.no_graph, "latents_finite": finite,
                       "loop_invariant_hoisting": "time_proj -> time_embedding MLP -> SiLU -> the 27 time_emb_proj layers depend on "
                                                  "the timestep only and are tabulated once per 50-step schedule in DenoiseEngine.__init__ "
                                                  "(outside the timed region; < 0.1 % of the step's flops, 6 launches): a timed step copies "
                                                  "its table row instead of recomputing them as the reference does per step",
                       "timing": f"median of {len(regions)} timed regions of {args.steps} steps, each bracketed by barrier + "
                                 "synchronize, MAX over ranks",
                       "regions_ms_per_step": [round(1e3 * r / args.steps, 4) for r in regions]},
        }
        out["rccl"] = rccl if rccl is not None else {"world_size": 1, "backend": None,
                                                     "note": "single process: no process group, no collective"}
        out["box"] = box_record(dev)
        if not args.no_roofline:
            roof, fam = roofline_pass(unet, B, dtype)
            peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
            rate = roof.pop("step_dense_tflop") / (dt / args.steps)
            out["step_dense"] = {"tflop_per_step": round(rate * dt / args.steps, 4), "tflops": round(rate, 1), "peak": peak,
                                 "frac": round(rate / peak, 4), "af_filter_tflop_per_step": roof.pop("step_af_filter_tflop"),
                                 "note": "algorithmic flops of the step's convolutions + linear layers + attention over the timed step"}
            if roof.get("bound") == "mfma" and dtype == torch.bfloat16 and out["box"].get("mfma_tflops_random_operands"):
                # next to the datasheet peak: the MFMA rate THIS box sustains on random bf16 operands (the chip clocks down
                # under data that toggles the multipliers; box.mfma_tflops is the same loop on near-constant operands)
                rp = out["box"]["mfma_tflops_random_operands"]
                roof["peak_random_operands"] = rp
                roof["frac_of_random_operand_peak"] = round(roof["achieved"] / rp, 4)
            out["roofline"] = roof
            out["kernel_families"] = fam
            fam_sum = sum(f["ms_per_step"] for f in fam.values())
            out["kernel_families_note"] = (f"every family is timed as a back-to-back HIP-graph replay of its own launches of one step: the families "
                                           f"sum to {fam_sum:.3f} ms against the {1e3 * dt / args.steps:.3f} ms timed step (MFMA-heavy launches replayed "
                                           "back to back run slower than interleaved with the step's other kernels), so per-family ms / tflops and "
                                           "roofline.frac are slightly pessimistic, never optimistic")
        del eng
        if world == 1 and not args.no_extras:
            # north_star: "batch 1/8/64 on 1 GPU" + the reference's precision, and configs[3], driver-timed in this run
            del unet
            torch.cuda.empty_cache()
            out["other_configs"] = [side_config(1, torch.bfloat16), side_config(8, torch.bfloat16),
                                    side_config(64, torch.float32, steps=10), side_config(1, torch.float32)]
            out["concurrent_jobs"] = [concurrent_engines(2), concurrent_engines(3)]
            out["vae_c4"] = vae_workload()
            out["harness_c1"] = harness_c1()
            out["i2sb_c5"] = i2sb_c5()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    parallel.barrier()


if __name__ == "__main__":
    main()
