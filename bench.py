#!/usr/bin/env python
"""bench.py — denoise-steps/sec of the FFHQ-256 alias-free UNet + DDIM update on MI355X.

A "step" is one denoising step of the whole local batch: timestep select -> NCHW->NHWC ->
AF-UNet forward (~450 HIP kernels from libafldm_hip.so) -> DDIM update, replayed as one
captured HIP graph.  Workload = BASELINE.json configs[1] (batch 64 per GPU, bf16 storage /
MFMA with fp32 accumulation), synthetic: seeded random weights of the FFHQ architecture
(conv_out x0.1), CPU-seeded noise, inputs resident in HBM before the timed region.
N > 1: launched by torch.distributed.run, one rank per GPU, batch-sharded (weak scaling: 64 per
GPU), no per-step communication, ONE RCCL all-gather of the final latents inside the timed
region.  Rank 0 prints one JSON line.
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
    dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
    name, d = dom
    # HBM bytes per launch of the dominant family from the PMC counters (FETCH_SIZE / WRITE_SIZE in
    # separate rocprofv3 passes, gfx950 x2 read correction): collected offline by
    # profiles/run_pmc_conv3x3.sh (a PMC pass cannot run inside the timed process) and committed.
    traffic = None
    try:
        tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "conv3x3_traffic.json")))
        if tj.get("family") == name and batch == 64 and dtype == torch.bfloat16:
            traffic = round(tj["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        traffic = None
    mfma_bound = name.startswith("conv") or name == "linear" or name == "attention" or name.startswith("af_act_N3") \
        or name.startswith("af_act_N16")
    if mfma_bound:
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        achieved = d["flops"] / d["ms"] / 1e9
        roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=traffic,
                    launches_per_step=d["launches"] // 3, avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2),
                    flops_per_launch=d["flops"] / d["launches"], algorithmic_bytes_per_launch=round(d["bytes"] / d["launches"]))
    else:
        achieved = d["bytes"] / d["ms"] / 1e6
        roof = dict(kernel=name, bound="hbm", achieved=round(achieved, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(achieved / PEAK_HBM_GBS, 4), traffic=None,
                    launches_per_step=d["launches"] // 3, avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2),
                    bytes_per_launch=d["bytes"] / d["launches"])
    return roof, fam


def cpu_baseline(batch=4, budget_s=12.0):
    """The oracle (CPU restatement, eager PyTorch fp32: torch.fft filters, F.conv2d, SDPA) on the host
    cores: a bounded sample of the bench workload (batch 4 instead of 64, as many denoise steps as fit
    ~12 s, at most 49).  A reported baseline, never the thing measured above."""
    from oracle import configs as oc, pipeline as op, unet as ou
    # eager ops at this size do not scale past a few tens of threads (256 threads measured 100x SLOWER
    # than 8 on the MI355X host: oversubscribed OpenMP barriers), so the thread count is capped.
    cores = min(os.cpu_count() or 1, 16)
    sd = ou.init_unet_params(oc.FFHQ_UNET, seed=0, conv_out_scale=0.1)
    sec, steps = op.time_denoise_steps(sd, oc.FFHQ_UNET, batch=batch, steps=49, threads=cores, budget_s=budget_s)
    return dict(value=round(batch / sec, 4), unit="denoise-steps/s", cores=cores, kind="port",
                ms_per_step=round(sec * 1e3, 2),
                sample=f"oracle CPU restatement, FFHQ AF-UNet + DDIM update, batch {batch} fp32, {steps} steps "
                       f"({sec * steps:.1f} s) after 1 warm-up step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
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
    B = args.batch
    total = B * world

    unet = build_unet(dtype, dev)
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), B, 50, use_graph=not args.no_graph)
    s, e = parallel.shard_range(total, rank, world)
    noise = parallel.global_noise(total, (4, 32, 32), 1234)[s:e].to(dev)     # resident before timing
    final = torch.empty((total, 4, 32, 32), dtype=torch.float32, device=dev) if world > 1 else None

    def run_steps(k):
        done = 0
        while done < k:
            pos = int(done % 50)
            if pos == 0:
                eng.reset(noise)                  # device->device copy of 1 MiB + counter reset
            n = min(50 - pos, k - done)
            eng.step(n)
            done += n

    run_steps(args.warmup if args.warmup > 0 else 1)       # includes graph capture
    if world > 1:
        # RCCL builds its communicator rings / channels lazily on the first collective of a kind:
        # that one-off set-up belongs to the warm-up, not to the timed steps
        torch.distributed.all_gather_into_tensor(final, eng.lat)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    if world > 1:
        torch.distributed.all_gather_into_tensor(final, eng.lat)        # the ONE collective of the sampler
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(eng.lat).all().item())

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
                       "batch_per_gpu": B, "global_batch": total, "sharding": "batch, one all-gather of final latents",
                       "hip_graph": not args.no_graph, "latents_finite": finite},
        }
        if not args.no_roofline:
            roof, fam = roofline_pass(unet, B, dtype)
            out["roofline"] = roof
            out["kernel_families"] = fam
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    parallel.barrier()


if __name__ == "__main__":
    main()
