#!/usr/bin/env python
"""One batch-64 engine against TWO free-running batch-32 engines on two streams (each replaying its own HIP graph; no fork / join
between them): do the half-size kernels of one stream fill what the other leaves idle?  ms per 64-sample step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from afldm_amd.engine import DenoiseEngine
from afldm_amd import parallel
from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dtype = torch.bfloat16
unet = bench.build_unet(dtype, dev)
noise = parallel.global_noise(64, (4, 32, 32), 1234).to(dev)
K = 40
NS = int(os.environ.get("NSTREAMS", "2"))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    return sorted(ts)[len(ts) // 2], ts


one = DenoiseEngine(unet, ffhq_ddim_scheduler(), 64, 50)
def run_one():
    one.reset(noise)
    one.step(K)
m, ts = timed(run_one)
print(f"one engine, batch 64:            {m:.4f} ms/step  {['%.4f' % t for t in ts]}", flush=True)

per = int(os.environ.get("PER", str(64 // NS)))          # PER=64 NSTREAMS=2: two batch-64 engines (128 samples per "step")
noise = parallel.global_noise(per * NS, (4, 32, 32), 1234).to(dev) if per * NS != 64 else noise
streams = [torch.cuda.Stream() for _ in range(NS)]
engs = []
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        e = DenoiseEngine(unet, ffhq_ddim_scheduler(), per, 50)
        e.reset(noise[i * per:(i + 1) * per])
        e.step(1)
        engs.append(e)
torch.cuda.synchronize()
def run_two():
    main = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            engs[i].reset(noise[i * per:(i + 1) * per])
    # interleave the replays so that both queues are fed from the start
    spg = engs[0].steps_per_graph
    for _ in range(K // spg):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                engs[i].step(spg)
    for s in streams:
        main.wait_stream(s)
m2, ts2 = timed(run_two)
print(f"{NS} engines, batch {per} each, {NS} streams: {m2:.4f} ms per {per * NS}-sample step = {m2 * 64 / (per * NS):.4f} per 64 samples  {['%.4f' % t for t in ts2]}", flush=True)
if per * NS != 64:
    sys.exit(0)
# same result?
one.reset(noise); one.step(10); torch.cuda.synchronize()
ref = one.lat.clone()
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        engs[i].reset(noise[i * per:(i + 1) * per]); engs[i].step(10)
torch.cuda.synchronize()
got = torch.cat([e.lat for e in engs], 0)
print("max |difference| after 10 steps:", float((got - ref).abs().max()), " rel-RMS:", float(((got - ref).pow(2).mean() / ref.pow(2).mean()).sqrt()))
