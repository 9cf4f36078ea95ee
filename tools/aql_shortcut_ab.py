#!/usr/bin/env python
"""conv_shortcut beside its neighbour: the 17 shortcut GEMMs of a step do not depend on the activation they are launched
next to; with the barrier bit of the marked launch cleared (afldm_amd/aql.py) the two run concurrently.
AFLDM_SHORTCUT_ORDER=0/1/2 picks the neighbour.  Prints ms/step without and with the policy and the difference of the
results after 10 steps (must be 0: same kernels, same inputs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from afldm_amd import aql
aql.install()
import numpy as np
import torch
import bench
from afldm_amd.engine import DenoiseEngine
from afldm_amd import parallel
from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler

B = int(os.environ.get("B", "64"))
K = int(os.environ.get("K", "40"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
unet = bench.build_unet(torch.bfloat16, dev)
noise = parallel.global_noise(B, (4, 32, 32), 1234).to(dev)
eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), B, 50)
eng.reset(noise)
eng.step(6)                      # captures
torch.cuda.synchronize()
assert aql.loaded()
# one eager pass with the trace on
eng.reset(noise)
torch.cuda.synchronize()
aql.trace_begin()
eng._step()
torch.cuda.synchronize()
n, marks = aql.trace_end()
print(f"order {os.environ.get('AFLDM_SHORTCUT_ORDER', '0')}: {n} dispatches per step, {len(marks)} independent regions: {marks[:6]} ...", flush=True)
aql.record(True); eng.reset(noise); torch.cuda.synchronize(); c0 = aql.counts()["dispatch"]; eng.graph.replay(); torch.cuda.synchronize()
ng = aql.counts()["dispatch"] - c0
aql.record(False)
assert ng == n, (ng, n)
pol = aql.policy_from_marks(n, marks)


def run(policy, steps):
    eng.reset(noise)
    torch.cuda.synchronize()
    if policy is not None:
        aql.arm(policy, steps * n)
    t0 = time.perf_counter()
    eng.step(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    assert policy is None or aql.armed_left() == 0
    aql.disarm()
    return dt


def timed(policy, reps=5):
    ts = [run(policy, K) for _ in range(reps + 1)][1:]
    return sorted(ts)[len(ts) // 2], ts


for name, p in (("base", None), ("policy", pol), ("base", None), ("policy", pol)):
    med, ts = timed(p)
    print(f"{name:8s} {med:.4f} ms/step  {['%.4f' % t for t in ts]}", flush=True)
run(None, 10); ref = eng.lat.clone()
run(pol, 10); got = eng.lat.clone()
print("max |difference| after 10 steps:", float((got - ref).abs().max()), flush=True)
