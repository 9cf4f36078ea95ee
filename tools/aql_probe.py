#!/usr/bin/env python
"""What do the packet headers of the step cost?  (csrc/aqlq.cpp, afldm_amd/aql.py)

The batch-64 step graph is replayed with the dispatch packets re-headed on their way to the hardware queue:
  base      as the runtime emits them (barrier bit + acquire / release fences on every launch)
  nofence   barrier bits kept, no acquire / release fences        (results may be stale: TIMING ONLY)
  nobarrier barrier bits cleared on every launch                  (launches overlap freely: results invalid, TIMING ONLY)
  both      neither
The invalid modes bound what ANY dependency-aware schedule on this queue could gain.  MODES env selects a subset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from afldm_amd import aql
if not os.environ.get("NOHOOK"):
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
eng.step(6)
torch.cuda.synchronize()
print("tool loaded:", aql.loaded(), aql.counts(), flush=True)
if os.environ.get("NOHOOK"):
    for _ in range(3):
        eng.graph.replay()
    torch.cuda.synchronize()
    eng.reset(noise); eng.step(50); torch.cuda.synchronize()
    print("no hook: replays fine", flush=True)
    sys.exit(0)
if not aql.loaded():
    sys.exit("HSA_TOOLS_LIB was not honoured")

aql.record(True)
eng.graph.replay()
torch.cuda.synchronize()
aql.record(False)
recs = aql.records()
n = len(recs)
print(f"dispatch packets of one step graph: {n}")
hdrs, cnt = np.unique(recs["header"], return_counts=True)
for h, c in zip(hdrs, cnt):
    print(f"  header 0x{int(h):04x} x {c}: type {int(h) & 0xff} barrier {(int(h) >> 8) & 1} acquire {(int(h) >> 9) & 3} release {(int(h) >> 11) & 3}")
print("  completion signals set on", int((recs["completion"] != 0).sum()), "packets")
for r in recs[:12]:
    print(f"  hdr 0x{int(r['header']):04x} grid {tuple(int(g) for g in r['grid'])} wg {tuple(int(g) for g in r['wg'])} lds {int(r['group_bytes'])} kobj 0x{int(r['kernel_object']):x}")


def timed(policy, reps=5):
    ts = []
    for _ in range(reps + 1):
        eng.reset(noise)
        torch.cuda.synchronize()
        if policy is not None:
            aql.arm(policy, K * n)
        t0 = time.perf_counter()
        eng.step(K)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
        left = aql.armed_left()
        aql.disarm()
        assert policy is None or left == 0, left
    ts = ts[1:]
    return sorted(ts)[len(ts) // 2], ts


modes = {"base": None, "nofence": np.full(n, 6, np.uint8), "nobarrier": np.full(n, 1, np.uint8), "both": np.full(n, 7, np.uint8),
         "norelease": np.full(n, 4, np.uint8), "noacquire": np.full(n, 2, np.uint8)}
sel = os.environ.get("MODES", "base,nofence,norelease,noacquire,nobarrier,both,base").split(",")
for m in sel:
    med, ts = timed(modes[m])
    print(f"{m:10s} {med:.4f} ms/step  {['%.4f' % t for t in ts]}   {aql.counts()}", flush=True)
