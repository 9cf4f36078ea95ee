#!/usr/bin/env python
"""Checksums + timings of the UNet's small resampling sites (run once with AFLDM_NO_RESAMPLE_SMALL=1 and once without: the
one-launch form must print the same checksums as the two-pass form)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import ops
from bench_kernels import timeit_graph
g = torch.Generator().manual_seed(3)
for dt in (torch.bfloat16, torch.float32):
    for kind, N, C in (("down", 16, 384), ("down", 8, 768), ("down", 4, 768), ("up", 2, 768), ("up", 4, 768), ("up", 8, 384)):
        x = torch.randn(64, N, N, C, generator=g).cuda().to(dt)
        if kind == "down":
            f = lambda: ops.af_lpf_down2(x, want_stats=True)
        else:
            f = lambda: ops.af_up2(x)
        y = f()
        torch.cuda.synchronize()
        h = hashlib.sha1(y.float().cpu().numpy().tobytes()).hexdigest()[:12]
        st = getattr(y, "gn_partial", None)
        hs = hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:12] if st is not None else "-"
        t = timeit_graph(f) if dt == torch.bfloat16 else 0.0
        print(f"{str(dt)[6:]:9s} {kind:4s} N={N:2d} C={C}: out {h} stats {hs}  {t:6.1f} us", flush=True)
