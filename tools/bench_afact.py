#!/usr/bin/env python
"""GroupNorm + WarpedNonlinearity at batch 64 per plane size / channel count, us per launch (HIP events over 50 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import ops
from bench_kernels import timeit
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("AFLDM_AF"))
for N, C in ((8, 384), (8, 768), (16, 384), (16, 768), (16, 192), (32, 192), (32, 384)):
    B, G = 64, 32
    x = (torch.randn(B, N, N, C) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    st = ops.gn_stats(x, G)
    if N >= 16:      # as a producing convolution leaves them: few splits
        S = 2 if N == 16 else 4
        rows = x.float().view(B, S, N * N // S, C)
        st = ops.GNStats(torch.stack([rows.sum(2), (rows * rows).sum(2)], -1).contiguous(), None)
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    out = torch.empty_like(x)
    t = timeit(lambda: ops.af_act(x, None, st, gamma, beta, G, 1e-5, out=out), iters=50)
    mb = 2 * x.numel() * 2 / 1e6
    print(f"[{tag}] N={N:2d} C={C:3d}: {t:7.1f} us  ({mb / t * 1e3 / 1e3:5.2f} TB/s of in + out)", flush=True)
