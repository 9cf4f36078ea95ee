#!/usr/bin/env python
"""GroupNorm + WarpedNonlinearity over the up blocks' virtual concats (x1 | x2) at batch 64: us per launch, HIP-graph timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import ops
from bench_kernels import timeit_graph
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("AFLDM_AF"))
for N, C1, C2 in ((16, 768, 384), (16, 384, 384), (16, 384, 192), (32, 384, 192), (32, 192, 192), (16, 384, 0), (32, 192, 0)):
    B, G = 64, 32
    x1 = (torch.randn(B, N, N, C1) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    x2 = (torch.randn(B, N, N, C2) * 1.3 + 0.2).to(torch.bfloat16).cuda() if C2 else None
    st = ops.gn_stats(x1, G, x2=x2)
    C = C1 + C2
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    out = torch.empty(B, N, N, C, dtype=torch.bfloat16, device="cuda")
    t = timeit_graph(lambda: ops.af_act(x1, x2, st, gamma, beta, G, 1e-5, out=out))
    mb = 2 * out.numel() * 2 / 1e6
    print(f"[{tag}] N={N:2d} C={C1}+{C2}: {t:7.1f} us  ({mb / t:5.2f} TB/s of in + out)", flush=True)
