#!/usr/bin/env python
"""Plane activation at batch 64: NHWC in/out, NHWC in -> blocks out, blocks in -> blocks out (us per launch, HIP-graph timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import ops
from bench_kernels import timeit_graph
for N, C in ((32, 192), (16, 384), (16, 192)):
    B, G = 64, 32
    x = (torch.randn(B, N, N, C) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    st = ops.gn_stats(x, G)
    xb = x.clone(); xb.c8 = True; xb.gn_partial = x.gn_partial          # (same bytes read as blocks: timing only)
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    out = torch.empty_like(x)
    t0 = timeit_graph(lambda: ops.af_act(x, None, st, gamma, beta, G, 1e-5, out=out))
    t1 = timeit_graph(lambda: ops.af_act(x, None, st, gamma, beta, G, 1e-5, out=out, out_c8=True))
    t2 = timeit_graph(lambda: ops.af_act(xb, None, st, gamma, beta, G, 1e-5, out=out, out_c8=True))
    t3 = timeit_graph(lambda: ops.af_act(xb, None, st, gamma, beta, G, 1e-5, out=out, out_c8=False))
    print(f"N={N} C={C}: NHWC->NHWC {t0:.1f} | NHWC->blocks {t1:.1f} | blocks->blocks {t2:.1f} | blocks->NHWC {t3:.1f} us", flush=True)
