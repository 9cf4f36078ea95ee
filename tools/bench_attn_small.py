#!/usr/bin/env python
"""Small-plane attention sites at batch 64: the fused launch against projection GEMM + attention (HIP-graph timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import ops
from bench_kernels import timeit_graph
for T, C, heads in ((64, 384, 16), (16, 768, 32)):
    B = 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, C, generator=g).cuda().to(torch.bfloat16)
    wp = ops.pack_weight((torch.randn(3 * C, C, generator=g) * C ** -0.5).cuda(), torch.bfloat16)
    b = torch.randn(3 * C, generator=g).cuda()

    def two():
        qk, vt = ops.linear_split(x, wp, b, 2 * C)
        return ops.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, scale=24 ** -0.5)
    t2 = timeit_graph(two)
    t1 = timeit_graph(lambda: ops.attn_small_fused(x, wp, b, heads, 24 ** -0.5))
    print(f"T={T} C={C}: projection + attention {t2:6.1f} us | fused {t1:6.1f} us", flush=True)
