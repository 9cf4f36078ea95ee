#!/usr/bin/env python
"""Bit-compare two conv variants (afldm_conv2d_tune) on the low-level shapes and time them (HIP graph, 20 launches).
usage: check_conv_variants.py N VA VB [splitk]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import _lib, ops
from bench_kernels import timeit_graph
N, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sk = int(sys.argv[4]) if len(sys.argv) > 4 else -1
B = 64
shapes = [(384, 384), (768, 384), (1152, 384), (768, 768)] if N == 8 else [(384, 768), (768, 768), (1152, 768), (1536, 768)]
g = torch.Generator().manual_seed(0)
for Ct, Cout in shapes:
    x = torch.randn(B, N, N, Ct, generator=g).cuda().to(torch.bfloat16)
    w = ops.pack_weight((torch.randn(Cout, Ct, 3, 3, generator=g) * (9 * Ct) ** -0.5).cuda(), torch.bfloat16)
    bias = torch.randn(Cout, generator=g).cuda()
    temb = torch.randn(B, Cout, generator=g).cuda().to(torch.bfloat16)
    res = torch.randn(B, N, N, Cout, generator=g).cuda().to(torch.bfloat16)
    outs, ts = [], []
    for v in (va, vb):
        _lib.lib.afldm_conv2d_tune(v, sk)
        run = lambda: ops.conv2d(x, w, bias, temb=temb, temb_stride=Cout, residual=res, want_stats=True)
        y = run()
        torch.cuda.synchronize()
        outs.append((y.clone(), y.gn_partial.clone()))
        ts.append(timeit_graph(run))
    _lib.lib.afldm_conv2d_tune(-1, -1)
    same = torch.equal(outs[0][0], outs[1][0])
    st_same = outs[0][1].shape == outs[1][1].shape and bool(torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-3))
    d = float((outs[0][0].float() - outs[1][0].float()).abs().max())
    print(f"N={N} {Ct}->{Cout}: variant {va} {ts[0]:6.1f} us | variant {vb} {ts[1]:6.1f} us | outputs identical {same} (max diff {d:.3g}) stats close {st_same}", flush=True)
