#!/usr/bin/env python
"""Fixed cost versus per-K-step cost of the 3x3 halo kernels: the same output shape is run with the input channel count
varied, time = a + b * (K steps).  `a` is what a one-tile-per-CU launch pays outside its K loop (launch ramp, first patch,
epilogue, statistics); b * steps is the loop itself."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from afldm_amd import _lib, ops
from bench_kernels import timeit_graph


# PAIRS=1: each case also on a sibling variant (e.g. 41 -> 47, the 32x32x16 MFMA form), outputs compared bit for bit
SIB = {41: 47, 43: 48}
CASES = (  # B, H, Cout, variant, input channel counts
    (64, 32, 192, 41, (64, 128, 192, 256, 384, 576)),
    (64, 16, 384, 43, (128, 256, 384, 512, 768)),
    (64, 8, 384, 51, (128, 256, 384, 512, 768)),
    (64, 4, 768, 52, (256, 512, 768, 1024, 1536)),
)
if os.environ.get("ONLY"):                        # e.g. ONLY=32,16
    CASES = tuple(c for c in CASES if str(c[1]) in os.environ["ONLY"].split(","))
if os.environ.get("PAIRS"):
    CASES = tuple(c for c in CASES if c[3] in SIB)
    CASES = tuple(x for c in CASES for x in (c, (c[0], c[1], c[2], SIB[c[3]], c[4])))
ref_out = {}
for (B, H, Cout, v, cins) in CASES:
    pts = []
    for Cin in cins:
        g = torch.Generator().manual_seed(Cin + H)
        x = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).cuda()
        w = ops.pack_weight((torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda(), torch.bfloat16)
        b = torch.zeros(Cout).cuda()
        y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device="cuda")
        try:
            _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
            a = ops.conv_args(x, w, b, out=y)
            need = _lib.lib.afldm_conv2d_workspace(ctypes.byref(a))
            ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device="cuda")
            a.workspace, a.workspace_bytes = ops.ptr(ws), need
            S = _lib.lib.afldm_conv2d_stats_splits(ctypes.byref(a))
            st = torch.empty((B, S, Cout, 2), dtype=torch.float32, device="cuda")
            a.stats_out = ops.ptr(st)
            got = _lib.lib.afldm_conv2d_variant(ctypes.byref(a))
            t = timeit_graph(lambda: ops.conv2d_launch(a))
        finally:
            _lib.lib.afldm_conv2d_tune(-1, -1)
        torch.cuda.synchronize()
        key = (H, Cin)
        if key in ref_out:
            same = bool(torch.equal(ref_out[key][0], y)) and bool(torch.equal(ref_out[key][1], st))
            print(f"    output + statistics identical to variant {ref_out[key][2]}: {same}", flush=True)
        else:
            ref_out[key] = (y.clone(), st.clone(), got & 255)
        steps = 9 * Cin // 64
        fl = 2.0 * B * H * H * Cout * Cin * 9
        pts.append((steps, t))
        print(f"  B={B} {H}x{H} {Cin:4d}->{Cout} variant {got & 255} slabs {(got >> 8) & 255}: {steps:4d} steps {t:7.1f} us "
              f"{fl / t / 1e6:6.0f} TFLOP/s", flush=True)
    s, t = np.array(pts, dtype=np.float64).T
    bb, aa = np.polyfit(s, t, 1)
    print(f"{H}x{H} variant {v}: fixed {aa:6.2f} us + {bb * 1e3:6.1f} ns per 64-channel tap step", flush=True)
