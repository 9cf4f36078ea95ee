#!/usr/bin/env python
"""One AF-VAE-sized 3x3 convolution on the one-tile-per-workgroup halo kernel (variant 58) and on persistent tiles (63)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import _lib, ops
from bench_kernels import timeit
for (B, H, Cin, Cout) in ((16, 256, 128, 128), (32, 128, 256, 256), (32, 128, 128, 128), (64, 64, 512, 512), (64, 32, 192, 192)):
    x = torch.randn(B, H, H, Cin).to(torch.bfloat16).cuda()
    w = ops.pack_weight((torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)).cuda(), torch.bfloat16)
    b = torch.zeros(Cout).cuda()
    y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device="cuda")
    res = {}
    resid = torch.randn(B, H, H, Cout).to(torch.bfloat16).cuda() if os.environ.get("RES") else None
    for v in ((58, 63) if H > 32 else (41, 46, 64)):
        try:
            _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
            a = ops.conv_args(x, w, b, out=y)
            got = _lib.lib.afldm_conv2d_variant(ctypes.byref(a)) & 255
            t = timeit(lambda: ops.conv2d(x, w, b, out=y, residual=resid, want_stats=True), iters=20)
        finally:
            _lib.lib.afldm_conv2d_tune(-1, -1)
        fl = 2.0 * B * H * H * Cout * Cin * 9
        res[v] = f"variant {v} (ran {got}): {t:8.1f} us = {fl / t / 1e6:6.0f} TFLOP/s"
    print(f"B={B} {H}x{H} {Cin}->{Cout}: " + " | ".join(res.values()), flush=True)
