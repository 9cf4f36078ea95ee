#!/usr/bin/env python
"""Phase stamps of k_af_act_plane (afldm_af_act_trace): where a workgroup's item time goes.  Only differences inside one
workgroup are meaningful (the shader clocks of the XCDs are not aligned); ticks are calibrated against the launch time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from afldm_amd import _lib, ops
from bench_kernels import timeit

NAMES = ["start -> item top (constants, fetch issue)", "stats in line + barrier", "tile -> LDS (waits for the fetch)",
         "barrier", "MFMA passes (+ next fetch issue)", "stats fold + barrier", "output staging", "barrier",
         "output stores issued"]
SHAPES = ((32, 192), (16, 384))
if os.environ.get("SHAPES"):          # e.g. SHAPES=32x576,32x384
    SHAPES = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(","))
for N, C in SHAPES:
    B, G = 64, 32
    x = (torch.randn(B, N, N, C) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    S = 2 if N == 16 else 4
    rows = x.float().view(B, S, N * N // S, C)
    st = ops.GNStats(torch.stack([rows.sum(2), (rows * rows).sum(2)], -1).contiguous(), None)
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    out = torch.empty_like(x)
    fn = lambda: ops.af_act(x, None, st, gamma, beta, G, 1e-5, out=out)
    t = timeit(fn, iters=50)
    tr = torch.zeros(4096 * 4 * 2 * 10, dtype=torch.int64, device="cuda")
    _lib.lib.afldm_af_act_trace(tr.data_ptr())
    try:
        fn(); torch.cuda.synchronize()
        tr.zero_()
        fn(); torch.cuda.synchronize()
    finally:
        _lib.lib.afldm_af_act_trace(None)
    a = tr.cpu().numpy().reshape(4096, 4, 2, 10)
    used = a[:, 0, 0, 0] != 0
    a = a[used].astype(np.float64)
    nwg = a.shape[0]
    items = int((a[:, 0, 1, 9] != 0).sum())
    # tick: the longest workgroup life against the launch time is a lower bound of ticks per us
    life = (a[:, :, :, 9].max(axis=(1, 2)) - a[:, :, 0, 0].min(axis=1))
    print(f"N={N} C={C}: {t:.1f} us per launch, {nwg} workgroups, {items} with a second item; workgroup life "
          f"median {np.median(life):.0f} max {life.max():.0f} shader clocks (max / launch time = {life.max() / t:.0f} per us)")
    for it in range(2):
        sel = a[:, :, it, :]
        ok = sel[:, 0, 9] != 0
        if not ok.any():
            continue
        d = np.diff(sel[ok], axis=2)              # [wg, wave, 9]
        d[:, :, 0] = sel[ok][:, :, 1] - sel[ok][:, :, 0] if it == 0 else 0
        print(f"  item {it} (wave 0 / wave 3 medians, shader clocks):")
        for k in range(9):
            print(f"    {NAMES[k]:48s} {np.median(d[:, 0, k]):8.0f} {np.median(d[:, 3, k]):8.0f}")
        tot = sel[ok][:, 0, 9] - sel[ok][:, 0, 1 if it else 0]
        print(f"    {'item total':48s} {np.median(tot):8.0f}")
