#!/usr/bin/env python
"""Phase stamps of the merged launches (afldm_af_act_conv2d_trace): where a workgroup's time goes in each phase, and how
far apart the members of a cluster arrive.  CASE=<index into bench_actconv.CASES>  FORM=pre|post|both"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from afldm_amd import _exp, _lib, ops  # noqa: E402
from bench_actconv import CASES  # noqa: E402

NAMES = {(0, 1): "pre: prologue (constants, statistics, first tile requested)", (1, 2): "pre: activation items",
         (2, 3): "hand-over 1: stores acknowledged + barrier", (3, 4): "hand-over 1: arrive -> cluster complete",
         (4, 5): "hand-over 1: leave + barrier", (5, 6): "convolution tile (after pre)", (0, 6): "convolution tile (no pre)",
         (6, 7): "hand-over 2: stores acknowledged + barrier", (7, 8): "hand-over 2: arrive -> cluster complete",
         (8, 9): "hand-over 2: leave + barrier", (9, 10): "post: prologue", (10, 11): "post: activation items"}


def main():
    B = int(os.environ.get("B", "64"))
    form = os.environ.get("FORM", "pre")
    name, N, C1, C2, Cout, use_temb, use_res, S = CASES[int(os.environ.get("CASE", "0"))]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    Ct = C1 + C2
    x1 = torch.randn(B, N, N, C1, generator=g).to(dev, torch.bfloat16)
    x2 = torch.randn(B, N, N, C2, generator=g).to(dev, torch.bfloat16) if C2 else None
    w = ops.pack_weight((torch.randn(Cout, Ct, 3, 3, generator=g) * (9 * Ct) ** -0.5).to(dev), torch.bfloat16)
    bias = torch.randn(Cout, generator=g).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(Ct, generator=g)).to(dev), (0.1 * torch.randn(Ct, generator=g)).to(dev)
    g2, b2 = (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev), (0.1 * torch.randn(Cout, generator=g)).to(dev)
    temb = torch.randn(B, Cout, generator=g).to(dev, torch.bfloat16) if use_temb else None
    res = torch.randn(B, N, N, Cout, generator=g).to(dev, torch.bfloat16) if use_res else None
    stats = ops.gn_stats(x1, 32, x2=x2)
    a_in = ops.af_act(x1, x2, stats, gamma, beta, 32, 1e-6)
    pre = (stats, gamma, beta, 32, 1e-6) if form in ("pre", "both") else None
    post = (g2, b2, 32, 1e-6) if form in ("post", "both") else None
    run = lambda: ops.act_conv_act(x1 if pre else a_in, x2 if pre else None, pre, w, bias, temb, Cout if use_temb else 0, res, True, post=post)
    for _ in range(3):
        assert run() is not None
    nwg = B * (N * N // (256 if N == 32 else 128)) * (Cout // 192)
    buf = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
    _exp.lib().afldm_af_act_conv2d_trace(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    _exp.lib().afldm_af_act_conv2d_trace(None)
    st = buf.cpu().double()
    used = [i for i in range(12) if float(st[:, i].min()) > 0]
    pre_, post_ = pre, post
    t0 = st[:, 0].min()
    last = used[-1]
    span = st[:, last].max() - t0
    us = e0.elapsed_time(e1) * 1e3
    print(f"{name} B={B} form={form}: launch (events, eager) {us:.1f} us; first start -> last stamp {span:.0f} ticks")
    for a, b in zip(used[:-1], used[1:]):
        d = st[:, b] - st[:, a]
        print(f"  {NAMES.get((a, b), f'{a}->{b}'):64s} min {d.min():8.0f}  median {d.median():8.0f}  max {d.max():8.0f} ticks")
    if float(st[:, 12].min()) > 0:
        base = 9 if post and not pre else 0
        seq = [base, 12, 13, 14, 15, 10 if post else 1]
        lab = ["small loads issued", "constants + first tile issued", "statistics folded", "barrier", "tables written"]
        if not (pre and post):
            for (a, b), l in zip(zip(seq[:-1], seq[1:]), lab):
                d = st[:, b] - st[:, a]
                print(f"    prologue detail: {l:40s} min {d.min():8.0f}  median {d.median():8.0f}  max {d.max():8.0f} ticks")
    print(f"  start skew over workgroups {st[:, 0].max() - t0:.0f} ticks; last-stamp skew {st[:, last].max() - st[:, last].min():.0f} ticks; error word {ops.actconv_error()}")


if __name__ == "__main__":
    main()
