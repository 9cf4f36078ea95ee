#!/usr/bin/env python
"""Do parallel branches of a captured HIP graph overlap on the device?  Two independent chains of a tiny-grid kernel
(afldm_silu on 256 KB: one wave of workgroups on a fraction of the CUs), captured on one stream vs forked on two."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd import ops  # noqa: E402


def chain(x, y, n):
    for _ in range(n):
        ops.silu(x, out=y)
        ops.silu(y, out=x)


def timed(g, reps=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    n = 100
    for numel in (1 << 12, 1 << 16, 1 << 20):
        xs = [torch.randn(numel, device="cuda") for _ in range(4)]
        chain(xs[0], xs[1], 2)
        chain(xs[2], xs[3], 2)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            chain(xs[0], xs[1], n)
            chain(xs[2], xs[3], n)
        g2 = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.graph(g2):
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                chain(xs[2], xs[3], n)
            chain(xs[0], xs[1], n)
            main_s.wait_stream(side)
        print(f"numel {numel:8d}: serial graph {timed(g1):8.1f} us   forked graph {timed(g2):8.1f} us   ({4 * n} kernels)", flush=True)


if __name__ == "__main__":
    main()
