#!/usr/bin/env python
"""Prototype: does running two half-batches of the UNet forward on two streams (inside one HIP graph)
beat one full-batch chain?  (Latency-bound low-resolution levels overlap across the halves.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    B = 64
    unet = bench.build_unet(torch.bfloat16, "cuda")
    x = torch.randn(B, 32, 32, 4, device="cuda").to(torch.bfloat16)
    t = torch.tensor([501.0], device="cuda")

    def full():
        return unet.forward_nhwc(x, t)

    side = torch.cuda.Stream()

    def halves():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y1 = unet.forward_nhwc(x[B // 2:], t)
        y0 = unet.forward_nhwc(x[:B // 2], t)
        cur.wait_stream(side)
        return y0, y1

    for name, fn in (("full", full), ("two halves / two streams", halves)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:28s}: {e0.elapsed_time(e1) / 20:.3f} ms per UNet forward (B = {B})", flush=True)


if __name__ == "__main__":
    main()
