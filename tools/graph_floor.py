#!/usr/bin/env python
"""Per-kernel floor inside a replayed HIP graph: N dependent launches of a one-thread kernel (afldm_select_timestep)
captured into one graph; time per replay / N.  What one removed launch buys in the denoise step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd import ops  # noqa: E402


def main():
    tv = torch.arange(64, dtype=torch.float32, device="cuda")
    idx = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    x = torch.randn(1 << 20, device="cuda")
    y = torch.empty_like(x)
    for n in (50, 200, 800):
        for kind in ("one-thread", "silu-4MB"):
            def body():
                for _ in range(n):
                    if kind == "one-thread":
                        ops.select_timestep(tv, idx, out, pre_advance=False)
                    else:
                        ops.silu(x, out=y)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{kind:10s} n={n:4d}: {dt * 1e6:9.1f} us per replay = {dt * 1e6 / n:6.2f} us per kernel", flush=True)


if __name__ == "__main__":
    main()
