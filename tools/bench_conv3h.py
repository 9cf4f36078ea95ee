#!/usr/bin/env python
"""3x3 convolutions of the 32x32 / 16x16 levels at batch 64 (bf16): halo-patch variants (conv3h.hip, ids 41-46)
against the implicit-GEMM picks (29 / 33), interleaved rounds, HIP-event timing of back-to-back launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd import _lib, ops  # noqa: E402

SHAPES = [
    ("L32 192->192", 64, 32, 192, 192, (29, 41, 47)),
    ("L32 384->192", 64, 32, 384, 192, (29, 41, 47)),
    ("L32 576->192", 64, 32, 576, 192, (29, 41, 47)),
    ("L32 384->384", 64, 32, 384, 384, (29, 41, 47)),
    ("L16 384->384", 64, 16, 384, 384, (33, 42, 43, 48, 49)),
    ("L16 768->384", 64, 16, 768, 384, (33, 42, 43, 48, 49)),
    ("L16 192->384", 64, 16, 192, 384, (33, 42, 43, 48, 49)),
]


def main():
    dt = torch.bfloat16
    cold = torch.zeros(300 << 20, dtype=torch.float32, device="cuda") if os.environ.get("COLD") else None
    rounds = int(os.environ.get("ROUNDS", 5))
    iters = int(os.environ.get("ITERS", 20))
    only = [t for t in os.environ.get("SHAPES", "").split(",") if t]
    for name, B, N, Cin, Cout, variants in SHAPES:
        if only and not any(name.startswith(t) for t in only):
            continue
        if os.environ.get("VARIANTS"):
            variants = tuple(int(v) for v in os.environ["VARIANTS"].split(","))
        use_res = os.environ.get("NORES", "0") != "1"
        x = torch.randn(B, N, N, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, 3, 3, Cin, device="cuda") / (3 * Cin ** 0.5)).to(dt)
        bias = torch.randn(Cout, device="cuda")
        res = torch.randn(B, N, N, Cout, device="cuda").to(dt)
        y = torch.empty(B, N, N, Cout, device="cuda", dtype=dt)
        st = {}
        flops = 2.0 * B * N * N * Cout * 9 * Cin
        ref = None
        times = {v: [] for v in variants}
        for r in range(rounds):
            for v in variants:
                _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
                fn = lambda: ops.conv2d(x, w, bias, residual=res if use_res else None, out=y, want_stats=True)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                if r == 0:
                    if ref is None:
                        ref = y.float().clone()
                    st[v] = float((y.float() - ref).abs().max() / ref.abs().max())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    if cold is not None:
                        cold.add_(1.0)              # 1.2 GB read + written: evicts the L2s and the Infinity Cache
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / iters * 1e3)
        _lib.lib.afldm_conv2d_tune(-1, -1)
        line = f"{name:14s}"
        for v in variants:
            t = sorted(times[v])[len(times[v]) // 2]
            line += f" | v{v}: {t:6.1f} us {flops / t / 1e6:6.0f} TF (min {min(times[v]):5.1f}, dev {st[v]:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
