#!/usr/bin/env python
"""3x3 convolutions of the 8x8 / 4x4 levels (and the dense 2x2 form) at batch 64, bf16: automatic plan, timing of
back-to-back launches (conv + split-K reduction), optional AFLDM_CONV_DBG decomposition."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd import _lib, ops  # noqa: E402

SHAPES = [
    ("L8 384->384", 64, 8, 384, 384, 3),
    ("L8 768->384", 64, 8, 768, 384, 3),
    ("L4 768->768", 64, 4, 768, 768, 3),
    ("L4 1536->768", 64, 4, 1536, 768, 3),
    ("L8 1152->384", 64, 8, 1152, 384, 3),
    ("L8 768->768", 64, 8, 768, 768, 3),
    ("L2d 3072->3072", 64, 1, 3072, 3072, 1),
    ("L2d 6144->3072", 64, 1, 6144, 3072, 1),
]


def main():
    dt = torch.bfloat16
    iters = int(os.environ.get("ITERS", 30))
    forced = os.environ.get("FORCE")      # "variant/splitk"
    only = [t for t in os.environ.get("ONLY", "").split(",") if t]
    for name, B, N, Cin, Cout, KS in SHAPES:
        if only and not any(name.startswith(t) for t in only):
            continue
        x = torch.randn(B, N, N, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, KS, KS, Cin, device="cuda") / (KS * Cin ** 0.5)).to(dt)
        bias = torch.randn(Cout, device="cuda")
        y = torch.empty(B, N, N, Cout, device="cuda", dtype=dt)
        ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        flops = 2.0 * B * N * N * Cout * KS * KS * Cin
        if forced:
            v, sk = (int(t) for t in forced.split("/"))
            _lib.check(_lib.lib.afldm_conv2d_tune(v, sk), "tune")
        fn = lambda: ops.conv2d(x, w, bias, out=y, workspace=ws, want_stats=True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e3)
        _lib.lib.afldm_conv2d_tune(-1, -1)
        t = sorted(ts)[2]
        wbytes = Cout * KS * KS * Cin * 2
        print(f"{name:16s} {t:7.1f} us  {flops / t / 1e6:6.0f} TF  weights {wbytes / 1e6:5.1f} MB -> {wbytes / t / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
