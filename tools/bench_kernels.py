#!/usr/bin/env python
"""Kernel micro-benchmarks on MI355X (HIP-event timing, interleaved rounds).

  python tools/bench_kernels.py conv     # sweep igemm tile/pipeline variants per UNet layer shape
  python tools/bench_kernels.py afact    # fused GN + alias-free activation
  python tools/bench_kernels.py attn
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def timeit_graph(fn, reps=20, iters=10):
    """us per call with `reps` calls captured into one HIP graph: the Python call overhead (~20 us) hides anything shorter
    when the calls are issued eagerly."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


# (name, B, H, W, C1, C2, Cout, KS)  at batch 64
CONV_SHAPES = [
    ("L32 192->192 3x3", 64, 32, 32, 192, 0, 192, 3),
    ("L32 384+192->192 3x3", 64, 32, 32, 384, 192, 192, 3),
    ("L32 384->384 3x3 (up conv)", 64, 32, 32, 384, 0, 384, 3),
    ("L32 192->192 1x1 (q/k/v/o)", 64, 32, 32, 192, 0, 192, 1),
    ("L16 384->384 3x3", 64, 16, 16, 384, 0, 384, 3),
    ("L16 384+384->384 3x3", 64, 16, 16, 384, 384, 384, 3),
    ("L16 384->384 1x1", 64, 16, 16, 384, 0, 384, 1),
    ("L8 384->384 3x3", 64, 8, 8, 384, 0, 384, 3),
    ("L8 768+384->384 3x3", 64, 8, 8, 768, 384, 384, 3),
    ("L8 768->768 3x3 (up conv)", 64, 8, 8, 768, 0, 768, 3),
    ("L4 768->768 3x3", 64, 4, 4, 768, 0, 768, 3),
    ("L4 768+768->768 3x3", 64, 4, 4, 768, 768, 768, 3),
    ("L2 768->768 3x3", 64, 2, 2, 768, 0, 768, 3),
    ("L2 768+768->768 3x3", 64, 2, 2, 768, 768, 768, 3),
    ("L4 768->768 1x1", 64, 4, 4, 768, 0, 768, 1),
    ("conv_out 192->4", 64, 32, 32, 192, 0, 4, 3),
]


def bench_conv(dtype=torch.bfloat16):
    # CONV_VARIANTS=4,12,21  CONV_SHAPES=L32,L16  restrict the sweep
    variants = [int(v) for v in os.environ.get("CONV_VARIANTS", ",".join(map(str, range(40)))).split(",")]
    only = [t for t in os.environ.get("CONV_SHAPES", "").split(",") if t]
    out = {}
    for name, B, H, W, C1, C2, Cout, KS in CONV_SHAPES:
        if only and not any(name.startswith(t) for t in only):
            continue
        x1 = torch.randn(B, H, W, C1, device="cuda").to(dtype)
        x2 = torch.randn(B, H, W, C2, device="cuda").to(dtype) if C2 else None
        w = (torch.randn(Cout, KS, KS, C1 + C2, device="cuda") / (KS * (C1 + C2) ** 0.5)).to(dtype)
        bias = torch.randn(Cout, device="cuda")
        y = torch.empty(B, H, W, Cout, device="cuda", dtype=dtype)
        ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        flops = 2.0 * B * H * W * Cout * KS * KS * (C1 + C2)
        M = B * H * W
        res = {}
        ref = None
        for v in variants:
            for sk in ((-1,) if M >= 16384 else (1, 2, 4, 8, 16)):
                _lib.check(_lib.lib.afldm_conv2d_tune(v, sk), "tune")
                try:
                    fn = lambda: ops.conv2d(x1, w, bias, x2=x2, out=y, workspace=ws)
                    fn()
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = y.float().clone()
                    err = float((y.float() - ref).abs().max() / ref.abs().max())
                    t = timeit(fn)
                    res[f"v{v}/sk{sk}"] = (round(t, 1), round(flops / t / 1e6, 0), round(err, 5))
                except Exception as e:  # noqa
                    res[f"v{v}/sk{sk}"] = ("ERR", str(e)[:60])
        _lib.lib.afldm_conv2d_tune(-1, -1)
        auto = timeit(lambda: ops.conv2d(x1, w, bias, x2=x2, out=y, workspace=ws))
        best = sorted((v for v in res.items() if v[1][0] != "ERR"), key=lambda kv: kv[1][0])
        best = best if os.environ.get("CONV_ALL") else best[:4]
        print(f"{name:30s} auto {auto:8.1f}us {flops/auto/1e6:7.0f} TF | best: " +
              "  ".join(f"{k}:{v[0]}us/{v[1]:.0f}TF" for k, v in best), flush=True)
        bad = [k for k, v in res.items() if v[0] == "ERR" or v[2] > 2e-2]
        if bad:
            print("    BAD:", bad[:6], [res[k] for k in bad[:3]])
        out[name] = res
    return out


def bench_afact(dtype=torch.bfloat16):
    for N, C in ((32, 192), (32, 576), (16, 384), (16, 768), (8, 768), (4, 1536), (2, 1536)):
        x = torch.randn(64, N, N, C, device="cuda").to(dtype)
        gamma = torch.ones(C, device="cuda")
        beta = torch.zeros(C, device="cuda")
        st = ops.gn_stats(x, 32)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.af_act(x, None, st, gamma, beta, 32, 1e-5, out=y))
        t2 = timeit(lambda: ops.gn_stats(x, 32, out=st.st1))
        nbytes = 2 * x.numel() * x.element_size()
        print(f"af_act N={N:2d} C={C:4d}: {t:8.1f} us  {24.0*N**3*64*C/t/1e6:7.1f} TF(dense-eq)  {nbytes/t/1e3:7.1f} GB/s | "
              f"gn_stats {t2:6.1f} us {nbytes/2/t2/1e3:7.1f} GB/s", flush=True)


def bench_attn(dtype=torch.bfloat16):
    for T, heads in ((1024, 8), (256, 16), (64, 16), (16, 32), (4, 32)):
        C = heads * 24
        q = torch.randn(64, T, C, device="cuda").to(dtype)
        k = torch.randn(64, T, C, device="cuda").to(dtype)
        vt = torch.randn(64, C, T, device="cuda").to(dtype)
        o = torch.empty_like(q)
        t = timeit(lambda: ops.attention(q, k, vt, heads, out=o))
        print(f"attn T={T:4d} heads={heads:2d}: {t:8.1f} us  {4.0*64*heads*T*T*24/t/1e6:7.1f} TF", flush=True)


def bench_lin(dtype=torch.bfloat16):
    """The short-K 1x1 GEMMs of an attention site at the 32x32 level: fused q|k|v and to_out (+residual, +stats)."""
    B, T, C = 64, 1024, 192
    x = torch.randn(B, T, C, device="cuda").to(dtype)
    w3 = (torch.randn(3 * C, 1, 1, C, device="cuda") / C ** 0.5).to(dtype)
    b3 = torch.randn(3 * C, device="cuda")
    w1 = (torch.randn(C, 1, 1, C, device="cuda") / C ** 0.5).to(dtype)
    b1 = torch.randn(C, device="cuda")
    x4 = x.view(B, 32, 32, C)
    res = torch.randn(B, 32, 32, C, device="cuda").to(dtype)
    t = timeit(lambda: ops.linear_split(x, w3, b3, 2 * C))
    print(f"qkv  [65536x192]x[192x576]: {t:7.1f} us  {(B*T*C*2 + B*T*3*C*2)/t/1e3:7.1f} GB/s", flush=True)
    t = timeit(lambda: ops.conv2d(x4, w3, b3))
    print(f"1x1  [65536x192]x[192x576] all NHWC: {t:7.1f} us  {(B*T*C*2 + B*T*3*C*2)/t/1e3:7.1f} GB/s", flush=True)
    t = timeit(lambda: ops.conv2d(x4, w3, b3, out_mode=1))
    print(f"1x1  [65536x192]x[192x576] all channel-major: {t:7.1f} us", flush=True)
    t = timeit(lambda: ops.conv2d(x4, w1, b1))
    print(f"1x1  [65536x192]x[192x192]: {t:7.1f} us  {(B*T*C*2*2)/t/1e3:7.1f} GB/s", flush=True)
    t = timeit(lambda: ops.conv2d(x4, w1, b1, residual=res))
    print(f"1x1 + residual            : {t:7.1f} us  {(B*T*C*2*3)/t/1e3:7.1f} GB/s", flush=True)
    t = timeit(lambda: ops.conv2d(x4, w1, b1, residual=res, want_stats=True))
    print(f"1x1 + residual + stats    : {t:7.1f} us  {(B*T*C*2*3)/t/1e3:7.1f} GB/s", flush=True)


def bench_fir():
    """upfirdn2d (csrc/fir.hip) on the image shifter's sizes: HBM-bound, algorithmic bytes = in + out."""
    from afldm_amd.af_libs import equivariance as eq
    from afldm_amd.af_libs.torch_utils.ops import upfirdn2d as up
    from afldm_amd.shift_utils.shifters import ImageShifter
    for dtype in (torch.float32, torch.bfloat16):
        es = 4 if dtype == torch.float32 else 2
        img = torch.randn(64, 3, 256, 256, device="cuda").to(dtype)
        n = img.numel()
        f6 = torch.rand(1, 6, device="cuda")
        t = timeit(lambda: ops.upfirdn2d(img, f6, padx0=3, padx1=2))
        print(f"{dtype}: 6-tap row pass  64x3x256^2: {t:7.1f} us  {2*n*es/t/1e3:7.1f} GB/s", flush=True)
        f6c = f6.reshape(6, 1).contiguous()
        t = timeit(lambda: ops.upfirdn2d(img, f6c, pady0=3, pady1=2))
        print(f"{dtype}: 6-tap col pass  64x3x256^2: {t:7.1f} us  {2*n*es/t/1e3:7.1f} GB/s", flush=True)
        t = timeit(lambda: eq.apply_fractional_translation(img, 2.375 / 256, 0.5 / 256))
        print(f"{dtype}: lanczos shift (2 passes + mask): {t:7.1f} us  {4*n*es/t/1e3:7.1f} GB/s", flush=True)
        sh = ImageShifter()
        t = timeit(lambda: sh.shift(img, 0.0, 2.375))
        print(f"{dtype}: bilinear shift (2 passes + mask): {t:7.1f} us  {4*n*es/t/1e3:7.1f} GB/s", flush=True)
        fb = up.setup_filter([1, 3, 3, 1], device="cuda")
        x128 = img[:, :, :128, :128].contiguous()
        t = timeit(lambda: up.upsample2d(x128, fb, up=2))
        print(f"{dtype}: blur upsample2d 128^2 -> 256^2 (4x4): {t:7.1f} us  {(x128.numel()+n)*es/t/1e3:7.1f} GB/s", flush=True)
        t = timeit(lambda: up.downsample2d(img, fb, down=2))
        print(f"{dtype}: blur downsample2d 256^2 -> 128^2 (4x4): {t:7.1f} us  {(x128.numel()+n)*es/t/1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "conv"
    {"conv": bench_conv, "afact": bench_afact, "attn": bench_attn, "lin": bench_lin, "fir": bench_fir}[what]()
