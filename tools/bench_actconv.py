#!/usr/bin/env python
"""Merged launch (afldm_af_act_conv2d, csrc/actconv.hip) against the two launches it replaces, per UNet site shape at
batch 64: bit-equality of the outputs and HIP-graph timings (20 pairs per graph, so that the pair's own boundary is in
both figures).  CASES=32,16 restricts; B=<batch> overrides the batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd import ops  # noqa: E402
from bench_kernels import timeit_graph  # noqa: E402

# (name, N, C1, C2, Cout, temb, residual, S of the statistics)
CASES = [
    ("32 conv1 192->192", 32, 192, 0, 192, True, False, 4),
    ("32 conv2 192->192 +res", 32, 192, 0, 192, False, True, 4),
    ("32 up conv1 384+192->192", 32, 384, 192, 192, True, False, 4),
    ("32 up conv1 192+192->192", 32, 192, 192, 192, True, False, 4),
    ("16 conv1 192->384", 16, 192, 0, 384, True, False, 1),
    ("16 conv2 384->384 +res", 16, 384, 0, 384, False, True, 2),
    ("16 up conv1 768+384->384", 16, 768, 384, 384, True, False, 2),
    ("16 up conv1 384+192->384", 16, 384, 192, 384, True, False, 2),
]


def main():
    B = int(os.environ.get("B", "64"))
    only = [t for t in os.environ.get("CASES", "").split(",") if t]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    for name, N, C1, C2, Cout, use_temb, use_res, S in CASES:
        if only and not any(name.startswith(t) for t in only):
            continue
        Ct = C1 + C2
        x1 = torch.randn(B, N, N, C1, generator=g).to(dev, torch.bfloat16)
        x2 = torch.randn(B, N, N, C2, generator=g).to(dev, torch.bfloat16) if C2 else None
        w = ops.pack_weight((torch.randn(Cout, Ct, 3, 3, generator=g) * (9 * Ct) ** -0.5).to(dev), torch.bfloat16)
        bias = torch.randn(Cout, generator=g).to(dev)
        gamma, beta = (1 + 0.1 * torch.randn(Ct, generator=g)).to(dev), (0.1 * torch.randn(Ct, generator=g)).to(dev)
        temb = torch.randn(B, Cout, generator=g).to(dev, torch.bfloat16) if use_temb else None
        res = torch.randn(B, N, N, Cout, generator=g).to(dev, torch.bfloat16) if use_res else None
        stats = ops.gn_stats(x1, 32, x2=x2)

        def two():
            a = ops.af_act(x1, x2, stats, gamma, beta, 32, 1e-6)
            return a, ops.conv2d(a, w, bias, temb=temb, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True)

        def one():
            return ops.af_act_conv2d(x1, x2, stats, gamma, beta, 32, 1e-6, w, bias, temb=temb,
                                     temb_stride=Cout if use_temb else 0, residual=res, want_stats=True)

        a_ref, y_ref = two()
        y = one()
        if y is None:
            print(f"{name:32s} no merged kernel")
            continue
        torch.cuda.synchronize()
        same = (torch.equal(y, y_ref), torch.equal(y.act_input, a_ref), torch.equal(y.gn_partial, y_ref.gn_partial))
        t2, t1 = timeit_graph(two), timeit_graph(one)
        ta = timeit_graph(lambda: ops.af_act(x1, x2, stats, gamma, beta, 32, 1e-6))
        print(f"{name:32s} B={B} bit-identical (y, act, stats) {same}  two launches {t2:7.1f} us (act alone {ta:5.1f})  merged {t1:7.1f} us  "
              f"gain {t2 - t1:+6.1f} us  err={ops.actconv_error()}", flush=True)
        # the chain conv -> act (the activation BEHIND the convolution, its input hot in the XCD's L2) and act -> conv -> act
        g2, b2 = (1 + 0.1 * torch.randn(Cout, generator=g)).to(dev), (0.1 * torch.randn(Cout, generator=g)).to(dev)
        tc = temb if use_temb else None

        def conv_then_act():
            y_ = ops.conv2d(a_ref, w, bias, temb=tc, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True)
            return y_, ops.af_act(y_, None, ops.gn_stats(y_, 32), g2, b2, 32, 1e-6)

        def conv_act_merged():
            return ops.act_conv_act(a_ref, None, None, w, bias, tc, Cout if use_temb else 0, res, True, post=(g2, b2, 32, 1e-6))

        def three():
            a_ = ops.af_act(x1, x2, stats, gamma, beta, 32, 1e-6)
            y_ = ops.conv2d(a_, w, bias, temb=tc, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True)
            return ops.af_act(y_, None, ops.gn_stats(y_, 32), g2, b2, 32, 1e-6)

        def chain():
            return ops.act_conv_act(x1, x2, (stats, gamma, beta, 32, 1e-6), w, bias, tc, Cout if use_temb else 0, res, True,
                                    post=(g2, b2, 32, 1e-6))

        yr, ar = conv_then_act()
        got = conv_act_merged()
        got3 = chain()
        torch.cuda.synchronize()
        if got is None or got3 is None:
            print(f"{'':32s} no merged kernel for the chains")
            continue
        same2 = (torch.equal(got[0], yr), torch.equal(got[1], ar), torch.equal(got3[0], yr), torch.equal(got3[1], ar))
        tca, tcm, t3, tch = timeit_graph(conv_then_act), timeit_graph(conv_act_merged), timeit_graph(three), timeit_graph(chain)
        print(f"{'':32s} conv -> act: bit-identical {same2[:2]}  two launches {tca:7.1f} us  merged {tcm:7.1f} us  gain {tca - tcm:+6.1f} us"
              f"   |  act -> conv -> act: {same2[2:]}  three launches {t3:7.1f} us  merged {tch:7.1f} us  gain {t3 - tch:+6.1f} us  "
              f"err={ops.actconv_error()}", flush=True)


if __name__ == "__main__":
    main()
