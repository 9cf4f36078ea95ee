#!/usr/bin/env python
"""Attention block front end at batch 64: the fused launch (csrc/attnf.hip) against the three launches it replaces, per level;
AFLDM_ATTNF_DBG decomposition (1 no attention phase, 2 no exponentials, 4 no projection MFMAs, 5 = 1 + 4) in subprocesses."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from afldm_amd import ops
    from bench_kernels import timeit, timeit_graph
    if os.environ.get("GRAPH"):            # launches captured into a HIP graph (needed below ~25 us per launch)
        timeit = lambda fn, iters=50: timeit_graph(fn)
    B = int(os.environ.get("B", "64"))
    shapes = ((1024, 192, 8), (256, 384, 16))
    if os.environ.get("SHAPES"):           # e.g. SHAPES=64x384x16
        shapes = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(","))
    for T, C, heads in shapes:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, T, C, generator=g).to(torch.bfloat16).cuda()
        side = int(T ** 0.5)
        st = ops.gn_stats(x.view(B, side, side, C), 32)
        gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
        w = ops.pack_weight((torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), torch.bfloat16)
        b = torch.zeros(3 * C).cuda()
        scale = (C // heads) ** -0.5
        out = torch.empty_like(x)
        t_f = timeit(lambda: ops.attn_block_fused(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, out=out), iters=50)
        if os.environ.get("TRACE"):
            from afldm_amd import _lib
            nwg, nw = B * heads, 8
            tr = torch.zeros(nwg * nw * 12, dtype=torch.int64, device="cuda")
            _lib.lib.afldm_attn_block_fused_trace(tr.data_ptr())
            ops.attn_block_fused(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, out=out)
            torch.cuda.synchronize()
            _lib.lib.afldm_attn_block_fused_trace(None)
            tr = tr.view(nwg, nw, 12).cpu().double()
            # (the stamp counters of the 8 XCDs are not aligned: only differences inside a workgroup mean anything; the tick
            #  is calibrated on launch time / rounds of workgroups, one workgroup per CU)
            names = ["start", "prologue", "barrier", "projection", "barrier", "pass 1", "pass 2", "epilogue"]
            rel = tr - tr[:, :, :1]
            total = float(rel[:, :, 7].mean())
            rounds = -(-nwg // 256)
            tick_us = t_f / rounds / total
            seg, prev = [], 0
            for i in range(1, 8):
                if float(tr[:, :, i].max()) == 0:
                    continue
                d = float((rel[:, :, i] - rel[:, :, prev]).mean())
                seg.append(f"{names[i]} {d * tick_us:.1f} us ({100 * d / total:.0f} %)")
                prev = i
            print(f"   per workgroup ({rounds} rounds, {total:.0f} ticks = {total * tick_us:.1f} us): " + " | ".join(seg), flush=True)
            if float(tr[:, :, 8].max()) > 0:
                w1 = float((rel[:, :, 8] - rel[:, :, 1]).mean()) * tick_us
                fo = float((rel[:, :, 9] - rel[:, :, 8]).mean()) * tick_us
                w2 = float((rel[:, :, 2] - rel[:, :, 9]).mean()) * tick_us
                spread = float((rel[:, :, 1].max(1).values - rel[:, :, 1].min(1).values).mean()) * tick_us
                print(f"   fold phase: wait at barrier 1 {w1:.2f} us (arrival spread of a workgroup's waves {spread:.2f}) | x issue + fold {fo:.2f} | barrier 2 {w2:.2f}")
        if os.environ.get("AFLDM_ATTNF_DBG") or os.environ.get("QUICK"):
            print(f"  dbg={os.environ.get('AFLDM_ATTNF_DBG')} stagger={os.environ.get('AFLDM_ATTNF_STAGGER')} T={T} C={C}: fused {t_f:7.1f} us", flush=True)
            continue
        hn = torch.empty_like(x)
        t_gn = timeit(lambda: ops.gn_apply(x.view(B, side, side, C), st, gamma, beta, 32, 1e-5, act=0, out=hn.view(B, side, side, C)), iters=50)
        t_lin = timeit(lambda: ops.linear_split(hn, w, b, 2 * C), iters=50)
        qk, vt = ops.linear_split(hn, w, b, 2 * C)
        t_at = timeit(lambda: ops.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, scale=scale, out=out), iters=50)
        os.environ["AFLDM_ATTNF_SLOW"] = "1"
        t_s = timeit(lambda: ops.attn_block_fused(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, out=out), iters=50)
        del os.environ["AFLDM_ATTNF_SLOW"]
        print(f"T={T} C={C} heads={heads} B={B}: fused {t_f:7.1f} us (row-maxima loop {t_s:7.1f}) | gn_apply {t_gn:6.1f} + qkv {t_lin:6.1f} + attention "
              f"{t_at:6.1f} = {t_gn + t_lin + t_at:7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "decomp":
        for dbg in ("1", "2", "4"):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, AFLDM_ATTNF_DBG=dbg))
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        main()
