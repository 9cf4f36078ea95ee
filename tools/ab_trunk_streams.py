#!/usr/bin/env python
"""The LOW-RESOLUTION TRUNK of the UNet step (8^2 / 4^2 / 2^2 levels: down_blocks[2:], mid_block, up_blocks[:3] - ~150 launches
of 5 - 20 us, most of them latency-bound) as ONE batch-64 graph against TWO batch-32 graphs replayed concurrently on two
streams (separate graphs: parallel branches of one HIP graph do not overlap on this runtime, profiles/r02).  ms per trunk."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from afldm_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
unet = bench.build_unet(torch.bfloat16, dev)
B = int(os.environ.get("B", "64"))
NS = int(os.environ.get("NSTREAMS", "2"))
LO = int(os.environ.get("LO", "2"))              # first down block of the trunk (2: from the 8^2 level, 3: from 4^2)
nb = len(unet.down_blocks)
x = torch.randn(B, 32, 32, 4, generator=torch.Generator().manual_seed(1)).to(dev, torch.bfloat16)
slices = unet.time_embed(501, B)[0]


def split_slices():
    it = iter(slices)
    per_down = [[next(it) for _ in blk.resnets] for blk in unet.down_blocks]
    mid = [next(it) for _ in unet.mid_block.resnets]
    per_up = [[next(it) for _ in blk.resnets] for blk in unet.up_blocks]
    return per_down, mid, per_up


PD, MID, PU = split_slices()


def head(xx):
    from afldm_amd.models import blocks as Bk
    h = Bk.conv_forward(unet.conv_in, xx, want_stats=True)
    skips = (h,)
    for i in range(LO):
        h, outs = unet.down_blocks[i](h, PD[i])
        skips += outs
    return h, skips


def trunk(h, last_skip):
    """h: input of down_blocks[LO] (= last_skip: the downsampler output that is also a skip connection)."""
    skips = (last_skip,)
    for i in range(LO, nb):
        h, outs = unet.down_blocks[i](h, PD[i])
        skips += outs
    h = unet.mid_block(h, MID)
    for j in range(nb - LO):
        blk = unet.up_blocks[j]
        n = len(blk.resnets)
        res, skips = skips[-n:], skips[:-n]
        h = blk(h, res, PU[j])
    assert len(skips) == 0, len(skips)
    return h


def graph_of(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            out = fn()
    torch.cuda.synchronize()
    return g, out


h0, sk = head(x)
torch.cuda.synchronize()
last = sk[-1]
assert last is h0 or last.data_ptr() == h0.data_ptr()
ops_sync = [ops.new_sync_buffer(dev) for _ in range(NS + 1)]


def run_full():
    with ops.sync_scope(ops_sync[0]):
        return trunk(h0, h0)


g64, out64 = graph_of(run_full)
per = B // NS
parts, outs = [], []
for i in range(NS):
    hi = h0[i * per:(i + 1) * per]
    st = getattr(h0, "gn_partial", None)
    if st is not None:
        hi.gn_partial = st[i * per:(i + 1) * per]

    def run_part(hi=hi, i=i):
        with ops.sync_scope(ops_sync[1 + i]):
            return trunk(hi, hi)
    g, o = graph_of(run_part)
    parts.append(g)
    outs.append(o)
streams = [torch.cuda.Stream() for _ in range(NS)]
K = 30


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    return sorted(ts)[len(ts) // 2]


def replay_parts():
    main = torch.cuda.current_stream()
    for s, g in zip(streams, parts):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            g.replay()
    for s in streams:
        main.wait_stream(s)


def replay_parts_serial():
    for g in parts:
        g.replay()


t64 = timed(g64.replay)
tser = timed(replay_parts_serial)
tpar = timed(replay_parts)
got = torch.cat(outs, 0)
rel = float(((got.float() - out64.float()).pow(2).mean() / out64.float().pow(2).mean()).sqrt())
print(f"trunk from down block {LO} (B={B}): one batch-{B} graph {t64:.4f} ms | {NS} batch-{per} graphs back to back {tser:.4f} ms | "
      f"on {NS} streams with fork / join {tpar:.4f} ms | gain {t64 - tpar:+.4f} ms | rel-RMS of the outputs {rel:.2e}", flush=True)
