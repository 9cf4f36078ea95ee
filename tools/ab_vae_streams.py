#!/usr/bin/env python
"""AF-VAE C4 (encode + decode of 128 images 256x256): one pass over the batch against N concurrent sub-batches on N streams
(eager launches, one host thread): do the sub-batches' kernels fill each other's launch ramps / tails?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dtype = torch.bfloat16
vae = bench.build_vae(dtype, dev)
B = 128
imgs = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)


def one():
    z = vae.encode(imgs).latent_dist.mode()
    return vae.decode(z, return_dict=False)[0]


def split(n):
    streams = [torch.cuda.Stream() for _ in range(n)]
    per = B // n
    def run():
        main = torch.cuda.current_stream()
        outs = [None] * n
        for i, s in enumerate(streams):
            s.wait_stream(main)
        # encode of every sub-batch first, then the decodes: the host thread alternates between the streams
        zs = [None] * n
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                zs[i] = vae.encode(imgs[i * per:(i + 1) * per]).latent_dist.mode()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[i] = vae.decode(zs[i], return_dict=False)[0]
        for s in streams:
            main.wait_stream(s)
        return torch.cat(outs, 0)
    return run


def timed(fn, reps=3):
    out = fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], out


t1, ref = timed(one)
print(f"one pass, batch 128:              {t1 * 1e3:7.1f} ms  {B / t1:6.1f} img/s", flush=True)
for n in (2, 4):
    t, out = timed(split(n))
    same = bool(torch.equal(out, ref))
    rel = float(((out.float() - ref.float()).pow(2).mean() / ref.float().pow(2).mean()).sqrt())
    print(f"{n} sub-batches of {B // n} on {n} streams: {t * 1e3:7.1f} ms  {B / t:6.1f} img/s   identical: {same} (rel-RMS {rel:.2e})", flush=True)
