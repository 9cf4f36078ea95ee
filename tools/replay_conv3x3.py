#!/usr/bin/env python
"""Replays exactly the 3x3 convolutions of ONE FFHQ AF-UNet denoise step (batch 64, bf16) and
nothing else, so that a `rocprofv3 --pmc` pass over this process measures the HBM traffic of the
bench's dominant kernel family (conv3x3: 64 launches per step).

  python tools/replay_conv3x3.py record   # one eager step, logs the conv3x3 call signatures
  python tools/replay_conv3x3.py          # replays them (the process rocprofv3 wraps)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SIG = os.path.join(ROOT, "gpurun_out", "conv3x3_signatures.json")


def record(batch=64):
    import bench
    from afldm_amd import ops
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    unet = bench.build_unet(torch.bfloat16, "cuda")
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), batch, 50, use_graph=False)
    eng.reset(torch.randn(batch, 4, 32, 32, generator=torch.Generator().manual_seed(1)))
    eng.step(1)
    sigs = []
    orig = ops.conv2d

    def spy(x, w, bias=None, x2=None, temb=None, temb_stride=0, residual=None, **kw):
        if w.shape[1] == 3:
            sigs.append(dict(B=x.shape[0], H=x.shape[1], W=x.shape[2], C1=x.shape[3], C2=0 if x2 is None else x2.shape[3],
                             Cout=w.shape[0], bias=bias is not None, temb=temb is not None, temb_stride=temb_stride,
                             residual=residual is not None))
        return orig(x, w, bias, x2=x2, temb=temb, temb_stride=temb_stride, residual=residual, **kw)

    orig_slabs = ops.conv2d_slabs

    def spy_slabs(x, w, x2=None):
        # conv1 of a 2x2 / 4x4 resnet whose K slices go straight to the activation kernel (no reduction launch)
        got = orig_slabs(x, w, x2)
        if got is not None and w.shape[1] == 3:
            sigs.append(dict(B=x.shape[0], H=x.shape[1], W=x.shape[2], C1=x.shape[3], C2=0, Cout=w.shape[0], bias=False,
                             temb=False, temb_stride=0, residual=False, slabs=True))
        return got

    ops.conv2d = spy
    ops.conv2d_slabs = spy_slabs
    eng.step(1)
    torch.cuda.synchronize()
    ops.conv2d, ops.conv2d_slabs = orig, orig_slabs
    os.makedirs(os.path.dirname(SIG), exist_ok=True)
    json.dump(sigs, open(SIG, "w"))
    alg = 0
    for s in sigs:
        m = s["B"] * s["H"] * s["W"]
        alg += 2 * (m * (s["C1"] + s["C2"]) + s["Cout"] * 9 * (s["C1"] + s["C2"]) + m * s["Cout"] * (2 if s["residual"] else 1))
    print(json.dumps(dict(launches=len(sigs), algorithmic_bytes_per_step=alg)))


def replay(reps=3):
    from afldm_amd import ops
    sigs = json.load(open(SIG))
    bufs = []
    for s in sigs:
        dt, dev = torch.bfloat16, "cuda"
        x1 = torch.randn(s["B"], s["H"], s["W"], s["C1"], device=dev).to(dt)
        x2 = torch.randn(s["B"], s["H"], s["W"], s["C2"], device=dev).to(dt) if s["C2"] else None
        ct = s["C1"] + s["C2"]
        w = (torch.randn(s["Cout"], 3, 3, ct, device=dev) / (3 * ct ** 0.5)).to(dt)
        bias = torch.randn(s["Cout"], device=dev) if s["bias"] else None
        temb = torch.randn(s["B"], s["Cout"], device=dev).to(dt) if s["temb"] else None
        res = torch.randn(s["B"], s["H"], s["W"], s["Cout"], device=dev).to(dt) if s["residual"] else None
        y = torch.empty(s["B"], s["H"], s["W"], s["Cout"], device=dev, dtype=dt)
        bufs.append((x1, w, bias, x2, temb, s["Cout"] if s["temb"] else 0, res, y, bool(s.get("slabs"))))
    ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(reps):
        for x1, w, bias, x2, temb, ts, res, y, slabs in bufs:
            if slabs:
                ops.conv2d_slabs(x1, w)
            else:
                ops.conv2d(x1, w, bias, x2=x2, temb=temb, temb_stride=ts, residual=res, out=y, workspace=ws)
    torch.cuda.synchronize()
    print(json.dumps(dict(launches=len(bufs), reps=reps)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "record":
        record()
    else:
        replay()
