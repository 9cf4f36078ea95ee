#!/usr/bin/env python
"""Prices of the two primitives the sample-per-XCD persistent trunk stands on, measured on the
box before building it: (1) per-XCD weight streaming when all 8 XCDs read the same matrix, (2) an XCD-local barrier +
hand-over.  Output: gpurun_out/trunk_bounds.txt (copied to profiles/r03/)."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libtrunk_bounds.so")
if not os.path.exists(SO) or "--build" in sys.argv:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO,
                           os.path.join(HERE, "trunk_bounds.hip")])
    if "--build" in sys.argv:
        sys.exit(0)
lib = ctypes.CDLL(SO)
vp = ctypes.c_void_p
lib.tb_stream.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp, vp]
lib.tb_barrier.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


sink = torch.zeros(1024, dtype=torch.int32, device=dev)
say("== 1. weight streaming: one pass over a matrix of S bytes (bf16 weights of one trunk layer: 10.6 / 21 MB at 4x4)")
say("   mode 0 = read ONCE by the chip (256 WGs x S/256), mode 1 = read by EVERY XCD (32 WGs x S/32 per XCD, 8 S in total)")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
GRAN = 16 * 8 * 512 * 256
for mb in (2.7, 10.6, 21.2, 37.7, 460.0):
    n = max(int(mb * 1e6) // GRAN, 1) * GRAN
    w = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    for mode in (0, 1):
        for cold in (True, False):
            if cold:
                ts = []
                for _ in range(4):
                    flush.fill_(1)          # evict the matrix from the L2s / MALL (512 MB > 256 MB)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    lib.tb_stream(w.data_ptr(), n, mode, 1, sink.data_ptr(), st)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ms = sorted(ts)[len(ts) // 2]
            else:
                ms = timed(lambda: lib.tb_stream(w.data_ptr(), n, mode, 1, sink.data_ptr(), st))
            tot = n * (8 if mode else 1)
            say(f"   S = {n / 1e6:7.1f} MB  mode {mode}  {'cold (HBM)      ' if cold else 'warm (L2 / MALL)'}: {ms * 1e3:8.1f} us   "
                f"matrix rate {n / ms / 1e6:8.1f} GB/s   delivered to the L2s {tot / ms / 1e6:8.1f} GB/s"
                + (f"   = {n / ms / 1e6:7.1f} GB/s per XCD" if mode else ""))
    del w

say("== 2. XCD-local barrier + hand-over (256 WGs x 256 threads, 1 per CU; a round = write record, barrier, read the")
say("      neighbour's record and check every word, barrier; uneven load: 1 in 5 workgroups sleeps before publishing)")
nb = lib.tb_state_bytes()
forms = {0: "no fences + sc1 payload loads            ", 1: "no fences + PLAIN payload loads          ",
         2: "agent release/acquire fences, plain loads", 3: "agent fences + sc1 payload loads         "}
for words in (1024, 16384):
    buf = torch.zeros(8 * 64 * words, dtype=torch.int32, device=dev)
    for form in (0, 1, 2, 3):
        res = {}
        err = stale = 0
        tick = None
        for rounds in (200, 2200):
            state = torch.zeros(nb // 4, dtype=torch.int32, device=dev)

            def run():
                state.zero_()
                lib.tb_barrier(state.data_ptr(), buf.data_ptr(), rounds, words, form, 256, st)
            res[rounds] = timed(run, 3)
            host = state.cpu()
            tick = host[:8].tolist()
            err += int(host[8 + 256 + 32])
            stale += int(host[8 + 256 + 33])
        per = (res[2200] - res[200]) / 2000 * 1e3
        say(f"   {words * 4 // 1024:3d} KB records, form {form} ({forms[form]}): {per:6.2f} us per round"
            f"   WGs per XCD {tick}  timeouts {err}  stale words {stale}")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/trunk_bounds.txt", "w").write("\n".join(lines) + "\n")
