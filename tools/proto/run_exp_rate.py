#!/usr/bin/env python
"""Throughput of v_exp_f32 against plain / packed VALU on MI355X (tools/proto/exp_rate.hip)."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libexp_rate.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, os.path.join(HERE, "exp_rate.hip")])
lib = ctypes.CDLL(SO)
lib.er_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(1024 * 256, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 4000
names = {0: "8 x (v_exp_f32 + v_add)", 1: "16 x v_fma_f32", 2: "8 x (v_exp_f32 + v_add + 2 v_fma)", 3: "8 values through the packed cubic 2^x"}
lines = []
for mode in (0, 1, 2, 3):
    lib.er_run(out.data_ptr(), mode, 10, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.er_run(out.data_ptr(), mode, iters, st); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[1]
    # 1024 WGs x 4 waves on 1024 SIMDs = 4 waves per SIMD; per iteration per wave: the listed instructions
    cyc = ms * 1e-3 * 2.4e9 / iters / 4          # SIMD cycles per wave-iteration at 2.4 GHz
    s = f"mode {mode} ({names[mode]}): {ms:8.3f} ms  = {cyc:6.1f} SIMD cycles per wave-iteration of 8 values"
    print(s); lines.append(s)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/exp_rate.txt", "w").write("\n".join(lines) + "\n")
