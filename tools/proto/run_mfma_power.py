#!/usr/bin/env python
"""tools/proto/mfma_power.hip: MFMA rate and shader clock with constant / random operands / random + LDS reads."""
import ctypes, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmfma_power.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, os.path.join(HERE, "mfma_power.hip")])
lib = ctypes.CDLL(SO)
lib.mp_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cus = torch.cuda.get_device_properties(0).multi_processor_count
wgs, iters = cus * 4, 30000
out = torch.zeros(wgs * 256, dtype=torch.float32, device="cuda")
ticks = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "constant low-entropy operands", 1: "random bf16 operands, 4 rotating sets", 2: "operands replaced by one ds_read_b128 per 4 MFMAs",
         3: "random operands + 1 discarded ds_read_b128 per 4 MFMAs", 4: "random operands + 2 reads per 4 MFMAs",
         5: "random operands + 3 reads per 4 MFMAs", 6: "random operands + 4 reads per 4 MFMAs"}
for rep in range(2):
    for mode in (0, 1, 3, 4, 5, 6):
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.mp_run(out.data_ptr(), ticks.data_ptr(), mode, wgs, iters, st) == 0
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flops = wgs * 4.0 * iters * 16 * 32 * 32 * 16 * 2
        t = int(ticks.cpu()[0])
        print(f"mode {mode} ({names[mode]}): {ms:7.2f} ms  {flops / ms / 1e9:7.1f} TFLOP/s   s_memtime {t / (ms * 1e3):6.0f} ticks/us", flush=True)
