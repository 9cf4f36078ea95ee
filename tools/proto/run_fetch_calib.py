#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE calibration (tools/proto/fetch_calib.hip).  WRITE side (round 5): FC_WRITE=1 under
   rocprofv3 --kernel-trace --pmc WRITE_SIZE (and, in a second pass, --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum).
Run under the profiler:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fcal -o fc -- python tools/proto/run_fetch_calib.py
then `python tools/proto/run_fetch_calib.py report gpurun_out/fcal` prints counter bytes / true bytes per access width."""
import ctypes, csv, glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
P, ROW = 1 << 20, 384            # 1 Mi pixels x 192 bf16 channels = 384 MiB (> the 256 MiB Infinity Cache)
if len(sys.argv) > 1 and sys.argv[1] == "report":
    true = P * ROW
    lines = []
    unit = {"FETCH_SIZE": 1024.0, "TCC_MISS_sum": 128.0, "TCC_EA0_RDREQ_sum": 64.0, "TCC_EA0_RDREQ_32B_sum": 32.0, "TCC_BUBBLE_sum": 128.0,
            "TCC_REQ_sum": 128.0, "TCC_HIT_sum": 128.0, "WRITE_SIZE": 1024.0, "TCC_EA0_WRREQ_sum": 64.0, "TCC_EA0_WRREQ_64B_sum": 64.0}
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if ("k_pieces" in n or "k_stream" in n or "k_wpieces" in n or "k_wstream" in n) and r["Counter_Name"] in unit:
                v = float(r["Counter_Value"])
                lines.append(f"{n[:44]:44s} {r['Counter_Name']:22s} x {unit[r['Counter_Name']]:6.0f} B / true bytes = {v * unit[r['Counter_Name']] / true:.3f}")
    print("\n".join(lines))
    open(os.environ.get("FC_OUT", "gpurun_out/fetch_calib.txt"), "w").write("\n".join(lines) + "\n")
    sys.exit(0)
import torch
SO = os.path.join(HERE, "libfetch_calib.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, os.path.join(HERE, "fetch_calib.hip")])
lib = ctypes.CDLL(SO)
lib.fc_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
x = torch.randint(0, 2 ** 31 - 1, (P * ROW // 4,), dtype=torch.int32, device="cuda")
junk = torch.randint(0, 2 ** 31 - 1, (P * ROW // 4,), dtype=torch.int32, device="cuda")      # evicts the caches between modes
out = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
if os.environ.get("FC_WRITE"):
    lib.fc_wrun.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    for sc1 in (0, 1):
        for mode in (0, 128, 64, 32, 16):
            junk.add_(1)
            torch.cuda.synchronize()
            assert lib.fc_wrun(x.data_ptr(), mode, sc1, P, ROW, st) == 0
            torch.cuda.synchronize()
    print("done (write)")
    sys.exit(0)
for mode in (0, 128, 64, 32, 16, 0, 32):
    junk.add_(1)
    torch.cuda.synchronize()
    assert lib.fc_run(x.data_ptr(), out.data_ptr(), mode, P, ROW, st) == 0
    torch.cuda.synchronize()
print("done")
