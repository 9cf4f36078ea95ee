import os, sys, torch
sys.path.insert(0, os.getcwd())
from afldm_amd import ops
dt = torch.bfloat16
for (B, K, N) in [(64, 3072, 3072), (64, 6144, 3072), (8, 3072, 3072), (1, 3072, 3072)]:
    x = torch.randn(B, 1, 1, K, device="cuda").to(dt)
    w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn(N, device="cuda")
    y = torch.empty(B, 1, 1, N, device="cuda", dtype=dt)
    ws = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    fn = lambda: ops.conv2d(x, w, bias, out=y, workspace=ws, want_stats=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} K={K} N={N}: {e0.elapsed_time(e1)/50*1e3:7.1f} us per call", flush=True)
