import sys, torch
sys.path.insert(0, '.')
from afldm_amd import ops
from tools.bench_kernels import timeit
for (B, N, C) in ((8, 256, 128), (8, 128, 256), (8, 64, 512), (64, 32, 192), (64, 16, 384)):
    x = torch.randn(B, N, N, C, device='cuda').to(torch.bfloat16)
    st = ops.gn_stats(x)
    g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
    y = torch.empty_like(x)
    t = timeit(lambda: ops.gn_apply(x, st, g, b, 32, 1e-6, act=1, out=y))
    t2 = timeit(lambda: ops.gn_stats(x, out=st.st1))
    print(f"gn_apply B={B} N={N} C={C}: {t:8.1f} us {2*x.numel()*2/t/1e3:7.1f} GB/s | gn_stats {t2:8.1f} us {x.numel()*2/t2/1e3:7.1f} GB/s")
for n in (25 << 20, 100 << 20, 268 << 20):
    a = torch.empty(n // 2, dtype=torch.bfloat16, device='cuda').normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(f"torch copy {n >> 20} MB: {t:8.1f} us  {2*n/t/1e3:7.1f} GB/s")
