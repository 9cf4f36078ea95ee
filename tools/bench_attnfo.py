#!/usr/bin/env python
"""The attention block of the 32^2 level at batch 64: fused front end + to_out launch against the one launch that carries
to_out (afldm_attn_block_fused_out); phase stamps of the latter (7 attention done, 10 hand-over done, 11 end)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from afldm_amd import ops, _lib
from bench_kernels import timeit_graph
B = int(os.environ.get("B", "64"))
T, C, heads = 1024, 192, 8
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, C, generator=g).to(torch.bfloat16).cuda()
st = ops.gn_stats(x.view(B, 32, 32, C), 32)
gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
w = ops.pack_weight((torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), torch.bfloat16)
b = torch.zeros(3 * C).cuda()
wo = ops.pack_weight((torch.randn(C, C, generator=g) / C ** 0.5).cuda(), torch.bfloat16)
bo = torch.zeros(C).cuda()
scale = (C // heads) ** -0.5
o = torch.empty_like(x)
t_f = timeit_graph(lambda: ops.attn_block_fused(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, out=o))
t_o = timeit_graph(lambda: ops.conv2d(o.view(B, 32, 32, C), wo, bo, residual=x.view(B, 32, 32, C), want_stats=True))
t_both = timeit_graph(lambda: (ops.attn_block_fused(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, out=o),
                               ops.conv2d(o.view(B, 32, 32, C), wo, bo, residual=x.view(B, 32, 32, C), want_stats=True)))
t_fo = timeit_graph(lambda: ops.attn_block_fused_out(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, wo, bo))
print(f"B={B}: front end {t_f:.1f} us + to_out {t_o:.1f} us (both in one graph: {t_both:.1f}) | one launch {t_fo:.1f} us", flush=True)
nwg, nw = B * heads, 8
tr = torch.zeros(nwg * nw * 12, dtype=torch.int64, device="cuda")
_lib.lib.afldm_attn_block_fused_trace(tr.data_ptr())
ops.attn_block_fused_out(x, st, gamma, beta, 32, 1e-5, w, b, heads, scale, wo, bo)
torch.cuda.synchronize()
_lib.lib.afldm_attn_block_fused_trace(None)
tr = tr.view(nwg, nw, 12).cpu().double()
rel = tr - tr[:, :, :1].min(1, keepdim=True).values
life = rel[:, :, 11].max(1).values
tick = t_fo / (-(-nwg // 256)) / float(life.mean())
for name, a, c in (("start -> attention done", 0, 7), ("attention done -> hand-over done (Wo fetch, stores acknowledged, siblings)", 7, 10),
                   ("hand-over done -> end (GEMM, staging, rows out, statistics)", 10, 11)):
    d = (tr[:, :, c] - tr[:, :, a])
    print(f"   {name}: mean {float(d.mean()) * tick:.2f} us, min {float(d.min()) * tick:.2f}, max {float(d.max()) * tick:.2f}")
att_end = tr[:, :, 7]
print(f"   spread of a workgroup's waves at 'attention done': {float((att_end.max(1).values - att_end.min(1).values).mean()) * tick:.2f} us;"
      f" of a sample's workgroups (first to last wave): {float((att_end.view(B, heads * nw).max(1).values - att_end.view(B, heads * nw).min(1).values).mean()) * tick:.2f} us"
      "   (note: work ids, not block ids, index the stamps? no - block ids: a sample's workgroups are 8 apart)")
