import ctypes, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from afldm_amd import _lib, ops
from bench_kernels import timeit_graph
for (B, H, Cin, Cout, va, vb) in ((8, 32, 192, 192, 55, 65), (8, 16, 384, 384, 54, 66), (1, 32, 192, 192, 55, 65), (1, 16, 384, 384, 54, 66), (8, 32, 576, 192, 55, 65), (4, 16, 768, 384, 54, 66)):
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).cuda()
    w = ops.pack_weight((torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda(), torch.bfloat16)
    b = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(B, H, H, Cout, generator=g).to(torch.bfloat16).cuda()
    outs = {}
    for v in (va, vb):
        _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
        try:
            y = ops.conv2d(x, w, b, residual=res, want_stats=True)
            a = ops.conv_args(x, w, b, residual=res, out=torch.empty_like(y))
            need = _lib.lib.afldm_conv2d_workspace(ctypes.byref(a))
            ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device="cuda")
            a.workspace, a.workspace_bytes = ops.ptr(ws), need
            got = _lib.lib.afldm_conv2d_variant(ctypes.byref(a))
            t = timeit_graph(lambda: ops.conv2d_launch(a))
        finally:
            _lib.lib.afldm_conv2d_tune(-1, -1)
        st = y.gn_partial.sum(1)
        outs[v] = (y, st, got & 255, (got >> 8) & 255, t)
    ya, sa, *_ = outs[va]; yb, sb, gb, zb, tb = outs[vb]
    print(f"B={B} {H}^2 {Cin}->{Cout}: variant {outs[va][2]} z={outs[va][3]} {outs[va][4]:6.1f} us | variant {gb} z={zb} {tb:6.1f} us | y identical {bool(torch.equal(ya, yb))} "
          f"max|dy| {float((ya.float()-yb.float()).abs().max()):.3e} stats rel {float((sa-sb).abs().max()/sa.abs().max()):.2e}", flush=True)
