"""Per-kernel parity: every C-ABI entry point vs the CPU oracle on seeded inputs (MI355X only).

fp32 mode is exact-arithmetic class (f32 MFMA == fmaf chain): tolerance 2e-5 * max|ref|
(SURVEY.md 8d).  bf16 mode: inputs are rounded to bf16 first, the oracle runs in fp32 on
the rounded inputs, and the kernel may differ by its internal bf16 roundings: rel-RMS <= 1e-2.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from afldm_amd import ops
    return ops


def rnd(dtype, *t):
    """round CPU fp32 tensors through `dtype` (so oracle and kernel see identical inputs)"""
    out = [x.to(dtype).to(torch.float32) for x in t]
    return out[0] if len(out) == 1 else out


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(device="cuda", dtype=dtype)


def back(y):
    return y.to(torch.float32).permute(0, 3, 1, 2).contiguous().cpu()


def close(got, ref, dtype, what="", f32_tol=2e-5, bf16_rms=1e-2):
    got, ref = got.double(), ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    scale = ref.abs().max().clamp_min(1e-30)
    if dtype == torch.float32:
        err = (got - ref).abs().max() / scale
        assert err <= f32_tol, f"{what}: max-abs/scale {err:.3e} > {f32_tol}"
    else:
        rms = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30))
        assert rms <= bf16_rms, f"{what}: rel-RMS {rms:.3e} > {bf16_rms}"
        assert (got - ref).abs().max() / scale <= 8 * bf16_rms, f"{what}: max err too large"


def test_library_is_native_and_on_gfx950():
    from afldm_amd import _lib
    buf = (_lib.ctypes.c_char * 64)()
    cus = _lib.lib.afldm_device_info(buf, 64)
    assert cus >= 200 and b"gfx950" in buf.value


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_and_pack(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = rnd(dtype, torch.randn(3, 5, 6, 7, generator=g))
    y = ops.to_nhwc(x.cuda(), dtype)
    assert torch.equal(y.float().cpu(), x.permute(0, 2, 3, 1))
    assert torch.equal(ops.to_nchw(y).cpu(), x)
    w = rnd(dtype, torch.randn(10, 6, 3, 3, generator=g))
    assert torch.equal(ops.pack_weight(w.cuda(), dtype).float().cpu(), w.permute(0, 2, 3, 1))
    wl = rnd(dtype, torch.randn(10, 6, generator=g))
    assert torch.equal(ops.pack_weight(wl.cuda(), dtype).float().cpu().reshape(10, 6), wl)


def test_timestep_embedding_and_silu():
    from oracle.unet import timestep_embedding
    ops = _ops()
    t = torch.tensor([981.0, 501.0, 1.0])
    got = ops.timestep_embedding(t.cuda(), 192).cpu()
    ref = timestep_embedding(t, 192)
    assert (got - ref).abs().max() < 2e-4          # sin/cos of ~1e3 rad in fp32
    x = torch.randn(1000)
    assert (ops.silu(x.cuda()).cpu() - F.silu(x)).abs().max() < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C1,C2,G,HW", [(64, 0, 32, 64), (96, 48, 8, 16), (192, 0, 32, 1024), (768, 384, 32, 16)])
def test_groupnorm(dtype, C1, C2, G, HW):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    side = int(HW ** 0.5)
    x = rnd(dtype, torch.randn(2, C1 + C2, side, side, generator=g) * 2 + 0.5)
    gamma = torch.randn(C1 + C2, generator=g)
    beta = torch.randn(C1 + C2, generator=g)
    x1 = nhwc(x[:, :C1], dtype)
    x2 = nhwc(x[:, C1:], dtype) if C2 else None
    st = ops.gn_stats(x1, G, x2=x2)
    xg = x.view(2, G, -1)
    mean, rs = ops.gn_mean_rstd(st, G, xg.shape[-1], 1e-5)
    assert (mean.cpu() - xg.mean(-1)).abs().max() < 1e-5
    rstd = 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5)
    assert ((rs.cpu() - rstd).abs() / rstd).max() < 1e-5
    for act in (0, 1):
        y = ops.gn_apply(x1, st, gamma.cuda(), beta.cuda(), G, 1e-5, act=act, x2=x2)
        ref = F.group_norm(x, G, gamma, beta, 1e-5)
        ref = F.silu(ref) if act else ref
        close(back(y), ref, dtype, f"gn_apply act={act}", bf16_rms=4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N", [2, 4, 8, 16, 32])
def test_af_act_vs_reference_fixture(golden, dtype, N):
    """The committed input/output pair comes from the IMPORTED reference modules."""
    ops = _ops()
    g = golden("g2_filters.npz")
    x = torch.from_numpy(g[f"wx_{N}"])             # [2, 6, N, N]
    x = torch.cat([x, x.flip(1), x * 0.5, x[:, :, :, :].roll(1, 3)], 1)[:, :16]   # 16 channels
    from oracle import ideal_filters as idf
    xr = rnd(dtype, x)
    ref = idf.warped_nonlinearity(xr)
    if dtype == torch.float32:      # channels 0..5 are literally the reference's recorded output
        assert torch.allclose(ref[:, :6], torch.from_numpy(g[f"wy_{N}"]), atol=1e-6)
    y = ops.af_act(nhwc(xr, dtype))
    close(back(y), ref, dtype, f"af_act N={N}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C1,C2,G", [(32, 32, 16, 8), (16, 64, 32, 32), (8, 96, 48, 8), (4, 64, 0, 32), (2, 48, 96, 8),
                                       (32, 192, 0, 32)])
def test_af_act_fused_groupnorm_concat(dtype, N, C1, C2, G):
    from oracle import ideal_filters as idf
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = rnd(dtype, torch.randn(2, C1 + C2, N, N, generator=g) * 1.5 + 0.3)
    gamma = 1 + 0.2 * torch.randn(C1 + C2, generator=g)
    beta = 0.1 * torch.randn(C1 + C2, generator=g)
    x1 = nhwc(x[:, :C1], dtype)
    x2 = nhwc(x[:, C1:], dtype) if C2 else None
    st = ops.gn_stats(x1, G, x2=x2)
    y = ops.af_act(x1, x2, st, gamma.cuda(), beta.cuda(), G, 1e-5)
    ref = idf.warped_nonlinearity(F.group_norm(x, G, gamma, beta, 1e-5))
    close(back(y), ref, dtype, f"gn+af_act N={N}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C,S", [(16, 384, 2), (32, 192, 4), (16, 192, 1)])
def test_af_act_plane_pipelined_statistics_match_the_inline_path(dtype, N, C, S):
    """k_af_act_plane, round 3: a persistent workgroup requests the NEXT item's GroupNorm partial sums before the MFMA
    passes of the current one and folds them afterwards (its first item still does it in line).  At batch 64 every
    workgroup walks 2 - 3 items (pipelined path); the same samples run alone take the in-line path (one item per
    workgroup): the two must agree bit for bit, and both with GroupNorm -> WarpedNonlinearity of the oracle."""
    from oracle import ideal_filters as idf
    ops = _ops()
    g = torch.Generator().manual_seed(N + C)
    B, G = 64, 32
    x = rnd(dtype, torch.randn(B, C, N, N, generator=g) * 1.3 + 0.2)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    xh = nhwc(x, dtype)
    # partial sums in S row splits per sample, as a producing convolution's epilogue would leave them
    rows = xh.float().view(B, S, N * N // S, C)
    st = ops.GNStats(torch.stack([rows.sum(2), (rows * rows).sum(2)], -1).contiguous(), None)
    y = ops.af_act(xh, None, st, gamma, beta, G, 1e-5)
    for b0 in (0, 31, 62):
        sub = ops.GNStats(st.st1[b0:b0 + 2].contiguous(), None)
        y2 = ops.af_act(xh[b0:b0 + 2].contiguous(), None, sub, gamma, beta, G, 1e-5)
        assert torch.equal(y2, y[b0:b0 + 2]), b0
    ref = idf.warped_nonlinearity(F.group_norm(x[:3], G, gamma.cpu(), beta.cpu(), 1e-5))
    close(back(y[:3]), ref, dtype, f"pipelined gn+af_act N={N}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C", [(2, 64), (4, 96), (4, 128), (8, 64), (8, 384), (16, 64), (16, 384), (32, 192)])
def test_af_resample(dtype, N, C):
    from oracle import ideal_filters as idf
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = rnd(dtype, torch.randn(2, C, N, N, generator=g))
    if N <= 16:
        close(back(ops.af_up2(nhwc(x, dtype))), idf.upsample_rfft(x, 2), dtype, f"af_up2 N={N}", bf16_rms=4e-3)
    if N >= 4:
        ref = idf.lpf_rfft(x)[:, :, ::2, ::2]
        y = ops.af_lpf_down2(nhwc(x, dtype), want_stats=True)
        close(back(y), ref, dtype, f"af_lpf_down2 N={N}", bf16_rms=4e-3)
        st = getattr(y, "gn_partial", None)        # emitted by the resampling kernel itself (one split per row / plane)
        assert st is not None and st.shape[0] == 2 and st.shape[2:] == (C, 2)
        yv = y.float()
        assert (st.sum(1)[..., 0] - yv.sum((1, 2))).abs().max() <= 1e-3 * (1 + yv.sum((1, 2)).abs().max())
        assert (st.sum(1)[..., 1] - (yv * yv).sum((1, 2))).abs().max() <= 1e-3 * (1 + (yv * yv).sum((1, 2)).max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C,B", [(16, 96, 3), (32, 192, 2), (32, 48, 5), (16, 384, 2)])
def test_af_resample_plane(dtype, N, C, B):
    """The one-kernel MFMA resample (afldm_af_resample_plane: 16 -> 32 with U, 32 -> 16 with D) against
    the oracle's rfft definition and the two-pass path, and its emitted GroupNorm partial sums against
    the stand-alone statistics pass over its own output."""
    from oracle import ideal_filters as idf
    ops = _ops()
    g = torch.Generator().manual_seed(N + C)
    x = rnd(dtype, torch.randn(B, C, N, N, generator=g))
    xh = nhwc(x, dtype)
    if N == 16:
        y = ops.af_up2(xh)
        ref = idf.upsample_rfft(x, 2)
        M = ops.up_matrix(N, 2, xh.device)
        assert getattr(y, "gn_partial", None) is None
    else:
        y = ops.af_lpf_down2(xh, want_stats=True)
        ref = idf.lpf_rfft(x)[:, :, ::2, ::2]
        M = ops.down_matrix(N, xh.device)
        st = y.gn_partial
        assert st.shape == (B, 1, C, 2)
        want = ops.gn_stats(y, out=torch.empty((B, ops.lib.afldm_gn_stats_splits(y.shape[1] * y.shape[2]), C, 2),
                                               dtype=torch.float32, device="cuda")).st1.sum(1)
        torch.testing.assert_close(st[:, 0], want, rtol=2e-5, atol=2e-3)
    close(back(y), ref, dtype, f"resample_plane N={N}", bf16_rms=4e-3)
    two = ops.af_resample(xh, M)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    d = (y.float() - two.float()).pow(2).mean().sqrt() / two.float().pow(2).mean().sqrt()
    assert float(d) <= tol, float(d)


@pytest.mark.parametrize("C,B,HW", [(192, 4, 1024), (192, 8, 1024), (384, 16, 256), (384, 64, 64)])
def test_weights_in_registers_projections(C, B, HW):
    """The short-K attention projections (lin.hip: weights in registers, persistent workgroups) at
    eligible sizes (bf16, K = 192 / 384, M >= 4096): fused q|k|v with channel-major V^T, and to_out with
    residual + GroupNorm partial sums.  Reference 1: fp32 torch.  Reference 2 (bit-exact): the general
    implicit-GEMM kernel, reached by appending 16 rows so that M is no multiple of the token tile."""
    ops = _ops()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(C + B)
    T = HW
    x = torch.randn(B, T, C, generator=g).to(dt).cuda()
    w = (torch.randn(3 * C, 1, 1, C, generator=g) / C ** 0.5).to(dt).cuda()
    bias = torch.randn(3 * C, generator=g).cuda()
    qk, vt = ops.linear_split(x, w, bias, 2 * C)
    ref = x.float() @ w.view(3 * C, C).float().t() + bias
    assert (qk.float() - ref[..., :2 * C]).abs().max() <= 3e-2 * ref.abs().max()
    assert (vt.float() - ref[..., 2 * C:].transpose(1, 2)).abs().max() <= 3e-2 * ref.abs().max()
    # general kernel on the same rows (one long "sample" of B*T + 16 tokens is not tile-aligned)
    xl = torch.cat([x.view(1, B * T, C), torch.zeros(1, 16, C, dtype=dt, device="cuda")], 1).contiguous()
    qk2, vt2 = ops.linear_split(xl, w, bias, 2 * C)
    assert torch.equal(qk.view(B * T, 2 * C), qk2[0, :B * T])
    assert torch.equal(vt.transpose(1, 2).reshape(B * T, C), vt2[0].t()[:B * T])
    # to_out: residual + statistics
    side = int(round(HW ** 0.5))
    o = torch.randn(B, side, side, C, generator=g).to(dt).cuda()
    res = torch.randn(B, side, side, C, generator=g).to(dt).cuda()
    wo = (torch.randn(C, 1, 1, C, generator=g) / C ** 0.5).to(dt).cuda()
    bo = torch.randn(C, generator=g).cuda()
    y = ops.conv2d(o, wo, bo, residual=res, want_stats=True)
    yref = o.float() @ wo.view(C, C).float().t() + bo + res.float()
    assert (y.float() - yref).abs().max() <= 3e-2 * yref.abs().max()
    ol = torch.cat([o.view(1, B * HW, 1, C), torch.zeros(1, 16, 1, C, dtype=dt, device="cuda")], 1).contiguous()
    rl = torch.cat([res.view(1, B * HW, 1, C), torch.zeros(1, 16, 1, C, dtype=dt, device="cuda")], 1).contiguous()
    y2 = ops.conv2d(ol, wo, bo, residual=rl)
    assert torch.equal(y.view(B * HW, C), y2[0, :B * HW, 0])
    st = y.gn_partial
    assert st.shape[0] == B and st.shape[2:] == (C, 2)
    yv = y.float()
    assert (st.double().sum(1)[..., 0].cpu() - yv.sum((1, 2)).double().cpu()).abs().max() <= 1e-3 * (1 + yv.sum((1, 2)).abs().max().item())
    assert (st.double().sum(1)[..., 1].cpu() - (yv * yv).sum((1, 2)).double().cpu()).abs().max() <= 1e-3 * (1 + (yv * yv).sum((1, 2)).max().item())
    # repeated launches are deterministic (no cross-workgroup races in the ring / staging)
    for _ in range(3):
        qk3, vt3 = ops.linear_split(x, w, bias, 2 * C)
        assert torch.equal(qk3, qk) and torch.equal(vt3, vt)


CONV_CASES = [
    # B, H, W, C1, C2, Cout, KS, temb, residual
    (2, 16, 16, 64, 0, 64, 3, False, False),
    (2, 16, 16, 128, 64, 64, 3, True, True),       # concat + temb + residual
    (2, 8, 8, 128, 128, 128, 1, False, False),     # 1x1 shortcut on a concat
    (3, 5, 7, 64, 0, 96, 3, True, False),          # ragged M, Cout tail, H != W
    (2, 2, 2, 256, 0, 128, 3, False, True),        # tiny M -> split-K
    (1, 4, 4, 768, 768, 768, 3, True, False),      # FFHQ U0-like, split-K
    (2, 32, 32, 192, 0, 192, 3, True, False),      # FFHQ level-0 resnet conv
    (2, 32, 32, 4, 0, 192, 3, False, False),       # conv_in (direct small-Cin kernel)
    (2, 32, 32, 192, 0, 4, 3, False, False),       # conv_out (direct small-Cout kernel)
    (64, 1, 1, 192, 0, 768, 1, False, False),      # time-MLP linear
    (1, 1, 1, 768, 0, 1408, 1, False, False),      # M = 1 linear (batched time_emb_proj)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(dtype, case):
    ops = _ops()
    B, H, W, C1, C2, Cout, KS, use_temb, use_res = case
    g = torch.Generator().manual_seed(4)
    x = rnd(dtype, torch.randn(B, C1 + C2, H, W, generator=g))
    w = rnd(dtype, torch.randn(Cout, C1 + C2, KS, KS, generator=g) / (KS * (C1 + C2) ** 0.5))
    b = torch.randn(Cout, generator=g)
    temb = rnd(dtype, torch.randn(B, Cout, generator=g)) if use_temb else None
    res = rnd(dtype, torch.randn(B, Cout, H, W, generator=g)) if use_res else None
    ref = F.conv2d(x, w, b, padding=KS // 2)
    if use_temb:
        ref = ref + temb[:, :, None, None]
    if use_res:
        ref = ref + res
    y = ops.conv2d(nhwc(x[:, :C1], dtype), ops.pack_weight(w.cuda(), dtype), b.cuda(),
                   x2=nhwc(x[:, C1:], dtype) if C2 else None,
                   temb=temb.to(device="cuda", dtype=dtype) if use_temb else None, temb_stride=Cout if use_temb else 0,
                   residual=nhwc(res, dtype) if use_res else None)
    close(back(y), ref, dtype, f"conv {case}", bf16_rms=6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_every_gemm_variant(dtype):
    """Force each implicit-GEMM tile / pipeline variant (afldm_conv2d_tune) on one ragged shape:
    M = 2*12*12 = 288 (not a tile multiple), virtual concat, Cout = 200 (ragged N), split-K 1 and 3."""
    from afldm_amd import _lib
    ops = _ops()
    B, H, W, C1, C2, Cout, KS = 2, 12, 12, 128, 64, 200, 3
    g = torch.Generator().manual_seed(14)
    x = rnd(dtype, torch.randn(B, C1 + C2, H, W, generator=g))
    w = rnd(dtype, torch.randn(Cout, C1 + C2, KS, KS, generator=g) / (KS * (C1 + C2) ** 0.5))
    b = torch.randn(Cout, generator=g)
    res = rnd(dtype, torch.randn(B, Cout, H, W, generator=g))
    ref = F.conv2d(x, w, b, padding=1) + res
    x1, x2, wp, rp = nhwc(x[:, :C1], dtype), nhwc(x[:, C1:], dtype), ops.pack_weight(w.cuda(), dtype), nhwc(res, dtype)
    ws = torch.empty(4 * B * H * W * Cout, dtype=torch.float32, device="cuda")
    nvar = 0
    try:
        for v in range(64):
            if _lib.lib.afldm_conv2d_tune(v, 1) != 0:
                break
            nvar += 1
            for sk in (1, 3):
                _lib.check(_lib.lib.afldm_conv2d_tune(v, sk), "tune")
                y = ops.conv2d(x1, wp, b.cuda(), x2=x2, residual=rp, workspace=ws)
                close(back(y), ref, dtype, f"conv variant {v} splitk {sk}", bf16_rms=6e-3)
    finally:
        _lib.lib.afldm_conv2d_tune(-1, -1)
    assert nvar >= 55


@pytest.mark.parametrize("case", [
    # B, H, W, C1, C2, Cout, KS, temb, residual
    (6, 32, 32, 192, 0, 192, 3, True, False),     # 48 tiles on <= 256 workgroups: 1 tile each
    (80, 16, 16, 128, 64, 384, 3, False, True),   # 320 tiles: 64 workgroups walk 2 tiles, virtual concat, residual
    (72, 16, 16, 192, 0, 768, 1, True, True),     # 576 tiles (3 rounds for some), 1x1, temb + residual, m-fast order
])
def test_conv2d_persistent_tile_variant_bit_identical(case):
    """Variant 40 (k_igemm3: persistent workgroups, producers running ahead across tile boundaries, wave-private
    epilogue) against variant 29 (one tile per workgroup) on the same call: outputs bit-identical, GroupNorm
    partial sums equal after folding the splits, deterministic across repeats."""
    from afldm_amd import _lib
    ops = _ops()
    dt = torch.bfloat16
    B, H, W, C1, C2, Cout, KS, use_temb, use_res = case
    g = torch.Generator().manual_seed(B + Cout)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dt).cuda()
    x2 = torch.randn(B, H, W, C2, generator=g).to(dt).cuda() if C2 else None
    w = (torch.randn(Cout, KS, KS, C1 + C2, generator=g) / (KS * (C1 + C2) ** 0.5)).to(dt).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    temb = torch.randn(B, Cout, generator=g).to(dt).cuda() if use_temb else None
    res = torch.randn(B, H, W, Cout, generator=g).to(dt).cuda() if use_res else None
    outs = {}
    try:
        for v in (29, 40, 40):
            _lib.check(_lib.lib.afldm_conv2d_tune(v, 1), "tune")
            y = ops.conv2d(x1, w, bias, x2=x2, temb=temb, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True)
            outs.setdefault(v, []).append((y, y.gn_partial))
    finally:
        _lib.lib.afldm_conv2d_tune(-1, -1)
    y29, st29 = outs[29][0]
    for y40, st40 in outs[40]:
        assert torch.equal(y29, y40)
        assert st40.shape[1] == (H * W // 128) * 4
        yv = y40.float()
        assert (st40.double().sum(1)[..., 0].cpu() - yv.double().sum((1, 2)).cpu()).abs().max() <= 1e-3 * (1 + yv.abs().sum((1, 2)).max().item())
        torch.testing.assert_close(st40.sum(1), st29.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.equal(outs[40][0][1], outs[40][1][1])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, KS, temb, residual
    (64, 4, 4, 768, 768, 3, True, True),      # 128x192 tiles, 8 slices of one 4x4 sample each
    (64, 8, 8, 384, 384, 3, True, False),     # 128x192 tiles, 4 slices = half a sample each
    (64, 1, 1, 3072, 3072, 1, False, True),   # the dense form of a 3x3 convolution on a 2x2 plane: 64x64 tiles
    (8, 8, 8, 768, 384, 3, False, True),      # batch-8 shapes
    (8, 16, 16, 384, 384, 3, True, True),
    (3, 4, 4, 768, 768, 3, True, True),       # M = 48: ragged last tile
    (1, 16, 16, 384, 384, 3, True, True),     # batch 1
    (1, 32, 32, 192, 192, 3, True, True),
])
def test_conv2d_split_k_reduced_inside_the_launch(dtype, case):
    """afldm_conv_args.sync: the K slices of a tile are reduced by the GEMM launch itself (arrival counter, slabs
    written through, reduce-scatter over the slices) instead of by k_splitk_reduce*.  Same arithmetic in the same
    order: outputs bit-identical to the two-launch form, statistics equal after folding the splits, 30 repeats
    identical (the hand-off must not depend on timing), and the counter words are zero again afterwards."""
    import ctypes
    from afldm_amd import _lib
    ops = _ops()
    B, H, W, Cin, Cout, KS, use_temb, use_res = case
    g = torch.Generator().manual_seed(B * 7 + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, KS, KS, Cin, generator=g) / (KS * Cin ** 0.5)).to(dtype).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    temb = torch.randn(B, Cout, generator=g).to(dtype).cuda() if use_temb else None
    res = torch.randn(B, H, W, Cout, generator=g).to(dtype).cuda() if use_res else None

    def run(fused):
        y = torch.empty(B, H, W, Cout, dtype=dtype, device="cuda")
        a = ops.conv_args(x, w, bias, temb=temb, temb_stride=Cout if use_temb else 0, residual=res, out=y)
        if not fused:
            a.sync, a.sync_bytes = None, 0
        need = _lib.lib.afldm_conv2d_workspace(ctypes.byref(a))
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device="cuda")
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
        S = _lib.lib.afldm_conv2d_stats_splits(ctypes.byref(a))
        st = torch.zeros(B, S, Cout, 2, dtype=torch.float32, device="cuda")
        a.stats_out = _lib.ptr(st)
        ops.conv2d_launch(a)
        return y, st, need

    y0, st0, need = run(False)
    if need == 0:
        pytest.skip("the planner does not split K for this shape")
    _lib.check(_lib.lib.afldm_conv2d_fused_splitk(1), "fused_splitk")
    try:
        ys = [run(True) for _ in range(30)]
        torch.cuda.synchronize()
    finally:
        _lib.lib.afldm_conv2d_fused_splitk(0)
    assert not ops.fused_splitk_error()        # no slice timed out (the error word of the sync buffers stays clear)
    for y1, st1, _ in ys:
        assert torch.equal(y0, y1)
        torch.testing.assert_close(st1.sum(1), st0.sum(1), rtol=1e-5, atol=1e-3)
        assert torch.equal(st1, ys[0][1])
    assert int(ops._sync_words(x.device).abs().sum()) == 0
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w.float().permute(0, 3, 1, 2).cpu(), bias.cpu(), padding=KS // 2)
    if use_temb:
        ref = ref + temb.float().cpu()[:, :, None, None]
    if use_res:
        ref = ref + res.float().permute(0, 3, 1, 2).cpu()
    close(back(ys[0][0]), ref, dtype, f"fused split-K {case}", bf16_rms=6e-3)


H3_CASES = [
    # B, N (plane), Cin, Cout, temb, residual, variants (conv3h.hip ids)
    (2, 32, 192, 192, True, False, (41, 45, 46, 47, 50)),
    (3, 32, 384, 192, False, True, (41, 45, 46, 47, 50)),
    (2, 32, 64, 384, True, True, (41, 45, 46, 47, 50)),
    (3, 16, 384, 384, True, True, (42, 43, 44, 48, 49)),
    (5, 16, 128, 192, False, False, (42, 43, 44, 48, 49)),
    (5, 8, 384, 384, True, True, (51,)),          # small-plane variants: one 8x8 sample per tile, 3 taps per step
    (3, 8, 768, 96, False, True, (51,)),
    (8, 4, 768, 768, True, True, (52,)),           # four 4x4 samples per tile, channel blocks split over 2-4 slices
    (4, 4, 256, 96, False, False, (52,)),
    (2, 16, 384, 96, True, True, (54,)),
    (2, 32, 192, 288, True, True, (55,)),
    (3, 32, 192, 192, True, True, (61,)),          # 128 x 96 tiles sized for two workgroups per CU (round 3, A/B variants)
    (5, 16, 384, 384, True, True, (62,)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", H3_CASES)
def test_conv3x3_halo_patch_variants(dtype, case):
    """conv3h.hip (halo patch staged once per channel block, nine taps as shifted views, producer / consumer waves):
    every variant against F.conv2d in fp32 on the same (rounded) inputs, with time embedding, residual and the fused
    GroupNorm statistics; bit-identical across reruns; the variant really ran (not the fallback)."""
    from afldm_amd import _lib
    ops = _ops()
    B, N, Cin, Cout, use_temb, use_res, variants = case
    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = rnd(dtype, torch.randn(B, Cin, N, N, generator=g))
    w = rnd(dtype, torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g)
    temb = rnd(dtype, torch.randn(B, Cout, generator=g)) if use_temb else None
    res = rnd(dtype, torch.randn(B, Cout, N, N, generator=g)) if use_res else None
    ref = F.conv2d(x, w, b, padding=1)
    if use_temb:
        ref = ref + temb[:, :, None, None]
    if use_res:
        ref = ref + res
    xh, wp = nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype)
    th = temb.to(device="cuda", dtype=dtype) if use_temb else None
    rh = nhwc(res, dtype) if use_res else None
    try:
        for v in variants:
            _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
            ys = [ops.conv2d(xh, wp, b.cuda(), temb=th, temb_stride=Cout if use_temb else 0, residual=rh, want_stats=True)
                  for _ in range(2)]
            import ctypes
            probe = ops.conv_args(xh, wp, b.cuda(), temb=th, temb_stride=Cout if use_temb else 0, residual=rh, out=ys[0])
            probe.workspace, probe.workspace_bytes = _lib.ptr(torch.empty(1 << 24, device="cuda")), 1 << 26
            code = _lib.lib.afldm_conv2d_variant(ctypes.byref(probe))
            assert code & 255 == v, "the halo variant did not run this shape"
            split = (code >> 8) & 255
            bm = 256 if v in (41, 44, 47, 49) else 64 if v in (51, 52, 53) else 128
            if split == 1:    # (with K slices the statistics come from the reduction kernel)
                assert ys[0].gn_partial.shape == (B, N * N // bm, Cout, 2), (v, ys[0].gn_partial.shape)     # the halo kernel ran
            close(back(ys[0]), ref, dtype, f"conv3h variant {v} {case}", bf16_rms=6e-3)
            assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0].gn_partial, ys[1].gn_partial)
            yv = ys[0].float()
            got = ys[0].gn_partial.double().sum(1).cpu()
            s1, s2 = yv.sum((1, 2)).cpu(), (yv * yv).sum((1, 2)).cpu()
            assert (got[..., 0] - s1).abs().max() <= 1e-4 * (1 + s1.abs().max()), f"sum {v}"
            assert (got[..., 1] - s2).abs().max() <= 1e-4 * (1 + s2.abs().max()), f"sumsq {v}"
    finally:
        _lib.lib.afldm_conv2d_tune(-1, -1)


def test_conv2d_operand_beyond_2gib_runs_in_whole_sample_chunks():
    """A pixel operand of 2 GiB or more (the AF-VAE's 256^2 levels at batch 128; here 34 x 256^2 x 512 channels bf16 =
    2.28 GB) does not fit a buffer descriptor: afldm_conv2d issues it as launches over whole-sample chunks on the fast
    kernels (round 3; it used to fall back to the round-1 register-staged kernel).  Every sample of the chunked call
    must equal the same sample convolved on its own, bit for bit, statistics included."""
    ops = _ops()
    dt = torch.bfloat16
    B, H, W, Cin, Cout = 34, 256, 256, 512, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, H, W, Cin, generator=g, device="cuda", dtype=torch.float32).to(dt)
    assert x.numel() * 2 >= 1 << 31
    w = (torch.randn(Cout, 3, 3, Cin, generator=g, device="cuda") / (3 * Cin ** 0.5)).to(dt)
    b = torch.randn(Cout, generator=g, device="cuda")
    temb = torch.randn(B, Cout, generator=g, device="cuda").to(dt)
    y = ops.conv2d(x, w, b, temb=temb, temb_stride=Cout, want_stats=True)
    assert torch.isfinite(y.float()).all()
    for i in (0, 16, 17, 33):
        yi = ops.conv2d(x[i:i + 1].contiguous(), w, b, temb=temb[i:i + 1].contiguous(), temb_stride=Cout, want_stats=True)
        assert torch.equal(yi[0], y[i]), i
        assert torch.equal(yi.gn_partial[0], y.gn_partial[i]), i
    ref = F.conv2d(x[33:34, :64, :64].float().permute(0, 3, 1, 2).cpu(), w.float().permute(0, 3, 1, 2).cpu(), b.cpu(), padding=1)
    ref = ref + temb[33].float().cpu()[None, :, None, None]
    close(back(y[33:34, :60, :60]), ref[:, :, :60, :60], dt, "chunked conv vs F.conv2d (interior crop)", bf16_rms=6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, temb, residual
    (2, 64, 64, 128, 128, True, True),
    (1, 72, 96, 64, 256, False, True),        # non-square plane, 9 x 3 blocks of 8 x 32 pixels
    (3, 128, 128, 128, 128, False, False),
])
def test_conv3x3_halo_patch_on_blocks_of_large_planes(dtype, case):
    """conv3h.hip variant 58 (round 3): the halo-patch kernel on 8 x 32 pixel blocks of planes larger than a tile (the
    AF-VAE's 64^2 .. 256^2 levels) - zero padding only at the image border, neighbours' pixels elsewhere - against
    F.conv2d, with residual / time embedding and the fused GroupNorm statistics (one record per block)."""
    from afldm_amd import _lib
    import ctypes
    ops = _ops()
    B, H, W, Cin, Cout, use_temb, use_res = case
    g = torch.Generator().manual_seed(H + W + Cin)
    x = rnd(dtype, torch.randn(B, Cin, H, W, generator=g))
    w = rnd(dtype, torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g)
    temb = rnd(dtype, torch.randn(B, Cout, generator=g)) if use_temb else None
    res = rnd(dtype, torch.randn(B, Cout, H, W, generator=g)) if use_res else None
    ref = F.conv2d(x, w, b, padding=1)
    if use_temb:
        ref = ref + temb[:, :, None, None]
    if use_res:
        ref = ref + res
    xh, wp = nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype)
    th = temb.to(device="cuda", dtype=dtype) if use_temb else None
    rh = nhwc(res, dtype) if use_res else None
    try:
        _lib.check(_lib.lib.afldm_conv2d_tune(58, -1), "tune")
        ys = [ops.conv2d(xh, wp, b.cuda(), temb=th, temb_stride=Cout if use_temb else 0, residual=rh, want_stats=True)
              for _ in range(2)]
        probe = ops.conv_args(xh, wp, b.cuda(), temb=th, temb_stride=Cout if use_temb else 0, residual=rh, out=ys[0])
        assert _lib.lib.afldm_conv2d_variant(ctypes.byref(probe)) & 255 == 58, "variant 58 did not take this shape"
    finally:
        _lib.lib.afldm_conv2d_tune(-1, -1)
    assert ys[0].gn_partial.shape == (B, min(H * W // 256, 32), Cout, 2)       # one record per block (ops folds past 32)
    close(back(ys[0]), ref, dtype, f"conv3h on blocks {case}", bf16_rms=6e-3)
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0].gn_partial, ys[1].gn_partial)
    yv = ys[0].float()
    got = ys[0].gn_partial.double().sum(1).cpu()
    s1, s2 = yv.sum((1, 2)).cpu(), (yv * yv).sum((1, 2)).cpu()
    assert (got[..., 0] - s1).abs().max() <= 1e-4 * (1 + s1.abs().max())
    assert (got[..., 1] - s2).abs().max() <= 1e-4 * (1 + s2.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    (2, 32, 32, 64, 0, 192, 3, True),      # H*W % 128 == 0, no split-K: statistics from the GEMM epilogue
    (2, 16, 16, 128, 64, 192, 3, True),    # virtual concat input, epilogue
    (3, 8, 8, 128, 0, 64, 3, False),       # 64x64 tiles: one tile per sample
    (2, 4, 4, 256, 256, 128, 3, True),     # split-K: statistics from the reduction kernel
    (2, 2, 2, 256, 0, 128, 3, False),      # tiny plane, split-K
    (2, 12, 12, 64, 0, 72, 3, False),      # H*W = 144: no fused producer -> stand-alone pass
    (2, 32, 32, 4, 0, 192, 3, False),      # conv_in: MFMA kernel with fused statistics (bf16), direct kernel (fp32)
    (3, 16, 16, 4, 0, 64, 3, False),       # conv_in shape with 2 statistic splits per sample
    (2, 12, 12, 4, 0, 64, 3, False),       # conv_in shape whose plane is no multiple of 128 pixels: direct kernel
    (2, 16, 16, 192, 0, 192, 1, True),     # 1x1 (attention to_out + residual)
    (24, 4, 4, 768, 0, 768, 1, True),      # to_out at the 4x4 level: a 64x64 tile holds 4 whole samples (bf16: epilogue statistics)
    (40, 2, 2, 768, 0, 768, 1, True),      # to_out at the 2x2 level: 16 samples per tile, ragged last tile
])
def test_conv2d_emits_groupnorm_statistics(dtype, case):
    """conv2d(..., want_stats=True): the per-channel partial sums attached to the output must equal
    those of a stand-alone pass over the stored tensor, whichever kernel produced them, and feed
    gn_apply to the same result."""
    ops = _ops()
    B, H, W, C1, C2, Cout, KS, use_res = case
    g = torch.Generator().manual_seed(21)
    x = rnd(dtype, torch.randn(B, C1 + C2, H, W, generator=g))
    w = rnd(dtype, torch.randn(Cout, C1 + C2, KS, KS, generator=g) / (KS * (C1 + C2) ** 0.5))
    b = torch.randn(Cout, generator=g)
    res = rnd(dtype, torch.randn(B, Cout, H, W, generator=g)) if use_res else None
    y = ops.conv2d(nhwc(x[:, :C1], dtype), ops.pack_weight(w.cuda(), dtype), b.cuda(),
                   x2=nhwc(x[:, C1:], dtype) if C2 else None, residual=nhwc(res, dtype) if use_res else None,
                   want_stats=True)
    assert hasattr(y, "gn_partial") and y.gn_partial.shape[0] == B and y.gn_partial.shape[2:] == (Cout, 2)
    ref = F.conv2d(x, w, b, padding=KS // 2) + (res if use_res else 0)
    close(back(y), ref, dtype, f"conv {case}", bf16_rms=6e-3)
    yv = y.float()                                            # statistics are those of the STORED values
    s1 = yv.sum((1, 2)).cpu()
    s2 = (yv * yv).sum((1, 2)).cpu()
    got = y.gn_partial.double().sum(1).cpu()
    assert (got[..., 0] - s1).abs().max() <= 1e-4 * (1 + s1.abs().max()), f"sum {case}"
    assert (got[..., 1] - s2).abs().max() <= 1e-4 * (1 + s2.abs().max()), f"sumsq {case}"
    G = 8
    gamma, beta = torch.randn(Cout, generator=g).cuda(), torch.randn(Cout, generator=g).cuda()
    fused = ops.gn_apply(y, ops.gn_stats(y), gamma, beta, G, 1e-5)
    plain = torch.empty_like(y)
    plain.copy_(y)                                            # a copy carries no attached statistics
    alone = ops.gn_apply(plain, ops.gn_stats(plain), gamma, beta, G, 1e-5)
    assert (fused.float() - alone.float()).abs().max() <= (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("B,Cin,Cout", [(8, 256, 192), (6, 768, 96), (64, 768, 768)])
def test_conv3x3_halo_patch_several_samples_per_tile_no_split(B, Cin, Cout):
    """conv3h.hip variant 52 without K slices (bf16): a 64-pixel tile holds four whole 4x4 samples - per-sample time
    embedding, residual and GroupNorm records come from the epilogue (the batch-64 plan of the 4x4 level; a ragged
    last tile is not allowed: B % 4 == 0 or the plan falls back)."""
    import ctypes
    from afldm_amd import _lib
    ops = _ops()
    dtype, N = torch.bfloat16, 4
    g = torch.Generator().manual_seed(B + Cin)
    x = rnd(dtype, torch.randn(B, Cin, N, N, generator=g))
    w = rnd(dtype, torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g)
    temb = rnd(dtype, torch.randn(B, Cout, generator=g))
    res = rnd(dtype, torch.randn(B, Cout, N, N, generator=g))
    ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + res
    xh, wp, th, rh = nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), temb.to(device="cuda", dtype=dtype), nhwc(res, dtype)
    try:
        _lib.check(_lib.lib.afldm_conv2d_tune(52, 1), "tune")
        ys = [ops.conv2d(xh, wp, b.cuda(), temb=th, temb_stride=Cout, residual=rh, want_stats=True) for _ in range(2)]
        probe = ops.conv_args(xh, wp, b.cuda(), temb=th, temb_stride=Cout, residual=rh, out=ys[0])
        code = _lib.lib.afldm_conv2d_variant(ctypes.byref(probe))
    finally:
        _lib.lib.afldm_conv2d_tune(-1, -1)
    if B % 4 == 0:
        assert code & 255 == 52 and (code >> 8) & 255 == 1, "variant 52 without slices did not run"
        assert ys[0].gn_partial.shape == (B, 1, Cout, 2)
    close(back(ys[0]), ref, dtype, f"conv3h multi-sample tile B={B}", bf16_rms=6e-3)
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0].gn_partial, ys[1].gn_partial)
    yv = ys[0].float()
    got = ys[0].gn_partial.double().sum(1).cpu()
    s1, s2 = yv.sum((1, 2)).cpu(), (yv * yv).sum((1, 2)).cpu()
    assert (got[..., 0] - s1).abs().max() <= 1e-4 * (1 + s1.abs().max())
    assert (got[..., 1] - s2).abs().max() <= 1e-4 * (1 + s2.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,Cin,Cout", [(64, 2, 768, 768), (64, 4, 768, 768), (8, 4, 768, 768), (16, 4, 256, 256), (5, 4, 64, 128)])
def test_resnet_conv1_norm2_act_from_split_k_slabs(dtype, B, N, Cin, Cout):
    """ResnetBlock2D on 2x2 / 4x4 planes: conv1 -> norm2 -> WarpedNonlinearity with the activation kernel fed by the
    convolution's split-K slabs (afldm_af_act_slabs: no reduction launch, no stored intermediate) against the ordinary
    three-launch sequence (same rounding points: equal up to the order of the GroupNorm sums) and against PyTorch."""
    import torch.nn as nn
    from afldm_amd import ops as aops
    from afldm_amd.af_modules.af_blocks import WarpedNonlinearity
    from afldm_amd.models import blocks
    g = torch.Generator().manual_seed(B * N + Cin)
    blk = blocks.ResnetBlock2D(in_channels=Cin, out_channels=Cout, temb_channels=64, groups=8, eps=1e-5)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(rnd(dtype, torch.randn(prm.shape, generator=g) * (0.05 if prm.ndim > 1 else 0.3)))
        blk.norm1.weight.add_(1.0)
        blk.norm2.weight.add_(1.0)
    blk.nonlinearity = WarpedNonlinearity(nn.SiLU())
    blk = blk.cuda().to(dtype)
    x = rnd(dtype, torch.randn(B, Cin, N, N, generator=g))
    temb = rnd(dtype, torch.randn(B, Cout, generator=g)).to(device="cuda", dtype=dtype)
    xh = nhwc(x, dtype)
    h = blk._norm_act(blk.norm1, xh)
    fused = blk._conv1_norm2_act_fused(h, temb, Cout)
    plain = blk._norm_act(blk.norm2, blocks.conv_forward(blk.conv1, h, temb=temb, temb_stride=Cout, want_stats=True))
    if fused is None:
        pytest.skip("the plan of this shape does not split K: nothing to fuse")
    assert fused.shape == plain.shape
    assert rel_rms_t(fused, plain) <= (1e-5 if dtype == torch.float32 else 4e-3)
    assert torch.equal(fused, blk._conv1_norm2_act_fused(h, temb, Cout))
    # and the whole block against PyTorch in fp32 on the same (rounded) parameters
    w1, b1 = blk.conv1.weight.float().cpu(), blk.conv1.bias.float().cpu()
    h_ref = F.conv2d(back(h), w1, b1, padding=1) + temb.float().cpu()[:, :, None, None]
    hn = F.group_norm(h_ref, 8, blk.norm2.weight.float().cpu(), blk.norm2.bias.float().cpu(), 1e-5)
    U, D = aops.filter_matrices(N, "cuda")
    U, D = U.cpu(), D.cpu()
    act_ref = torch.einsum("ph,bchw,qw->bcpq", U, hn, U)
    act_ref = torch.einsum("hp,bcpq,wq->bchw", D, F.silu(act_ref), D)
    assert rel_rms_t(back(fused), act_ref) <= (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,C", [(64, 4, 768), (64, 2, 768), (8, 4, 768)])
def test_resnet_conv2_hands_attention_its_normalised_input(dtype, B, N, C, monkeypatch):
    """ResnetBlock2D -> Attention on 2x2 / 4x4 planes: conv2's split-K slabs (+ bias + residual) are finished, stored and
    GroupNorm-ed for the attention block in one launch (afldm_af_act_slabs, act = 0); block output and attention output
    against the ordinary sequence (reduction launch, gn_apply)."""
    import torch.nn as nn
    from afldm_amd.af_modules.af_blocks import WarpedNonlinearity
    from afldm_amd.models import blocks
    g = torch.Generator().manual_seed(B + N + C)
    blk = blocks.ResnetBlock2D(in_channels=C, out_channels=C, temb_channels=64, groups=32, eps=1e-5)
    attn = blocks.Attention(C, heads=C // 24, dim_head=24, eps=1e-5, norm_num_groups=32, residual_connection=True, bias=True)
    with torch.no_grad():
        for m in (blk, attn):
            for prm in m.parameters():
                prm.copy_(rnd(dtype, torch.randn(prm.shape, generator=g) * (0.03 if prm.ndim > 1 else 0.3)))
        for nrm in (blk.norm1, blk.norm2, attn.group_norm):
            nrm.weight.add_(1.0)
    blk.nonlinearity = WarpedNonlinearity(nn.SiLU())
    blk, attn = blk.cuda().to(dtype), attn.cuda().to(dtype)
    xh = nhwc(rnd(dtype, torch.randn(B, C, N, N, generator=g)), dtype)
    temb = rnd(dtype, torch.randn(B, C, generator=g)).to(device="cuda", dtype=dtype)
    y = blk(xh, temb, C, next_gn=blocks._next_gn(attn))
    if getattr(y, "gn_applied", None) is None:
        pytest.skip("the plan of conv2 does not split K for this shape: nothing to hand over")
    o = attn(y)
    monkeypatch.setenv("AFLDM_NO_FUSED_ACT", "1")
    y0 = blk(xh, temb, C, next_gn=blocks._next_gn(attn))
    assert getattr(y0, "gn_applied", None) is None
    o0 = attn(y0)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_rms_t(y, y0) <= tol and rel_rms_t(o, o0) <= tol


def rel_rms_t(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("B,C,Cout", [(3, 192, 4), (2, 64, 3), (5, 128, 1)])
def test_unet_tail_norm_silu_conv_out_in_one_launch(B, C, Cout):
    """afldm_conv_out_fused (bf16, 32x32): conv_norm_out -> SiLU -> conv_out against gn_apply + conv2d (same rounding
    of the activated tensor) and against PyTorch in fp32; the statistics come from the producer's partial sums."""
    ops = _ops()
    dtype, N, G = torch.bfloat16, 32, 32 if C % 32 == 0 else 16
    g = torch.Generator().manual_seed(C + Cout)
    x = rnd(dtype, torch.randn(B, C, N, N, generator=g) * 1.5 + 0.3)
    w = rnd(dtype, torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5))
    b = torch.randn(Cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.conv2d(F.silu(F.group_norm(x, G, gamma, beta, 1e-5)), w, b, padding=1)
    xh, wp = nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype)
    xh.gn_partial = ops.gn_stats(xh).st1                    # as the producing convolution would have attached them
    ys = [ops.conv_out_fused(xh, wp, b.cuda(), gamma.cuda(), beta.cuda(), G, 1e-5) for _ in range(2)]
    assert ys[0] is not None and ys[0].shape == (B, N, N, Cout) and torch.equal(ys[0], ys[1])
    close(back(ys[0]), ref, dtype, f"fused UNet tail C={C} Cout={Cout}", bf16_rms=8e-3)
    hn = ops.gn_apply(xh, ops.gn_stats(xh), gamma.cuda(), beta.cuda(), G, 1e-5, act=1)
    plain = ops.conv2d(hn, wp, b.cuda())
    assert rel_rms_t(ys[0], plain) <= 6e-3


SKINNY_CASES = [
    # B, H, W, C1, C2, Cout, residual, temb       (1x1 convolutions / dense layers over few rows: skinny.hip)
    (16, 1, 1, 3072, 0, 3072, True, True),     # a 2x2-level 3x3 convolution in its dense form (batch 16)
    (1, 1, 1, 768, 0, 768, False, False),      # one row
    (8, 2, 2, 768, 768, 768, True, False),     # two inputs (virtual concat), 8 samples of 4 pixels in one 64-row block
    (5, 4, 4, 256, 0, 96, True, True),         # 80 rows: a ragged second block, 4 samples per block
    (1, 8, 8, 384, 0, 384, True, False),       # one 8x8 sample = one block
    (3, 8, 8, 384, 0, 416, False, True),       # three blocks; bf16: K = 384 = 4 slices of 3 steps, 13 blocks of 32 couts
    (2, 16, 16, 256, 0, 64, True, True),       # H*W = 256: four statistic splits per sample
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", SKINNY_CASES)
def test_conv1x1_few_rows_operands_straight_to_registers(dtype, case):
    """skinny.hip (M <= 1024 rows, K a multiple of 8 MFMA steps: one launch, no LDS staging, no split-K slabs) against
    F.conv2d in fp32 on the same (rounded) inputs, with bias, time embedding, residual, two-pointer input and the
    attached GroupNorm statistics; bit-identical across reruns; the kernel really ran."""
    import ctypes
    from afldm_amd import _lib
    ops = _ops()
    B, H, W, C1, C2, Cout, use_res, use_temb = case
    g = torch.Generator().manual_seed(C1 + Cout + B)
    x = rnd(dtype, torch.randn(B, C1 + C2, H, W, generator=g))
    w = rnd(dtype, torch.randn(Cout, C1 + C2, 1, 1, generator=g) / (C1 + C2) ** 0.5)
    b = torch.randn(Cout, generator=g)
    temb = rnd(dtype, torch.randn(B, Cout, generator=g)) if use_temb else None
    res = rnd(dtype, torch.randn(B, Cout, H, W, generator=g)) if use_res else None
    ref = F.conv2d(x, w, b)
    if use_temb:
        ref = ref + temb[:, :, None, None]
    if use_res:
        ref = ref + res
    kw = dict(x2=nhwc(x[:, C1:], dtype) if C2 else None, temb=temb.to(device="cuda", dtype=dtype) if use_temb else None,
              temb_stride=Cout if use_temb else 0, residual=nhwc(res, dtype) if use_res else None)
    x1, wp, bias = nhwc(x[:, :C1], dtype), ops.pack_weight(w.cuda(), dtype), b.cuda()
    probe = ops.conv_args(x1, wp, bias, kw["x2"], kw["temb"], kw["temb_stride"], kw["residual"],
                          torch.empty(B, H, W, Cout, device="cuda", dtype=dtype))
    assert _lib.lib.afldm_conv2d_variant(ctypes.byref(probe)) == -16, "the skinny kernel does not take this shape"
    ys = [ops.conv2d(x1, wp, bias, want_stats=True, **kw) for _ in range(2)]
    close(back(ys[0]), ref, dtype, f"skinny {case}", bf16_rms=6e-3)
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0].gn_partial, ys[1].gn_partial)
    assert ys[0].gn_partial.shape == (B, max(1, H * W // 64), Cout, 2)
    yv = ys[0].float()
    got = ys[0].gn_partial.double().sum(1).cpu()
    s1, s2 = yv.sum((1, 2)).cpu(), (yv * yv).sum((1, 2)).cpu()
    assert (got[..., 0] - s1).abs().max() <= 1e-4 * (1 + s1.abs().max())
    assert (got[..., 1] - s2).abs().max() <= 1e-4 * (1 + s2.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C1,C2,Cout", [(64, 0, 64), (96, 32, 48)])
def test_conv3x3_on_2x2_plane_runs_as_one_dense_layer(dtype, C1, C2, Cout):
    """blocks.conv_forward on a 2x2 plane (the 2x2 level of the UNet): the dense re-formulation
    (no zero-padding taps) must equal F.conv2d, including virtual concat, the per-channel time
    embedding (temb_mod), the residual and the attached GroupNorm statistics."""
    import torch.nn as nn
    from afldm_amd.models import blocks
    ops = _ops()
    g = torch.Generator().manual_seed(33)
    B = 5
    conv = nn.Conv2d(C1 + C2, Cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(rnd(dtype, torch.randn(conv.weight.shape, generator=g) / (3 * (C1 + C2) ** 0.5)))
        conv.bias.copy_(torch.randn(Cout, generator=g))
    x = rnd(dtype, torch.randn(B, C1 + C2, 2, 2, generator=g))
    temb = rnd(dtype, torch.randn(B, Cout, generator=g))
    res = rnd(dtype, torch.randn(B, Cout, 2, 2, generator=g))
    ref = F.conv2d(x, conv.weight, conv.bias, padding=1) + temb[:, :, None, None] + res
    conv = conv.cuda()
    xin = nhwc(x[:, :C1], dtype) if C2 == 0 else (nhwc(x[:, :C1], dtype), nhwc(x[:, C1:], dtype))
    y = blocks.conv_forward(conv, xin, temb=temb.to(device="cuda", dtype=dtype), temb_stride=Cout,
                            residual=nhwc(res, dtype), want_stats=True)
    assert y.shape == (B, 2, 2, Cout) and ("dense2x2", dtype, C1, C2) in conv.__dict__["_afldm_cache"]
    close(back(y), ref, dtype, "dense 2x2 conv", bf16_rms=6e-3)
    got = y.gn_partial.double().sum(1).cpu()
    yv = y.float()
    assert (got[..., 0] - yv.sum((1, 2)).cpu()).abs().max() < 1e-3
    assert (got[..., 1] - (yv * yv).sum((1, 2)).cpu()).abs().max() < 1e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,K,N,use_res", [(4, 1024, 192, 192, True), (4, 1024, 192, 576, False), (16, 256, 384, 384, True)])
def test_short_k_linear_layers_at_attention_sizes(dtype, B, HW, K, N, use_res):
    """The attention blocks' 1x1 layers at their real row counts (M >= 4096, K = 192 / 384): NHWC output
    with bias + residual + GroupNorm statistics from the epilogue, and the fused q|k|v form (token-major
    q|k, channel-major v through the transposed staging tile)."""
    ops = _ops()
    g = torch.Generator().manual_seed(44)
    side = int(HW ** 0.5)
    x = rnd(dtype, torch.randn(B, K, side, side, generator=g))
    w = rnd(dtype, torch.randn(N, K, 1, 1, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    res = rnd(dtype, torch.randn(B, N, side, side, generator=g)) if use_res else None
    ref = F.conv2d(x, w, b) + (res if use_res else 0)
    xh, wp = nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype)
    y = ops.conv2d(xh, wp, b.cuda(), residual=nhwc(res, dtype) if use_res else None, want_stats=True)
    close(back(y), ref, dtype, "streaming 1x1", bf16_rms=6e-3)
    yv = y.float()
    got = y.gn_partial.double().sum(1).cpu()
    assert (got[..., 0] - yv.sum((1, 2)).cpu()).abs().max() <= 1e-4 * (1 + yv.sum((1, 2)).abs().max().item())
    assert (got[..., 1] - (yv * yv).sum((1, 2)).cpu()).abs().max() <= 1e-4 * (1 + (yv * yv).sum((1, 2)).max().item())
    if N % 192 == 0 and N >= 384 and not use_res:      # fused projection: last third channel-major
        split = 2 * N // 3
        tok = xh.view(B, HW, K)
        qk, vt = ops.linear_split(tok, wp, b.cuda(), split)
        refs = (F.conv2d(x, w, b)).flatten(2)              # [B, N, HW]
        close(qk.float().cpu(), refs[:, :split].transpose(1, 2), dtype, "q|k token-major", bf16_rms=6e-3)
        close(vt.float().cpu(), refs[:, split:], dtype, "v channel-major", bf16_rms=6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_channel_major_output(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = rnd(dtype, torch.randn(2, 64, 8, 8, generator=g))
    w = rnd(dtype, torch.randn(96, 64, 1, 1, generator=g) / 8)
    b = torch.randn(96, generator=g)
    yt = ops.conv2d(nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), b.cuda(), out_mode=1)
    assert yt.shape == (2, 96, 64)
    close(yt.float().cpu().view(2, 96, 8, 8), F.conv2d(x, w, b), dtype, "conv out_mode=1", bf16_rms=6e-3)


def test_conv2d_errors_are_reported_not_fatal():
    from afldm_amd._lib import AfldmError
    ops = _ops()
    x = torch.zeros(1, 4, 4, 72, device="cuda")          # Cin=72: not a multiple of 32, too big for direct
    w = torch.zeros(64, 3, 3, 72, device="cuda")
    with pytest.raises(AfldmError, match="Cin"):
        ops.conv2d(x, w)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.conv2d(x.cpu(), w.cpu())
    ops.conv2d(torch.zeros(1, 4, 4, 64, device="cuda"), torch.zeros(64, 3, 3, 64, device="cuda"))   # still usable


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Bk,heads,T,d", [(2, 2, 8, 1024, 24), (3, 1, 16, 256, 24), (2, 2, 16, 64, 24),
                                            (4, 2, 32, 16, 24), (2, 1, 32, 4, 24), (2, 2, 4, 64, 16),
                                            (64, 64, 16, 64, 24), (32, 16, 32, 16, 24), (40, 40, 32, 4, 24)])   # >= 1024 (batch, head) pairs: 2- / 1-wave workgroups
def test_attention(dtype, B, Bk, heads, T, d):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    C = heads * d
    q = rnd(dtype, torch.randn(B, T, C, generator=g))
    k = rnd(dtype, torch.randn(Bk, T, C, generator=g))
    v = rnd(dtype, torch.randn(Bk, T, C, generator=g))
    rep = B // Bk
    kk = k.repeat_interleave(rep, 0).view(B, T, heads, d).transpose(1, 2)
    vv = v.repeat_interleave(rep, 0).view(B, T, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q.view(B, T, heads, d).transpose(1, 2), kk, vv)
    ref = ref.transpose(1, 2).reshape(B, T, C)
    o = ops.attention(q.to(device="cuda", dtype=dtype), k.to(device="cuda", dtype=dtype),
                      v.transpose(1, 2).contiguous().to(device="cuda", dtype=dtype), heads)
    close(o.float().cpu(), ref, dtype, f"attention T={T}", f32_tol=5e-5, bf16_rms=1e-2)


def test_attention_sharp_softmax_rows():
    """A spiked key forces the online-softmax rescale path (guide 5.4 rule 26)."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    B, heads, T, d = 1, 2, 256, 24
    q = torch.randn(B, T, heads * d, generator=g)
    k = torch.randn(B, T, heads * d, generator=g)
    v = torch.randn(B, T, heads * d, generator=g)
    k[0, 200] = q[0, 5] * 6.0          # late key dominates row 5
    k[0, 3] = q[0, 77] * 6.0           # early key dominates row 77
    ref = F.scaled_dot_product_attention(q.view(B, T, heads, d).transpose(1, 2), k.view(B, T, heads, d).transpose(1, 2),
                                         v.view(B, T, heads, d).transpose(1, 2)).transpose(1, 2).reshape(B, T, -1)
    o = ops.attention(q.cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda(), heads)
    close(o.cpu(), ref, torch.float32, "attention spiked", f32_tol=5e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_ddim_step(dtype):
    from oracle.ddim import DDIM
    ops = _ops()
    s = DDIM()
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 4, 32, 32, generator=g)
    eps = rnd(dtype, torch.randn(3, 4, 32, 32, generator=g))
    coef = torch.tensor([c for t in s.timesteps for c in s.coefficients(t)], dtype=torch.float32).cuda()
    idx = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i in (0, 1, 49):
        idx.fill_(i)
        out = ops.ddim_step(x.cuda(), nhwc(eps, dtype), coef, idx, advance=True)
        ref = s.step(eps, s.timesteps[i], x)
        assert (out.cpu() - ref).abs().max() <= 2e-5 * ref.abs().max()
        assert int(idx.item()) == i + 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_qkv_split_gemm_and_strided_attention(dtype):
    """One GEMM -> Q|K token-major + V channel-major, consumed by attention through column slices."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    B, T, heads, d = 2, 64, 8, 24          # C = 192: a multiple of the bf16 K step (64)
    C = heads * d
    x = rnd(dtype, torch.randn(B, T, C, generator=g))
    w = rnd(dtype, torch.randn(3 * C, C, generator=g) / C ** 0.5)
    b = torch.randn(3 * C, generator=g)
    qk, vt = ops.linear_split(x.to(device="cuda", dtype=dtype), ops.pack_weight(w.cuda(), dtype), b.cuda(), 2 * C)
    ref = F.linear(x, w, b)
    close(qk.float().cpu(), ref[..., :2 * C], dtype, "qk", bf16_rms=6e-3)
    close(vt.float().cpu(), ref[..., 2 * C:].transpose(1, 2), dtype, "vt", bf16_rms=6e-3)
    o = ops.attention(qk[:, :, :C], qk[:, :, C:], vt, heads)
    q, k, v = (rnd(dtype, ref[..., i * C:(i + 1) * C]).view(B, T, heads, d).transpose(1, 2) for i in range(3))
    oref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, C)
    close(o.float().cpu(), oref, dtype, "attention on fused qkv", f32_tol=5e-5, bf16_rms=1.5e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C", [(32, 192), (16, 384), (8, 384), (4, 768), (2, 768)])
def test_alias_free_ops_full_size_properties(dtype, N, C):
    """Full-size (batch 64) properties of the alias-free kernels, independent of any reference run:
    * every operator is a circulant product, so it commutes with integer circular shifts:
      af_act(roll(x)) == roll(af_act(x))  (catches any pixel / plane ordering slip at every N);
    * the resampling operators are linear: op(a x + b z) == a op(x) + b op(z);
    * the fused GroupNorm path equals gn_apply followed by the plain activation."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    B = 64
    x = (torch.randn(B, N, N, C, generator=g) * 1.3 + 0.1).to(device="cuda", dtype=dtype)
    tol = 1e-5 if dtype == torch.float32 else 2e-2

    def rr(a, b):
        a, b = a.float(), b.float()
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))

    y = ops.af_act(x)
    ys = ops.af_act(torch.roll(x, shifts=(1, N - 1), dims=(1, 2)).contiguous())
    assert rr(ys, torch.roll(y, shifts=(1, N - 1), dims=(1, 2))) <= tol
    if N <= 16:
        z = torch.randn(B, N, N, C, generator=g).to(device="cuda", dtype=dtype)
        mix = (0.75 * x.float() - 1.5 * z.float()).to(dtype)
        for op in ((ops.af_up2, ops.af_lpf_down2) if N >= 4 else (ops.af_up2,)):
            lin = 0.75 * op(x).float() - 1.5 * op(z).float()
            assert rr(op(mix), lin) <= (1e-5 if dtype == torch.float32 else 2e-2), op.__name__
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    st = ops.gn_stats(x)
    fused = ops.af_act(x, None, st, gamma, beta, 32, 1e-5)
    two = ops.af_act(ops.gn_apply(x, st, gamma, beta, 32, 1e-5, act=0))
    assert rr(fused, two) <= (1e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("shape", [(64, 32, 32, 192, 192, 3), (64, 16, 16, 384, 384, 3), (64, 4, 4, 768, 768, 3),
                                   (64, 32, 32, 192, 576, 1)])
def test_conv_full_size_linearity(shape):
    """Batch-64 layer shapes of the FFHQ UNet (bf16): the convolution minus its bias is linear in the
    input - conv(a x + b z) - conv(0) == a (conv(x) - conv(0)) + b (conv(z) - conv(0)) - for the
    LDS-DMA tile variants, the producer/consumer variant, split-K and the staged epilogue alike."""
    ops = _ops()
    B, H, W, Cin, Cout, KS = shape
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, Cin, generator=g).to(device="cuda", dtype=dt)
    z = torch.randn(B, H, W, Cin, generator=g).to(device="cuda", dtype=dt)
    w = (torch.randn(Cout, KS, KS, Cin, generator=g) / (KS * Cin ** 0.5)).to(device="cuda", dtype=dt)
    b = torch.randn(Cout, generator=g).cuda()
    c0 = ops.conv2d(torch.zeros_like(x), w, b).float()
    mix = (0.5 * x.float() + 2.0 * z.float()).to(dt)
    lhs = ops.conv2d(mix, w, b).float() - c0
    rhs = 0.5 * (ops.conv2d(x, w, b).float() - c0) + 2.0 * (ops.conv2d(z, w, b).float() - c0)
    rel = float((lhs - rhs).pow(2).mean().sqrt() / rhs.pow(2).mean().sqrt())
    assert rel <= 1.5e-2, rel        # three bf16 roundings of O(1) values on each side


@pytest.mark.parametrize("T,heads", [(1024, 8), (256, 16)])
def test_attention_full_size_properties(T, heads):
    """Batch-64 attention (bf16): the output is invariant under a permutation of the key/value
    positions (the 64-key chunking and the lazy reference max see a different order), and a
    constant V comes back unchanged (rows of the softmax sum to one through the row of ones)."""
    ops = _ops()
    dt, B, d = torch.bfloat16, 64, 24
    C = heads * d
    g = torch.Generator().manual_seed(10)
    q = torch.randn(B, T, C, generator=g).to(device="cuda", dtype=dt)
    k = torch.randn(B, T, C, generator=g).to(device="cuda", dtype=dt)
    v = torch.randn(B, T, C, generator=g).to(device="cuda", dtype=dt)
    o = ops.attention(q, k, v.transpose(1, 2).contiguous(), heads).float()
    perm = torch.randperm(T, generator=g).cuda()
    op = ops.attention(q, k[:, perm].contiguous(), v[:, perm].transpose(1, 2).contiguous(), heads).float()
    assert float((o - op).pow(2).mean().sqrt() / o.pow(2).mean().sqrt()) <= 1e-2
    const = torch.full((B, C, T), 0.625, device="cuda", dtype=dt)
    oc = ops.attention(q, k, const, heads).float()
    assert (oc - 0.625).abs().max() <= 4e-3

