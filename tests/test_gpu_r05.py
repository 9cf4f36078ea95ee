"""Round-5 parity cases on MI355X.

VERDICT r04 item 2: the BENCH path - batch 64, bf16, the fused attention front end on by policy, the step replayed as
a HIP graph - tied to the oracle over a trajectory, not only over single forwards:
  * the 50-step C1 run, the 99-evaluation I2SB bridge and `ddim_inversion` a second time with every eligible attention
    block forced onto the fused launch (`_FUSED_ATTN_MIN_WGS = 0`: a batch of 1 takes it), against the same oracle
    fixtures the three-launch path is held to;
  * a batch-64 forward and a batch-64 50-step graph-replayed run whose sample 0 is the oracle fixture's input, asserted
    against the fixture (reference loop: ldm_pipeline.py:103-109).
Tolerances as in test_gpu_r02.py: bf16 forward <= 2e-2, multi-step <= 5e-2 rel-RMS."""
import numpy as np
import pytest
import torch

from test_gpu_r02 import build_unet, rel_rms

pytestmark = pytest.mark.gpu


def _force_fused(monkeypatch):
    """Every eligible attention block on afldm_attn_block_fused, whatever the batch; returns the call log."""
    from afldm_amd import ops
    monkeypatch.setattr(ops, "_FUSED_ATTN_MIN_WGS", 0)
    calls = []
    real = ops.attn_block_fused
    monkeypatch.setattr(ops, "attn_block_fused", lambda *a, **k: (calls.append(tuple(a[0].shape)), real(*a, **k))[1])
    return calls


def test_ffhq_full_50_step_ddim_fused_attention_vs_oracle(golden, monkeypatch):
    """C1 at full length (50 DDIM steps, FFHQ size, batch 1, graph replay) with the attention blocks of the 32^2 / 16^2
    levels on the fused front end - the kernel bench.py times - against the fp32 oracle (tests/golden/g13_r03.npz)."""
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    calls = _force_fused(monkeypatch)
    g = golden("g13_r03.npz")
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), 1, 50, use_graph=True)
    eng.reset(torch.from_numpy(g["ffhq_x"]))
    eng.step(25)
    mid = rel_rms(eng.lat, g["ffhq_ddim_step25"])
    eng.step(25)
    fin = rel_rms(eng.lat, g["ffhq_ddim_final"])
    print(f"[C1 50 steps, fused attention] bf16: rel-RMS vs oracle after 25 steps {mid:.3e}, after 50 steps {fin:.3e}; "
          f"{len(calls)} fused launches captured")
    assert len(calls) >= 10, calls                        # 10 eligible blocks per forward (warm-up + capture)
    assert mid <= 5e-2 and fin <= 5e-2, (mid, fin)


def test_ffhq_99_evaluation_bridge_and_inversion_fused_attention_vs_oracle(golden, monkeypatch):
    """C5's sampler at full length (99 UNet evaluations) and `ddim_inversion` with the fused attention front end forced
    (same fixtures and bounds as test_gpu_r02.py holds the three-launch path to)."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.pipelines.ldm_pipeline import MyLDMPipeline
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    calls = _force_fused(monkeypatch)
    g = golden("g14_r03.npz")
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    pipe = I2SBLDMPipeline(None, unet, I2SBScheduler.from_config(cfg))
    pipe.set_progress_bar_config(disable=True)
    out = pipe._bridge(torch.from_numpy(g["i2sb_start"]).cuda().to(torch.bfloat16), 100, True, None)
    r = rel_rms(out.float(), g["i2sb_final99"])
    n_bridge = len(calls)
    lp = MyLDMPipeline(None, unet, ffhq_ddim_scheduler())
    lp.scheduler.set_timesteps(6, device="cuda")
    inv = lp.ddim_inversion(torch.from_numpy(g["inv_in"]).cuda().to(torch.bfloat16), bar=False)
    ri = rel_rms(inv.float(), g["inv_out_6"])
    print(f"[C5 99 evaluations / inversion, fused attention] bf16: bridge {r:.3e} ({n_bridge} fused launches), inversion {ri:.3e}")
    assert n_bridge >= 10 and len(calls) > n_bridge
    assert r <= 6e-3 and ri <= 5e-2, (r, ri)


def test_bench_path_batch64_forward_sample0_vs_oracle_fixture(golden):
    """Exactly what bench.py times (batch 64, bf16, policy defaults: the fused attention front end is ON at this batch):
    sample 0 of the batch is the oracle fixture's input (g6_ffhq_unet.npz), its output must meet the fixture at the
    bf16 forward tolerance; the other 63 samples are noise."""
    from afldm_amd import ops
    g = golden("g6_ffhq_unet.npz")
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    x = torch.randn(64, 4, 32, 32, generator=torch.Generator().manual_seed(11))
    x[0] = torch.from_numpy(g["x"])[0]
    assert ops.attn_block_fused_ok(torch.empty(64, 1024, 192, dtype=torch.bfloat16, device="cuda"), 8, 32)
    y = unet(x.cuda(), 981).sample
    r = rel_rms(y[:1].float(), g["y_t981"])
    print(f"[bench path, batch 64 forward] sample 0 vs oracle fixture: rel-RMS {r:.3e}")
    assert r <= 2e-2, r


def test_bench_path_batch64_50_steps_sample0_vs_oracle(golden):
    """The bench configuration over the reference's whole loop: a batch-64 bf16 DenoiseEngine (HIP-graph replay of the
    step, fused attention on by policy) runs all 50 DDIM steps; sample 0 started from the oracle fixture's noise and
    must meet the oracle's step-25 and final latents (tests/golden/g13_r03.npz) at the multi-step tolerance."""
    from afldm_amd.engine import DenoiseEngine
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    g = golden("g13_r03.npz")
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    x = torch.randn(64, 4, 32, 32, generator=torch.Generator().manual_seed(12))
    x[0] = torch.from_numpy(g["ffhq_x"])[0]
    eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), 64, 50, use_graph=True)
    eng.reset(x)
    eng.step(25)
    mid = rel_rms(eng.lat[:1], g["ffhq_ddim_step25"])
    eng.step(25)
    fin = rel_rms(eng.lat[:1], g["ffhq_ddim_final"])
    assert torch.isfinite(eng.lat).all()
    print(f"[bench path, batch 64, 50 steps] sample 0 vs oracle: after 25 steps {mid:.3e}, after 50 steps {fin:.3e}")
    assert mid <= 5e-2 and fin <= 5e-2, (mid, fin)


# ------------------------------------------------------------------------------------------------ merged launches
@pytest.mark.parametrize("case", [
    # B, N, C1, C2, Cout, temb, residual
    (64, 32, 192, 0, 192, True, False),       # conv1 of a 32^2 resnet block: XCD-local hand-over, 4 workgroups per sample
    (64, 32, 192, 0, 192, False, True),       # conv2 (+ residual)
    (64, 32, 384, 192, 192, True, False),     # up block: virtual concat, groups straddle the two tensors
    (64, 16, 384, 0, 384, False, True),       # 16^2: 2 x 2 tiles per sample, 8-wave workgroups
    (64, 16, 768, 384, 384, True, False),
    (64, 16, 192, 0, 384, True, False),
    (40, 32, 192, 0, 192, True, True),        # 160 tiles: clusters straddle XCDs -> the general (agent-scope) hand-over
    (36, 16, 384, 384, 384, True, False),
    (3, 32, 192, 0, 192, True, True),         # small batches take other convolution tiles: the two launches (None)
    (16, 32, 192, 192, 192, False, False),    # two clusters per XCD
])
def test_af_act_conv2d_merged_launch_bit_identical_to_two_launches(case):
    """afldm_af_act_conv2d (csrc/actconv.hip): norm -> WarpedNonlinearity -> 3x3 conv of a ResnetBlock2D as ONE launch with
    a per-sample hand-over inside it, against afldm_af_act followed by afldm_conv2d on the same inputs: output,
    activated tensor and GroupNorm partial sums bit-identical, re-runs bit-identical, no hand-over error flag."""
    from afldm_amd import ops
    B, N, C1, C2, Cout, use_temb, use_res = case
    g = torch.Generator().manual_seed(B * 1000 + N + C1)
    Ct = C1 + C2
    x1 = (torch.randn(B, N, N, C1, generator=g) * 1.5 + 0.3).cuda().to(torch.bfloat16)
    x2 = torch.randn(B, N, N, C2, generator=g).cuda().to(torch.bfloat16) if C2 else None
    w = ops.pack_weight((torch.randn(Cout, Ct, 3, 3, generator=g) * (9 * Ct) ** -0.5).cuda(), torch.bfloat16)
    bias = torch.randn(Cout, generator=g).cuda()
    gamma, beta = (1 + 0.2 * torch.randn(Ct, generator=g)).cuda(), (0.2 * torch.randn(Ct, generator=g)).cuda()
    temb = torch.randn(B, Cout, generator=g).cuda().to(torch.bfloat16) if use_temb else None
    res = torch.randn(B, N, N, Cout, generator=g).cuda().to(torch.bfloat16) if use_res else None
    stats = ops.gn_stats(x1, 32, x2=x2)
    a_ref = ops.af_act(x1, x2, stats, gamma, beta, 32, 1e-6)
    y_ref = ops.conv2d(a_ref, w, bias, temb=temb, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True)
    outs = [ops.af_act_conv2d(x1, x2, stats, gamma, beta, 32, 1e-6, w, bias, temb=temb, temb_stride=Cout if use_temb else 0,
                              residual=res, want_stats=True) for _ in range(3)]
    if B != 64 and outs[0] is None:
        # batches whose convolution plan is not the one-tile-per-CU halo variant (64-pixel tiles of small batches, ...) have no
        # merged kernel by design: the caller runs the two launches
        assert all(o is None for o in outs)
        return
    assert all(o is not None for o in outs), "no merged kernel for a shape the UNet's 32^2 / 16^2 levels use"
    # the chains conv -> act and act -> conv -> act (the activation BEHIND the convolution takes the statistics its epilogue writes)
    g2, b2 = (1 + 0.2 * torch.randn(Cout, generator=g)).cuda(), (0.2 * torch.randn(Cout, generator=g)).cuda()
    a2_ref = ops.af_act(y_ref, None, ops.gn_stats(y_ref, 32), g2, b2, 32, 1e-6)
    kw = dict(temb=temb, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True, post=(g2, b2, 32, 1e-6))
    for got in (ops.act_conv_act(a_ref, None, None, w, bias, **kw), ops.act_conv_act(x1, x2, (stats, gamma, beta, 32, 1e-6), w, bias, **kw)):
        assert got is not None
        assert torch.equal(got[0], y_ref) and torch.equal(got[1], a2_ref) and torch.equal(got[0].gn_partial, y_ref.gn_partial)
    # the general (placement-independent) hand-over on the same problem
    from afldm_amd import _exp, _lib
    _exp.lib().afldm_af_act_conv2d_mode(1)
    try:
        outs.append(ops.af_act_conv2d(x1, x2, stats, gamma, beta, 32, 1e-6, w, bias, temb=temb, temb_stride=Cout if use_temb else 0,
                                      residual=res, want_stats=True))
    finally:
        _exp.lib().afldm_af_act_conv2d_mode(0)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o.act_input, a_ref) and torch.equal(o, y_ref) and torch.equal(o.gn_partial, y_ref.gn_partial)
    assert ops.actconv_error() == 0
    # the no-normalisation form (stats = None) is the activation alone in front of the convolution
    a0 = ops.af_act(x1, x2)
    y0 = ops.conv2d(a0, w, bias)
    o0 = ops.af_act_conv2d(x1, x2, None, None, None, 0, 0.0, w, bias)
    assert o0 is not None and torch.equal(o0, y0) and torch.equal(o0.act_input, a0)


# ------------------------------------------------------------------------------------------------ cooperative 2x2 level
@pytest.mark.parametrize("B", [1, 8, 64])
def test_cooperative_2x2_level_vs_separate_launches_and_oracle(golden, B, monkeypatch):
    """afldm_trunk_run (csrc/trunk.hip + afldm_amd/trunk.py): down_blocks[-1] -> mid_block -> up_blocks[0].resnets of the FFHQ AF-UNet
    as ONE cooperative launch (opt-in: it measured slower, profiles/r05/trunk_coop.txt) against the 51 separate launches of the
    same level and against the oracle fixture: finite, bit-identical reruns, no barrier time-out, bf16 forward tolerance."""
    from afldm_amd import ops, trunk
    g = golden("g6_ffhq_unet.npz")
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    x = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(40 + B))
    x[0] = torch.from_numpy(g["x"])[0]
    x = x.cuda()
    monkeypatch.setattr(trunk, "_ENABLED", False)
    y_sep = unet(x, 981).sample
    monkeypatch.setattr(trunk, "_ENABLED", True)
    assert trunk.eligible(unet, torch.empty(B, 2, 2, 768, dtype=torch.bfloat16, device="cuda"))
    y1 = unet(x, 981).sample
    y2 = unet(x, 981).sample
    torch.cuda.synchronize()
    assert torch.isfinite(y1).all() and torch.equal(y1, y2) and ops.actconv_error() == 0
    r_sep, r_orc = rel_rms(y1.float(), y_sep.float().cpu()), rel_rms(y1[:1].float(), g["y_t981"])
    print(f"[cooperative 2x2 level] B={B}: vs separate launches {r_sep:.3e}, sample 0 vs oracle fixture {r_orc:.3e} "
          f"(separate launches: {rel_rms(y_sep[:1].float(), g['y_t981']):.3e})")
    assert r_sep <= 2e-2 and r_orc <= 2e-2


def test_conv2_epilogue_applies_the_attention_groupnorm_at_8x8(golden, monkeypatch):
    """afldm_conv_args.y_norm: at the 8x8 level one halo tile is a whole sample x whole groups, so conv2's epilogue applies the
    GroupNorm of the attention block behind it (diffusers Attention.group_norm) instead of a stand-alone afldm_gn_apply launch:
    same UNet output as the separate launch, five
    launches fewer per forward, fixture tolerance kept."""
    from afldm_amd import ops
    g = golden("g6_ffhq_unet.npz")
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    x = torch.randn(64, 4, 32, 32, generator=torch.Generator().manual_seed(77))      # (full batch: small batches split K at 8x8)
    x[0] = torch.from_numpy(g["x"])[0]
    x = x.cuda()
    calls = []
    real = ops.gn_apply
    monkeypatch.setattr(ops, "gn_apply", lambda *a, **k: (calls.append(tuple(a[0].shape)), real(*a, **k))[1])
    monkeypatch.setattr(ops, "_CONV_NORM", False)
    y_sep = unet(x, 981).sample
    n_sep = len(calls)
    monkeypatch.setattr(ops, "_CONV_NORM", True)
    del calls[:]
    y_fus = unet(x, 981).sample
    n_fus = len(calls)
    r = rel_rms(y_fus.float(), y_sep.float().cpu())
    print(f"[conv2 -> attention GroupNorm in the epilogue] gn_apply launches {n_sep} -> {n_fus}; fused vs separate rel-RMS {r:.3e}; "
          f"vs oracle fixture {rel_rms(y_fus[:1].float(), g['y_t981']):.3e}")
    assert n_sep - n_fus == 5 and r <= 2e-3 and rel_rms(y_fus[:1].float(), g["y_t981"]) <= 2e-2


@pytest.mark.parametrize("T,C,heads,B", [(64, 384, 16, 64), (16, 768, 32, 64), (64, 384, 16, 4), (16, 768, 32, 8)])
def test_attn_small_fused_vs_two_launch_path_and_reference(T, C, heads, B):
    """afldm_attn_small_fused (csrc/attns.hip): to_q | to_k | to_v + scaled_dot_product_attention of the 8x8 / 4x4 levels in ONE
    launch against (a) the fp32 restatement of diffusers' processor on the same normalised tokens and (b) the two launches it
    replaces (projection GEMM + afldm_attention); bf16 per-op tolerance 2e-2, bit-identical reruns."""
    from afldm_amd import ops
    g = torch.Generator().manual_seed(T + C + B)
    x = torch.randn(B, T, C, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(3 * C, C, generator=g) * C ** -0.5).cuda()
    b = (0.1 * torch.randn(3 * C, generator=g)).cuda()
    wp = ops.pack_weight(w, torch.bfloat16)
    from afldm_amd import _exp, _lib
    assert _exp.lib().afldm_attn_small_fused_supported(B, T, C, heads) == 1      # (the policy keeps it off: measured no faster in the step)
    scale = 24 ** -0.5
    o1 = ops.attn_small_fused(x, wp, b, heads, scale)
    o2 = ops.attn_small_fused(x, wp, b, heads, scale)
    qk, vt = ops.linear_split(x, wp, b, 2 * C)
    o_two = ops.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, scale=scale)
    torch.cuda.synchronize()
    xf, wf = x.float(), wp.reshape(3 * C, C).float()
    q, k, v = [(xf @ wf[i * C:(i + 1) * C].T + b[i * C:(i + 1) * C]).view(B, T, heads, 24).transpose(1, 2) for i in range(3)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, T, C)
    r_ref, r_two = rel_rms(o1.float(), ref.cpu()), rel_rms(o1.float(), o_two.float().cpu())
    print(f"[attn_small_fused] T={T} C={C} B={B}: vs fp32 reference {r_ref:.3e} (two launches: {rel_rms(o_two.float(), ref.cpu()):.3e}), vs two launches {r_two:.3e}")
    assert torch.equal(o1, o2) and r_ref <= 2e-2 and r_two <= 2e-2
    assert _exp.lib().afldm_attn_small_fused_supported(3, T, C, heads) == 0


@pytest.mark.parametrize("B", [8, 64])
def test_attn_block_fused_out_vs_front_end_plus_to_out(B):
    """afldm_attn_block_fused_out (the attention launch that also runs to_out + the residual connection + the next
    GroupNorm's partial sums, 32^2 level; reference: diffusers AttnProcessor2_0 as configured by cross_frame_attn.py:66-130
    IDLE branch) against the two launches it replaces on the same bf16 inputs - the front end's o is identical, the
    GEMM accumulates the same products in another order - and against the fp32 form of to_out on that o; the statistics
    against sums formed from the stored y itself; bit-identical reruns; hand-over counters back at zero, no error word."""
    from afldm_amd import ops
    T, C, heads, G, eps = 1024, 192, 8, 32, 1e-5
    gen = torch.Generator().manual_seed(B + 5)
    x = (torch.randn(B, T, C, generator=gen) * (0.7 + torch.rand(1, 1, C, generator=gen)) + 0.3 * torch.randn(1, 1, C, generator=gen))
    xg = x.to(torch.bfloat16).cuda()
    gamma, beta = (0.5 + torch.rand(C, generator=gen)).cuda(), (0.3 * torch.randn(C, generator=gen)).cuda()
    wq = ops.pack_weight((torch.randn(3 * C, C, generator=gen) / C ** 0.5).cuda(), torch.bfloat16)
    bq = (0.2 * torch.randn(3 * C, generator=gen)).cuda()
    wo_f = torch.randn(C, C, generator=gen) / C ** 0.5
    wo = ops.pack_weight(wo_f.cuda(), torch.bfloat16)
    bo = (0.2 * torch.randn(C, generator=gen)).cuda()
    scale = (C // heads) ** -0.5
    stats = ops.gn_stats(xg.view(B, 32, 32, C), G)
    assert ops.attn_block_fused_out_ok(xg, heads, G)
    o = ops.attn_block_fused(xg, stats, gamma, beta, G, eps, wq, bq, heads, scale)
    y_two = ops.conv2d(o.view(B, 32, 32, C), wo, bo, residual=xg.view(B, 32, 32, C), want_stats=True)
    sync = ops.new_sync_buffer(xg.device)
    with ops.sync_scope(sync):
        y = ops.attn_block_fused_out(xg, stats, gamma, beta, G, eps, wq, bq, heads, scale, wo, bo)
        y2 = ops.attn_block_fused_out(xg, stats, gamma, beta, G, eps, wq, bq, heads, scale, wo, bo)
    torch.cuda.synchronize()
    assert int(sync.abs().sum().item()) == 0, "hand-over counters / error word must be back at zero"
    assert torch.equal(y, y2) and torch.equal(y.gn_partial, y2.gn_partial)
    y32 = o.float() @ wo.view(C, C).float().t() + bo + xg.float()
    r_two, r_32, r_two32 = rel_rms(y.float(), y_two.view(B, T, C).float().cpu()), rel_rms(y.float(), y32.cpu()), rel_rms(y_two.view(B, T, C).float(), y32.cpu())
    same = float((y == y_two.view(B, T, C)).float().mean())
    print(f"[attn fused + to_out] B={B}: vs two launches {r_two:.3e} ({same * 100:.2f} % of the elements bit-equal), vs fp32 to_out {r_32:.3e} "
          f"(two launches {r_two32:.3e})")
    assert r_32 <= 3e-3 and r_32 <= 1.2 * r_two32 + 1e-4 and r_two <= 3e-3 and same >= 0.98
    st = y.gn_partial.double().sum(1).cpu()                          # [B, C, 2]
    yd = y.double().cpu()
    direct = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1)
    assert y.gn_partial.shape == (B, heads, C, 2)
    assert float((st - direct).abs().max() / direct.abs().max()) <= 1e-5


def _c8_to_nhwc(t):
    B, N, _, C = t.shape
    return t.reshape(B, C // 8, N, N, 8).permute(0, 2, 3, 1, 4).reshape(B, N, N, C)


def _nhwc_to_c8(t):
    B, N, _, C = t.shape
    out = t.reshape(B, N, N, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous().view(B, N, N, C)
    out.c8 = True
    return out


@pytest.mark.parametrize("N,C1,C2,Cout,B", [(32, 192, 0, 192, 64), (16, 384, 192, 384, 64), (32, 384, 192, 192, 8), (16, 192, 0, 384, 16)])
def test_c8_layout_activation_and_convolution_bit_identical(N, C1, C2, Cout, B):
    """The 8-channel-block layout [B][C/8][N][N][8] between the alias-free activations and the 3x3 convolutions of a
    ResnetBlock2D (afldm_af_act_c8, afldm_conv_args.x_layout / y_layout): same values, bit for bit, as the NHWC tensors -
    activation NHWC -> c8, c8 -> c8; convolution c8 -> c8 (with time embedding and statistics), c8 -> NHWC (with residual)."""
    from afldm_amd import ops
    gen = torch.Generator().manual_seed(N + C1 + C2)
    G, eps = 32, 1e-5
    x1 = (torch.randn(B, N, N, C1, generator=gen) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    x2 = (torch.randn(B, N, N, C2, generator=gen) * 0.7 - 0.1).to(torch.bfloat16).cuda() if C2 else None
    C = C1 + C2
    gamma, beta = (0.5 + torch.rand(C, generator=gen)).cuda(), (0.3 * torch.randn(C, generator=gen)).cuda()
    st = ops.gn_stats(x1, G, x2=x2)
    a_ref = ops.af_act(x1, x2, st, gamma, beta, G, eps)
    a_c8 = ops.af_act(x1, x2, st, gamma, beta, G, eps, out_c8=True)
    assert ops.is_c8(a_c8) and torch.equal(_c8_to_nhwc(a_c8), a_ref)
    w1 = ops.pack_weight((torch.randn(Cout, C, 3, 3, generator=gen) / (3 * C ** 0.5)).cuda(), torch.bfloat16)
    b1 = (0.1 * torch.randn(Cout, generator=gen)).cuda()
    temb = (0.5 * torch.randn(B, Cout, generator=gen)).to(torch.bfloat16).cuda()
    if not ops.conv2d_c8_ok(a_ref, w1, b1, temb=temb, temb_stride=Cout):
        pytest.skip("no 8-channel-block convolution for this shape (policy)")
    h_ref = ops.conv2d(a_ref, w1, b1, temb=temb, temb_stride=Cout, want_stats=True)
    h_c8 = ops.conv2d(a_c8, w1, b1, temb=temb, temb_stride=Cout, want_stats=True, out_c8=True)
    assert torch.equal(_c8_to_nhwc(h_c8), h_ref) and torch.equal(h_c8.gn_partial, h_ref.gn_partial)
    g2, be2 = (0.5 + torch.rand(Cout, generator=gen)).cuda(), (0.3 * torch.randn(Cout, generator=gen)).cuda()
    a2_ref = ops.af_act(h_ref, None, ops.gn_stats(h_ref, G), g2, be2, G, eps)
    a2_c8 = ops.af_act(h_c8, None, ops.gn_stats(h_c8, G), g2, be2, G, eps, out_c8=True)
    assert torch.equal(_c8_to_nhwc(a2_c8), a2_ref)
    w2 = ops.pack_weight((torch.randn(Cout, Cout, 3, 3, generator=gen) / (3 * Cout ** 0.5)).cuda(), torch.bfloat16)
    res = (torch.randn(B, N, N, Cout, generator=gen)).to(torch.bfloat16).cuda()
    y_ref = ops.conv2d(a2_ref, w2, None, residual=res, want_stats=True)
    y_c8in = ops.conv2d(a2_c8, w2, None, residual=res, want_stats=True)
    assert not ops.is_c8(y_c8in) and torch.equal(y_c8in, y_ref) and torch.equal(y_c8in.gn_partial, y_ref.gn_partial)


def test_resnet_block_c8_flow_bit_identical_to_nhwc(monkeypatch):
    """ResnetBlock2D at the 32^2 level (batch 64, bf16, alias-free): the default flow - tensors between activations and
    convolutions in 8-channel blocks - equals the all-NHWC flow bit for bit, with and without a conv_shortcut."""
    from afldm_amd import ops
    from afldm_amd.models.blocks import ResnetBlock2D
    from afldm_amd.af_modules.af_api import wrap_nonlinearity
    torch.manual_seed(3)
    for cin, cout in ((192, 192), (384, 192)):
        blk = ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=768, groups=32, eps=1e-5).to(torch.bfloat16).cuda()
        blk.nonlinearity = wrap_nonlinearity(blk.nonlinearity)
        x = (torch.randn(64, 32, 32, cin) * 1.2).to(torch.bfloat16).cuda()
        temb = (0.3 * torch.randn(1, cout)).to(torch.bfloat16).cuda()
        calls = []
        real = ops.af_act
        monkeypatch.setattr(ops, "af_act", lambda *a, **k: (calls.append(bool(k.get("out_c8"))), real(*a, **k))[1])
        monkeypatch.setattr(ops, "_C8", True)
        y1 = blk(x, temb, 0)
        assert calls == [True, True], calls          # both activations write 8-channel blocks (AFLDM_C8_EDGES default 5)
        monkeypatch.setattr(ops, "_C8", False)
        blk.__dict__.pop("_afldm_c8", None)
        y0 = blk(x, temb, 0)
        assert calls[2:] == [False, False]
        assert torch.equal(y0, y1) and torch.equal(y0.gn_partial, y1.gn_partial)
        monkeypatch.undo()
